// Pointwise (1x1) convolution GEMM for the wide shapes (Cout > 64) on the bf16 matrix pipe WITHOUT giving up
// fp32 results: every fp32 operand is split exactly into three bf16 numbers, x = h + m + l (truncation split,
// see pw_gemm.h), and the six partial products that can reach the fp32 result,
//     a*b ~= ah*bl + ah*bm + ah*bh + am*bh + am*bm + al*bh          (dropped terms <= 3 * 2^-24 |a||b|)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Each bf16 x bf16 product is exact in fp32, so the
// result carries the same rounding error as an fp32 FMA chain (measured on MI355X: 4e-7 relative to fp64 at
// K = 128...256, fp32 chain 3...6e-7; tools/probe_bf16split.py) while one K=16 step costs 6 x 32 cycles on
// the matrix pipe instead of 8 x 64 for v_mfma_f32_32x32x2_f32.  The MFMA phase drops below the HBM time of
// these (fused, <= 43 FLOP/B) GEMMs, which is what makes them stream-bound.
//
// Data flow (block = 4 waves = 128 px x COUTP channels of one frame; wave = CT x 32 channels x 128 px):
//   * activations: global (fp32 NCHW rows, float4 along px) -> registers (prologue: norm / GELU / SE scale /
//     norm-backward, once per element) -> split -> LDS as bf16 in MFMA-operand order.  A thread owns 4
//     consecutive ci x 4 consecutive px, so each LDS write is 8 B = 4 k-consecutive bf16 of one pixel; an MFMA
//     lane reads its whole operand (8 k-values of one pixel) with ONE ds_read_b128 that lands in the register
//     quad the MFMA consumes (no transposes, no sub-dword accesses, no register shuffles).  LDS bytes:
//     (((part*2+ks)*2+kg)*4+e)*512 + j*16 + half*8, pixel px = 4*j + e -- a lane keeps four consecutive pixels
//     in four accumulator tiles, so the epilogue stores float4 rows exactly like the fp32 kernel.
//   * weights: pre-split and pre-swizzled by pack (Wp[ks][cotile][part][lane] = 16 B = the lane's 8 bf16 of
//     A[co = lane&31][ci = 16 ks + 8 (lane>>5) + 0..7]); a wave's A load is one contiguous 1 KB read from L2.
//     A parts are re-loaded for the next k-step right after their last use (no second register set).
#include "pw_gemm.h"
// Non-temporal hints per access class, each measured inside the training step with interleaved A/B runs (history: NOTES_next_round.md):
//   * raw activation loads of the main loop: ON (-0.09 ms / step); not for the accumulate epilogue 4, which re-reads its own output
//   * stores of the fused pass-B epilogue (du2 of the dz GEMM: 268 MB at N = 4, one third of that kernel's traffic): ON (-0.09 ms)
//   * stores of the skip epilogues 5 / 6 / 8 (dx): ON (-0.09 ms)
//   * every other store and the epilogue operand loads: plain (the next kernel re-reads them; the hint cost +0.1 ... +0.15 ms)
#define PWS_NT_LOADS 1
#define PWS_NT_STORE_PASSB 1
#define PWS_NT_STORE_SKIP 1
#include <type_traits>
#include <cstdlib>
// four consecutive activation elements as loaded: fp32 storage keeps the float4, bf16 storage keeps the raw 8 bytes (half the
// prefetch registers) and is widened at staging
template <typename TA> struct PwsRaw { using type = float4; };
template <> struct PwsRaw<bf16_t> { using type = uncr_u2; };
template <bool NT, typename TA> __device__ __forceinline__ typename PwsRaw<TA>::type pws_ld(const TA* p) {
    if constexpr (sizeof(TA) == 4) return ld4<TA, NT>(p);
    else {
        if constexpr (NT) return __builtin_nontemporal_load((const uncr_u2*)p);
        else return *(const uncr_u2*)p;
    }
}
__device__ __forceinline__ float pws_get(const float4& r, int e) { return ((const float*)&r)[e]; }
__device__ __forceinline__ float pws_get(const uncr_u2& r, int e) {      // e is a compile-time constant after unrolling
    const unsigned w = e < 2 ? r.x : r.y;
    return (e & 1) ? bf16_hi(w) : bf16_lo(w);
}
template <bool NT, typename TA> __device__ __forceinline__ void pws_st(TA* p, const float4& v) { st4<TA, NT>(p, v); }

#ifndef PWS_ABL
#define PWS_ABL 0   // development ablations (tools/ablate_split.sh): 1 no stores, 2 no staging, 4 no activation loads, 8 no A reloads, 16 no MFMA, 32 no B reads, 64 no GELU
#endif
#ifndef PWS_MAP
#define PWS_MAP 1
#endif
#ifndef PWS_TROT
#define PWS_TROT 0      // tile-order rotation per frame (development knob: see tile_of)
#endif
#define PWS_TP 128
#define PWS_BUF 24576   // bytes per LDS stage: 3 parts x 32 ci x 128 px x 2 B (bf16 activations: 1 part, 8192 B)
#define PWS_A16_WPARTS 2   // bf16 activations: leading weight parts used (2 = 16 significant bits: the weights stay fp32-grade,
                           // the only rounding is the activations' bf16 storage; 1 product per part and k-step)

// DEPTH = chunks of raw activations in flight in registers (2 wherever the register budget allows: a chunk is
// only ~1.3 us of MFMA work, one chunk of prefetch distance does not cover HBM latency under load).
// PRO / EPI are compile-time: with run-time switches every staged element and every stored row carries a branch
// tree, the kernel grows to ~60 KB of code (instruction-cache misses at every jump) and the per-row coefficient
// loads each get their own vmcnt(0) -- measured: an 11k-cycle epilogue next to a 14k-cycle MFMA loop.
//
// Persistent: grid = (blocks per frame, frames) with about two blocks per CU in total; block b of a frame walks the
// pixel tiles b, b + G, b + 2G, ... and the chunk pipeline (raw chunk c+1+DEPTH in flight in registers, chunk c+1
// staged in LDS, chunk c under the MFMAs, A/B operands of the next k-step rolling in) runs straight across tile
// boundaries: when a tile's epilogue stores are issued, the next tile's first chunks and weights are already there.
// A block's two HBM streams therefore never stop for a prologue, a block relaunch or a store acknowledgement.
//
// TA = bf16_t ("bf16 activations, fp32 accumulate"): activations are read and written as bf16, the prologue still runs in fp32 and
// its result is rounded ONCE to bf16 (one operand part instead of three); the weights keep their PWS_A16_WPARTS leading parts, so
// a k-step is PWS_A16_WPARTS products instead of six and the kernel is purely stream-bound.  A fragments are double-buffered over
// two k-steps (one k-step of MFMAs no longer covers an L2 round trip).
// The 256-channel (CT = 2) variants keep ONE raw chunk in flight (a second one spills in every storage mode); two blocks per CU.
// H2 = true (fp32 storage, forward GEMMs behind a norm prologue): two fp16 parts per operand, three products (pw_gemm.h).
template <int CT, int PRO, int EPI, int DEPTH, typename TA, bool H2 = false>
__global__ __launch_bounds__(256, 2) void pw_gemm_split_kernel(PwArgs g) {
    constexpr int NT = 256, WN = 4;
    constexpr bool BF = sizeof(TA) == 2;
    static_assert(!(H2 && BF), "the fp16 two-part split is for fp32 storage");
    constexpr int NPA = BF ? 1 : (H2 ? 2 : 3);            // activation parts in LDS
    constexpr int NW = PWS_A16_WPARTS;         // weight parts used with bf16 activations
    using Raw = typename PwsRaw<TA>::type;
    constexpr bool PRE2 = PRO == PRO_NORMBWD;
    constexpr int COUTP = 32 * CT * WN;
    constexpr int NCT = CT * WN;

    __shared__ __attribute__((aligned(16))) unsigned char xs[2][NPA * 8192];
    __shared__ float cf[PRO == PRO_NORMBWD ? 4 : 3][256];      // [3]: the norm's mean (centred norm backward)
    __shared__ float red[COUTP][2];
    __shared__ float ecf[EPI == 6 ? 7 : ((EPI == 5 || EPI == 8) ? 6 : (EPI == 3 ? 5 : ((EPI == 9 || EPI == 10) ? 3 : 1)))][COUTP];   // epilogue per-channel scalars: bias, then the epi-3 A, B, S, D ([5]: epi 5 / 6 / 8 mean); epi 9 / 10: A, B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index as a scalar: row addresses stay in SGPRs
    const int j = lane & 31, kg = lane >> 5;
    const int n = blockIdx.y;
    const int Cin = g.Cin, Cout = g.Cout, P = g.P;
    const int ntile = P / PWS_TP;                                    // pixel tiles of the frame
    const int G = gridDim.x, bx = blockIdx.x;
    const int nt = (ntile - bx + G - 1) / G;                         // tiles of this block (>= 1: G <= ntile)
    const int nk = (Cin + PWS_KC - 1) / PWS_KC;          // chunks that hold data
    const int nkp = (nk + DEPTH - 1) / DEPTH * DEPTH;     // chunks computed per tile (padding chunks are all-zero)
#ifdef PWS_STAMP
    const unsigned long long ts0 = __builtin_readcyclecounter();
    unsigned long long tph[4] = {0, 0, 0, 0};     // k-step 0 | staging | request + barrier | k-step 1
#define PWS_T(i) { const unsigned long long tn_ = __builtin_readcyclecounter(); tph[i] += tn_ - tl_; tl_ = tn_; }
#else
#define PWS_T(i)
#endif

    // Every optional pointer is read branch-free (a null pointer reads a dummy location and the value is replaced by
    // a select): a load under a branch makes hipcc fall back to s_waitcnt vmcnt(0) at the join, and on gfx9 the
    // stores share that counter -- in the epilogue that serialised every row group behind all earlier stores.
    if constexpr (PRO != PRO_NONE) {
        const float* p0 = g.k0 ? g.k0 + (size_t)n * Cin : g.Wt;      // dummy location: any readable floats
        const float* p1 = g.k1 ? g.k1 + (size_t)n * Cin : g.Wt;
        const float* p2 = g.k2 ? g.k2 + (size_t)n * Cin : g.Wt;
        const float* p3 = g.k3 ? g.k3 + (size_t)n * Cin : g.Wt;
        // rows Cin .. 255 (a chunk's padding rows when Cin is not a multiple of 32) get ZERO coefficients: every prologue is linear in
        // them, so such a row stages zeros without a select per staged element (it re-reads channel 0, finite wherever that is)
        for (int i = tid; i < 256; i += NT) {
            const int ii = i < Cin ? i : 0;
            const float a = p0[ii], b = p1[ii], c = p2[ii];
            const bool in = i < Cin;
            cf[0][i] = in ? (g.k0 ? a : 1.f) : 0.f;
            cf[1][i] = in ? (g.k1 ? b : 0.f) : 0.f;
            cf[2][i] = in ? (g.k2 ? c : (PRO == PRO_AFFINE_GELU ? 1.f : 0.f)) : 0.f;
            if constexpr (PRO == PRO_NORMBWD) { const float m = p3[ii]; cf[3][i] = (in && g.k3) ? m : 0.f; }
        }
    }
    {
        const float* pb = g.bias ? g.bias + (size_t)n * g.bias_stride_n : g.Wt;
        for (int c = tid; c < COUTP; c += NT) {
            const int cc = c < Cout ? c : Cout - 1;
            const float b = pb[g.bias ? cc : 0];
            red[c][0] = 0.f; red[c][1] = 0.f;
            ecf[0][c] = g.bias ? b : 0.f;
            if constexpr (EPI == 3) {
                const int ci = n * Cout + cc;
                ecf[1][c] = g.e0[ci]; ecf[2][c] = g.e1[ci]; ecf[3][c] = g.e2[ci]; ecf[4][c] = g.e3[ci];
            }
            if constexpr (EPI == 5 || EPI == 6 || EPI == 8) {
                const int ci = n * Cout + cc;
                ecf[1][c] = g.e0[ci]; ecf[2][c] = g.e1[ci]; ecf[3][c] = g.e2[ci];
                { const float m = (g.emu ? g.emu : g.e2)[ci]; ecf[5][c] = g.emu ? m : 0.f; }
                if constexpr (EPI == 6) {      // ReLU mask: e3*aux3 + bias > 0; [6]: the pivot of sum out*(aux3 - pivot)
                    ecf[0][c] = g.bias[ci]; ecf[4][c] = g.e3[ci];
                    const float m = (g.rmu ? g.rmu : g.e3)[ci];
                    ecf[6][c] = g.rmu ? m : 0.f;
                }
            }
            if constexpr (EPI == 9 || EPI == 10) {      // 9: out = relu(A*(v + bias) + B): a ConvLayer's norm + ReLU on the fresh accumulator; 10: out = aux + A*(v + bias) + B
                const int ci = n * Cout + cc;
                ecf[1][c] = g.e0[ci]; ecf[2][c] = g.e1[ci];
            }
        }
    }

    // staging ownership: rows 4*cig .. 4*cig+3 of the chunk, pixels 4*sj .. 4*sj+3 of the tile
#if PWS_MAP
    // lane bits [2:0] = low pixel-quad bits, [3] = which 8-byte half of the 16-byte operand slot, [5:4] = high pixel-quad bits: the 16
    // consecutive lanes that one ds_write_b64 cycle serves then fill 128 CONTIGUOUS bytes (8 slots x 2 halves) -- with the plain
    // mapping (16 lanes = 16 slots, one half each: stride 16 B) lanes s and s + 8 meet on the same banks ((a / 4) mod 32 for
    // 8-byte writes), a 2-way conflict on every staging write.  Global loads keep whole 128-byte lines per 8 lanes.
    const int sj = (lane & 7) | ((lane >> 4) << 3), cig = 2 * wn + ((lane >> 3) & 1);
#else
    const int sj = tid & 31, cig = tid >> 5;
#endif
    const TA* inb = (const TA*)g.in + (size_t)n * Cin * P + 4 * sj;
    const TA* in2b = g.in2 ? (const TA*)g.in2 + (size_t)n * Cin * P + 4 * sj : inb;
    // LDS byte offset of this thread's 8-B half-slot for (part 0, e 0): ks = cig>>2, kg = (cig>>1)&1, half = cig&1
    const int st_off = ((cig >> 2) * 2 + ((cig >> 1) & 1)) * 2048 + sj * 16 + (cig & 1) * 8;

    // position in the block's chunk stream: chunk c of its ti-th tile.  Positions past the end are clamped to the
    // last tile for the loads (re-reads, never consumed) -- no branches around memory operations.
    struct Pos { int c, ti; };
    auto advance = [&](Pos& p) { const bool wrap = p.c + 1 == nkp; p.c = wrap ? 0 : p.c + 1; p.ti += wrap ? 1 : 0; };
#if PWS_TROT
    // the frames of a launch walk their tiles in rotated order (frame n starts at its block's (n mod nt)-th tile): without it the
    // blocks (bx, 0 ... N-1) request addresses that differ in the frame bits only, at every moment
    const int trot = nt > 0 ? (int)((unsigned)(n * PWS_TROT) % (unsigned)nt) : 0;
#else
    const int trot = 0;
#endif
    auto tile_of = [&](int ti) { const int t = ti + trot; return bx + (t < nt ? t : t - nt) * G; };
    auto tile_px = [&](int ti) { return tile_of(ti < nt ? ti : nt - 1) * PWS_TP; };

    Raw pre[DEPTH][4], pre2[PRE2 ? DEPTH : 1][4];
    // rows past Cin re-read row 0 (branch-free; they are zeroed at staging)
    auto load_chunk = [&](const Pos& p, auto slot) {
        constexpr int S = decltype(slot)::value;
        const int px = tile_px(p.ti);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = p.c * PWS_KC + 4 * cig + r;
            const int kk = k < Cin ? k : 0;
            pre[S][r] = pws_ld<PWS_NT_LOADS && EPI != 4, TA>(inb + (size_t)kk * P + px);
            if constexpr (PRE2) pre2[S][r] = pws_ld<PWS_NT_LOADS && EPI != 4, TA>(in2b + (size_t)kk * P + px);
        }
    };
    auto stage_chunk = [&](int kc, int buf, auto slot) {
        constexpr int S = decltype(slot)::value;
        // pixel-major so that only one pixel's 4 rows x 3 parts are live at a time (register pressure)
        float c0[4], c1[4], c2[4], c3[4];
        bool valid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kc * PWS_KC + 4 * cig + r;
            const int kk = k < Cin ? k : 0;
            valid[r] = k < Cin;
            // k < 256 whenever the chunk holds data; the padding chunk of a DEPTH-padded tile (nkp - nk <= DEPTH - 1 = 1 chunk, k < 288) wraps onto
            // rows 0..31 only where Cin = 256, and is never consumed as data there (nk = nkp for an even chunk count).
            // Non-finite inputs: a padding row stages 0 * x[channel 0]; an inf / NaN in channel 0 therefore reaches every output channel --
            // which a dense 1x1 convolution does with it anyway (every output depends on channel 0)
            static_assert(DEPTH <= 2, "kz = k & 255 assumes at most one padding chunk per tile");
            const int kz = k & 255;
            if constexpr (PRO != PRO_NONE) { c0[r] = cf[0][kz]; c1[r] = cf[1][kz]; c2[r] = cf[2][kz]; }
            if constexpr (PRO == PRO_NORMBWD) {
                c3[r] = cf[3][kz];
                if constexpr (!BF) {
                    // fp32 storage: centre the second operand in place right away, so that the means are dead before the
                    // register-hungry split section (the 256-channel variants spilled 120 B per lane with four live
                    // coefficient rows)
                    pre2[PRE2 ? S : 0][r].x -= c3[r]; pre2[PRE2 ? S : 0][r].y -= c3[r];
                    pre2[PRE2 ? S : 0][r].z -= c3[r]; pre2[PRE2 ? S : 0][r].w -= c3[r];
                }
            }
        }
        unsigned char* b = &xs[buf][0] + st_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hh[4], mm[4], ll[4];
            float vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = pws_get(pre[S][r], e);
                if constexpr (PRO == PRO_AFFINE) v = fmaf(c0[r], v, c1[r]);
                else if constexpr (PRO == PRO_AFFINE_GELU) v = (PWS_ABL & 64) ? c2[r] * fmaf(c0[r], v, c1[r]) : c2[r] * gelu_f(fmaf(c0[r], v, c1[r]));
                else if constexpr (PRO == PRO_NORMBWD) {
                    if constexpr (BF) v = fmaf(c0[r], v, fmaf(c1[r], pws_get(pre2[PRE2 ? S : 0][r], e) - c3[r], c2[r]));
                    else v = fmaf(c0[r], v, fmaf(c1[r], pws_get(pre2[PRE2 ? S : 0][r], e), c2[r]));
                }
                else if constexpr (PRO == PRO_AFFINE_RELU) v = fmaxf(fmaf(c0[r], v, c1[r]), 0.f);
                if constexpr (PRO == PRO_NONE) { if (!valid[r]) v = 0.f; }
                if constexpr (BF || H2) vv[r] = v;
                else split3_bf16(v, hh[r], mm[r], ll[r]);
            }
            if constexpr (BF) {      // one operand part: the prologue's fp32 result rounded to bf16 (RNE)
                *(u32x2_t*)(b + e * 512) = u32x2_t{cvt_pk_bf16(vv[0], vv[1]), cvt_pk_bf16(vv[2], vv[3])};
            } else if constexpr (H2) {
                unsigned h01, l01, h23, l23;
                split2_f16_pair(vv[0], vv[1], h01, l01);
                split2_f16_pair(vv[2], vv[3], h23, l23);
                *(u32x2_t*)(b + e * 512) = u32x2_t{h01, h23};
                *(u32x2_t*)(b + 8192 + e * 512) = u32x2_t{l01, l23};
            } else {
                *(u32x2_t*)(b + e * 512) = u32x2_t{pack_bf16x2(hh[0], hh[1]), pack_bf16x2(hh[2], hh[3])};
                *(u32x2_t*)(b + 8192 + e * 512) = u32x2_t{pack_bf16x2(mm[0], mm[1]), pack_bf16x2(mm[2], mm[3])};   // part stride 8192
                *(u32x2_t*)(b + 16384 + e * 512) = u32x2_t{pack_bf16x2(ll[0], ll[1]), pack_bf16x2(ll[2], ll[3])};
            }
        }
    };

    f32x16 acc[4][CT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][ct][r] = 0.f;
    };
    zero_acc();

    // A fragments: Wp[ks][cotile][part][lane] (16 B each)
    const u32x4_t* wp = (const u32x4_t*)g.Wt + (size_t)(wn * CT) * PWS_NSLOT * 64 + lane;
    auto lda = [&](int ks, int ct, int part) { return wp[((size_t)(ks * NCT + ct) * PWS_NSLOT + (H2 ? 3 : 0) + part) * 64]; };
    u32x4_t ah[CT], am[CT], al[CT];
    u32x4_t a2[BF ? 2 : 1][BF ? NW : 1][CT];      // bf16 activations: the A fragments of two consecutive k-steps
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, DEPTH - 1>;
    Pos lp{0, 0};                       // next chunk to request from HBM
    load_chunk(lp, S0{}); advance(lp);
    if constexpr (BF) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) a2[q][w][ct] = lda(q, ct, w);
    } else if constexpr (H2) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { ah[ct] = lda(0, ct, 0); al[ct] = lda(0, ct, 1); }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { ah[ct] = lda(0, ct, 0); am[ct] = lda(0, ct, 1); al[ct] = lda(0, ct, 2); }
    }
    __syncthreads();   // cf / ecf visible
    // fp16 two-part split (H2): a per-frame power-of-two scale brings a rigorous bound on the staged operand to 2^14 (a factor 4
    // below the fp16 maximum: headroom for the rounding of the bound itself); it multiplies the prologue coefficients here and
    // leaves in the epilogue together with the weights' per-channel scale (hsc).  Nothing can overflow, and everything above
    // 2^-29 of the frame's bound keeps 22 bits.  The bound:
    //   norm-backward prologue (gradient GEMM): |C1*v + C2*(v2 - mu) + C3| <= max|C1| max|v| + max|C2| (max|v2| + max|mu|) + max|C3|
    //     with the per-block maxima the producers of v and v2 left (in_amax, in2_amax);
    //   affine prologue (forward GEMM behind a norm): in_amax = [N][Cin] per-plane bounds on |A*h + B| from the statistics
    //     finalisation (uncr_norm_finalize_fwd: |h| <= sqrt(max over blocks of the block's sum h^2));
    //   affine + GELU (+ SE scale): the same times max|S| (|gelu(u)| <= |u|).
    __shared__ float hsc[H2 ? COUTP : 1];
    if constexpr (H2) {
        float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (PRO == PRO_NORMBWD) {
            for (int i = tid; i < Cin; i += NT) {
                m[0] = fmaxf(m[0], fabsf(cf[0][i])); m[1] = fmaxf(m[1], fabsf(cf[1][i]));
                m[2] = fmaxf(m[2], fabsf(cf[2][i])); m[3] = fmaxf(m[3], fabsf(cf[3][i]));
            }
            for (int i = tid; i < g.in2_amax_n; i += NT) m[5] = fmaxf(m[5], g.in2_amax[(size_t)n * g.in2_amax_n + i]);
        } else if constexpr (PRO == PRO_AFFINE_GELU) {
            for (int i = tid; i < Cin; i += NT) m[2] = fmaxf(m[2], fabsf(cf[2][i]));
        }
        if (g.in_amax)
            for (int i = tid; i < g.in_amax_n; i += NT) {
                const float v = g.in_amax[(size_t)n * g.in_amax_n + i];
                m[4] = v > m[4] || !(v == v) ? v : m[4];         // a NaN bound stays (no scaling below)
            }
        __shared__ float bred[4][6];
        __shared__ float bscale;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) { const float o = __shfl_xor(m[q], sft, 64); m[q] = o > m[q] || !(o == o) ? o : m[q]; }
            if (lane == 0) bred[wn][q] = m[q];
        }
        __syncthreads();
        if (tid == 0) {
            float mm[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                mm[q] = bred[0][q];
#pragma unroll
                for (int w = 1; w < 4; ++w) mm[q] = bred[w][q] > mm[q] || !(bred[w][q] == bred[w][q]) ? bred[w][q] : mm[q];
            }
            float bound;
            if constexpr (PRO == PRO_NORMBWD) bound = mm[0] * mm[4] + mm[1] * (mm[5] + mm[3]) + mm[2];
            else if constexpr (PRO == PRO_AFFINE_GELU) bound = mm[4] * mm[2];
            else bound = mm[4];
            float sc = 1.f;
            if (bound > 0.f && bound < 3.0e38f) {
                int e;
                (void)frexpf(bound, &e);                       // bound = f * 2^e, f in [0.5, 1)
                e = 14 - e;
                e = e > 100 ? 100 : (e < -100 ? -100 : e);
                sc = ldexpf(1.f, e);
            }
            bscale = sc;
        }
        __syncthreads();
        const float sc = bscale;
        if constexpr (PRO == PRO_AFFINE_GELU) {
            for (int i = tid; i < Cin; i += NT) cf[2][i] *= sc;
        } else {
            for (int i = tid; i < Cin; i += NT) { cf[0][i] *= sc; cf[1][i] *= sc; if constexpr (PRO == PRO_NORMBWD) cf[2][i] *= sc; }
        }
        // the weights' per-output-channel scale (tail of the packed buffer, pack_wt_split_tile) and the frame's scale leave together
        const float* wtail = g.Wt + (size_t)pws_nks(Cin) * NCT * PWS_NSLOT * 64 * 4;
        const float isc = 1.f / sc;
        for (int c = tid; c < COUTP; c += NT) hsc[c] = wtail[c] * isc;
        __syncthreads();
    }
    float amx = 0.f;          // max |stored output| of this block (CT = 1 statistics / skip epilogues)
    constexpr bool AMAXK = CT == 1 && (EPI == 1 || EPI == 2 || EPI == 5);
    constexpr bool SKIP = EPI == 5 || EPI == 6 || EPI == 8;      // the skip + PreNorm-backward epilogues
    stage_chunk(0, 0, S0{});
    load_chunk(lp, S0{}); advance(lp);
    if constexpr (DEPTH == 2) { load_chunk(lp, S1{}); advance(lp); }
    __syncthreads();

    const int rd_off = kg * 2048 + j * 16;   // + (part*2+ks)*4096 + e*512
    const int nks = 2 * nkp;
    // B operand registers roll like the A registers: each part is re-read for the NEXT k-step right after its
    // last use, so no MFMA waits on LDS latency (hipcc otherwise sinks the ds_reads next to their consumers).
    u32x4_t bh[4], bm[4], bl[4];
    auto ldb = [&](const unsigned char* p0, int part, u32x4_t (&b)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = *(const u32x4_t*)(p0 + part * 8192 + e * 512);   // one ds_read_b128 = the lane's 8 k-values
    };
    if constexpr (!BF && !H2) ldb(&xs[0][0] + rd_off, 1, bm);
    ldb(&xs[0][0] + rd_off, 0, bh);

#define PWS_MF(A, B)                                                                                          \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) _Pragma("unroll") for (int e = 0; e < 4; ++e)           \
        if (!(PWS_ABL & 16)) acc[e][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[ct]), \
                                                             __builtin_bit_cast(bf16x8_t, B[e]), acc[e][ct], 0, 0, 0)
    // one k-step: six (A part, B part) products; cb / nb = LDS address of this / the next k-step's B operand
    auto kstep = [&](int ksn, const unsigned char* cb, const unsigned char* nb, auto roll) {
        constexpr bool ROLL = decltype(roll)::value;   // false on a tile's last k-step: operands are re-read after the epilogue
        // sched_barrier(0): hipcc must not move anything across -- left alone it batches the A re-loads right before
        // their consumers and sinks the ds_reads to theirs, which serialises every latency behind the MFMAs.
        // The low B part is needed by one product only: it is read at the top of its own k-step (one product of
        // cover) so that only bm/bh stay live across the staging section between the k-steps.
#define PWS_SB() __builtin_amdgcn_sched_barrier(0)
        if (!(PWS_ABL & 32)) ldb(cb, 2, bl);
        PWS_SB();
        PWS_MF(ah, bm); PWS_SB();
        PWS_MF(ah, bl); PWS_SB();
        PWS_MF(ah, bh); PWS_SB();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) if (ROLL && !(PWS_ABL & 8)) ah[ct] = lda(ksn, ct, 0);
        PWS_SB();
        PWS_MF(am, bh); PWS_SB();
        PWS_MF(am, bm); PWS_SB();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) if (ROLL && !(PWS_ABL & 8)) am[ct] = lda(ksn, ct, 1);
        if (ROLL && !(PWS_ABL & 32)) ldb(nb, 1, bm);
        PWS_SB();
        PWS_MF(al, bh); PWS_SB();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) if (ROLL && !(PWS_ABL & 8)) al[ct] = lda(ksn, ct, 2);
        if (ROLL && !(PWS_ABL & 32)) ldb(nb, 0, bh);
        PWS_SB();
#undef PWS_SB
    };
#define PWS_MF16(A, B)                                                                                        \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) _Pragma("unroll") for (int e = 0; e < 4; ++e)           \
        if (!(PWS_ABL & 16)) acc[e][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, A[ct]),              \
                                                            __builtin_bit_cast(f16x8_t, B[e]), acc[e][ct], 0, 0, 0)
    // fp16 two-part k-step: three products; the low B part is read at the top of its own k-step, everything else rolls
    auto kstep_h2 = [&](int ksn, const unsigned char* cb, const unsigned char* nb, auto roll) {
        constexpr bool ROLL = decltype(roll)::value;
#define PWS_SB() __builtin_amdgcn_sched_barrier(0)
        ldb(cb, 1, bl);
        PWS_SB();
        PWS_MF16(ah, bh); PWS_SB();
        PWS_MF16(al, bh); PWS_SB();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) if (ROLL) al[ct] = lda(ksn, ct, 1);
        PWS_SB();
        PWS_MF16(ah, bl); PWS_SB();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) if (ROLL) ah[ct] = lda(ksn, ct, 0);
        if (ROLL) ldb(nb, 0, bh);
        PWS_SB();
#undef PWS_SB
    };
    // bf16 activations: k-step with fragment set q (= its parity): NW products, then the set is re-loaded with the weights of
    // k-step `ksn` (two k-steps ahead, wrapping into the next tile) and bh with the next k-step's B operand
    auto kstep_a16 = [&](auto qc, int ksn, const unsigned char* nb) {
        constexpr int Q = decltype(qc)::value;
#define PWS_SB() __builtin_amdgcn_sched_barrier(0)
        PWS_SB();
#pragma unroll
        for (int w = 0; w < (BF ? NW : 0); ++w) { PWS_MF(a2[BF ? Q : 0][w], bh); PWS_SB(); }
#pragma unroll
        for (int w = 0; w < (BF ? NW : 0); ++w)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) if (!(PWS_ABL & 8)) a2[BF ? Q : 0][w][ct] = lda(ksn, ct, w);
        if (!(PWS_ABL & 32)) ldb(nb, 0, bh);
        PWS_SB();
#undef PWS_SB
    };
    int par = 0;   // LDS buffer holding the chunk under the MFMAs
    using Roll = std::true_type;
    using NoRoll = std::false_type;
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    auto compute_chunk = [&](int c, auto slot, auto roll_last) {
        // k-step 0 of chunk c | stage the raw chunk held in the register slot (the stream's next chunk) into the other
        // buffer, refill the slot from HBM | barrier | k-step 1 (its rolling B reads already come from the freshly
        // staged buffer).  One barrier per chunk: a buffer is re-written only after every wave has passed the barrier
        // that follows the last reads from it.
        const unsigned char* xb = &xs[par][0] + rd_off;
        const unsigned char* xn = &xs[par ^ 1][0] + rd_off;
        const int cn = c + 1 == nkp ? 0 : c + 1;                 // the next chunk of the stream (wraps into the next tile)
#ifdef PWS_STAMP
        unsigned long long tl_ = __builtin_readcyclecounter();
#endif
        if constexpr (BF) kstep_a16(Q0{}, 2 * c + 2 >= nks ? 2 * c + 2 - nks : 2 * c + 2, xb + 4096);
        else if constexpr (H2) kstep_h2(2 * c + 1, xb, xb + 4096, Roll{});
        else kstep(2 * c + 1, xb, xb + 4096, Roll{});
        PWS_T(0)
        if (!(PWS_ABL & 2)) stage_chunk(cn, par ^ 1, slot);
        PWS_T(1)
        if (!(PWS_ABL & 4)) { load_chunk(lp, slot); advance(lp); }
        __syncthreads();
        PWS_T(2)
        if constexpr (BF) kstep_a16(Q1{}, 2 * c + 3 >= nks ? 2 * c + 3 - nks : 2 * c + 3, xn);
        else if constexpr (H2) kstep_h2(2 * c + 2 == nks ? 0 : 2 * c + 2, xb + 4096, xn, roll_last);
        else kstep(2 * c + 2 == nks ? 0 : 2 * c + 2, xb + 4096, xn, roll_last);
        PWS_T(3)
        par ^= 1;
    };

#ifdef PWS_STAMP
    const unsigned long long ts1 = __builtin_readcyclecounter();
    unsigned long long tepi = 0;
#endif
    for (int ti = 0; ti < nt; ++ti) {
        for (int kc = 0; kc + DEPTH < nkp; kc += DEPTH) {
            compute_chunk(kc, S0{}, Roll{});
            if constexpr (DEPTH == 2) compute_chunk(kc + 1, S1{}, Roll{});
        }
        // the tile's last chunk(s): no operand roll-over on the final k-step (frees 56 registers for the epilogue)
        if constexpr (DEPTH == 2) { compute_chunk(nkp - 2, S0{}, Roll{}); compute_chunk(nkp - 1, S1{}, NoRoll{}); }
        else compute_chunk(nkp - 1, S0{}, NoRoll{});

#ifdef PWS_STAMP
        const unsigned long long te0 = __builtin_readcyclecounter();
#endif
        // ---- tile epilogue (same accumulator layout as the fp32 kernel) ----
        // pass 1: bias / fused backward transform in place in the accumulators + statistics (aux loads in batches)
        // pass 2: float4 row stores, fire and forget; the next tile's chunks and weights were requested before them
        const int tile = tile_of(ti);
        const bool tile_ok = g.Pv <= 0 || (tile + 1) * PWS_TP <= g.Pv;      // (any-size planes: no statistics from a tile that reaches into the tail)
        // address of (row, lane) = uniform row base (SGPR arithmetic) + one per-lane element offset
        const int loff = 4 * kg * P + tile * PWS_TP + 4 * j;
        int nco = n * Cout;
        asm volatile("" : "+s"(nco));   // keep the 2 x 32 row bases from being hoisted out of the tile loop (SGPR spills)
        auto row_of = [&](int ct, int r) { return (wn * CT + ct) * 32 + (r & 3) + 8 * (r >> 2); };   // + 4*kg per lane
        {
            constexpr int RB = SKIP ? 4 : 8;     // rows per request batch (the skip epilogues read three rows per output row)
            constexpr bool AUX = EPI == 2 || EPI == 3 || EPI == 4 || EPI == 10 || SKIP;
            constexpr int NBAT = CT * 16 / RB;                      // batches of the tile: (ct, rb) = (bi / (16 / RB), RB * (bi % (16 / RB)))
            constexpr int NSET = 1;
            float4 xa[NSET][AUX ? RB : 1];
            float4 xb[NSET][SKIP ? RB : 1], xc[NSET][(EPI == 5 || EPI == 6) ? RB : 1];
            auto issue = [&](auto bic, auto setc) {
                constexpr int bi = decltype(bic)::value, S = decltype(setc)::value;
                constexpr int ct = bi / (16 / RB), rb = RB * (bi % (16 / RB));
                if constexpr (SKIP) {      // skip + PreNorm backward: x, dy, and the producing block's h3 (statistics)
                    const TA* a3 = g.aux3 ? (const TA*)g.aux3 : (const TA*)g.aux2;
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        const int rw = row_of(ct, rb + q);
                        const int rc = rw + 4 < Cout ? rw : (Cout > 8 ? Cout - 8 : 0);
                        const size_t o = (size_t)(nco + rc) * P + loff;
                        xa[S][q] = ld4<TA, false>((const TA*)g.aux + o);
                        xb[S][q] = ld4<TA, false>((const TA*)g.aux2 + o);
                        if constexpr (EPI != 8) xc[S][q] = ld4<TA, false>(a3 + o);      // (epi 8 masks with x itself: no third stream)
                    }
                }
                if constexpr (EPI == 4) {      // accumulate: out += result (dense 3x3 as nine shifted 1x1 GEMMs)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rw = row_of(ct, rb + q);
                        const int rc = rw + 4 < Cout ? rw : (Cout > 8 ? Cout - 8 : 0);
                        xa[S][q] = ld4<TA>((const TA*)g.out + (size_t)(nco + rc) * P + loff);
                    }
                }
                if constexpr (EPI == 2 || EPI == 3 || EPI == 10) {
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        const int rw = row_of(ct, rb + q);            // rows past Cout (padded tiles) re-read the last valid row
                        const int rc = rw + 4 < Cout ? rw : (Cout > 8 ? Cout - 8 : 0);
                        xa[S][q] = ld4<TA, false>((const TA*)g.aux + (size_t)(nco + rc) * P + loff);
                    }
                }
            };
            auto transform = [&](auto bic, auto setc) {
                constexpr int bi = decltype(bic)::value, S = decltype(setc)::value;
                constexpr int ct = bi / (16 / RB), rb = RB * (bi % (16 / RB));
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int r = rb + q;
                    const int col = row_of(ct, r) + 4 * kg;
                    const float bb = (EPI == 6 || EPI == 8) ? 0.f : ecf[0][col];
                    float4 v;
                    if constexpr (H2) {     // the weights' pack-time scale and the frame's operand scale leave here
                        const float hinv = hsc[col];
                        v = make_float4(fmaf(acc[0][ct][r], hinv, bb), fmaf(acc[1][ct][r], hinv, bb),
                                        fmaf(acc[2][ct][r], hinv, bb), fmaf(acc[3][ct][r], hinv, bb));
                    }
                    else v = make_float4(acc[0][ct][r] + bb, acc[1][ct][r] + bb, acc[2][ct][r] + bb, acc[3][ct][r] + bb);
                    float s0 = 0.f, s1 = 0.f;
                    if constexpr (EPI == 3) {
                        // du2 = gelu'(A*h2 + B) * (S*dz + D): the SE / GELU backward applied to the fresh accumulator
                        const float4 x = xa[S][q];
                        const float eA = ecf[1][col], eB = ecf[2][col], eS = ecf[3][col], eD = ecf[4][col];
                        v.x = gelu_grad_f(fmaf(eA, x.x, eB)) * fmaf(eS, v.x, eD);
                        v.y = gelu_grad_f(fmaf(eA, x.y, eB)) * fmaf(eS, v.y, eD);
                        v.z = gelu_grad_f(fmaf(eA, x.z, eB)) * fmaf(eS, v.z, eD);
                        v.w = gelu_grad_f(fmaf(eA, x.w, eB)) * fmaf(eS, v.w, eD);
                        v = rnd4<TA>(v);      // statistics of the values as stored
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                    } else if constexpr (EPI == 2) {
                        const float4 x = xa[S][q];
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                    } else if constexpr (EPI == 1) {
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    } else if constexpr (EPI == 10) {
                        // an eval-mode MBConv's closing BatchNorm (running statistics: a fixed affine map) and its skip on the fresh
                        // accumulator: y = x + A*h3 + B (uncrtaints.py:145: x + self.conv(x)), the arithmetic of ew_kernel<EW_RESIDUAL>
                        // on an h3 that is never stored; statistics of y as that kernel takes them
                        const float4 x = xa[S][q];
                        const float eA = ecf[1][col], eB = ecf[2][col];
                        v.x = x.x + fmaf(eA, v.x, eB); v.y = x.y + fmaf(eA, v.y, eB);
                        v.z = x.z + fmaf(eA, v.z, eB); v.w = x.w + fmaf(eA, v.w, eB);
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    } else if constexpr (EPI == 4) {
                        v.x += xa[S][q].x; v.y += xa[S][q].y; v.z += xa[S][q].z; v.w += xa[S][q].w;
                    } else if constexpr (EPI == 5) {
                        // dx = dy + C1*da + C2*x + C3 (PreNorm backward + skip) on the fresh accumulator da
                        const float4 x = xa[S][q], y = xb[S][q], h = xc[S][q];
                        const float e1 = ecf[1][col], e2 = ecf[2][col], e3 = ecf[3][col], em = ecf[5][col];
                        v.x = y.x + fmaf(e1, v.x, fmaf(e2, x.x - em, e3));
                        v.y = y.y + fmaf(e1, v.y, fmaf(e2, x.y - em, e3));
                        v.z = y.z + fmaf(e1, v.z, fmaf(e2, x.z - em, e3));
                        v.w = y.w + fmaf(e1, v.w, fmaf(e2, x.w - em, e3));
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * h.x + v.y * h.y + v.z * h.z + v.w * h.w;
                    } else if constexpr (EPI == 6) {
                        // as 5, then the producing ConvLayer's ReLU backward: du0 = dx * [rA*c0 + rB > 0], aux3 = c0
                        const float4 x = xa[S][q], y = xb[S][q], h = xc[S][q];
                        const float e1 = ecf[1][col], e2 = ecf[2][col], e3 = ecf[3][col], em = ecf[5][col];
                        const float rA = ecf[4][col], rB = ecf[0][col];
                        v.x = fmaf(rA, h.x, rB) > 0.f ? y.x + fmaf(e1, v.x, fmaf(e2, x.x - em, e3)) : 0.f;
                        v.y = fmaf(rA, h.y, rB) > 0.f ? y.y + fmaf(e1, v.y, fmaf(e2, x.y - em, e3)) : 0.f;
                        v.z = fmaf(rA, h.z, rB) > 0.f ? y.z + fmaf(e1, v.z, fmaf(e2, x.z - em, e3)) : 0.f;
                        v.w = fmaf(rA, h.w, rB) > 0.f ? y.w + fmaf(e1, v.w, fmaf(e2, x.w - em, e3)) : 0.f;
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        const float rm = ecf[EPI == 6 ? 6 : 0][col];
                        s1 = v.x * (h.x - rm) + v.y * (h.y - rm) + v.z * (h.z - rm) + v.w * (h.w - rm);
                    } else if constexpr (EPI == 8) {
                        // as 6 with the mask taken from x itself: x = relu(.) of the producing ConvLayer, so [x > 0] IS its ReLU mask and
                        // the layer's pre-norm tensor is not read (csrc/inconv.hip); statistics (sum du0, sum du0*x)
                        const float4 x = xa[S][q], y = xb[S][q];
                        const float e1 = ecf[1][col], e2 = ecf[2][col], e3 = ecf[3][col], em = ecf[5][col];
                        v.x = x.x > 0.f ? y.x + fmaf(e1, v.x, fmaf(e2, x.x - em, e3)) : 0.f;
                        v.y = x.y > 0.f ? y.y + fmaf(e1, v.y, fmaf(e2, x.y - em, e3)) : 0.f;
                        v.z = x.z > 0.f ? y.z + fmaf(e1, v.z, fmaf(e2, x.z - em, e3)) : 0.f;
                        v.w = x.w > 0.f ? y.w + fmaf(e1, v.w, fmaf(e2, x.w - em, e3)) : 0.f;
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                    } else if constexpr (EPI == 9) {
                        const float eA = ecf[1][col], eB = ecf[2][col];
                        v.x = fmaxf(fmaf(eA, v.x, eB), 0.f); v.y = fmaxf(fmaf(eA, v.y, eB), 0.f);
                        v.z = fmaxf(fmaf(eA, v.z, eB), 0.f); v.w = fmaxf(fmaf(eA, v.w, eB), 0.f);
                        v = rnd4<TA>(v);
                        s0 = v.x + v.y + v.z + v.w;
                        s1 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    }
                    acc[0][ct][r] = v.x; acc[1][ct][r] = v.y; acc[2][ct][r] = v.z; acc[3][ct][r] = v.w;
                    if constexpr (AMAXK) amx = fmaxf(amx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    if constexpr (EPI != 0 && EPI != 4) {
                        s0 = half_wave_sum_dpp(s0);
                        s1 = half_wave_sum_dpp(s1);
                        if (j == 31 && tile_ok) { red[col][0] += s0; red[col][1] += s1; }   // this lane owns column col in the block
                    }
                }
            };
            // compile-time walk over the batches (register sets are static after unrolling)
            auto walk = [&](auto self, auto bic) -> void {
                constexpr int bi = decltype(bic)::value;
                if constexpr (bi < NBAT) {
                    using Cur = std::integral_constant<int, 0>;
                    if constexpr (AUX) {
                        issue(bic, Cur{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    transform(bic, Cur{});
                    self(self, std::integral_constant<int, bi + 1>{});
                }
            };
            walk(walk, std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = row_of(ct, r);
                const float4 v = make_float4(acc[0][ct][r], acc[1][ct][r], acc[2][ct][r], acc[3][ct][r]);
                if (rw + 4 * kg < Cout && (!(PWS_ABL & 1) || v.x == 1.2345e-30f)) {
                    pws_st<(PWS_NT_STORE_PASSB && EPI == 3) || (PWS_NT_STORE_SKIP && SKIP), TA>((TA*)g.out + (size_t)(nco + rw) * P + loff, v);
                }
            }
        }
        zero_acc();
        if constexpr (!BF) {      // (bf16 activations: the rolling operand loads already wrapped into the next tile)
            // operands of the next tile's first k-step (its chunk 0 is already staged in xs[par])
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                ah[ct] = lda(0, ct, 0);
                if constexpr (H2) al[ct] = lda(0, ct, 1);
                else { am[ct] = lda(0, ct, 1); al[ct] = lda(0, ct, 2); }
            }
            if constexpr (!H2) ldb(&xs[par][0] + rd_off, 1, bm);
            ldb(&xs[par][0] + rd_off, 0, bh);
        }
#ifdef PWS_STAMP
        tepi += __builtin_readcyclecounter() - te0;
#endif
    }
    if constexpr (EPI != 0 && EPI != 4) {
        // one statistics slot per block (its tiles were summed in a fixed order): G slots per frame instead of P/128
        __syncthreads();
        if (!SKIP || g.part)
            for (int c = tid; c < COUTP; c += NT)
                if (c < Cout) g.part[((size_t)n * Cout + c) * G + bx] = make_float2(red[c][0], red[c][1]);
    }
    if constexpr (AMAXK) {
        if (g.amax_out) {        // kernel-uniform
            __shared__ float amr[4];
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) amx = fmaxf(amx, __shfl_xor(amx, sft, 64));
            if (lane == 0) amr[wn] = amx;
            __syncthreads();
            if (tid == 0) g.amax_out[(size_t)n * G + bx] = fmaxf(fmaxf(amr[0], amr[1]), fmaxf(amr[2], amr[3]));
        }
    }
#undef PWS_MF
#undef PWS_MF16
#ifdef PWS_STAMP
    const unsigned long long ts2 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
    __syncthreads();
    if (tid == 0 && g.e3) {   // development: per-block phase durations (cycles) -> e3 (epi 3: overwrites its D coefficients at the very end)
        const unsigned long long ts3 = __builtin_readcyclecounter();
        float* o = const_cast<float*>(g.e3) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = (float)(ts1 - ts0); o[1] = (float)(ts2 - ts1); o[2] = (float)tepi; o[3] = (float)nt;
        o[4] = (float)tph[0]; o[5] = (float)tph[1]; o[6] = (float)tph[2]; o[7] = (float)tph[3];
    }
#endif
}

#ifndef PWS_PRO
#error "compile with -DPWS_PRO=<0..4> (uncrtaints_amd/build.py builds one object per prologue kind)"
#endif

#if PWS_PRO == 0
// Wp[ks][cotile][slot][lane][8 x 16 bit] from W[co][k] (transpose=1) or W[k][co] (transpose=0); zero padded to
// Kp = 32*ceil(rows_k/32) and cp output channels; slots 0-2 = the exact bf16 parts h, m, l, slots 3-4 = the two fp16 parts of the
// SCALED weight, followed by a tail of cp floats: the inverse of the per-output-channel scale.
// fp16 scale (range safety of the two-part split): every output channel co gets its own power of two S[co] with
// S*max_k|W[k][co]| in [2^14, 2^15) -- nothing can overflow fp16 whatever the checkpoint holds, every weight within 2^17 of its
// row's maximum keeps 22 significant bits, smaller ones an absolute error of 2^-39 of the row maximum.  The scale leaves in the
// GEMM epilogue (exact: a power of two).  An all-zero row and a row holding inf / NaN get S = 1 (inf / NaN propagate like in fp32).
// One block = one co tile (32 output channels) of one weight: phase 1 = the 32 row maxima, phase 2 = the fragments of all k-steps.
__device__ __forceinline__ float pws_pow2_scale(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);                        // amax = f * 2^e, f in [0.5, 1)
    e = 15 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);     // 2^100 * (a weight < 2^-86) underflows the low part: as irrelevant as in fp32
    return ldexpf(1.f, e);
}
__device__ __forceinline__ void pack_wt_split_tile(const float* __restrict__ W, int rows_k, int cols_co, int ld,
                                                   int transpose, int nks, int nct, u32x4_t* __restrict__ out, int cot) {
    __shared__ float rmax[8][32];
    __shared__ float rscale[32];
    const int tid = threadIdx.x;
    auto wat = [&](int k, int co) -> float {
        return (k < rows_k && co < cols_co) ? (transpose ? W[(size_t)co * ld + k] : W[(size_t)k * ld + co]) : 0.f;
    };
    {
        const int co = cot * 32 + (tid & 31), sl = tid >> 5;
        float m = 0.f;
        bool bad = false;
        for (int k = sl; k < rows_k; k += 8) { const float v = fabsf(wat(k, co)); bad |= !(v < 3.0e38f); m = fmaxf(m, v); }
        rmax[sl][tid & 31] = bad ? __builtin_inff() : m;
    }
    __syncthreads();
    if (tid < 32) {
        float m = rmax[0][tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) m = fmaxf(m, rmax[q][tid]);
        const float sc = pws_pow2_scale(m);
        rscale[tid] = sc;
        float* tail = (float*)(out + (size_t)nks * nct * PWS_NSLOT * 64);
        tail[cot * 32 + tid] = 1.f / sc;
    }
    __syncthreads();
    const int lane = tid & 63;
    const int co = cot * 32 + (lane & 31);
    const float sc = rscale[lane & 31];
    for (int ks = tid >> 6; ks < nks; ks += 4) {
        const int kb = 16 * ks + 8 * (lane >> 5);
        unsigned h[8], m[8], l[8], fh[4], fl[4];
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[q] = wat(kb + q, co); split3_bf16(v[q], h[q], m[q], l[q]); }
#pragma unroll
        for (int q = 0; q < 8; q += 2) split2_f16_pair(v[q] * sc, v[q + 1] * sc, fh[q >> 1], fl[q >> 1]);
        u32x4_t* o = out + (size_t)(ks * nct + cot) * PWS_NSLOT * 64 + lane;
        o[192] = u32x4_t{fh[0], fh[1], fh[2], fh[3]};
        o[256] = u32x4_t{fl[0], fl[1], fl[2], fl[3]};
        o[0] = u32x4_t{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]), pack_bf16x2(h[4], h[5]), pack_bf16x2(h[6], h[7])};
        o[64] = u32x4_t{pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7])};
        o[128] = u32x4_t{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]), pack_bf16x2(l[4], l[5]), pack_bf16x2(l[6], l[7])};
    }
}
__global__ __launch_bounds__(256) void pack_wt_split_kernel(const float* __restrict__ W, int rows_k, int cols_co,
                                                            int ld, int transpose, int nks, int nct,
                                                            u32x4_t* __restrict__ out) {
    pack_wt_split_tile(W, rows_k, cols_co, ld, transpose, nks, nct, out, blockIdx.x);
}

// all 1x1-conv weights of a model in ONE launch (33 pack launches per step otherwise): grid = (8, items),
// descriptor = 8 x int64 {W, out, rows_k, cols_co, ld, transpose, -, -} in device memory (graph-replay safe).
// Wide items (cp >= 128): block x = co tile; narrow items (plain fp32 [Kp][cp]): the 8 blocks stride over the elements.
__global__ __launch_bounds__(256) void pack_wt_batch_kernel(const long long* __restrict__ desc) {
    const long long* d = desc + (size_t)blockIdx.y * 8;
    const float* W = (const float*)d[0];
    float* out = (float*)d[1];
    const int rows_k = (int)d[2], cols_co = (int)d[3], ld = (int)d[4], transpose = (int)d[5];
    const int cp = cols_co > 128 ? 256 : (cols_co > 64 ? 128 : (cols_co > 32 ? 64 : 32));
    if (cp >= 128) {
        if ((int)blockIdx.x < cp / 32)
            pack_wt_split_tile(W, rows_k, cols_co, ld, transpose, pws_nks(rows_k), cp / 32, (u32x4_t*)out, blockIdx.x);
    } else {
        const int Kp = (rows_k + 31) / 32 * 32;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < Kp * cp; idx += gridDim.x * 256) {
            const int k = idx / cp, co = idx % cp;
            float v = 0.f;
            if (k < rows_k && co < cols_co) v = transpose ? W[(size_t)co * ld + k] : W[(size_t)k * ld + co];
            out[idx] = v;
        }
    }
}
int pw_pack_batch(const long long* desc, int n_items, hipStream_t stream) {
    hipLaunchKernelGGL(pack_wt_batch_kernel, dim3(8, n_items), dim3(256), 0, stream, desc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

size_t pw_split_wt_floats(int rows_k, int cp) {
    const int nks = pws_nks(rows_k), nct = cp / 32;
    return (size_t)nks * nct * PWS_NSLOT * 64 * 4 + cp;   // 16 B = 4 floats per lane entry; tail: 1 / fp16 scale per output channel
}

int pw_split_pack(const float* W, int rows_k, int cols_co, int ld, int transpose, float* out, hipStream_t stream) {
    const int cp = cols_co > 128 ? 256 : 128;
    const int nks = pws_nks(rows_k), nct = cp / 32;
    hipLaunchKernelGGL(pack_wt_split_kernel, dim3(nct), dim3(256), 0, stream, W, rows_k, cols_co, ld, transpose, nks, nct,
                       (u32x4_t*)out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
#endif

template <int EPI, typename TA>
static void pws_launch_epi(const PwArgs& g, dim3 grid, int cp, hipStream_t stream) {
    // prefetch depth 2 where registers allow (CT = 1); the 256-channel tile keeps one chunk in flight with fp32 storage
    // (bf16 storage alike: depth 2 spills with the double-buffered A fragments)
#if PWS_PRO == 1 || PWS_PRO == 2
    // forward GEMMs behind a norm prologue, fp32 storage: the fp16 two-part split (three products instead of six)
    if constexpr ((EPI == 0 || EPI == 1 || EPI == 10) && sizeof(TA) == 4) {      // epi 0: the same GEMMs in eval mode behind a BatchNorm (running statistics)
        if (g.h2 && g.in_amax && g.in_amax_n > 0) {
            if (cp == 256) hipLaunchKernelGGL((pw_gemm_split_kernel<2, PWS_PRO, EPI, 1, TA, true>), grid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, EPI, 2, TA, true>), grid, dim3(256), 0, stream, g);
            return;
        }
    }
#endif
    if (cp == 256) hipLaunchKernelGGL((pw_gemm_split_kernel<2, PWS_PRO, EPI, 1, TA>), grid, dim3(256), 0, stream, g);
    else {
#if PWS_PRO == 0
        // in_conv (epi 9, Cin <= 15): ONE chunk holds the whole contraction axis; with two chunks in flight the tile is padded to two
        // (DEPTH-padding: the second is all zeros) and half of the staging and matrix work of this write-bound GEMM is spent on them
        if constexpr (EPI == 9) {
            if (g.Cin <= PWS_KC) { hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, EPI, 1, TA>), grid, dim3(256), 0, stream, g); return; }
        }
#endif
        hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, EPI, 2, TA>), grid, dim3(256), 0, stream, g);
    }
}

#define PWS_CAT2(a, b) a##b
#define PWS_CAT(a, b) PWS_CAT2(a, b)
#if PWS_PRO == 0
// persistent grid: about two resident blocks per CU in total, split evenly over the frames; also the number of
// statistics slots per (frame, channel) the kernels write
int pw_split_blocks_per_frame(int N, int P) {
    static int slots = 0;
    if (!slots) {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
            ncu = 256;
        slots = 2 * ncu;
    }
    const int ntile = P / PWS_TP;
    int bpf = slots / N;
    if (bpf < 1) bpf = 1;
    if (bpf > ntile) bpf = ntile;
    return bpf;
}
#endif

template <typename TA>
static int pws_launch_t(const PwArgs& g, int N, int cp, hipStream_t stream) {
    dim3 grid(pw_split_blocks_per_frame(N, g.P), N);
    switch (g.epi) {
        case 0: pws_launch_epi<0, TA>(g, grid, cp, stream); break;
        case 1: pws_launch_epi<1, TA>(g, grid, cp, stream); break;
        case 2: pws_launch_epi<2, TA>(g, grid, cp, stream); break;
        case 3:
#if PWS_PRO == 3
            // the dz GEMM of an MBConv backward with both operand bounds at hand: fp16 two-part split with a per-frame scale
            if (sizeof(TA) == 4 && cp == 256 && g.h2 && g.in_amax && g.in2_amax && g.in_amax_n > 0 && g.in2_amax_n > 0) {
                hipLaunchKernelGGL((pw_gemm_split_kernel<2, PWS_PRO, 3, 1, float, true>), grid, dim3(256), 0, stream, g);
                break;
            }
#endif
            pws_launch_epi<3, TA>(g, grid, cp, stream); break;
        case 4:      // accumulate (dense 3x3 as nine shifted GEMMs): fp32 storage only
            if (sizeof(TA) != 4) return UNCR_EINVAL;
            pws_launch_epi<4, float>(g, grid, cp, stream);
            break;
#if PWS_PRO == 3
        case 5:      // the backward of pw1 only: 256 -> 128 channels
            if (cp != 128) return UNCR_EINVAL;
            if (sizeof(TA) == 4 && g.h2 && g.in_amax && g.in2_amax && g.in_amax_n > 0 && g.in2_amax_n > 0)
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 5, 2, float, true>), grid, dim3(256), 0, stream, g);
            else
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 5, 2, TA>), grid, dim3(256), 0, stream, g);
            break;
        case 6:
            if (cp != 128) return UNCR_EINVAL;
            if (sizeof(TA) == 4 && g.h2 && g.in_amax && g.in2_amax && g.in_amax_n > 0 && g.in2_amax_n > 0)
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 6, 2, float, true>), grid, dim3(256), 0, stream, g);
            else
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 6, 2, TA>), grid, dim3(256), 0, stream, g);
            break;
        case 8:
            if (cp != 128) return UNCR_EINVAL;
            if (sizeof(TA) == 4 && g.h2 && g.in_amax && g.in2_amax && g.in_amax_n > 0 && g.in2_amax_n > 0)
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 8, 2, float, true>), grid, dim3(256), 0, stream, g);
            else
                hipLaunchKernelGGL((pw_gemm_split_kernel<1, PWS_PRO, 8, 2, TA>), grid, dim3(256), 0, stream, g);
            break;
#endif
#if PWS_PRO == 2
        case 10:     // eval-mode MBConv tail: closing BatchNorm + skip on the accumulator of pw2 (256 -> <= 128 channels)
            if (cp != 128) return UNCR_EINVAL;
            pws_launch_epi<10, TA>(g, grid, cp, stream); break;
#endif
#if PWS_PRO == 0
        case 9:      // a ConvLayer's norm + ReLU on the accumulator (in_conv without its pre-norm tensor, csrc/inconv.hip)
            pws_launch_epi<9, TA>(g, grid, cp, stream); break;
#endif
        default: return UNCR_EINVAL;
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

int PWS_CAT(pw_split_launch_p, PWS_PRO)(const PwArgs& g, int N, int cp, int act, hipStream_t stream) {
    if (g.P % PWS_TP) return UNCR_ESHAPE;
    if (act == UNCR_BF16) return pws_launch_t<bf16_t>(g, N, cp, stream);
    return pws_launch_t<float>(g, N, cp, stream);
}
