// Weight gradient of the wide 1x1 convolutions on the bf16 matrix pipe with fp32 results (same exact 3-way
// operand split and six products as pw_gemm_split.hip):
//
//     dW[co, ci] = sum_{p} fD(d[n, co, p]) * fX(x[n, ci, p])        (one partial [COP][CIP] per block)
//
// The contraction axis (pixels) is the contiguous one of BOTH operands, so after the prologue + split a thread's
// float4 (4 pixels of one row) is 3 x 8 B of k-consecutive bf16 and an MFMA lane's operand (8 k-values of one
// row) is one ds_read_b128 -- no transposes anywhere.
//
// Block = 8 waves (one block per CU, 256 VGPRs per wave), each wave owns a 64 x 64 corner of the [COP][CIP]
// product (4 accumulator tiles), the block walks its share of the frame's 32-pixel chunks:
//   raw chunk c+2 in flight (registers) | chunk c+1 being staged into the other LDS buffer, one float4 piece after
//   each 4-MFMA group of k-step 0 | barrier | k-step 1, whose operand prefetches already read the new buffer.
// The MFMA stream never waits on HBM: the only vm-counter traffic in the loop are those raw loads, each consumed a
// full chunk after it was issued.  LDS bytes of one buffer: [part 3][plane 4 = ks*2+kg][row R][16 B], plane stride
// padded by 32 B (bank spread of the 8-B staging writes).
//
// H2 (fp32 storage, bounds at hand): two fp16 parts per operand and three products, as the forward GEMMs and the dz GEMM run
// (pw_gemm.h, split2_f16_pair).  The contraction runs over pixels here, so every ROW of either operand has its own power-of-two
// scale: a rigorous bound on the staged row -- norm-backward rows |c0| max|d| + |c1| (max|d2| + |mu|) + |c2| from the per-block
// maxima the producers of d and d2 left, GELU rows the statistics finalisation's bound on |A h + B| (|gelu(u)| <= |u|) times |s| --
// brought to 2^14.  The scales multiply the rows' prologue coefficients and leave the [COP][CIP] partial in the epilogue (exact).
#include "pw_gemm.h"
#include <type_traits>

#ifndef WGS_PF
#define WGS_PF 1       // raw chunks in flight per thread in the fp16-split kernel (1 | 2).  2 (254 VGPRs, no spill): 146 -> 138 us per
                       // launch in isolation at N = 4, 390 -> 386 at N = 12, the training step unchanged (4 + 3 interleaved pairs)
#endif
#ifndef WGS_ROT
#define WGS_ROT 5      // chunks by which consecutive blocks of a frame are rotated inside their pixel ranges (0: every block starts at its
                       // range's first chunk).  The ranges start P / gridDim.x pixels apart -- 4 KB per row at N = 4 -- so un-rotated
                       // blocks request the same low address bits at every moment.  Measured inside the training step, 9 interleaved
                       // pairs (profiles/r06_ab_wgrot2.log): 11.11 -> 10.93 ms with 5, 10.96 with 7, 10.95 with an even spread; in
                       // isolation on cold operands N = 4 does not move and N = 8 gains 11 % (r06_time_wgrad2.log)
#endif
#ifndef WGS_ABL
#define WGS_ABL 0      // development ablations (timing only, results wrong): 1 no global loads after the first two chunks, 2 no MFMA,
#endif                 // 4 no staging (prologue arithmetic, split, LDS writes) after the first chunk

struct WgsArgs {
    const float* d;
    const float* d2;
    const float* x;
    const float* dk0; const float* dk1; const float* dk2;   // [N*Cd]
    const float* xk0; const float* xk1; const float* xk2;   // [N*Cx]
    float* part;       // [N*G][COP][CIP]
    int Cd, Cx, P;
    const float* dk3;  // the norm's mean per (n, co) (centred norm backward on d) or null
    // H2: per-block maxima of |d| and |d2| ([N][d_amax_n], [N][d2_amax_n]) and per-plane bounds on the affine input of x ([N*Cx])
    const float* d_amax; int d_amax_n;
    const float* d2_amax; int d2_amax_n;
    const float* x_ub;
    int Pv;            // > 0: padded planes of an any-size image -- the whole 32-pixel chunks below Pv only (uncr_wgrad_boundary adds the rest)
};

// power of two bringing `bound` to [2^13, 2^14); 1 for a zero / non-finite bound (inf / NaN then propagate as in fp32)
__device__ __forceinline__ float wgs_scale14(float bound) {
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(bound, &e);
    e = 14 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, e);
}

template <int PRO>
__device__ __forceinline__ float wgs_pro(float v, float v2, float c0, float c1, float c2, float c3 = 0.f) {
    if constexpr (PRO == PRO_AFFINE) return fmaf(c0, v, c1);
    else if constexpr (PRO == PRO_AFFINE_GELU) return c2 * gelu_f(fmaf(c0, v, c1));
    else if constexpr (PRO == PRO_NORMBWD) return fmaf(c0, v, fmaf(c1, v2 - c3, c2));
    else if constexpr (PRO == PRO_AFFINE_RELU) return fmaxf(fmaf(c0, v, c1), 0.f);
    else return v;
}

// two pixels of one row on the packed fp32 ops (same operations per component as wgs_pro)
template <int PRO>
__device__ __forceinline__ f32x2 wgs_pro2(f32x2 v, f32x2 v2, float c0, float c1, float c2, float c3 = 0.f) {
    if constexpr (PRO == PRO_AFFINE) return fma2(f2(c0), v, f2(c1));
    else if constexpr (PRO == PRO_AFFINE_GELU) return f2(c2) * gelu_f2(fma2(f2(c0), v, f2(c1)));
    else if constexpr (PRO == PRO_NORMBWD) return fma2(f2(c0), v, fma2(f2(c1), v2 - f2(c3), f2(c2)));
    else if constexpr (PRO == PRO_AFFINE_RELU) return f2(fmaxf(fmaf(c0, v.x, c1), 0.f), fmaxf(fmaf(c0, v.y, c1), 0.f));
    else return v;
}

template <int WCO, int WCI, int PRO_D, int PRO_X, bool H2 = false>
__global__ __launch_bounds__(512, 1) void pw_wgrad_split_kernel(WgsArgs g) {
    constexpr int NT = 512;
    constexpr int COP = 64 * WCO, CIP = 64 * WCI, R = COP + CIP;
    constexpr int PS = R * 16 + 32;          // plane stride (bytes)
    constexpr int PART = 4 * PS, BUF = 3 * PART;
    constexpr int ND = COP / 64, NX = CIP / 64;   // float4 pieces per thread per chunk (512 threads x 8 float4 per row)
    constexpr bool D2 = PRO_D == PRO_NORMBWD;
    static_assert(WCO * WCI == 8, "8 waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                           // [2][BUF]
#ifdef WGS_STAMP
    const unsigned long long ts0 = __builtin_readcyclecounter();       // development: per-block phase stamps, see the kernel's end
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wv / WCI, wci = wv % WCI;
    const int n = blockIdx.y;
    const int P = g.P;
    const int nch = (g.Pv > 0 ? g.Pv : P) / 32;
    const int cbeg = (int)((long long)blockIdx.x * nch / gridDim.x), cend = (int)((long long)(blockIdx.x + 1) * nch / gridDim.x);
    const int nc = cend - cbeg;
    // the block starts rot0 chunks into its range and wraps around (see WGS_ROT)
    const int rot0 = nc > 0 ? (int)((unsigned)(blockIdx.x * WGS_ROT + n * 3) % (unsigned)nc) : 0;
    auto rot_chunk = [&](int ch) { const int c = ch < 0 ? 0 : ch; return c + rot0 < nc ? c + rot0 : (c + rot0 < 2 * nc ? c + rot0 - nc : 0); };

    // loader mapping: piece i covers rows (tid>>3) + 64*i, float4 column c4 = tid & 7 of the 32-pixel chunk
    const int lrow = tid >> 3, c4 = tid & 7;
    const float* dbase = g.d + ((size_t)n * COP + lrow) * P + 4 * c4;
    const float* d2base = D2 ? g.d2 + ((size_t)n * COP + lrow) * P + 4 * c4 : dbase;
    const float* xbase = g.x + ((size_t)n * CIP + lrow) * P + 4 * c4;
    const int st_off = (c4 >> 1) * PS + lrow * 16 + (c4 & 1) * 8;   // + part*PART + 64*i*16 (+ COP*16 for x rows)

    // per-row prologue coefficients of this thread's six rows: chunk-invariant, kept in registers (optional pointers
    // are read branch-free: a null pointer reads a dummy location and the value is replaced by a select)
    float k0[ND + NX], k1[ND + NX], k2[ND + NX], k3[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const float m = (g.dk3 ? g.dk3 : g.d)[g.dk3 ? n * COP + lrow + 64 * i : 0];
        k3[i] = g.dk3 ? m : 0.f;
    }
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) {
        const bool isd = i < ND;
        const int idx = isd ? n * COP + lrow + 64 * i : n * CIP + lrow + 64 * (i - ND);
        const float* q0 = isd ? g.dk0 : g.xk0;
        const float* q1 = isd ? g.dk1 : g.xk1;
        const float* q2 = isd ? g.dk2 : g.xk2;
        const float a = (q0 ? q0 : g.d)[q0 ? idx : 0], b = (q1 ? q1 : g.d)[q1 ? idx : 0], c = (q2 ? q2 : g.d)[q2 ? idx : 0];
        k0[i] = q0 ? a : 1.f;
        k1[i] = q1 ? b : 0.f;
        k2[i] = q2 ? c : ((isd ? PRO_D : PRO_X) == PRO_AFFINE_GELU ? 1.f : 0.f);
    }

    __shared__ float rsc[H2 ? R : 1];      // H2: 1 / scale of every staged row
    if constexpr (H2) {
        static_assert(PRO_D == PRO_NORMBWD && PRO_X == PRO_AFFINE_GELU, "bounds are derived for these prologues");
        float a1 = 0.f, a2 = 0.f;       // the frame's max |d|, max |d2| (a NaN maximum stays: no scaling then)
        for (int j = lane; j < g.d_amax_n; j += 64) { const float v = g.d_amax[(size_t)n * g.d_amax_n + j]; a1 = v > a1 || !(v == v) ? v : a1; }
        for (int j = lane; j < g.d2_amax_n; j += 64) { const float v = g.d2_amax[(size_t)n * g.d2_amax_n + j]; a2 = v > a2 || !(v == v) ? v : a2; }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const float o1 = __shfl_xor(a1, sft, 64), o2 = __shfl_xor(a2, sft, 64);
            a1 = o1 > a1 || !(o1 == o1) ? o1 : a1;
            a2 = o2 > a2 || !(o2 == o2) ? o2 : a2;
        }
#pragma unroll
        for (int i = 0; i < ND + NX; ++i) {
            float sc;
            if (i < ND) {
                sc = wgs_scale14(fabsf(k0[i]) * a1 + fabsf(k1[i]) * (a2 + fabsf(k3[i < ND ? i : 0])) + fabsf(k2[i]));
                k0[i] *= sc; k1[i] *= sc; k2[i] *= sc;
            } else {
                sc = wgs_scale14(g.x_ub[n * CIP + lrow + 64 * (i - ND)] * fabsf(k2[i]));
                k2[i] *= sc;
            }
            if (c4 == 0) rsc[(i < ND ? 0 : COP) + lrow + 64 * (i < ND ? i : i - ND)] = 1.f / sc;
        }
    }

    // raw chunks in flight (registers): PF sets (WGS_PF; the exact-split kernels have no registers left for a second one)
    constexpr int PF = H2 ? WGS_PF : 1;
    float4 dv[PF][ND], dv2[PF][D2 ? ND : 1], xv[PF][NX];
    bool abl_first = true;
    auto load_piece = [&](int i, int ch, int set = 0) {   // i, set compile-time after unrolling; ch clamped by the caller
        if ((WGS_ABL & 1) && !abl_first) return;
        const size_t po = (size_t)(cbeg + rot_chunk(ch)) * 32;
        if (i < ND) {
            dv[set][i] = ld_nt4(dbase + (size_t)(64 * i) * P + po);
            if constexpr (D2) dv2[set][i] = ld_nt4(d2base + (size_t)(64 * i) * P + po);
        } else {
            xv[set][i - ND] = ld_nt4(xbase + (size_t)(64 * (i - ND)) * P + po);
        }
    };
    auto stage_piece = [&](int i, int buf, int set = 0) {
        if ((WGS_ABL & 4) && !abl_first) return;
        const bool isd = i < ND;
        const int row = isd ? lrow + 64 * i : COP + lrow + 64 * (i - ND);
        const float c0 = k0[i], c1 = k1[i], c2 = k2[i];
        float4 v = isd ? dv[set][i] : xv[set][i - ND];
        float4 w = (isd && D2) ? dv2[set][D2 ? i : 0] : v;
        if (WGS_ABL & 1) {      // keep the arithmetic inside the loop although its inputs no longer change
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w));
        }
        unsigned char* b = xs + buf * BUF + st_off + (row - lrow) * 16;
        if constexpr (H2) {
            unsigned hh[2], ll[2];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const f32x2 a = f2(((const float*)&v)[q], ((const float*)&v)[q + 1]), bb = f2(((const float*)&w)[q], ((const float*)&w)[q + 1]);
                const f32x2 t = isd ? wgs_pro2<PRO_D>(a, bb, c0, c1, c2, k3[isd ? i : 0]) : wgs_pro2<PRO_X>(a, bb, c0, c1, c2);
                split2_f16_pair(t.x, t.y, hh[q >> 1], ll[q >> 1]);
            }
            *(u32x2_t*)(b) = u32x2_t{hh[0], hh[1]};
            *(u32x2_t*)(b + PART) = u32x2_t{ll[0], ll[1]};
        } else {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const f32x2 a = f2(((const float*)&v)[q], ((const float*)&v)[q + 1]), bb = f2(((const float*)&w)[q], ((const float*)&w)[q + 1]);
                const f32x2 t = isd ? wgs_pro2<PRO_D>(a, bb, c0, c1, c2, k3[isd ? i : 0]) : wgs_pro2<PRO_X>(a, bb, c0, c1, c2);
                split3_bf16(t.x, h[q], m[q], l[q]);
                split3_bf16(t.y, h[q + 1], m[q + 1], l[q + 1]);
            }
            *(u32x2_t*)(b) = u32x2_t{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
            *(u32x2_t*)(b + PART) = u32x2_t{pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3])};
            *(u32x2_t*)(b + 2 * PART) = u32x2_t{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
        }
    };

#ifdef WGS_STAMP
    const unsigned long long ts1 = __builtin_readcyclecounter();       // coefficients + bounds done
#endif
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // operand addresses: plane = ks*2 + (lane>>5); A rows of this wave's two co tiles, B rows of its two ci tiles
    const int arow = (wco * 2) * 32 + (lane & 31), brow = COP + (wci * 2) * 32 + (lane & 31);
    const int aoff = (lane >> 5) * PS + arow * 16, boff = (lane >> 5) * PS + brow * 16;
    auto ldop = [&](int buf, int ks, int part, int off, u32x4_t (&o)[2]) {
        const unsigned char* p = xs + buf * BUF + part * PART + ks * 2 * PS + off;
        o[0] = *(const u32x4_t*)p;
        o[1] = *(const u32x4_t*)(p + 32 * 16);
    };

    // prologue: chunk 0 -> buffer 0, chunk 1 raw in registers
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) load_piece(i, 0);
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) {
        stage_piece(i, 0);
        if constexpr (PF == 2) { load_piece(i, nc > 1 ? 1 : 0, 1); load_piece(i, nc > 2 ? 2 : nc - 1, 0); }
        else load_piece(i, nc > 1 ? 1 : 0);
    }
    __syncthreads();
#ifdef WGS_STAMP
    const unsigned long long ts1b = __builtin_readcyclecounter();      // first chunk staged (HBM latency of the block's first loads)
#endif
    abl_first = false;

    // rolling operands (ah, am, bh re-read in place after their last use) and double-buffered single-use ones
    u32x4_t ah[2], am[2], bh[2], al[2][2], bl[2][2], bm[2][2];
    if constexpr (H2) {
        // parts: 0 = high, 1 = low.  Products per k-step: ah*bl, ah*bh, al*bh; two staging pieces ride behind each in k-step 0.
        ldop(0, 0, 0, aoff, ah); ldop(0, 0, 0, boff, bh); ldop(0, 0, 1, aoff, al[0]); ldop(0, 0, 1, boff, bl[0]);
#define WGS_MFH(A, B)                                                                                            \
    if (!(WGS_ABL & 2)) _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, A[a]),                    \
                                                           __builtin_bit_cast(f16x8_t, B[b]), acc[a][b], 0, 0, 0)
#define WGS_SBH() __builtin_amdgcn_sched_barrier(0)
        // SET: the register set holding raw chunk c+1 (PF == 2: (c + 1) & 1, the loop is unrolled over chunk pairs so that it is static)
        auto body = [&](int c, auto SETC) {
            constexpr int SET = decltype(SETC)::value;
            const int cur = PF == 2 ? (SET ^ 1) : (c & 1);
            const int c2 = c + 1 + PF < nc ? c + 1 + PF : nc - 1;
            WGS_MFH(ah, bl[0]); WGS_SBH();
            ldop(cur, 1, 1, aoff, al[1]); ldop(cur, 1, 1, boff, bl[1]);
            if (0 < ND + NX) { stage_piece(0, cur ^ 1, SET); load_piece(0, c2, SET); }
            if (1 < ND + NX) { stage_piece(1, cur ^ 1, SET); load_piece(1, c2, SET); }
            WGS_SBH();
            WGS_MFH(ah, bh); WGS_SBH();
            ldop(cur, 1, 0, aoff, ah);
            if (2 < ND + NX) { stage_piece(2, cur ^ 1, SET); load_piece(2, c2, SET); }
            if (3 < ND + NX) { stage_piece(3, cur ^ 1, SET); load_piece(3, c2, SET); }
            WGS_SBH();
            WGS_MFH(al[0], bh); WGS_SBH();
            ldop(cur, 1, 0, boff, bh);
            if (4 < ND + NX) { stage_piece(4, cur ^ 1, SET); load_piece(4, c2, SET); }
            if (5 < ND + NX) { stage_piece(5, cur ^ 1, SET); load_piece(5, c2, SET); }
            WGS_SBH();
            __syncthreads();
            WGS_MFH(ah, bl[1]); WGS_SBH();
            ldop(cur ^ 1, 0, 1, aoff, al[0]); ldop(cur ^ 1, 0, 1, boff, bl[0]);
            WGS_SBH();
            WGS_MFH(ah, bh); WGS_SBH();
            ldop(cur ^ 1, 0, 0, aoff, ah);
            WGS_SBH();
            WGS_MFH(al[1], bh); WGS_SBH();
            ldop(cur ^ 1, 0, 0, boff, bh);
            WGS_SBH();
        };
        if constexpr (PF == 2) {
            for (int c = 0; c < nc; c += 2) {
                body(c, std::integral_constant<int, 1>{});
                if (c + 1 < nc) body(c + 1, std::integral_constant<int, 0>{});
            }
        } else {
            for (int c = 0; c < nc; ++c) body(c, std::integral_constant<int, 0>{});
        }
#undef WGS_MFH
#undef WGS_SBH
    } else {
    ldop(0, 0, 0, aoff, ah); ldop(0, 0, 1, aoff, am); ldop(0, 0, 0, boff, bh);
    ldop(0, 0, 2, aoff, al[0]); ldop(0, 0, 2, boff, bl[0]); ldop(0, 0, 1, boff, bm[0]);

#define WGS_MF(A, B)                                                                                             \
    if (!(WGS_ABL & 2)) _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[a]),                  \
                                                            __builtin_bit_cast(bf16x8_t, B[b]), acc[a][b], 0, 0, 0)
#define WGS_SB() __builtin_amdgcn_sched_barrier(0)
    for (int c = 0; c < nc; ++c) {
        const int cur = c & 1;
        const int c2 = c + 2 < nc ? c + 2 : nc - 1;     // raw chunk to request (clamped, branch-free)
        // ---- k-step 0 (operands of k-step 1 come from the same buffer); one staging piece after each product ----
        {
            constexpr int Q = 0;   // parity of the double-buffered operands
            WGS_MF(ah, bl[Q]); WGS_SB();
            ldop(cur, 1, 2, aoff, al[Q ^ 1]); ldop(cur, 1, 2, boff, bl[Q ^ 1]);
            if (0 < ND + NX) { stage_piece(0, cur ^ 1); load_piece(0, c2); }
            WGS_SB();
            WGS_MF(ah, bm[Q]); WGS_SB();
            ldop(cur, 1, 1, boff, bm[Q ^ 1]);
            if (1 < ND + NX) { stage_piece(1, cur ^ 1); load_piece(1, c2); }
            WGS_SB();
            WGS_MF(ah, bh); WGS_SB();
            ldop(cur, 1, 0, aoff, ah);
            if (2 < ND + NX) { stage_piece(2, cur ^ 1); load_piece(2, c2); }
            WGS_SB();
            WGS_MF(am, bh); WGS_SB();
            if (3 < ND + NX) { stage_piece(3, cur ^ 1); load_piece(3, c2); }
            WGS_SB();
            WGS_MF(al[Q], bh); WGS_SB();
            ldop(cur, 1, 0, boff, bh);
            if (4 < ND + NX) { stage_piece(4, cur ^ 1); load_piece(4, c2); }
            WGS_SB();
            WGS_MF(am, bm[Q]); WGS_SB();
            ldop(cur, 1, 1, aoff, am);
            if (5 < ND + NX) { stage_piece(5, cur ^ 1); load_piece(5, c2); }
            WGS_SB();
        }
        __syncthreads();   // chunk c+1 staged by every wave; nobody still reads buffer cur^1's old contents
        // ---- k-step 1 (operands of the next chunk's k-step 0 come from the freshly staged buffer) ----
        {
            constexpr int Q = 1;
            WGS_MF(ah, bl[Q]); WGS_SB();
            ldop(cur ^ 1, 0, 2, aoff, al[Q ^ 1]); ldop(cur ^ 1, 0, 2, boff, bl[Q ^ 1]);
            WGS_SB();
            WGS_MF(ah, bm[Q]); WGS_SB();
            ldop(cur ^ 1, 0, 1, boff, bm[Q ^ 1]);
            WGS_SB();
            WGS_MF(ah, bh); WGS_SB();
            ldop(cur ^ 1, 0, 0, aoff, ah);
            WGS_SB();
            WGS_MF(am, bh); WGS_SB();
            WGS_MF(al[Q], bh); WGS_SB();
            ldop(cur ^ 1, 0, 0, boff, bh);
            WGS_SB();
            WGS_MF(am, bm[Q]); WGS_SB();
            ldop(cur ^ 1, 0, 1, aoff, am);
            WGS_SB();
        }
    }
#undef WGS_MF
#undef WGS_SB
    }

#ifdef WGS_STAMP
    const unsigned long long ts2 = __builtin_readcyclecounter();       // chunk loop done
#endif
    float* po = g.part + ((size_t)n * gridDim.x + blockIdx.x) * COP * CIP;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wco * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = (wci * 2 + b) * 32 + (lane & 31);
                float v = acc[a][b][r];
                if constexpr (H2) v = v * rsc[co] * rsc[COP + ci];      // powers of two: exact
                po[co * CIP + ci] = v;
            }
#ifdef WGS_STAMP
    __builtin_amdgcn_s_waitcnt(0);       // stores acknowledged
    __syncthreads();
    if (tid == 0) {   // (overwrites the first values of this block's partial: timing builds only)
        const unsigned long long ts3 = __builtin_readcyclecounter();
        po[0] = (float)(ts1 - ts0); po[1] = (float)(ts2 - ts1b); po[2] = (float)(ts3 - ts2); po[3] = (float)nc;
        po[4] = (float)(ts0 & 0xFFFFFF); po[5] = (float)(ts3 & 0xFFFFFF); po[6] = (float)(ts1b - ts1);
        po[7] = (float)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF);      // HW_REG_XCC_ID[3:0]
    }
#endif
}

static int wgs_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
            ncu = 256;
    }
    return ncu;
}

// blocks per frame of the split kernel: one block per CU in total, never more than one block per 32-pixel chunk
int pw_wgrad_split_nbx(int N, int P) {
    int g = wgs_ncu() / N;
    if (g < 1) g = 1;
    if (g > P / 32) g = P / 32;
    return g;
}

// shape 0: (Cd, Cx) = (256, 128), shape 1: (128, 256); (pro_d, pro_x) in {(NORMBWD, AFFINE), (NORMBWD, AFFINE_GELU)}
bool pw_wgrad_split_supported(int Cd, int Cx, int pro_d, int pro_x, bool rowsum) {
    if (rowsum || pro_d != PRO_NORMBWD) return false;
    if (!((Cd == 256 && Cx == 128) || (Cd == 128 && Cx == 256))) return false;
    return pro_x == PRO_AFFINE || pro_x == PRO_AFFINE_GELU;
}

template <int WCO, int WCI, int PRO_X, bool H2 = false>
static int wgs_launch(const WgsArgs& g, dim3 grid, hipStream_t stream) {
    constexpr int R = 64 * WCO + 64 * WCI;
    constexpr size_t lds = 2 * 3 * 4 * (size_t)(R * 16 + 32);
    auto kern = pw_wgrad_split_kernel<WCO, WCI, PRO_NORMBWD, PRO_X, H2>;
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return UNCR_EINVAL;
        once = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, g);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

int pw_wgrad_split_launch(const float* d, const float* d2, const float* x, const float* dk0, const float* dk1,
                          const float* dk2, const float* dkmu, const float* xk0, const float* xk1, const float* xk2, float* part,
                          int N, int Cd, int Cx, int P, int nbx, int pro_x, const float* d_amax, int d_amax_n,
                          const float* d2_amax, int d2_amax_n, const float* x_ub, int Pv, hipStream_t stream) {
    if (P % 32 || nbx < 1 || nbx > P / 32) return UNCR_ESHAPE;
    WgsArgs g{d, d2, x, dk0, dk1, dk2, xk0, xk1, xk2, part, Cd, Cx, P, dkmu, d_amax, d_amax_n, d2_amax, d2_amax_n, x_ub, Pv};
    dim3 grid(nbx, N);
    // every bound at hand (and the shape the bounds are derived for): two fp16 parts, three products
    if (Cd == 128 && pro_x == PRO_AFFINE_GELU && d_amax && d_amax_n > 0 && d2_amax && d2_amax_n > 0 && x_ub)
        return wgs_launch<2, 4, PRO_AFFINE_GELU, true>(g, grid, stream);
    if (Cd == 256) {
        if (pro_x == PRO_AFFINE) return wgs_launch<4, 2, PRO_AFFINE>(g, grid, stream);
        return wgs_launch<4, 2, PRO_AFFINE_GELU>(g, grid, stream);
    }
    if (pro_x == PRO_AFFINE) return wgs_launch<2, 4, PRO_AFFINE>(g, grid, stream);
    return wgs_launch<2, 4, PRO_AFFINE_GELU>(g, grid, stream);
}
