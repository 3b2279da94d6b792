// Weight gradient of the wide 1x1 convolutions from bf16 activations ("bf16 activations, fp32 accumulate", BASELINE config 3):
//
//     dW[co, ci] = sum_p fD(d[n, co, p]) * fX(x[n, ci, p])        (one fp32 partial [COP][CIP] per block)
//
// d, d2, x are stored as bf16; the prologues fD (norm backward of two tensors) and fX (norm apply [+ GELU]) run in fp32 and
// their results are rounded once to bf16: one v_mfma_f32_32x32x16_bf16 product per MAC with fp32 accumulation -- the kernel is
// a pure stream (16 MFMAs against 80 KB of operands per 64-pixel chunk and CU).
//
// The contraction axis (pixels) is the contiguous one of BOTH operands: a thread's 16-byte load is 8 consecutive pixels of one
// row = after the prologue one 16-byte LDS write = exactly the 8 k-values an MFMA lane needs (one ds_read_b128), no transposes.
// Block = 8 waves, one block per CU; each wave owns a 64 x 64 corner of the [COP][CIP] product; the block walks its share of the
// frame's 64-pixel chunks: raw chunk c+2 in flight in registers | chunk c+1 being staged into the other LDS buffer between the
// k-steps of chunk c | one barrier per chunk.  LDS bytes of one buffer: [plane 8 = pixel octet][row R][16 B], plane stride
// padded by 16 B (the eight lanes that write one row's octets hit eight different bank groups).
#include "pw_gemm.h"

struct WgaArgs {
    const bf16_t* d;
    const bf16_t* d2;
    const bf16_t* x;
    const float* dk0; const float* dk1; const float* dk2;   // [N*Cd]
    const float* xk0; const float* xk1; const float* xk2;   // [N*Cx]
    float* part;       // [N*G][COP][CIP]
    int Cd, Cx, P;
    const float* dk3;  // the norm's mean per (n, co) (centred norm backward on d) or null
};

template <int PRO>
__device__ __forceinline__ float wga_pro(float v, float v2, float c0, float c1, float c2, float c3 = 0.f) {
    if constexpr (PRO == PRO_AFFINE) return fmaf(c0, v, c1);
    else if constexpr (PRO == PRO_AFFINE_GELU) return c2 * gelu_f(fmaf(c0, v, c1));
    else if constexpr (PRO == PRO_NORMBWD) return fmaf(c0, v, fmaf(c1, v2 - c3, c2));
    else return v;
}

template <int WCO, int WCI, int PRO_X>
__global__ __launch_bounds__(512, 1) void pw_wgrad_a16_kernel(WgaArgs g) {
    constexpr int COP = 64 * WCO, CIP = 64 * WCI, R = COP + CIP;
    constexpr int PS = R * 16 + 16;          // plane stride (bytes)
    constexpr int BUF = 8 * PS;
    constexpr int ND = COP / 64, NX = CIP / 64;   // 16-byte pieces per thread per chunk (512 threads = 64 rows x 8 octets)
    static_assert(WCO * WCI == 8, "8 waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][BUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wv / WCI, wci = wv % WCI;
    const int n = blockIdx.y;
    const int P = g.P;
    const int nch = P / 64;
    const int cbeg = (int)((long long)blockIdx.x * nch / gridDim.x), cend = (int)((long long)(blockIdx.x + 1) * nch / gridDim.x);
    const int nc = cend - cbeg;
    if (nc <= 0) return;      // (grid <= chunks: never taken; keeps the clamps below well defined)

    const int lrow = tid >> 3, c8 = tid & 7;
    const bf16_t* dbase = g.d + ((size_t)n * COP + lrow) * P + 8 * c8;
    const bf16_t* d2base = g.d2 + ((size_t)n * COP + lrow) * P + 8 * c8;
    const bf16_t* xbase = g.x + ((size_t)n * CIP + lrow) * P + 8 * c8;
    unsigned char* st_base = smem + c8 * PS + lrow * 16;      // + buf*BUF + 64*i*16 (+ COP*16 for x rows)

    // per-row prologue coefficients of this thread's rows (chunk-invariant); optional pointers read branch-free
    float k0[ND + NX], k1[ND + NX], k2[ND + NX], k3[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const float m = (g.dk3 ? g.dk3 : g.dk0)[g.dk3 ? n * COP + lrow + 64 * i : 0];
        k3[i] = g.dk3 ? m : 0.f;
    }
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) {
        const bool isd = i < ND;
        const int idx = isd ? n * COP + lrow + 64 * i : n * CIP + lrow + 64 * (i - ND);
        const float* q0 = isd ? g.dk0 : g.xk0;
        const float* q1 = isd ? g.dk1 : g.xk1;
        const float* q2 = isd ? g.dk2 : g.xk2;
        const float a = (q0 ? q0 : g.dk0)[q0 ? idx : 0], b = (q1 ? q1 : g.dk0)[q1 ? idx : 0], c = (q2 ? q2 : g.dk0)[q2 ? idx : 0];
        k0[i] = q0 ? a : 1.f;
        k1[i] = q1 ? b : 0.f;
        k2[i] = q2 ? c : ((!isd && PRO_X == PRO_AFFINE_GELU) ? 1.f : 0.f);
    }

    u32x4_t dv[ND], dv2[ND], xv[NX];
    // consecutive blocks of a frame start 5 chunks further into their ranges and wrap (pw_wgrad_split.hip, WGS_ROT: un-rotated blocks,
    // whose ranges start a power of two apart, request the same low address bits at every moment)
    const int rot0 = nc > 0 ? (int)((unsigned)(blockIdx.x * 5 + n * 3) % (unsigned)nc) : 0;
    auto load_piece = [&](int i, int ch) {   // i compile-time after unrolling; ch clamped by the caller
        const int cr = ch + rot0 < nc ? ch + rot0 : ch + rot0 - nc;
        const size_t po = (size_t)(cbeg + cr) * 64;
        if (i < ND) {
            dv[i] = __builtin_nontemporal_load((const u32x4_t*)(dbase + (size_t)(64 * i) * P + po));
            dv2[i] = __builtin_nontemporal_load((const u32x4_t*)(d2base + (size_t)(64 * i) * P + po));
        } else {
            xv[i - ND] = __builtin_nontemporal_load((const u32x4_t*)(xbase + (size_t)(64 * (i - ND)) * P + po));
        }
    };
    auto stage_piece = [&](int i, int buf) {
        const bool isd = i < ND;
        const float c0 = k0[i], c1 = k1[i], c2 = k2[i];
        const u32x4_t v = isd ? dv[i] : xv[i - ND];
        const u32x4_t w = isd ? dv2[i] : v;
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a0 = bf16_lo(v[q]), a1 = bf16_hi(v[q]), b0 = bf16_lo(w[q]), b1 = bf16_hi(w[q]);
            const float cm = k3[isd ? i : 0];
            const float t0 = isd ? wga_pro<PRO_NORMBWD>(a0, b0, c0, c1, c2, cm) : wga_pro<PRO_X>(a0, b0, c0, c1, c2);
            const float t1 = isd ? wga_pro<PRO_NORMBWD>(a1, b1, c0, c1, c2, cm) : wga_pro<PRO_X>(a1, b1, c0, c1, c2);
            o[q] = cvt_pk_bf16(t0, t1);
        }
        const int rofs = isd ? 64 * i : COP + 64 * (i - ND);
        *(u32x4_t*)(st_base + buf * BUF + rofs * 16) = u32x4_t{o[0], o[1], o[2], o[3]};
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // operand addresses: plane = 2*ks + (lane>>5); A rows of this wave's two co tiles, B rows of its two ci tiles
    const int arow = (wco * 2) * 32 + (lane & 31), brow = COP + (wci * 2) * 32 + (lane & 31);
    const int aoff = (lane >> 5) * PS + arow * 16, boff = (lane >> 5) * PS + brow * 16;
    auto ldop = [&](int buf, int ks, int off, u32x4_t (&o)[2]) {
        const unsigned char* p = smem + buf * BUF + ks * 2 * PS + off;
        o[0] = *(const u32x4_t*)p;
        o[1] = *(const u32x4_t*)(p + 32 * 16);
    };

    // prologue: chunk 0 -> buffer 0, chunk 1 raw in registers
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) load_piece(i, 0);
#pragma unroll
    for (int i = 0; i < ND + NX; ++i) { stage_piece(i, 0); load_piece(i, nc > 1 ? 1 : 0); }
    __syncthreads();

#define WGA_MF(A, B)                                                                                             \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)                  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[a]),                  \
                                                            __builtin_bit_cast(bf16x8_t, B[b]), acc[a][b], 0, 0, 0)
    constexpr int NP = ND + NX;
    for (int c = 0; c < nc; ++c) {
        const int cur = c & 1;
        const int c2 = c + 2 < nc ? c + 2 : nc - 1;     // raw chunk to request (clamped, branch-free)
        u32x4_t af[2][2], bfr[2][2];
        ldop(cur, 0, aoff, af[0]); ldop(cur, 0, boff, bfr[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) { ldop(cur, ks + 1, aoff, af[(ks + 1) & 1]); ldop(cur, ks + 1, boff, bfr[(ks + 1) & 1]); }
            WGA_MF(af[ks & 1], bfr[ks & 1]);
            // the staging pieces of chunk c+1 are spread over the k-steps; each frees its raw registers for chunk c+2
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i * 4 / NP == ks) { stage_piece(i, cur ^ 1); load_piece(i, c2); }
        }
        __syncthreads();   // chunk c+1 staged by every wave; every wave is done reading buffer cur
    }
#undef WGA_MF

    float* po = g.part + ((size_t)n * gridDim.x + blockIdx.x) * COP * CIP;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wco * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = (wci * 2 + b) * 32 + (lane & 31);
                po[co * CIP + ci] = acc[a][b][r];
            }
}

static int wga_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
            ncu = 256;
    }
    return ncu;
}

// blocks per frame: one block per CU in total, never more than one block per 64-pixel chunk
int pw_wgrad_a16_nbx(int N, int P) {
    int g = wga_ncu() / N;
    if (g < 1) g = 1;
    if (g > P / 64) g = P / 64;
    return g;
}

// (Cd, Cx) = (256, 128) or (128, 256); (pro_d, pro_x) in {(NORMBWD, AFFINE), (NORMBWD, AFFINE_GELU)}
bool pw_wgrad_a16_supported(int Cd, int Cx, int pro_d, int pro_x, bool rowsum) {
    if (rowsum || pro_d != PRO_NORMBWD) return false;
    if (!((Cd == 256 && Cx == 128) || (Cd == 128 && Cx == 256))) return false;
    return pro_x == PRO_AFFINE || pro_x == PRO_AFFINE_GELU;
}

template <int WCO, int WCI, int PRO_X>
static int wga_launch(const WgaArgs& g, dim3 grid, hipStream_t stream) {
    constexpr int R = 64 * WCO + 64 * WCI;
    constexpr size_t lds = 2 * 8 * (size_t)(R * 16 + 16);
    auto kern = pw_wgrad_a16_kernel<WCO, WCI, PRO_X>;
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

int pw_wgrad_a16_launch(const void* d, const void* d2, const void* x, const float* dk0, const float* dk1, const float* dk2,
                        const float* dkmu, const float* xk0, const float* xk1, const float* xk2, float* part, int N, int Cd, int Cx,
                        int P, int nbx, int pro_x, hipStream_t stream) {
    if (P % 64 || nbx < 1 || nbx > P / 64 || !d2) return UNCR_ESHAPE;
    WgaArgs g{(const bf16_t*)d, (const bf16_t*)d2, (const bf16_t*)x, dk0, dk1, dk2, xk0, xk1, xk2, part, Cd, Cx, P, dkmu};
    dim3 grid(nbx, N);
    if (Cd == 256) {
        if (pro_x == PRO_AFFINE) return wga_launch<4, 2, PRO_AFFINE>(g, grid, stream);
        return wga_launch<4, 2, PRO_AFFINE_GELU>(g, grid, stream);
    }
    if (pro_x == PRO_AFFINE) return wga_launch<2, 4, PRO_AFFINE>(g, grid, stream);
    return wga_launch<2, 4, PRO_AFFINE_GELU>(g, grid, stream);
}
