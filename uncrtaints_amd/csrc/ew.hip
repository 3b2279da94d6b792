// Element-wise family over NCHW planes with fused per-(n,c) coefficient application and
// per-block (sum, sum*) partial statistics.  All of these are HBM-streaming kernels: one block =
// one (n,c) plane chunk of EW_CHUNK pixels, 16-byte loads/stores, stats reduced with wave shuffles.
//
// grid = (P / EW_CHUNK, N*C); partials are written to part[(n*C+c)*NP + blockIdx.x], NP = P/EW_CHUNK.
#include "common.h"

#define EW_CHUNK 1024   // pixels per block (256 threads x float4)

enum : int {
    EW_STATS_SQ = 0,     // stats only: (sum a, sum a^2)
    EW_STATS_AUX = 1,    // stats only: (sum a, sum a*b)
    EW_AFFINE_RELU = 2,  // out = relu(A*a + B);              stats (sum out, sum out^2)
    EW_RESIDUAL = 3,     // out = a + A*b + B;                stats (sum out, sum out^2)   a=x, b=h3
    EW_PASSB = 4,        // out = gelu'(A*b + B) * (S*a + D); stats (sum out, sum out*b)   a=dz, b=h2
    EW_PASSE = 5,        // out = a + C1*b + C2*(c - M) + C3; stats (sum out, sum out*aux) a=dy, b=da, c=x (M = k3 or 0)
    EW_RELU_BWD = 6,     // out = a * [A*b + B > 0];          stats (sum out, sum out*(b - M))   a=d(a0), b=c0, M = k2 or 0
    EW_SE_POOL = 7,      // stats only: (sum gelu(A*a + B), 0)
    EW_HEAD_FWD = 8,     // out = c < n_mean ? scale*sigmoid(a) : softplus(a)+eps   (per-plane channel test)
    EW_HEAD_BWD = 9,     // out = a * f'(b)   a = d(out), b = pre-activation
    EW_RESIDUAL_RELU = 10,  // out = a + relu(A*b + B)   (ResidualConvBlock skip, uncrtaints.py:67)
    // the other variance nonlinearities of get_nonlinearity (uncrtaints.py:223-228): 'elu' -> elu(a)+1+eps, else identity
    EW_HEAD_FWD_ELU = 11, EW_HEAD_BWD_ELU = 12, EW_HEAD_FWD_ID = 13, EW_HEAD_BWD_ID = 14,
    // stand-alone calls of the norm layers / SE (a caller using PreNorm or SE outside MBConv; inside it they are prologues)
    EW_AFFINE = 15,      // out = A*a + B;                    stats (sum out, sum out^2)
    EW_NORMBWD = 16,     // out = C1*a + C2*(b - M) + C3;     (M = k3 or 0)   a=dy, b=x
    EW_SE_POOL4 = 17     // EW_SE_POOL with four chunks of a plane per block (se_pool4_kernel): same slots, same values
};
__host__ __device__ constexpr bool ew_is_head_fwd(int op) { return op == EW_HEAD_FWD || op == EW_HEAD_FWD_ELU || op == EW_HEAD_FWD_ID; }
__host__ __device__ constexpr bool ew_is_head_bwd(int op) { return op == EW_HEAD_BWD || op == EW_HEAD_BWD_ELU || op == EW_HEAD_BWD_ID; }
__host__ __device__ constexpr int ew_var_mode(int op) {      // 0 softplus, 1 elu + 1, 2 identity
    return (op == EW_HEAD_FWD_ELU || op == EW_HEAD_BWD_ELU) ? 1 : ((op == EW_HEAD_FWD_ID || op == EW_HEAD_BWD_ID) ? 2 : 0);
}

struct EwArgs {
    const void* a;       // activation tensors: fp32 or bf16 (the kernel's storage type T)
    const void* b;
    const void* c;
    const void* aux;     // optional second stats operand for PASSE
    void* out;
    const float* k0;     // per-plane coefficient arrays [N*C] (meaning depends on op)
    const float* k1;
    const float* k2;
    const float* k3;
    float2* part;        // optional [N*C][NP]
    int P;
    int Pv;              // pixels of a plane that carry data (P: all of them).  Pv < P = padded planes of an any-size image (anysize.hip):
                         // pixels Pv .. P-1 hold zeros on input and are written as zeros; the statistics run over the first Pv pixels
    int C;               // channels per frame (HEAD ops)
    int n_mean;          // HEAD: number of mean channels
    float scale;         // HEAD: scale_by
    float eps;           // HEAD: variance epsilon
};

template <int OP, typename T, typename TO = T>      // T: storage of the inputs, TO: of the output (differs for HEAD_BWD only)
__global__ __launch_bounds__(256) void ew_kernel(EwArgs g) {
    const int plane = blockIdx.y;
    const size_t off = (size_t)plane * g.P + (size_t)blockIdx.x * EW_CHUNK + threadIdx.x * 4;
    float s0 = 0.f, s1 = 0.f;
    float4 va = ld_nt4t((const T*)g.a + off);
    float4 vo;
    float* o = (float*)&vo;
    const float* pa = (const float*)&va;
    // padded planes: this thread's pixels from `nval` on belong to the zero tail (kernel-uniform test first: the dense sizes pay one
    // scalar compare).  A point-wise result f(0) there is put back to zero BEFORE the statistics are taken from it.
    const bool padded = g.Pv < g.P;
    const int nval = g.Pv - (int)(blockIdx.x * EW_CHUNK + threadIdx.x * 4);
#define EW_ZERO_TAIL()                                                         \
    if (padded && nval < 4) {                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) o[i] = i < nval ? o[i] : 0.f; \
    }
    if constexpr (OP == EW_STATS_SQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += pa[i]; s1 += pa[i] * pa[i]; }
    } else if constexpr (OP == EW_STATS_AUX) {
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += pa[i]; s1 += pa[i] * pb[i]; }
    } else if constexpr (OP == EW_AFFINE_RELU) {
        const float A = g.k0[plane], B = g.k1[plane];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaf(A, pa[i], B), 0.f);
        EW_ZERO_TAIL()
        vo = rnd4<T>(vo);       // statistics of the values as stored
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += o[i]; s1 += o[i] * o[i]; }
    } else if constexpr (OP == EW_AFFINE) {
        const float A = g.k0[plane], B = g.k1[plane];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaf(A, pa[i], B);
        EW_ZERO_TAIL()
        vo = rnd4<T>(vo);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += o[i]; s1 += o[i] * o[i]; }
    } else if constexpr (OP == EW_NORMBWD) {
        const float C1 = g.k0[plane], C2 = g.k1[plane], C3 = g.k2[plane];
        const float M = g.k3 ? g.k3[plane] : 0.f;
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaf(C1, pa[i], fmaf(C2, pb[i] - M, C3));
        EW_ZERO_TAIL()
    } else if constexpr (OP == EW_RESIDUAL) {
        const float A = g.k0[plane], B = g.k1[plane];
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pa[i] + fmaf(A, pb[i], B);
        EW_ZERO_TAIL()
        vo = rnd4<T>(vo);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += o[i]; s1 += o[i] * o[i]; }
    } else if constexpr (OP == EW_RESIDUAL_RELU) {
        const float A = g.k0[plane], B = g.k1[plane];
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pa[i] + fmaxf(fmaf(A, pb[i], B), 0.f);
        EW_ZERO_TAIL()
    } else if constexpr (OP == EW_PASSB) {
        const float A = g.k0[plane], B = g.k1[plane], S = g.k2[plane], D = g.k3[plane];
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const f32x2 r = gelu_grad_f2(fma2(f2(A), f2(pb[i], pb[i + 1]), f2(B))) * fma2(f2(S), f2(pa[i], pa[i + 1]), f2(D));
            o[i] = r.x; o[i + 1] = r.y;
        }
        EW_ZERO_TAIL()
        vo = rnd4<T>(vo);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s0 += o[i]; s1 += o[i] * pb[i]; }
    } else if constexpr (OP == EW_PASSE) {
        const float C1 = g.k0[plane], C2 = g.k1[plane], C3 = g.k2[plane];
        const float M = g.k3 ? g.k3[plane] : 0.f;      // centred norm backward
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float4 vc = ld_nt4t((const T*)g.c + off);
        const float* pb = (const float*)&vb;
        const float* pc = (const float*)&vc;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pa[i] + fmaf(C1, pb[i], fmaf(C2, pc[i] - M, C3));
        EW_ZERO_TAIL()
        vo = rnd4<T>(vo);
        if (g.part) {
            const float4 vx = ld_nt4t((const T*)g.aux + off);
            const float* px = (const float*)&vx;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0 += o[i]; s1 += o[i] * px[i]; }
        }
    } else if constexpr (OP == EW_RELU_BWD) {
        const float A = g.k0[plane], B = g.k1[plane];
        const float M = g.k2 ? g.k2[plane] : 0.f;      // pivot of the second statistic (the norm's mean: centred norm backward)
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = fmaf(A, pb[i], B) > 0.f ? pa[i] : 0.f;      // a is already a stored value: nothing to round
            s0 += o[i]; s1 += o[i] * (pb[i] - M);
        }
    } else if constexpr (OP == EW_SE_POOL) {
        const float A = g.k0[plane], B = g.k1[plane];
        // polynomial and affine on the packed ops; the accumulation keeps the fused form s0 = fma(u/2, 1 + erf, s0) (one rounding
        // per term)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const f32x2 u = fma2(f2(A), f2(pa[i], pa[i + 1]), f2(B));
            const f32x2 hu = f2(0.5f) * u, pe = f2(1.0f) + erf_f2(u * f2(0.70710678118654752440f));
            if (padded && nval < 4) {         // a tail pixel would add gelu(B)
                if (i < nval) s0 = fmaf(hu.x, pe.x, s0);
                if (i + 1 < nval) s0 = fmaf(hu.y, pe.y, s0);
            } else {
                s0 = fmaf(hu.x, pe.x, s0);
                s0 = fmaf(hu.y, pe.y, s0);
            }
        }
    } else if constexpr (ew_is_head_fwd(OP)) {
        // n_mean > 0: first n_mean channels get scale*sigmoid; n_mean < 0: first |n_mean| channels identity
        const int ch = plane % g.C;
        const int nm = g.n_mean < 0 ? -g.n_mean : g.n_mean;
        if (ch < nm) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = g.n_mean > 0 ? g.scale * sigmoid_f(pa[i]) : pa[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = pa[i];
                if constexpr (ew_var_mode(OP) == 0) o[i] = (x > 20.f ? x : log1pf(expf(x))) + g.eps;   // nn.Softplus(beta=1, threshold=20)
                else if constexpr (ew_var_mode(OP) == 1) o[i] = (x > 0.f ? x : expm1f(x)) + 1.f + g.eps;  // nn.ELU() + 1 + eps
                else o[i] = x;                                                                            // nn.Identity()
            }
        }
    } else if constexpr (ew_is_head_bwd(OP)) {
        // C > 0: b holds the pre-activation; C < 0: b holds the head's OUTPUT (fused forward, uncr_head_fwd) and the
        // derivative is recovered from it: sigmoid: s(1-s) with s = out/scale; softplus: 1 - exp(-(out - eps));
        // elu + 1: out - eps for out - eps <= 1, else 1
        const bool from_out = g.C < 0;
        const int Cc = from_out ? -g.C : g.C;
        const int ch = plane % Cc;
        const float4 vb = ld_nt4t((const T*)g.b + off);
        const float* pb = (const float*)&vb;
        const int nm = g.n_mean < 0 ? -g.n_mean : g.n_mean;
        if (ch < nm) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sg = from_out ? pb[i] / g.scale : sigmoid_f(pb[i]);
                o[i] = g.n_mean > 0 ? pa[i] * g.scale * sg * (1.f - sg) : pa[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ew_var_mode(OP) == 0) {
                    const float d = from_out ? -expm1f(-(pb[i] - g.eps)) : (pb[i] > 20.f ? 1.f : sigmoid_f(pb[i]));
                    o[i] = pa[i] * d;
                } else if constexpr (ew_var_mode(OP) == 1) {
                    const float d = from_out ? fminf(pb[i] - g.eps, 1.f) : (pb[i] > 0.f ? 1.f : expf(pb[i]));
                    o[i] = pa[i] * d;
                } else o[i] = pa[i];
            }
        }
    }
    if constexpr (OP != EW_STATS_SQ && OP != EW_STATS_AUX && OP != EW_SE_POOL) st_nt4t((TO*)g.out + off, vo);
    if constexpr (!ew_is_head_fwd(OP) && !ew_is_head_bwd(OP)) {
        if (g.part) {
            __shared__ float red[8];
            block_sum2<256>(s0, s1, red);
            if (threadIdx.x == 0) g.part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
        }
    }
}

// EW_SE_POOL with FOUR chunks of one plane per block: every lane has its four loads in flight before the first GELU.  The per-chunk
// arithmetic, the wave sums and the four-wave combination are those of ew_kernel<EW_SE_POOL> / block_sum2 and the partial slots are
// the same ones, so the results are bit-identical.  Measured (tools/bench_sepool.py, operand rewritten by a copy kernel before each
// call as in the step): bf16 storage 40.6 -> 34.2 us at N = 4 and 120 -> 96 us at N = 12; fp32 storage 57 -> 61 us and 149 -> 157 us
// (SLOWER: the engine uses this kernel for bf16 storage only).  grid = (P / (4 * EW_CHUNK), planes).
template <typename T>
__global__ __launch_bounds__(256) void se_pool4_kernel(EwArgs g) {
    const int plane = blockIdx.y;
    const size_t off = (size_t)plane * g.P + (size_t)blockIdx.x * (4 * EW_CHUNK) + threadIdx.x * 4;
    float4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = ld_nt4t((const T*)g.a + off + (size_t)c * EW_CHUNK);
    const float A = g.k0[plane], B = g.k1[plane];
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* pa = (const float*)&v[c];
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const f32x2 u = fma2(f2(A), f2(pa[i], pa[i + 1]), f2(B));
            const f32x2 hu = f2(0.5f) * u, pe = f2(1.0f) + erf_f2(u * f2(0.70710678118654752440f));
            s0 = fmaf(hu.x, pe.x, s0);
            s0 = fmaf(hu.y, pe.y, s0);
        }
        s0 = wave_sum_dpp(s0);
        if (lane == 63) red[c][w] = s0;
    }
    __syncthreads();
    if (threadIdx.x < 4 && g.part) {
        float sa = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sa += red[threadIdx.x][i];
        g.part[(size_t)plane * (gridDim.x * 4) + blockIdx.x * 4 + threadIdx.x] = make_float2(sa, 0.f);
    }
}

// dst = src converted between storage types (the model input -> bf16 activations; a bf16 input gradient -> fp32)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) st4<TD>(dst + i, ld4<TS>(src + i));
    else
        for (long long j = i; j < n; ++j) st1<TD>(dst + j, ld1<TS>(src + j));
}
extern "C" int uncr_cast(const void* src, void* dst, long long n, int src_dt, int dst_dt, hipStream_t stream) {
    if (n <= 0 || !src || !dst) return UNCR_EINVAL;
    const dim3 grid((unsigned)((n + 1023) / 1024)), blk(256);
    if (src_dt == UNCR_F32 && dst_dt == UNCR_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, blk, 0, stream, (const float*)src, (bf16_t*)dst, n);
    else if (src_dt == UNCR_BF16 && dst_dt == UNCR_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, blk, 0, stream, (const bf16_t*)src, (float*)dst, n);
    else return UNCR_EINVAL;
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ew_slots(int P) { return P / EW_CHUNK; }

// per-plane totals of a [planes][slots] (sum0, sum1) partial array, fp64 accumulation in a fixed order
__global__ __launch_bounds__(256) void part_sums_kernel(const float2* __restrict__ part, int slots, int planes,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= planes) return;
    double a = 0.0, b = 0.0;
    for (int j = 0; j < slots; ++j) { const float2 v = part[(size_t)pl * slots + j]; a += (double)v.x; b += (double)v.y; }
    if (out0) out0[pl] = (float)a;
    if (out1) out1[pl] = (float)b;
}
extern "C" int uncr_part_sums(const float* part, int slots, int planes, float* out0, float* out1, hipStream_t stream) {
    if (!part || slots <= 0 || planes <= 0 || (!out0 && !out1)) return UNCR_EINVAL;
    hipLaunchKernelGGL(part_sums_kernel, dim3((planes + 255) / 256), dim3(256), 0, stream, (const float2*)part, slots, planes, out0, out1);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ew(int op, const void* a, const void* b, const void* c, const void* aux, void* out,
                       const float* k0, const float* k1, const float* k2, const float* k3, float* part, int planes,
                       int P, int C, int n_mean, float scale, float eps, int act, int Pv, hipStream_t stream) {
    if (planes <= 0 || P <= 0 || (P % EW_CHUNK) != 0 || Pv <= 0 || Pv > P) return UNCR_ESHAPE;
    if (Pv < P && (act != UNCR_F32 || op == EW_SE_POOL4)) return UNCR_EINVAL;      // padded planes: fp32 storage, one chunk per block
    if (!a || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    // HEAD_FWD* / RESIDUAL_RELU exist for fp32 storage only; HEAD_BWD*: a, b are fp32 (model output side), `act` is the storage
    // of the OUTPUT (the gradient w.r.t. the head's pre-activation, an activation gradient)
    if (act == UNCR_BF16 && (ew_is_head_fwd(op) || op == EW_RESIDUAL_RELU)) return UNCR_EINVAL;
    EwArgs g{a, b, c, aux, out, k0, k1, k2, k3, (float2*)part, P, Pv, C, n_mean, scale, eps};
    dim3 grid(P / EW_CHUNK, planes), blk(256);
#define EW_CASE(OPV)                                                                                       \
    case OPV:                                                                                              \
        hipLaunchKernelGGL((ew_kernel<OPV, float>), grid, blk, 0, stream, g);                              \
        break;
#define EW_CASE_HB(OPV)     /* head backward: fp32 inputs, output in the activation storage */                \
    case OPV:                                                                                              \
        if (act == UNCR_BF16) hipLaunchKernelGGL((ew_kernel<OPV, float, bf16_t>), grid, blk, 0, stream, g); \
        else hipLaunchKernelGGL((ew_kernel<OPV, float, float>), grid, blk, 0, stream, g);                  \
        break;
#define EW_CASE_A(OPV)      /* ops on activation tensors: fp32 and bf16 storage */                         \
    case OPV:                                                                                              \
        UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL((ew_kernel<OPV, T>), grid, blk, 0, stream, g));       \
        break;
    if (op == EW_SE_POOL4) {
        if ((P / EW_CHUNK) % 4 != 0) return UNCR_ESHAPE;
        const dim3 grid4(P / (4 * EW_CHUNK), planes);
        UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL((se_pool4_kernel<T>), grid4, blk, 0, stream, g));
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
    switch (op) {
        EW_CASE_A(EW_STATS_SQ)
        EW_CASE_A(EW_STATS_AUX)
        EW_CASE_A(EW_AFFINE_RELU)
        EW_CASE_A(EW_RESIDUAL)
        EW_CASE_A(EW_PASSB)
        EW_CASE_A(EW_PASSE)
        EW_CASE_A(EW_RELU_BWD)
        EW_CASE_A(EW_SE_POOL)
        EW_CASE_A(EW_AFFINE)
        EW_CASE_A(EW_NORMBWD)
        EW_CASE(EW_HEAD_FWD)
        EW_CASE_HB(EW_HEAD_BWD)
        EW_CASE(EW_RESIDUAL_RELU)
        EW_CASE(EW_HEAD_FWD_ELU)
        EW_CASE_HB(EW_HEAD_BWD_ELU)
        EW_CASE(EW_HEAD_FWD_ID)
        EW_CASE_HB(EW_HEAD_BWD_ID)
        default:
            return UNCR_EINVAL;
    }
#undef EW_CASE
#undef EW_CASE_A
#undef EW_CASE_HB
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
