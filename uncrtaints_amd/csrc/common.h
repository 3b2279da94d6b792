// Shared device helpers for libuncr_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UNCR_OK 0
#define UNCR_EINVAL (-1)
#define UNCR_ESHAPE (-2)

#define UNCR_LAUNCH_CHECK()                       \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// prologue kinds shared by the pointwise GEMM, the weight-gradient GEMM and the element-wise family
enum : int {
    PRO_NONE = 0,        // v
    PRO_AFFINE = 1,      // A*v + B
    PRO_AFFINE_GELU = 2, // S * gelu(A*v + B)        (S == 1 when no scale array is given)
    PRO_NORMBWD = 3,     // C1*v + C2*v2 + C3        (norm backward: v = d(norm out), v2 = raw norm input)
    PRO_AFFINE_RELU = 4  // max(A*v + B, 0)
};

enum : int { NORM_GROUP = 0, NORM_BATCH_TRAIN = 1, NORM_BATCH_EVAL = 2 };

// erf for the exact (erf-form) GELU of nn.GELU().  ocml's erff is a two-branch ~40-instruction routine and made
// every GELU-carrying streaming kernel VALU-bound (3.3 TB/s vs 5.7 TB/s without it).  This is a branch-free
// fit  erf(t) = 1 - 2^(-t*Q(t)),  t = min(|x|, 4),  Q of degree 8 (weighted least squares + Lawson iterations
// against scipy's fp64 erfc, tools/fit_erf.py): the fit itself is good to 2.2e-9, so what remains in fp32 arithmetic
// (max abs error 8.5e-8, the rounding floor of values near 1) is rounding noise, not a smooth bias -- a degree-7 fit
// (1.6e-8 systematic) was enough to push ill-conditioned gradient sums over the parity tolerance.  8 FMAs + one
// v_exp_f32.  -DUNCR_EXACT_ERF restores erff.
__device__ __forceinline__ float erf_f(float x) {
#ifdef UNCR_EXACT_ERF
    return erff(x);
#else
    const float t = fminf(fabsf(x), 4.0f);
    float q = -1.16047604024061e-05f;
    q = fmaf(q, t, 0.00015296389756258577f);
    q = fmaf(q, t, -0.0008482325938530266f);
    q = fmaf(q, t, 0.002274781931191683f);
    q = fmaf(q, t, -8.480128599330783e-05f);
    q = fmaf(q, t, -0.027724478393793106f);
    q = fmaf(q, t, 0.1483079046010971f);
    q = fmaf(q, t, 0.9184429049491882f);
    q = fmaf(q, t, 1.6279072761535645f);
    const float e = 1.0f - __builtin_amdgcn_exp2f(-t * q);
    return copysignf(e, x);
#endif
}
// Round 4: the tail of GELU / GELU' re-associated around the SAME erf fit -- u Phi(u) = u/2 + |u/2| (1 - 2^(-tQ)) and
// Phi(u) = 1/2 + copysign(1/2, u) (1 - 2^(-tQ)): no sign transfer onto the erf, no "1 + erf" (which cancels for u < 0), two / one VALU
// instructions fewer per element.  Every big kernel of the step runs at the socket power cap, where instructions per element are time:
// fp32 step -0.08 ms (3 of 4 interleaved pairs), bf16-storage step -0.135 ms (3 of 3).  Accuracy against fp64 over 4 M arguments
// (tools/probe_gelu_diet.py): gelu max abs 4.5e-7 -> 3.1e-7, rms 6.1e-8 -> 5.1e-8; gelu' unchanged (1.4e-7 / 2.9e-8).
// -DUNCR_GELU_DIET=0 restores 0.5 u (1 + erf).
#ifndef UNCR_GELU_DIET
#define UNCR_GELU_DIET 1
#endif
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
// 1 - |erf(x)| = 2^(-t Q(t)): what the fit computes before the sign is put back
__device__ __forceinline__ float erfc_abs_f(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float q = -1.16047604024061e-05f;
    q = fmaf(q, t, 0.00015296389756258577f);
    q = fmaf(q, t, -0.0008482325938530266f);
    q = fmaf(q, t, 0.002274781931191683f);
    q = fmaf(q, t, -8.480128599330783e-05f);
    q = fmaf(q, t, -0.027724478393793106f);
    q = fmaf(q, t, 0.1483079046010971f);
    q = fmaf(q, t, 0.9184429049491882f);
    q = fmaf(q, t, 1.6279072761535645f);
    return __builtin_amdgcn_exp2f(-t * q);
}
#endif
__device__ __forceinline__ float gelu_f(float u) {
    // exact (erf) GELU, as nn.GELU() default
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
    // u Phi(u) = u/2 + |u/2| |erf|: no sign transfer, no 1 + erf (for u < 0 this is (u/2) * (1 - |erf|) without the cancellation)
    const float hu = 0.5f * u;
    return fmaf(fabsf(hu), 1.0f - erfc_abs_f(u * 0.70710678118654752440f), hu);
#else
    return 0.5f * u * (1.0f + erf_f(u * 0.70710678118654752440f));
#endif
}
__device__ __forceinline__ float gelu_grad_f(float u) {
    // d/du [u * Phi(u)] = Phi(u) + u * phi(u);  phi(u) = exp(-u^2/2)/sqrt(2 pi) = 2^(-u^2 * log2(e)/2)/sqrt(2 pi)
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
    const float cdf = fmaf(copysignf(0.5f, u), 1.0f - erfc_abs_f(u * 0.70710678118654752440f), 0.5f);
#else
    const float cdf = 0.5f * (1.0f + erf_f(u * 0.70710678118654752440f));
#endif
#ifdef UNCR_EXACT_EXP
    const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
#else
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
#endif
    return cdf + u * pdf;
}
// Two-wide forms on the packed fp32 VALU ops of gfx950 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 results per lane
// per issue; per component the same operations in the same order as the scalar forms above, so the values are identical).  The
// GELU-carrying prologues / epilogues of the streaming kernels were issue-bound on VALU (pw2 forward: 33 VALU instructions per
// element, 19 of them the GELU; profiles/r02_pipes.json): the polynomial and the affine around it run packed, only min, v_exp_f32
// and the sign transfer stay per component.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 f2(float a, float b) { f32x2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f32x2 f2(float a) { f32x2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 erf_f2(f32x2 x) {
#ifdef UNCR_EXACT_ERF
    return f2(erff(x.x), erff(x.y));
#else
    const f32x2 t = f2(fminf(fabsf(x.x), 4.0f), fminf(fabsf(x.y), 4.0f));
    f32x2 q = f2(-1.16047604024061e-05f);
    q = fma2(q, t, f2(0.00015296389756258577f));
    q = fma2(q, t, f2(-0.0008482325938530266f));
    q = fma2(q, t, f2(0.002274781931191683f));
    q = fma2(q, t, f2(-8.480128599330783e-05f));
    q = fma2(q, t, f2(-0.027724478393793106f));
    q = fma2(q, t, f2(0.1483079046010971f));
    q = fma2(q, t, f2(0.9184429049491882f));
    q = fma2(q, t, f2(1.6279072761535645f));
    const f32x2 a = -t * q;
    return f2(copysignf(1.0f - __builtin_amdgcn_exp2f(a.x), x.x), copysignf(1.0f - __builtin_amdgcn_exp2f(a.y), x.y));
#endif
}
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
__device__ __forceinline__ f32x2 erfc_abs_f2(f32x2 x) {       // per component: 1 - |erf|, the same operations as erfc_abs_f
    const f32x2 t = f2(fminf(fabsf(x.x), 4.0f), fminf(fabsf(x.y), 4.0f));
    f32x2 q = f2(-1.16047604024061e-05f);
    q = fma2(q, t, f2(0.00015296389756258577f));
    q = fma2(q, t, f2(-0.0008482325938530266f));
    q = fma2(q, t, f2(0.002274781931191683f));
    q = fma2(q, t, f2(-8.480128599330783e-05f));
    q = fma2(q, t, f2(-0.027724478393793106f));
    q = fma2(q, t, f2(0.1483079046010971f));
    q = fma2(q, t, f2(0.9184429049491882f));
    q = fma2(q, t, f2(1.6279072761535645f));
    const f32x2 a = -t * q;
    return f2(__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y));
}
#endif
__device__ __forceinline__ f32x2 gelu_f2(f32x2 u) {
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
    const f32x2 hu = f2(0.5f) * u;
    return fma2(f2(fabsf(hu.x), fabsf(hu.y)), f2(1.0f) - erfc_abs_f2(u * f2(0.70710678118654752440f)), hu);
#else
    return f2(0.5f) * u * (f2(1.0f) + erf_f2(u * f2(0.70710678118654752440f)));
#endif
}
__device__ __forceinline__ f32x2 gelu_grad_f2(f32x2 u) {
#if UNCR_GELU_DIET && !defined(UNCR_EXACT_ERF)
    const f32x2 cdf = fma2(f2(copysignf(0.5f, u.x), copysignf(0.5f, u.y)), f2(1.0f) - erfc_abs_f2(u * f2(0.70710678118654752440f)), f2(0.5f));
#else
    const f32x2 cdf = f2(0.5f) * (f2(1.0f) + erf_f2(u * f2(0.70710678118654752440f)));
#endif
#ifdef UNCR_EXACT_EXP
    const f32x2 pdf = f2(0.39894228040143267794f * expf(-0.5f * u.x * u.x), 0.39894228040143267794f * expf(-0.5f * u.y * u.y));
#else
    const f32x2 a = f2(-0.72134752044448170368f) * u * u;
    const f32x2 pdf = f2(0.39894228040143267794f) * f2(__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y));
#endif
    return fma2(u, pdf, cdf);
}
// gelu(A*h + B) / gelu'(A*h + B) on the four values of a float4
__device__ __forceinline__ float4 gelu_affine4(float A, float B, const float4& h) {
    const f32x2 a = gelu_f2(fma2(f2(A), f2(h.x, h.y), f2(B))), b = gelu_f2(fma2(f2(A), f2(h.z, h.w), f2(B)));
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float4 gelu_grad_affine4(float A, float B, const float4& h) {
    const f32x2 a = gelu_grad_f2(fma2(f2(A), f2(h.x, h.y), f2(B))), b = gelu_grad_f2(fma2(f2(A), f2(h.z, h.w), f2(B)));
    return make_float4(a.x, a.y, b.x, b.y);
}

// Used by the output head (26 channels) and the squeeze-excite MLP only -- never on a bandwidth-critical stream -- so the accurate
// expf.  The head needs it: a random-init MGNLL is dominated by the few pixels whose variance sits at the 1e-8 clamp, and there
// the residual mean - target is a difference of O(1) numbers: the fast __expf's ~3e-7 relative error in the sigmoid reached every
// gradient of such a model as ~1e-4 (tools/probe_grad_noise.py).
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// sum over the 64 lanes of a wave (all lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// 16-byte accesses for tensors that are streamed once.  A translation unit opts in to the non-temporal form with
// `#define UNCR_NT 1` before this header.  Measured per kernel family inside the training step (tools/ab_variants.sh,
// one GPU session, 2 x 20 steps each): depthwise row kernels -0.30 ms/step (dw_bwd 0.22 -> 0.18-0.19 ms: the three input
// streams no longer evict each other) -- ON; split GEMM activations/outputs +0.17 ms -- OFF; weight-gradient GEMM,
// element-wise, aggregate and residual+pool kernels within +-0.06 ms (noise) -- OFF.
#ifndef UNCR_NT
#define UNCR_NT 0
#endif
typedef float uncr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt4(const float* p) {
#if UNCR_NT
    const uncr_f4 v = __builtin_nontemporal_load((const uncr_f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *(const float4*)p;
#endif
}
__device__ __forceinline__ void st_nt4(float* p, const float4& v) {
#if UNCR_NT
    const uncr_f4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, (uncr_f4*)p);
#else
    *(float4*)p = v;
#endif
}

// ---- activation storage types -------------------------------------------------------------------------------------
// Every [frames][channels][pixels] activation / activation-gradient tensor is stored as fp32 (float) or as bf16 (bf16_t:
// "bf16 activations, fp32 accumulate", BASELINE config 3).  Kernels are templated on the storage type and always compute
// in fp32: ld4<T> widens four consecutive elements, st4<T> narrows them (round-to-nearest-even, v_cvt_pk_bf16_f32), and
// rnd4<T> returns the values AS THEY WILL BE STORED -- producers take their (sum, sum^2) statistics from those, so a
// normalisation is applied to exactly the tensor its statistics describe.
typedef unsigned short bf16_t;
typedef __bf16 uncr_bf2 __attribute__((ext_vector_type(2)));
typedef float uncr_f2 __attribute__((ext_vector_type(2)));
#define UNCR_F32 0
#define UNCR_BF16 1
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {       // {lo16 = bf16(a), hi16 = bf16(b)}, RNE
    return __builtin_bit_cast(unsigned, __builtin_convertvector(uncr_f2{a, b}, uncr_bf2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_round(float a) { return bf16_lo(cvt_pk_bf16(a, 0.f)); }
typedef unsigned uncr_u2 __attribute__((ext_vector_type(2)));
template <typename T, bool NT = false> __device__ __forceinline__ float4 ld4(const T* p) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (NT) { const uncr_f4 v = __builtin_nontemporal_load((const uncr_f4*)p); return make_float4(v.x, v.y, v.z, v.w); }
        else return *(const float4*)p;
    } else {
        uncr_u2 r;
        if constexpr (NT) r = __builtin_nontemporal_load((const uncr_u2*)p);
        else r = *(const uncr_u2*)p;
        return make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
    }
}
template <typename T, bool NT = false> __device__ __forceinline__ void st4(T* p, const float4& v) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (NT) { const uncr_f4 q = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(q, (uncr_f4*)p); }
        else *(float4*)p = v;
    } else {
        const uncr_u2 q = {cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
        if constexpr (NT) __builtin_nontemporal_store(q, (uncr_u2*)p);
        else *(uncr_u2*)p = q;
    }
}
template <typename T> __device__ __forceinline__ float4 rnd4(const float4& v) {
    if constexpr (sizeof(T) == 4) return v;
    else {
        const unsigned a = cvt_pk_bf16(v.x, v.y), b = cvt_pk_bf16(v.z, v.w);
        return make_float4(bf16_lo(a), bf16_hi(a), bf16_lo(b), bf16_hi(b));
    }
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) {
    if constexpr (sizeof(T) == 4) return *p;
    else return __uint_as_float((unsigned)(*p) << 16);
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v) {
    if constexpr (sizeof(T) == 4) *p = v;
    else *p = (bf16_t)(cvt_pk_bf16(v, 0.f) & 0xFFFFu);
}
// raw (not yet widened) four elements: prefetch rings keep these, so that a bf16 ring holds twice the rows in the same registers
// and the widening shifts sit at the point of use, not behind the load
template <typename T> struct raw4 { typedef float4 type; };
template <> struct raw4<bf16_t> { typedef uncr_u2 type; };
template <typename T, bool NT = false> __device__ __forceinline__ typename raw4<T>::type ld4raw(const T* p) {
    if constexpr (sizeof(T) == 4) return ld4<T, NT>(p);
    else {
        if constexpr (NT) return __builtin_nontemporal_load((const uncr_u2*)p);
        else return *(const uncr_u2*)p;
    }
}
__device__ __forceinline__ float4 widen4(const float4& r) { return r; }
__device__ __forceinline__ float4 widen4(const uncr_u2& r) { return make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y)); }
// the streamed-once forms follow the translation unit's UNCR_NT switch like ld_nt4 / st_nt4
template <typename T> __device__ __forceinline__ float4 ld_nt4t(const T* p) { return ld4<T, (UNCR_NT != 0)>(p); }
template <typename T> __device__ __forceinline__ void st_nt4t(T* p, const float4& v) { st4<T, (UNCR_NT != 0)>(p, v); }
// host-side dispatch on the storage code of the C ABI (UNCR_F32 / UNCR_BF16)
#define UNCR_DISPATCH_ACT(code, T, ...)                                   \
    do {                                                                  \
        if ((code) == UNCR_BF16) { using T = bf16_t; __VA_ARGS__; }       \
        else { using T = float; __VA_ARGS__; }                            \
    } while (0)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// sum over each 32-lane half of a wave (lanes 0-31 and 32-63 separately)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// DPP (data-parallel primitive) adds: full-rate VALU cross-lane moves inside a row of 16 lanes, no LDS
// crossbar (ds_bpermute) round trip.  dpp_ctrl: quad_perm 0x00-0xFF, row_ror:n = 0x120+n, row_bcast15 = 0x142,
// row_bcast31 = 0x143.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                                 BANK_MASK, BOUND));
}
// sum over each 32-lane half; the result is valid in lane 31 (lanes 0-31) and lane 63 (lanes 32-63) ONLY
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_mov<0x124>(v);                // row_ror:4
    v += dpp_mov<0x128>(v);                // row_ror:8  -> every lane of a row holds the row sum
    v += dpp_mov<0x142, 0xA, 0xF, false>(v);   // row_bcast15 into rows 1 and 3: lane 31 / 63 = half sums
    return v;
}

// sum over the whole wave; the result is valid in lane 63 ONLY
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = half_wave_sum_dpp(v);
    v += dpp_mov<0x143, 0xC, 0xF, false>(v);   // row_bcast31 into rows 2 and 3: lane 63 = wave sum
    return v;
}

// max over the whole wave of NON-NEGATIVE values (the update_dpp fill for lanes without a source is 0); valid in lane 63 ONLY
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    v = fmaxf(v, dpp_mov<0x142, 0xA, 0xF, false>(v));
    v = fmaxf(v, dpp_mov<0x143, 0xC, 0xF, false>(v));
    return v;
}

// Block-wide sum of two floats; result valid in thread 0.  `red` must hold >= 2*nwaves floats.
template <int NTHREADS>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    constexpr int NW = NTHREADS / 64;
    a = wave_sum_dpp(a);
    b = wave_sum_dpp(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) { sa += red[2 * i]; sb += red[2 * i + 1]; }
        a = sa; b = sb;
    }
    __syncthreads();
}

// reflect index for padding 1 (PyTorch 'reflect': -1 -> 1, n -> n-2)
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// counter-based RNG for the attention dropout: 32-bit mix of (seed, index); uniform in [0,1)
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)((uint32_t)(z >> 40)) * (1.0f / 16777216.0f);
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
