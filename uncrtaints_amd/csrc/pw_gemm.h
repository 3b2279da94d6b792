// Shared between the two pointwise-GEMM translation units (pw_gemm.hip: fp32-MFMA kernels, weight gradient;
// pw_gemm_split.hip: the bf16x3-split kernels for the wide shapes).
#pragma once
#include "common.h"

struct PwArgs {
    // activation tensors (in, in2, out, aux, aux2, aux3) are fp32 or bf16: the kernels are templated on the storage type
    const void* in;
    const void* in2;     // PRO_NORMBWD second operand
    const float* Wt;     // packed weights (format depends on the kernel variant, see uncr_pack_wt)
    void* out;           // [N][Cout][P]
    const float* k0;     // prologue coefficients, [N*Cin] each
    const float* k1;
    const float* k2;
    const float* bias;   // [Cout] or [N][Cout] (bias_stride_n = Cout) or null
    const void* aux;     // EPI_AUX operand [N][Cout][P] (storage type of `out`)
    const float* e0;     // epi 3 (fused SE/GELU backward): per-(n,co) A, B, S, D  -> out = gelu'(A*aux+B)*(S*v+D)
    const float* e1;
    const float* e2;
    const float* e3;
    float2* part;        // [N*Cout][NP] or null
    int bias_stride_n;
    int Cin, Cout, P;
    int pro;             // PRO_*
    int epi;             // 0 none, 1 (sum, sum^2), 2 (sum, sum*aux), 3 fused pass-B + (sum, sum*aux), 4 accumulate,
                         // 5 skip + PreNorm backward: out = aux2 + e0*v + e1*aux + e2, statistics (sum, sum*aux3) if part
    const void* aux2;    // epi 5: dy
    const void* aux3;    // epi 5: h3 of the producing block (or null: no statistics); epi 6: c0 of the producing ConvLayer,
                         // whose ReLU backward is applied to the output: out *= [e3*aux3 + bias > 0] (bias, e3: [N*Cout])
    // epi 7 (narrow fp32-MFMA kernels only): head nonlinearities on the fresh accumulator (uncrtaints.py:441-445):
    //   channel < |head_nm|: head_nm > 0 ? head_scale*sigmoid(v) : v;  else variance f(v) (+ head_eps), head_var 0 softplus,
    //   1 elu + 1, 2 identity
    int head_nm = 0, head_var = 0;
    float head_scale = 1.f, head_eps = 0.f;
    float* head_pre = nullptr;   // optional second output: the pre-activation [N][Cout][P]
    const float* k3 = nullptr;    // PRO_NORMBWD: the norm's mean per (n, ci) -- centred form C1*v + C2*(v2 - mean) + C3; null: 0
    int h2 = 0;                   // wide kernels, fp32 storage: fp16 two-part split wherever the bounds below are given
    // magnitude bounds for the fp16 two-part split ([N][n] floats each; the kernel derives a per-frame power-of-two scale from them
    // and from its own coefficient rows).  PRO_NORMBWD + EPI 3 (gradient GEMM): per-block max |value| arrays written by the
    // producers of the two prologue operands.  PRO_AFFINE / PRO_AFFINE_GELU + EPI 0 / 1 (forward GEMM behind a norm): in_amax =
    // per-plane bounds on |A*h + B| ([N][Cin], uncr_norm_finalize_fwd), in2_amax unused.  Without them: the exact bf16 split.
    // amax_out: this launch's per-block max |stored output| ([N][blocks per frame]) or null.
    float* amax_out = nullptr;
    const float* in_amax = nullptr;
    const float* in2_amax = nullptr;
    int in_amax_n = 0, in2_amax_n = 0;
    // padded planes of an any-size image (anysize.hip): pixels Pv .. P-1 of every plane are a tail that carries no data.  The statistics
    // epilogues leave out every pixel tile that reaches into it (uncr_fix_tail behind the launch adds the boundary tile's valid pixels
    // and zeroes the tail); 0 = P: dense planes
    int Pv = 0;
    const float* rmu = nullptr;   // epi 6: per-(n, co) pivot of the statistics' second component, sum out*(aux3 - rmu) (the norm's mean); null: 0
    const float* emu = nullptr;   // epi 5 / 6: mean of the PreNorm per (n, co): out = dy + e0*v + e1*(x - emu) + e2; null: 0
};

// x = h + m + l exactly, each part a bf16 (kept in the upper half of a 32-bit word).  Truncation split: h takes
// the leading 8 significant bits, m the next 8 starting at the residual's leading one, l the (<= 8) rest.
__device__ __forceinline__ void split3_bf16(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));
}
// two upper-half bf16 -> one dword {lo16 = a, hi16 = b}
__device__ __forceinline__ unsigned pack_bf16x2(unsigned a, unsigned b) {
    return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// Two-part fp16 split for the FORWARD wide GEMMs (fp32 storage): x*SC = h + l + r with h, l fp16 (round to nearest) and
// |r| <= max(2^-22 |x*SC|, 2^-25) -- fp16 has 11 significant bits, so two parts carry 22; the three products h*h', h*l', l*h'
// reach 2^-22 relative accuracy, the same order as an fp32 FMA chain, at HALF the matrix-pipe work of the exact 3 x bf16 split.
// fp16's narrow exponent is why this is used only where the range is known: activation operands are outputs of a norm prologue
// and are scaled per frame from a rigorous bound the statistics finalisation derives (PwArgs::in_amax), weights are scaled per
// output channel at pack time (pack_wt_split_tile); both scales are powers of two and leave in the epilogue.  Gradient GEMMs
// take the same route where their producers leave magnitude bounds (dz GEMM), and the exact bf16 split otherwise.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2v_t __attribute__((ext_vector_type(2)));
// (a, b) -> packed fp16 pairs {lo16 = a, hi16 = b}: hi parts and lo parts.  No clamp: every caller scales its operand from a rigorous
// bound to at most 2^15 first (weights at pack time, activations per frame), so the conversion cannot overflow on finite data; an inf /
// NaN input (or a frame whose bound is not finite: no scaling) gives inf / NaN parts and the result is NaN, as in fp32.
#ifndef UNCR_SPLIT_MIX
#define UNCR_SPLIT_MIX 1      // residuals a - h on v_fma_mix_f32 (the fp16 half is widened inside the FMA): 4 instead of 6 VALU instructions per pair
#endif
__device__ __forceinline__ void split2_f16_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2_t h = __builtin_convertvector(f32x2v_t{a, b}, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
#if UNCR_SPLIT_MIX
    // la = fma(float(h.lo), -1, a), lb = fma(float(h.hi), -1, b): exact (a - h needs no rounding), the same bits as the two-step form.
    // hipcc does not select v_fma_mix_f32 for this pattern (it emits v_cvt_f32_f16 x 2 + v_sub x 2): inline asm.
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    const f16x2_t l = __builtin_convertvector(f32x2v_t{la, lb}, f16x2_t);
#else
    const f32x2v_t hf = __builtin_convertvector(h, f32x2v_t);
    const f16x2_t l = __builtin_convertvector(f32x2v_t{a - hf.x, b - hf.y}, f16x2_t);
#endif
    lo = __builtin_bit_cast(unsigned, l);
}
#define PWS_NSLOT 5      // packed weight slots per (k-step, co tile): bf16 h, m, l and fp16 h, l (scaled)
#define PWS_KC 32
// k-steps in the packed weights: chunk count padded to even (the DEPTH = 2 kernels compute chunk pairs)
__host__ __device__ static inline int pws_nks(int rows_k) { return 2 * (((rows_k + PWS_KC - 1) / PWS_KC + 1) / 2 * 2); }

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// pw_gemm_split.hip
// one object per prologue kind (pw_gemm_split.hip compiled with -DPWS_PRO=0..4); cp in {128, 256}
// act: UNCR_F32 (exact 3-way split of both operands, six products) | UNCR_BF16 (bf16 activations: one activation part,
// the two leading weight parts)
int pw_split_launch_p0(const PwArgs& g, int N, int cp, int act, hipStream_t stream);
int pw_split_launch_p1(const PwArgs& g, int N, int cp, int act, hipStream_t stream);
int pw_split_launch_p2(const PwArgs& g, int N, int cp, int act, hipStream_t stream);
int pw_split_launch_p3(const PwArgs& g, int N, int cp, int act, hipStream_t stream);
int pw_split_launch_p4(const PwArgs& g, int N, int cp, int act, hipStream_t stream);
int pw_split_pack(const float* W, int rows_k, int cols_co, int ld, int transpose, float* out, hipStream_t stream);
size_t pw_split_wt_floats(int rows_k, int cp);
int pw_split_blocks_per_frame(int N, int P);
int pw_pack_batch(const long long* desc, int n_items, hipStream_t stream);

// pw_wgrad_split.hip
int pw_wgrad_split_nbx(int N, int P);
bool pw_wgrad_split_supported(int Cd, int Cx, int pro_d, int pro_x, bool rowsum);
int pw_wgrad_split_launch(const float* d, const float* d2, const float* x, const float* dk0, const float* dk1,
                          const float* dk2, const float* dkmu, const float* xk0, const float* xk1, const float* xk2, float* part,
                          int N, int Cd, int Cx, int P, int nbx, int pro_x, const float* d_amax, int d_amax_n,
                          const float* d2_amax, int d2_amax_n, const float* x_ub, int Pv, hipStream_t stream);

// pw_wgrad_a16.hip: the same weight gradients from bf16 operands (one bf16 x bf16 product per MAC, fp32 accumulation)
int pw_wgrad_a16_nbx(int N, int P);
bool pw_wgrad_a16_supported(int Cd, int Cx, int pro_d, int pro_x, bool rowsum);
int pw_wgrad_a16_launch(const void* d, const void* d2, const void* x, const float* dk0, const float* dk1, const float* dk2,
                        const float* dkmu, const float* xk0, const float* xk1, const float* xk2, float* part, int N, int Cd, int Cx,
                        int P, int nbx, int pro_x, hipStream_t stream);
