// in_conv = Conv2d(Cin -> Cout, k = 1, bias) + GroupNorm + ReLU (utae.py:453-520 as built by uncrtaints.py:310-314) WITHOUT its
// pre-norm tensor.  c0 = W x + b has K = Cin = 15: everything the GroupNorm needs from c0 is a quadratic form in the second-moment
// matrix of the input frame, so c0 (402 MB at 12 frames of 256 x 256) is never written or read:
//
//   forward   M~[n] = sum_p x~ x~^T,  x~ = [x; 1]  (16 x 16 per frame: second moments, channel sums, pixel count)     uncr_inconv_moments
//             sum_p c0[k]   = W[k].Sx + b[k] P ,   sum_p c0[k]^2 = W[k] Mx W[k]^T + 2 b[k] W[k].Sx + b[k]^2 P
//             -> mean / rstd per (frame, group), A = gamma rstd, B = beta - mean A                                 uncr_inconv_norm_from_moments
//             a0 = relu(A (W x + b) + B) straight from the GEMM's accumulators                                      uncr_pw_gemm(epi = 9)
//   backward  du = d a0 * [a0 > 0] (the consumer's dx epilogue masks with x = a0 > 0, uncr_pw_gemm_dx),  R[n] = sum_p du x^T  (one
//             plain weight-gradient GEMM), and with S1 = sum_p du:
//             sum_p du c0 = W[k].R[n,k] + b[k] S1     -> the GroupNorm backward's coefficients c1, c2, c3 and d gamma, d beta
//             dW = sum_n c1 R + c2 (W Mx + b Sx^T - mu Sx^T) + c3 Sx^T ,   db = sum_n c1 S1 + c2 (W.Sx + b P - mu P) + c3 P      uncr_inconv_bwd_finish
// All small algebra in fp64 (products of fp32 inputs are exact in fp64), fixed order.  The gradient w.r.t. the model INPUT needs c0
// per pixel again; the host layer then recomputes it with the plain GEMM (rare: the input of a training run has no gradient).
#include "common.h"

#define ICM_TP 256          // pixels per LDS tile
#define ICM_LD (ICM_TP + 1) // row stride (floats): 16 rows of one pixel fall into 16 different banks
#define ICM_A 16            // augmented channel count (Cin + 1 <= 16)

__device__ __forceinline__ float icm_ld(const float* p) { return *p; }
__device__ __forceinline__ float icm_ld(const bf16_t* p) { return __uint_as_float((unsigned)(*p) << 16); }

// grid = (nblk, N), block = 256 = 16 x 16 entries (i, j) of the frame's augmented moment matrix; part[(n*nblk + b)*256 + i*16 + j]
// T: storage of x (fp32, or bf16: the moments of the values as stored)
template <typename T>
__global__ __launch_bounds__(256) void inconv_moments_kernel(const T* __restrict__ x, int Cin, int P, int px_per_block,
                                                             double* __restrict__ part) {
    __shared__ float xs[ICM_A][ICM_LD];
    const int tid = threadIdx.x, i = tid >> 4, j = tid & 15;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * px_per_block, p1 = min(P, p0 + px_per_block);
    const T* xb = x + (size_t)n * Cin * P;
    double acc = 0.0;
    for (int t0 = p0; t0 < p1; t0 += ICM_TP) {
        const int np = min(ICM_TP, p1 - t0);
        __syncthreads();                        // the previous tile has been consumed
        for (int c = 0; c < ICM_A; ++c) {
            float v = 0.f;
            if (tid < np) v = c < Cin ? icm_ld(xb + (size_t)c * P + t0 + tid) : (c == Cin ? 1.f : 0.f);
            xs[c][tid] = v;                     // pixels past the range contribute zeros (also to the count)
        }
        __syncthreads();
        const float* ri = xs[i];
        const float* rj = xs[j];
#pragma unroll 8
        for (int p = 0; p < ICM_TP; ++p) acc = fma((double)ri[p], (double)rj[p], acc);
    }
    part[((size_t)n * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}

extern "C" int uncr_inconv_moment_blocks(int P) {
    const int b = (P + 1023) / 1024;
    return b < 1 ? 1 : (b > 256 ? 256 : b);
}
extern "C" int uncr_inconv_moments(const void* x, int N, int Cin, int P, double* part, int act, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || P <= 0) return UNCR_ESHAPE;
    if (!x || !part || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    const int nblk = uncr_inconv_moment_blocks(P);
    const int ppb = ((P + nblk - 1) / nblk + ICM_TP - 1) / ICM_TP * ICM_TP;
    if (act == UNCR_BF16)
        hipLaunchKernelGGL(inconv_moments_kernel<bf16_t>, dim3(nblk, N), dim3(256), 0, stream, (const bf16_t*)x, Cin, P, ppb, part);
    else
        hipLaunchKernelGGL(inconv_moments_kernel<float>, dim3(nblk, N), dim3(256), 0, stream, (const float*)x, Cin, P, ppb, part);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// grid = N * G, block = 256.  mom [N][256] (out): the frame's reduced augmented moment matrix (written by the block of group 0).
__global__ __launch_bounds__(256) void inconv_norm_from_moments_kernel(
    const double* __restrict__ part, int nblk, int Cin, int Cout, int G, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ coefA, float* __restrict__ coefB,
    float* __restrict__ save_mean, float* __restrict__ save_rstd, double* __restrict__ mom) {
    __shared__ double M[256];
    __shared__ double t1s[256], t2s[256];
    __shared__ float sh_mean, sh_rstd;
    const int tid = threadIdx.x, n = blockIdx.x / G, g = blockIdx.x % G, Cg = Cout / G;
    {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += part[((size_t)n * nblk + b) * 256 + tid];
        M[tid] = a;
        if (g == 0) mom[(size_t)n * 256 + tid] = a;
    }
    __syncthreads();
    const double Pn = M[Cin * 16 + Cin];          // pixel count of the frame
    for (int c = tid; c < Cg; c += 256) {       // (Cg <= 256: one pass)
        const int k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0;
        double ws = 0.0, q = 0.0;
        for (int a = 0; a < Cin; ++a) {
            const double wa = (double)w[a];
            ws += wa * M[a * 16 + Cin];
            double r = 0.0;
            for (int e = 0; e < Cin; ++e) r += (double)w[e] * M[a * 16 + e];
            q += wa * r;
        }
        t1s[c] = ws + b * Pn;
        t2s[c] = q + 2.0 * b * ws + b * b * Pn;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0, ss = 0.0;
        for (int c = 0; c < Cg; ++c) { s += t1s[c]; ss += t2s[c]; }
        const double Mn = (double)Cg * Pn;
        const double mean = s / Mn;
        double var = ss / Mn - mean * mean;
        if (var < 0) var = 0;
        sh_mean = (float)mean;
        sh_rstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[n * G + g] = sh_mean;
        save_rstd[n * G + g] = sh_rstd;
    }
    __syncthreads();
    for (int c = tid; c < Cg; c += 256) {
        const int k = g * Cg + c;
        const float a = gamma[k] * sh_rstd;
        coefA[n * Cout + k] = a;
        coefB[n * Cout + k] = beta[k] - sh_mean * a;
    }
}

extern "C" int uncr_inconv_norm_from_moments(const double* part, int nblk, int N, int Cin, int Cout, int groups, const float* W,
                                             const float* bias, const float* gamma, const float* beta, float eps, float* coefA,
                                             float* coefB, float* save_mean, float* save_rstd, double* mom, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || Cout <= 0 || groups <= 0 || Cout % groups || Cout / groups > 256 || nblk <= 0)
        return UNCR_ESHAPE;
    if (!part || !W || !gamma || !beta || !coefA || !coefB || !save_mean || !save_rstd || !mom) return UNCR_EINVAL;
    hipLaunchKernelGGL(inconv_norm_from_moments_kernel, dim3(N * groups), dim3(256), 0, stream, part, nblk, Cin, Cout, groups, W, bias,
                       gamma, beta, eps, coefA, coefB, save_mean, save_rstd, mom);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// grid = G, block = 512.  Dynamic LDS: S1, S2 [N][Cg] doubles, c2, c3, m-terms [N] doubles.
__global__ __launch_bounds__(512) void inconv_bwd_finish_kernel(
    const float* __restrict__ R /* [N][Cout][Cin] */, const float2* __restrict__ part /* [N*Cout][NP]: .x = sum du */, int NP,
    const double* __restrict__ mom /* [N][256] */, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_rstd, int N, int Cin, int Cout,
    int G, float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, g = blockIdx.x, Cg = Cout / G;
    double* S1 = sm;                       // [N][Cg]
    double* S2 = S1 + (size_t)N * Cg;      // [N][Cg]
    double* c2 = S2 + (size_t)N * Cg;      // [N]
    double* c3 = c2 + N;                   // [N]
    for (int q = tid; q < N * Cg; q += 512) {
        const int n = q / Cg, c = q - n * Cg, k = g * Cg + c;
        const float2* src = part + ((size_t)n * Cout + k) * NP;
        double s1 = 0.0;
        for (int j = 0; j < NP; ++j) s1 += (double)src[j].x;
        const float* r = R + ((size_t)n * Cout + k) * Cin;
        const float* w = W + (size_t)k * Cin;
        double s2 = 0.0;
        for (int a = 0; a < Cin; ++a) s2 += (double)w[a] * (double)r[a];
        S1[q] = s1;
        S2[q] = s2 + (bias ? (double)bias[k] : 0.0) * s1;       // sum_p du * c0
    }
    __syncthreads();
    for (int n = tid; n < N; n += 512) {
        const double mu = (double)save_mean[n * G + g], r = (double)save_rstd[n * G + g];
        const double Mn = (double)Cg * mom[(size_t)n * 256 + Cin * 16 + Cin];
        double a = 0.0, b = 0.0;
        for (int c = 0; c < Cg; ++c) {
            const double gm = (double)gamma[g * Cg + c];
            a += gm * S1[n * Cg + c];
            b += gm * r * (S2[n * Cg + c] - mu * S1[n * Cg + c]);
        }
        c2[n] = -r * r * (b / Mn);       // dc0 = c1*du + c2*(c0 - mu) + c3 (centred form, norm.hip::gn_finalize_bwd_kernel)
        c3[n] = -r * (a / Mn);
    }
    __syncthreads();
    // d gamma, d beta, d bias: one thread per channel of the group
    for (int c = tid; c < Cg; c += 512) {
        const int k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0, gm = (double)gamma[k];
        double dg = 0.0, dbt = 0.0, dbias = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mu = (double)save_mean[n * G + g], r = (double)save_rstd[n * G + g];
            const double* M = mom + (size_t)n * 256;
            const double Pn = M[Cin * 16 + Cin];
            const double s1 = S1[n * Cg + c], s2 = S2[n * Cg + c];
            dg += r * (s2 - mu * s1);
            dbt += s1;
            double T = b * Pn;                                      // sum_p c0[k]
            for (int a = 0; a < Cin; ++a) T += (double)w[a] * M[a * 16 + Cin];
            dbias += r * gm * s1 + c2[n] * (T - mu * Pn) + c3[n] * Pn;
        }
        dgamma[k] = (float)dg;
        dbeta[k] = (float)dbt;
        if (db) db[k] = (float)dbias;
    }
    // d W[k][a]
    for (int q = tid; q < Cg * Cin; q += 512) {
        const int c = q / Cin, a = q - c * Cin, k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0, gm = (double)gamma[k];
        double acc = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mu = (double)save_mean[n * G + g], r = (double)save_rstd[n * G + g];
            const double* M = mom + (size_t)n * 256;
            const double Sxa = M[a * 16 + Cin];
            double Q = b * Sxa;                                     // sum_p c0[k] * x[a]
            for (int e = 0; e < Cin; ++e) Q += (double)w[e] * M[e * 16 + a];
            acc += r * gm * (double)R[((size_t)n * Cout + k) * Cin + a] + c2[n] * (Q - mu * Sxa) + c3[n] * Sxa;
        }
        dW[(size_t)k * Cin + a] = (float)acc;
    }
}

extern "C" int uncr_inconv_bwd_finish(const float* R, const float* part, int NP, const double* mom, const float* W, const float* bias,
                                      const float* gamma, const float* save_mean, const float* save_rstd, int N, int Cin, int Cout,
                                      int groups, float* dW, float* db, float* dgamma, float* dbeta, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || Cout <= 0 || groups <= 0 || Cout % groups || NP <= 0) return UNCR_ESHAPE;
    if (!R || !part || !mom || !W || !gamma || !save_mean || !save_rstd || !dW || !dgamma || !dbeta) return UNCR_EINVAL;
    const size_t lds = ((size_t)2 * N * (Cout / groups) + 2 * (size_t)N) * sizeof(double);
    if (lds > 60 * 1024) return UNCR_ESHAPE;
    hipLaunchKernelGGL(inconv_bwd_finish_kernel, dim3(groups), dim3(512), lds, stream, R, (const float2*)part, NP, mom, W, bias, gamma,
                       save_mean, save_rstd, N, Cin, Cout, groups, dW, db, dgamma, dbeta);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
