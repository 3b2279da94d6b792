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

#define ICM_A 16            // augmented channel count (Cin + 1 <= 16)
typedef double icm_f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float icm_w(float v) { return v; }
__device__ __forceinline__ float icm_w(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
template <typename T> struct icm_vec4;
template <> struct icm_vec4<float> { typedef float4 type; };
template <> struct icm_vec4<bf16_t> { typedef ushort4 type; };

// M~ = X~ X~^T is a 16 x 16 x (pixels) product with fp64 accumulation: exactly v_mfma_f64_16x16x4_f64 with A = B (the product is
// symmetric, so one register feeds both operands: lane l holds channel l & 15 of pixel 4-group l >> 4).  No LDS tile: a lane loads four
// consecutive pixels of its channel (one 16- / 8-byte load per 16 pixels of the wave), widens them to fp64 (exact) and issues four
// MFMAs; the products of fp32 values are exact in fp64.  C/D layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4*reg.
// grid = (nblk, N), block = 256 (4 waves, the block's pixel range dealt to them in 16-pixel steps); part[(n*nblk + b)*256 + i*16 + j]
// T: storage of x (fp32, or bf16: the moments of the values as stored)
template <typename T>
__global__ __launch_bounds__(256) void inconv_moments_kernel(const T* __restrict__ x, int Cin, int P, int px_per_block,
                                                             double* __restrict__ part, int pstride) {
    typedef typename icm_vec4<T>::type V4;
    __shared__ double red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int n = blockIdx.y;
    // P = pixels that carry data (the augmented channel counts them one by one); pstride = the plane stride (> P: the padded planes of
    // an any-size image, whose tail holds zeros: a 4-group that straddles P adds nothing but its valid pixels' count)
    const int p0 = blockIdx.x * px_per_block, p1 = min((P + 3) & ~3, p0 + px_per_block);
    const T* xb = x + ((size_t)n * Cin + (c < Cin ? c : 0)) * pstride;
    icm_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    // two 16-pixel steps per iteration on independent accumulators (an MFMA's dependent latency is 4 passes)
    for (int s = p0 + 32 * wv; s < p1; s += 128) {
        float v[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int px = s + 16 * h + 4 * q;
            const bool in = px < p1;                 // (p1 and px are multiples of 4: a 4-group is inside or outside as a whole)
            V4 raw = {};
            if (in && c < Cin) raw = *(const V4*)(xb + px);
            const bool aug = in && c == Cin;
            v[h][0] = c < Cin ? icm_w(raw.x) : ((aug && px + 0 < P) ? 1.f : 0.f); v[h][1] = c < Cin ? icm_w(raw.y) : ((aug && px + 1 < P) ? 1.f : 0.f);
            v[h][2] = c < Cin ? icm_w(raw.z) : ((aug && px + 2 < P) ? 1.f : 0.f); v[h][3] = c < Cin ? icm_w(raw.w) : ((aug && px + 3 < P) ? 1.f : 0.f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double a0 = (double)v[0][e], a1 = (double)v[1][e];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][(q + 4 * r) * 16 + c] = acc0[r] + acc1[r];
    __syncthreads();
    part[((size_t)n * gridDim.x + blockIdx.x) * 256 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

extern "C" int uncr_inconv_moment_blocks(int P) {
    const int b = (P + 1023) / 1024;
    return b < 1 ? 1 : (b > 256 ? 256 : b);
}
extern "C" int uncr_inconv_moments(const void* x, int N, int Cin, int P, double* part, int act, int pstride, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || P <= 0) return UNCR_ESHAPE;
    if (!x || !part || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    if (pstride <= 0) pstride = P;
    const int nblk = uncr_inconv_moment_blocks(P);
    if (pstride % 4 || pstride < ((P + 3) & ~3)) return UNCR_ESHAPE;      // 16-byte loads; a straddling 4-group stays inside the plane
    const int ppb = ((P + nblk - 1) / nblk + 127) / 128 * 128;
    if (act == UNCR_BF16)
        hipLaunchKernelGGL(inconv_moments_kernel<bf16_t>, dim3(nblk, N), dim3(256), 0, stream, (const bf16_t*)x, Cin, P, ppb, part, pstride);
    else
        hipLaunchKernelGGL(inconv_moments_kernel<float>, dim3(nblk, N), dim3(256), 0, stream, (const float*)x, Cin, P, ppb, part, pstride);
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
    __shared__ int sh_flat;
    const int tid = threadIdx.x, n = blockIdx.x / G, g = blockIdx.x % G, Cg = Cout / G;
    {
        // up to 32 loads in flight per thread (a plain loop serialises nblk L2 round trips), summed in block order
        const double* src = part + (size_t)n * nblk * 256 + tid;
        double a = 0.0;
        int b = 0;
        for (; b + 32 <= nblk; b += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(b + u) * 256];
#pragma unroll
            for (int u = 0; u < 32; ++u) a += v[u];
        }
        for (; b + 8 <= nblk; b += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + u) * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; b < nblk; ++b) a += src[(size_t)b * 256];
        M[tid] = a;
        if (g == 0) mom[(size_t)n * 256 + tid] = a;
    }
    __syncthreads();
    const double Pn = M[Cin * 16 + Cin];          // pixel count of the frame
    for (int c = tid; c < Cg; c += 256) {       // (Cg <= 256: one pass)
        const int k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0;
        double wr[ICM_A];                         // the weight row in registers (a loop over global memory re-reads it 15 x 15 times)
#pragma unroll
        for (int a = 0; a < ICM_A; ++a) wr[a] = a < Cin ? (double)w[a < Cin ? a : 0] : 0.0;
        double ws = 0.0, q = 0.0;
#pragma unroll
        for (int a = 0; a < ICM_A - 1; ++a) {     // (rows / columns past Cin multiply zero weights; M's augmented row is excluded below)
            double r = 0.0;
#pragma unroll
            for (int e = 0; e < ICM_A - 1; ++e) r += wr[e] * (e < Cin ? M[a * 16 + e] : 0.0);
            ws += wr[a] * (a < Cin ? M[a * 16 + Cin] : 0.0);
            q += wr[a] * r;
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
        // InstanceNorm (Cg == 1) over a constant plane -- a zero-padded date: c0 = bias everywhere: the exact result is 0, which
        // A*c0 + B with a rounded B does not give (norm.hip::gn_finalize_fwd_kernel has the same rule and the reasoning)
        const bool flat = Cg == 1 && var <= ldexp(mean * mean, -17);
        if (flat) var = 0;
        sh_flat = flat ? 1 : 0;
        sh_mean = (float)mean;
        sh_rstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[n * G + g] = sh_mean;
        save_rstd[n * G + g] = sh_rstd;
    }
    __syncthreads();
    for (int c = tid; c < Cg; c += 256) {
        const int k = g * Cg + c;
        const float a = sh_flat ? 0.f : gamma[k] * sh_rstd;
        coefA[n * Cout + k] = a;
        coefB[n * Cout + k] = sh_flat ? beta[k] : beta[k] - sh_mean * a;
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

// grid = G, block = 512.  Dynamic LDS: S1, S2 [N][Cg], c2, c3 [N], Mc [256], v2, v3 [16], mean, rstd [N] doubles.
// The sums over frames are taken once per group, not once per output: with c2, c3 per (frame, group)
//   Mc = sum_n c2[n] M~[n]  (16 x 16),  v2[a] = sum_n c2[n] mu[n] Sx[n][a],  v3[a] = sum_n c3[n] Sx[n][a]
//   dW[k][a] = gamma[k] sum_n rstd[n] R[n][k][a] + sum_e W[k][e] Mc[e][a] + b[k] Mc[a][Cin] - v2[a] + v3[a]
//   db[k]    = gamma[k] sum_n rstd[n] S1[n][k]   + sum_a W[k][a] Mc[a][Cin] + b[k] Mc[Cin][Cin] - v2[Cin] + v3[Cin]
// (Sx[n][a] = M~[n][a][Cin], the pixel count = M~[n][Cin][Cin]: the augmented row serves the bias like a 16th input channel).
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
    double* Mc = c3 + N;                   // [256]
    double* v2 = Mc + 256;                 // [16]
    double* v3 = v2 + 16;                  // [16]
    double* mus = v3 + 16;                 // [N] the group's mean / rstd per frame (read in every loop below)
    double* rss = mus + N;                 // [N]
    for (int n = tid; n < N; n += 512) { mus[n] = (double)save_mean[n * G + g]; rss[n] = (double)save_rstd[n * G + g]; }
    for (int q = tid; q < N * Cg; q += 512) {
        const int n = q / Cg, c = q - n * Cg, k = g * Cg + c;
        const float2* src = part + ((size_t)n * Cout + k) * NP;
        double s1 = 0.0;
        int j = 0;
        for (; j + 16 <= NP; j += 16) {        // loads in flight in batches, summed in slot order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[j + u].x;
#pragma unroll
            for (int u = 0; u < 16; ++u) s1 += (double)v[u];
        }
        for (; j + 4 <= NP; j += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[j + u].x;
#pragma unroll
            for (int u = 0; u < 4; ++u) s1 += (double)v[u];
        }
        for (; j < NP; ++j) s1 += (double)src[j].x;
        const float* r = R + ((size_t)n * Cout + k) * Cin;
        const float* w = W + (size_t)k * Cin;
        double s2 = 0.0;
        for (int a = 0; a < Cin; ++a) s2 += (double)w[a] * (double)r[a];
        S1[q] = s1;
        S2[q] = s2 + (bias ? (double)bias[k] : 0.0) * s1;       // sum_p du * c0
    }
    __syncthreads();
    for (int n = tid; n < N; n += 512) {
        const double mu = mus[n], r = rss[n];
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
    if (tid < 256) {
        double a = 0.0;
        for (int n = 0; n < N; ++n) a += c2[n] * mom[(size_t)n * 256 + tid];
        Mc[tid] = a;
    } else if (tid < 256 + 16) {
        const int a = tid - 256;
        double s2 = 0.0, s3 = 0.0;
        for (int n = 0; n < N; ++n) {
            const double sx = mom[(size_t)n * 256 + a * 16 + Cin];
            s2 += c2[n] * mus[n] * sx;
            s3 += c3[n] * sx;
        }
        v2[a] = s2;
        v3[a] = s3;
    }
    __syncthreads();
    // d gamma, d beta, d bias: one thread per channel of the group
    for (int c = tid; c < Cg; c += 512) {
        const int k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0, gm = (double)gamma[k];
        double dg = 0.0, dbt = 0.0, rs1 = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mu = mus[n], r = rss[n];
            const double s1 = S1[n * Cg + c], s2 = S2[n * Cg + c];
            dg += r * (s2 - mu * s1);
            dbt += s1;
            rs1 += r * s1;
        }
        double t = b * Mc[Cin * 16 + Cin];
        for (int a = 0; a < Cin; ++a) t += (double)w[a] * Mc[a * 16 + Cin];
        dgamma[k] = (float)dg;
        dbeta[k] = (float)dbt;
        if (db) db[k] = (float)(gm * rs1 + t - v2[Cin] + v3[Cin]);
    }
    // d W[k][a]
    for (int q = tid; q < Cg * Cin; q += 512) {
        const int c = q / Cin, a = q - c * Cin, k = g * Cg + c;
        const float* w = W + (size_t)k * Cin;
        const double b = bias ? (double)bias[k] : 0.0, gm = (double)gamma[k];
        double rr = 0.0;
        int n = 0;
        for (; n + 4 <= N; n += 4) {
            float rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rv[u] = R[((size_t)(n + u) * Cout + k) * Cin + a];
#pragma unroll
            for (int u = 0; u < 4; ++u) rr += rss[n + u] * (double)rv[u];
        }
        for (; n < N; ++n) rr += rss[n] * (double)R[((size_t)n * Cout + k) * Cin + a];
        float wv[ICM_A];
#pragma unroll
        for (int e = 0; e < ICM_A; ++e) wv[e] = w[e < Cin ? e : 0];
        double t = b * Mc[a * 16 + Cin];
#pragma unroll
        for (int e = 0; e < ICM_A - 1; ++e) t += e < Cin ? (double)wv[e] * Mc[e * 16 + a] : 0.0;
        dW[(size_t)k * Cin + a] = (float)(gm * rr + t - v2[a] + v3[a]);
    }
}

extern "C" int uncr_inconv_bwd_finish(const float* R, const float* part, int NP, const double* mom, const float* W, const float* bias,
                                      const float* gamma, const float* save_mean, const float* save_rstd, int N, int Cin, int Cout,
                                      int groups, float* dW, float* db, float* dgamma, float* dbeta, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || Cout <= 0 || groups <= 0 || Cout % groups || NP <= 0) return UNCR_ESHAPE;
    if (!R || !part || !mom || !W || !gamma || !save_mean || !save_rstd || !dW || !dgamma || !dbeta) return UNCR_EINVAL;
    const size_t lds = ((size_t)2 * N * (Cout / groups) + 4 * (size_t)N + 256 + 32) * sizeof(double);
    if (lds > 60 * 1024) return UNCR_ESHAPE;
    hipLaunchKernelGGL(inconv_bwd_finish_kernel, dim3(groups), dim3(512), lds, stream, R, (const float2*)part, NP, mom, W, bias, gamma,
                       save_mean, save_rstd, N, Cin, Cout, groups, dW, db, dgamma, dbeta);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- the same for a train-mode BatchNorm behind the convolution (encoder_norm = 'batch'): the statistics set is a channel over ALL
// frames, so the frames' moment matrices add up first:  M~ = sum_n M~[n],
//   sum c0[k] = W[k].Sx + b[k] P,  sum c0[k]^2 = W[k] Mx W[k]^T + 2 b[k] W[k].Sx + b[k]^2 P   (Sx, Mx, P of the total),
// mean / rstd per channel (+ the running statistics' update, nn.BatchNorm2d), A = gamma rstd, B = beta - mean A for every frame.
// fp64 throughout: the channel means of c0 = W x + b over non-negative inputs sit several standard deviations from zero, where the
// fp32 slot sums of a statistics epilogue leave 1e-7 in the mean -- ATen's CPU path accumulates in fp64 too.
// stage 1: grid = N, block = 256: mom[n][256] = the frame's reduced matrix.  stage 2: grid = ceil(Cout / 256), block = 256.
__global__ __launch_bounds__(256) void inconv_moment_reduce_kernel(const double* __restrict__ part, int nblk, double* __restrict__ mom) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const double* src = part + (size_t)n * nblk * 256 + tid;
    double a = 0.0;
    int b = 0;
    for (; b + 16 <= nblk; b += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(b + u) * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) a += v[u];
    }
    for (; b < nblk; ++b) a += src[(size_t)b * 256];
    mom[(size_t)n * 256 + tid] = a;
}
__global__ __launch_bounds__(256) void inconv_bn_from_moments_kernel(
    const double* __restrict__ mom, int N, int Cin, int Cout, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ save_mean,
    float* __restrict__ save_rstd, double* __restrict__ momtot) {
    __shared__ double M[256];
    const int tid = threadIdx.x;
    {
        double a = 0.0;
        for (int n = 0; n < N; ++n) a += mom[(size_t)n * 256 + tid];      // frame order
        M[tid] = a;
        if (blockIdx.x == 0) momtot[tid] = a;
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + tid;
    if (k >= Cout) return;
    const double Pn = M[Cin * 16 + Cin];
    const float* w = W + (size_t)k * Cin;
    const double b = bias ? (double)bias[k] : 0.0;
    double wr[ICM_A];
#pragma unroll
    for (int a = 0; a < ICM_A; ++a) wr[a] = a < Cin ? (double)w[a < Cin ? a : 0] : 0.0;
    double ws = 0.0, q = 0.0;
#pragma unroll
    for (int a = 0; a < ICM_A - 1; ++a) {
        double r = 0.0;
#pragma unroll
        for (int e = 0; e < ICM_A - 1; ++e) r += wr[e] * (e < Cin ? M[a * 16 + e] : 0.0);
        ws += wr[a] * (a < Cin ? M[a * 16 + Cin] : 0.0);
        q += wr[a] * r;
    }
    const double mean = (ws + b * Pn) / Pn;
    double var = (q + 2.0 * b * ws + b * b * Pn) / Pn - mean * mean;
    if (var < 0) var = 0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[k] = fm;
    save_rstd[k] = fr;
    if (running_mean) {
        const double unb = var * (Pn / (Pn > 1 ? Pn - 1 : 1));
        running_mean[k] = (float)((1.0 - momentum) * running_mean[k] + momentum * mean);
        running_var[k] = (float)((1.0 - momentum) * running_var[k] + momentum * unb);
    }
    const float a = gamma[k] * fr, bb = beta[k] - fm * a;
    for (int n = 0; n < N; ++n) { coefA[(size_t)n * Cout + k] = a; coefB[(size_t)n * Cout + k] = bb; }
}
extern "C" int uncr_inconv_bn_from_moments(const double* part, int nblk, int N, int Cin, int Cout, const float* W, const float* bias,
                                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                                           float momentum, float eps, float* coefA, float* coefB, float* save_mean, float* save_rstd,
                                           double* mom /* [N][256] */, double* momtot /* [256] */, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || Cout <= 0 || nblk <= 0) return UNCR_ESHAPE;
    if (!part || !W || !gamma || !beta || !coefA || !coefB || !save_mean || !save_rstd || !mom || !momtot) return UNCR_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return UNCR_EINVAL;
    hipLaunchKernelGGL(inconv_moment_reduce_kernel, dim3(N), dim3(256), 0, stream, part, nblk, mom);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(inconv_bn_from_moments_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, mom, N, Cin, Cout, W, bias, gamma,
                       beta, running_mean, running_var, momentum, eps, coefA, coefB, save_mean, save_rstd, momtot);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// backward of the same: with S1[k] = sum_{n,p} du, Rt[k][a] = sum_n R[n][k][a], S2[k] = W[k].Rt[k] + b[k] S1[k] = sum du*c0,
//   d gamma = r (S2 - mu S1), d beta = S1;  dc0 = c1 du + c2 (c0 - mu) + c3 with c1 = r gamma, c2 = -r^2 (gamma r (S2 - mu S1) / P),
//   c3 = -r gamma S1 / P (P = all pixels of all frames);
//   dW[k][a] = c1 Rt[k][a] + c2 (sum_e W[k][e] Mx[e][a] + (b[k] - mu) Sx[a]) + c3 Sx[a],   db[k] = c1 S1 + c2 (W[k].Sx + (b[k] - mu) P) + c3 P.
// One thread per channel (a few hundred loads each: latency, off the critical path of the full-resolution kernels).
__global__ __launch_bounds__(64) void inconv_bwd_finish_bn_kernel(
    const float* __restrict__ R /* [N][Cout][Cin] */, const float2* __restrict__ part /* [N*Cout][NP]: .x = sum du */, int NP,
    const double* __restrict__ momtot /* [256] */, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_rstd, int N, int Cin, int Cout,
    float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= Cout) return;
    double s1 = 0.0;
    for (int n = 0; n < N; ++n) {
        const float2* src = part + ((size_t)n * Cout + k) * NP;
        int j = 0;
        for (; j + 8 <= NP; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[j + u].x;
#pragma unroll
            for (int u = 0; u < 8; ++u) s1 += (double)v[u];
        }
        for (; j < NP; ++j) s1 += (double)src[j].x;
    }
    double rt[ICM_A];
#pragma unroll
    for (int a = 0; a < ICM_A; ++a) rt[a] = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* r = R + ((size_t)n * Cout + k) * Cin;
#pragma unroll
        for (int a = 0; a < ICM_A - 1; ++a) rt[a] += a < Cin ? (double)r[a < Cin ? a : 0] : 0.0;
    }
    const float* w = W + (size_t)k * Cin;
    double wr[ICM_A];
#pragma unroll
    for (int a = 0; a < ICM_A; ++a) wr[a] = a < Cin ? (double)w[a < Cin ? a : 0] : 0.0;
    const double b = bias ? (double)bias[k] : 0.0, gm = (double)gamma[k];
    const double mu = (double)save_mean[k], r = (double)save_rstd[k];
    double s2 = b * s1;
#pragma unroll
    for (int a = 0; a < ICM_A - 1; ++a) s2 += wr[a] * rt[a];
    const double Pn = momtot[Cin * 16 + Cin];
    const double dg = r * (s2 - mu * s1);
    const double c1 = r * gm, c2 = -r * r * (gm * dg / Pn), c3 = -r * (gm * s1 / Pn);
    dgamma[k] = (float)dg;
    dbeta[k] = (float)s1;
    double wsx = 0.0;
#pragma unroll
    for (int a = 0; a < ICM_A - 1; ++a) wsx += wr[a] * (a < Cin ? momtot[a * 16 + Cin] : 0.0);
    if (db) db[k] = (float)(c1 * s1 + c2 * (wsx + (b - mu) * Pn) + c3 * Pn);
#pragma unroll
    for (int a = 0; a < ICM_A - 1; ++a) {
        if (a < Cin) {
            double t = 0.0;
#pragma unroll
            for (int e = 0; e < ICM_A - 1; ++e) t += wr[e] * (e < Cin ? momtot[e * 16 + a] : 0.0);
            const double sx = momtot[a * 16 + Cin];
            dW[(size_t)k * Cin + a] = (float)(c1 * rt[a] + c2 * (t + (b - mu) * sx) + c3 * sx);
        }
    }
}
extern "C" int uncr_inconv_bwd_finish_bn(const float* R, const float* part, int NP, const double* momtot, const float* W,
                                         const float* bias, const float* gamma, const float* save_mean, const float* save_rstd, int N,
                                         int Cin, int Cout, float* dW, float* db, float* dgamma, float* dbeta, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin + 1 > ICM_A || Cout <= 0 || NP <= 0) return UNCR_ESHAPE;
    if (!R || !part || !momtot || !W || !gamma || !save_mean || !save_rstd || !dW || !dgamma || !dbeta) return UNCR_EINVAL;
    hipLaunchKernelGGL(inconv_bwd_finish_bn_kernel, dim3((Cout + 63) / 64), dim3(64), 0, stream, R, (const float2*)part, NP, momtot, W,
                       bias, gamma, save_mean, save_rstd, N, Cin, Cout, dW, db, dgamma, dbeta);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
