// Normalisation statistics -> per-(frame,channel) affine coefficients.
//
// Every GroupNorm / BatchNorm on the path (uncrtaints.py:16-22, utae.py:470-473) is applied by the
// CONSUMING kernel as  u = A[n,c]*h + B[n,c]; the producers only emit per-block partial sums
// (sum h, sum h^2) per (n,c).  These kernels turn the partials into A/B (forward) and into the three
// backward coefficients  dh = C1*du + C2*h + C3  (+ d gamma, d beta).  Final combines run in fp64 in a
// fixed order, so results are deterministic.
#include "common.h"
#include "bn_inline.h"

// ---------------------------------------------------------------------------------------------
// forward: GroupNorm.  grid = N*G blocks; block reduces the contiguous range of Cg*NP partials.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_fwd_kernel(
    const float2* __restrict__ part, int NP, int C, int G, int P, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ coefA, float* __restrict__ coefB,
    float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ ub, float* __restrict__ hb,
    const T* __restrict__ xsrc /* nullable: the tensor itself, for statistics sets far from zero (bn_inline.h) */, size_t pstride) {
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const float2* src = part + ((size_t)n * C + (size_t)g * Cg) * NP;
    const int cnt = Cg * NP;
    // ub (optional): per-plane upper bound on |A*h + B|.  |h| <= sqrt(sum of the block's h^2) for every element of a block, so
    // the largest partial sum of squares of a plane bounds its values (a few bits loose, rigorous); the consumer GEMM derives the
    // power-of-two scale of its fp16 operand split from these (pw_gemm_split.hip).  Non-negative floats order like their bits.
    __shared__ unsigned smax[256];
    if (ub) {
        for (int c = threadIdx.x; c < Cg; c += 256) smax[c] = 0u;
        __syncthreads();
    }
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const float2 v = src[i];
        s += (double)v.x;
        ss += (double)v.y;
        if (ub) atomicMax(&smax[i / NP], __float_as_uint(v.y));
    }
    __shared__ double red[8];
    __shared__ float sh_mean, sh_rstd;
    __shared__ int sh_flat;
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = s; red[2 * w + 1] = ss; }
    __syncthreads();
    __shared__ double sh_md, sh_vd;
    __shared__ int sh_need;
    const double M = (double)Cg * (double)P;
    if (threadIdx.x == 0) {
        double S = 0, SS = 0;
        for (int i = 0; i < 4; ++i) { S += red[2 * i]; SS += red[2 * i + 1]; }
        const double mean = S / M;
        double var = SS / M - mean * mean;
        if (var < 0) var = 0;
        sh_md = mean;
        sh_vd = var;
        sh_need = (xsrc && mean != 0.0 && var <= ldexp(mean * mean, UNCR_REPAIR_SHIFT)) ? 1 : 0;
    }
    __syncthreads();
    const bool repaired = sh_need != 0;          // block-uniform
    if (repaired) {                              // a set >= 32 sigma from zero: second moment about the mean, from the tensor itself
        const float m0 = (float)sh_md;
        double S1, S2;
        centred_partials<T>(xsrc + ((size_t)n * C + (size_t)g * Cg) * pstride, Cg, pstride, P, m0, threadIdx.x, 256, S1, S2);
        S1 = wave_sum_d(S1);
        S2 = wave_sum_d(S2);
        __syncthreads();
        if (lane == 0) { red[2 * w] = S1; red[2 * w + 1] = S2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            S1 = S2 = 0.0;
            for (int i = 0; i < 4; ++i) { S1 += red[2 * i]; S2 += red[2 * i + 1]; }
            const double e1 = S1 / M;
            sh_md = (double)m0 + e1;
            sh_vd = fmax(S2 / M - e1 * e1, 0.0);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = sh_md;
        double var = sh_vd;
        // InstanceNorm (one plane per statistics set) over a CONSTANT plane -- a zero-padded date behind in_conv: every plane of that
        // frame holds its channel's bias, and every later plane of the frame a constant again: exact arithmetic (and the reference,
        // whose Welford mean of a constant is that constant, uncrtaints.py:16-22) gives (h - mean) * rstd = 0 there, so the frame's
        // activations are zero and it contributes nothing to any weight gradient.  The raw moments of fp32 slot sums cannot tell
        // var = 0 from var ~ 1e-6 mean^2, and A*h + B with a rounded B would leave 1e-6-sized activations behind, which the frame's
        // gradient (amplified by rstd = 1/sqrt(eps) = 316 per norm) turns into O(1) errors of the encoder's weight gradients.  A plane
        // whose variance is below the resolution of its statistics (2^-17 mean^2) is therefore TREATED as constant: var = 0 and the
        // coefficients of the exact result, A = 0, B = beta.
        // (a recomputed variance resolves 2^-34 mean^2: a constant plane gives exactly 0 there)
        const bool flat = Cg == 1 && var <= ldexp(mean * mean, repaired ? -34 : -17);
        if (flat) var = 0;
        sh_flat = flat ? 1 : 0;
        sh_mean = (float)mean;
        sh_rstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[n * G + g] = sh_mean;
        save_rstd[n * G + g] = sh_rstd;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cg; c += 256) {
        const int ch = g * Cg + c;
        const float a = sh_flat ? 0.f : gamma[ch] * sh_rstd;
        const float b = sh_flat ? beta[ch] : beta[ch] - sh_mean * a;
        coefA[n * C + ch] = a;
        coefB[n * C + ch] = b;
        if (ub) ub[n * C + ch] = fmaf(fabsf(a), sqrtf(__uint_as_float(smax[c])), fabsf(b));
        if (hb) hb[n * C + ch] = sqrtf(__uint_as_float(smax[c]));          // the raw bound on |h| itself (needs ub)
    }
}

// forward: BatchNorm (train: batch statistics + running update; eval: running statistics). grid = C.
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(
    const float2* __restrict__ part, int NP, int N, int C, int P, int train, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ coefA, float* __restrict__ coefB,
    float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ ub, float* __restrict__ hb,
    const T* __restrict__ src /* nullable, as in gn_finalize_fwd_kernel */, size_t pstride) {
    const int c = blockIdx.x;
    __shared__ double red[8];
    __shared__ float sh_mean, sh_rstd;
    // ub: see gn_finalize_fwd_kernel; one bound per frame (more than 256 frames share one)
    __shared__ unsigned smax[256];
    const bool per_frame = N <= 256;
    if (ub) {
        for (int i = threadIdx.x; i < 256; i += 256) smax[i] = 0u;
        __syncthreads();
        if (!train)      // eval mode: the partials (of the same tensor) only serve the bound
            for (int i = threadIdx.x; i < N * NP; i += 256) {
                const int n = i / NP, j = i - n * NP;
                atomicMax(&smax[per_frame ? n : 0], __float_as_uint(part[((size_t)n * C + c) * NP + j].y));
            }
    }
    if (train) {
        double s = 0.0, ss = 0.0;
        const int cnt = N * NP;
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const int n = i / NP, j = i - n * NP;
            const float2 v = part[((size_t)n * C + c) * NP + j];
            s += (double)v.x;
            ss += (double)v.y;
            if (ub) atomicMax(&smax[per_frame ? n : 0], __float_as_uint(v.y));
        }
        s = wave_sum_d(s);
        ss = wave_sum_d(ss);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { red[2 * w] = s; red[2 * w + 1] = ss; }
        __syncthreads();
        __shared__ double sh_md, sh_vd;
        __shared__ int sh_need;
        const double M = (double)N * (double)P;
        if (threadIdx.x == 0) {
            double S = 0, SS = 0;
            for (int i = 0; i < 4; ++i) { S += red[2 * i]; SS += red[2 * i + 1]; }
            const double mean = S / M;
            sh_md = mean;
            sh_vd = fmax(SS / M - mean * mean, 0.0);
            sh_need = (src && mean != 0.0 && sh_vd <= ldexp(mean * mean, UNCR_REPAIR_SHIFT)) ? 1 : 0;
        }
        __syncthreads();
        if (sh_need) {                               // block-uniform: a channel >= 32 sigma from zero (bn_inline.h)
            const float m0 = (float)sh_md;
            double S1, S2;
            centred_partials<T>(src + (size_t)c * pstride, N, (size_t)C * pstride, P, m0, threadIdx.x, 256, S1, S2);
            S1 = wave_sum_d(S1);
            S2 = wave_sum_d(S2);
            __syncthreads();
            if (lane == 0) { red[2 * w] = S1; red[2 * w + 1] = S2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                S1 = S2 = 0.0;
                for (int i = 0; i < 4; ++i) { S1 += red[2 * i]; S2 += red[2 * i + 1]; }
                const double e1 = S1 / M;
                sh_md = (double)m0 + e1;
                sh_vd = fmax(S2 / M - e1 * e1, 0.0);
            }
        }
        if (threadIdx.x == 0) {
            const double mean = sh_md, var = sh_vd;
            sh_mean = (float)mean;
            sh_rstd = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unb = var * (M / (M > 1 ? M - 1 : 1));
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
        }
    } else if (threadIdx.x == 0) {
        sh_mean = running_mean[c];
        sh_rstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    __syncthreads();
    if (threadIdx.x == 0) { save_mean[c] = sh_mean; save_rstd[c] = sh_rstd; }
    const float a = gamma[c] * sh_rstd;
    const float b = beta[c] - sh_mean * a;
    for (int n = threadIdx.x; n < N; n += 256) {
        coefA[n * C + c] = a;
        coefB[n * C + c] = b;
        if (ub) ub[n * C + c] = fmaf(fabsf(a), sqrtf(__uint_as_float(smax[per_frame ? n : 0])), fabsf(b));
        if (hb) hb[n * C + c] = sqrtf(__uint_as_float(smax[per_frame ? n : 0]));
    }
}

// ---------------------------------------------------------------------------------------------
// backward.  part holds (S1, S2) = (sum_p du, sum_p du*h) per (n,c).
//   GN  : grid = G blocks (loop over n so d gamma / d beta come out in a fixed order)
//   BN  : grid = C blocks
// dh = C1*du + C2*h + C3 with  C1 = r*gamma, C2 = -r^2*m2, C3 = r*(-m1 + r*mu*m2),
//   m1 = mean_group(gamma*du), m2 = mean_group(gamma*du*hhat).
// ---------------------------------------------------------------------------------------------
// grid = N*G blocks; per-(n,c) d gamma / d beta contributions go to scratch[2][N][C], reduced over n by
// gn_dgb_reduce_kernel in a fixed order (deterministic).
__global__ __launch_bounds__(256) void gn_finalize_bwd_kernel(
    const float2* __restrict__ part, int NP, int N, int C, int G, int P, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_rstd, float* __restrict__ c1,
    float* __restrict__ c2, float* __restrict__ c3, float* __restrict__ cmu, float* __restrict__ scratch, int centered) {
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;   // <= 256
    __shared__ double s1[256], s2[256];
    __shared__ double sm1, sm2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // per-channel sums over the NP partials: one wave per channel, round-robin
    for (int c = w; c < Cg; c += 4) {
        const float2* src = part + ((size_t)n * C + g * Cg + c) * NP;
        double a = 0.0, b = 0.0;
        for (int j = lane; j < NP; j += 64) {
            const float2 v = src[j];
            a += (double)v.x;
            b += (double)v.y;
        }
        a = wave_sum_d(a);
        b = wave_sum_d(b);
        if (lane == 0) { s1[c] = a; s2[c] = b; }
    }
    __syncthreads();
    const double mu = (double)save_mean[n * G + g], r = (double)save_rstd[n * G + g];
    const double mus = centered ? 0.0 : mu;   // centered partials already hold sum du*(h - mean)
    if (w == 0) {
        double a = 0.0, b = 0.0;
        for (int c = lane; c < Cg; c += 64) {
            const double gm = (double)gamma[g * Cg + c];
            a += gm * s1[c];
            b += gm * r * (s2[c] - mus * s1[c]);
        }
        a = wave_sum_d(a);
        b = wave_sum_d(b);
        if (lane == 0) {
            const double M = (double)Cg * (double)P;
            sm1 = a / M;
            sm2 = b / M;
        }
    }
    __syncthreads();
    if (threadIdx.x < Cg) {
        const int c = threadIdx.x, ch = g * Cg + c;
        const double gm = (double)gamma[ch];
        c1[n * C + ch] = (float)(r * gm);
        c2[n * C + ch] = (float)(-r * r * sm2);
        // cmu given: the centred form dh = C1*du + C2*(h - mu) + C3 with C3 = -r*m1.  The raw form's constant
        // r*(-m1 + r*mu*m2) is rounded to fp32 with an error ~ eps*|C2*mu|: a per-plane offset of dh that adds up coherently
        // in every sum over the plane (bias-type gradients downstream lose |mu|/std * sqrt(P) digits).
        if (cmu) { c3[n * C + ch] = (float)(-r * sm1); cmu[n * C + ch] = (float)mu; }
        else c3[n * C + ch] = (float)(r * (-sm1 + r * mu * sm2));
        scratch[(size_t)n * C + ch] = (float)(r * (s2[c] - mus * s1[c]));
        scratch[(size_t)N * C + (size_t)n * C + ch] = (float)s1[c];
    }
}

__global__ __launch_bounds__(256) void gn_dgb_reduce_kernel(const float* __restrict__ scratch, int N, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) {
        a += (double)scratch[(size_t)n * C + c];
        b += (double)scratch[(size_t)N * C + (size_t)n * C + c];
    }
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
}

__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(
    const float2* __restrict__ part, int NP, int N, int C, int P, int train, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_rstd, float* __restrict__ c1,
    float* __restrict__ c2, float* __restrict__ c3, float* __restrict__ cmu, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int centered) {
    const int c = blockIdx.x;
    __shared__ double red[8];
    __shared__ float k1, k2, k3, kmu;
    double a = 0.0, b = 0.0;
    const int cnt = N * NP;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int n = i / NP, j = i - n * NP;
        const float2 v = part[((size_t)n * C + c) * NP + j];
        a += (double)v.x;
        b += (double)v.y;
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S1 = 0, S2 = 0;
        for (int i = 0; i < 4; ++i) { S1 += red[2 * i]; S2 += red[2 * i + 1]; }
        const double mu = (double)save_mean[c], r = (double)save_rstd[c], gm = (double)gamma[c];
        const double dg = r * (centered ? S2 : S2 - mu * S1);   // centered: S2 = sum du*(h - mean)
        dgamma[c] = (float)dg;
        dbeta[c] = (float)S1;
        k1 = (float)(r * gm);
        if (train) {
            const double M = (double)N * (double)P;
            const double m1 = gm * S1 / M, m2 = gm * dg / M;
            k2 = (float)(-r * r * m2);
            k3 = cmu ? (float)(-r * m1) : (float)(r * (-m1 + r * mu * m2));      // centred / raw form (see the GroupNorm kernel)
        } else {
            k2 = 0.f;
            k3 = 0.f;
        }
        kmu = (float)mu;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        c1[n * C + c] = k1;
        c2[n * C + c] = k2;
        c3[n * C + c] = k3;
        if (cmu) cmu[n * C + c] = kmu;
    }
}

// ---------------------------------------------------------------------------------------------
// PreNorm backward statistics and the pw1 weight gradient from per-frame raw products (MBConv backward).
//   R[n][k][c] = sum_p du1n[n,k,p] * x[n,c,p]        (weight-gradient GEMM with the UN-normalised block input x)
// da = W1^T du1n is linear in du1n, so the sums its norm backward needs take no pass over da:
//   sum_p da[n,c,p]          = sum_k W1[k,c] * S[n,k],        S[n,k] = sum_p du1n[n,k,p]
//   sum_p da[n,c,p]*x[n,c,p] = sum_k W1[k,c] * R[n,k,c]
// and, with pw1's input a = A0*x + B0,   dW1[k,c] = sum_n A0[n,c]*R[n,k,c] + B0[n,c]*S[n,k].
// S needs no pass either: du1n = c1*du1 + c2*h1 + c3  =>  S = c1*sum du1 + c2*sum h1 + c3*P, from the partial sums the
// depthwise backward (part_b) and the forward pw1 GEMM (part_f) wrote.  fp64, fixed order.
// One launch: the block of (4 rows k) x (32 columns c) reduces the weight-gradient GEMM's per-block partials to R itself (what a
// separate reduction launch did before), frame by frame, derives the S values of its rows, and leaves dW1 and one statistics
// partial per (n, c).  grid = (C/32, Ch/4) -- 256 blocks at the model's shape, the reduction is bound by what one CU can pull --,
// block = 1024 = 32 c x 4 k x 8 slices of the partial range (latency-bound: loads in flight).
// (Measured and dropped: one block per frame with the last of a tile's blocks adding the frames' terms.  The device-scope release
// that pattern needs writes the XCD's L2 back; 35 us per launch against 20 us for the three launches this kernel replaces.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prenorm_bwd_finish_kernel(
    const float* __restrict__ wpart, int nbx, int COP, int CIP, const float* __restrict__ W1,
    const float2* __restrict__ part_b, int NPB, const float2* __restrict__ part_f, int NPF, const float* __restrict__ c1,
    const float* __restrict__ c2, const float* __restrict__ c3, const float* __restrict__ cmu, const float* __restrict__ A0,
    const float* __restrict__ B0, float2* __restrict__ part0, float* __restrict__ dW1, int N, int Ch, int C, int P,
    const float* __restrict__ xmu) {
    // four frames per round: their loads are all in flight before the round's first barrier
    __shared__ double comb[4][8][128];
    __shared__ double sS[4][4];
    __shared__ double red[4][2][128];
    __shared__ double dwt[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cl = tid & 31, kk = (tid >> 5) & 3, sl = tid >> 7, e = tid & 127;
    const int c = blockIdx.x * 32 + cl, k = blockIdx.y * 4 + kk;
    const int NP0 = Ch / 2;
    const size_t SW = (size_t)COP * CIP;
    const double w = (double)W1[(size_t)k * C + c];
    double dw = 0.0;
    for (int n0 = 0; n0 < N; n0 += 4) {
        const int nf = min(4, N - n0);
        {       // S of the block's four rows for the round's frames: one wave per plane, the lanes walk the partial slots
            const int f = wv >> 2, r = wv & 3;
            if (f < nf) {
                const int pl = (n0 + f) * Ch + blockIdx.y * 4 + r;
                double sb = 0.0, sf = 0.0;
                for (int j = lane; j < NPB; j += 64) sb += (double)part_b[(size_t)pl * NPB + j].x;
                if (part_f)
                    for (int j = lane; j < NPF; j += 64) sf += (double)part_f[(size_t)pl * NPF + j].x;
                sb = wave_sum_d(sb);
                sf = wave_sum_d(sf);
                // cmu: centred coefficients, du1n = c1*du1 + c2*(h1 - mu) + c3
                if (lane == 0)
                    sS[f][r] = (double)c1[pl] * sb + (double)c2[pl] * (sf - (cmu ? (double)cmu[pl] * (double)P : 0.0)) +
                               (double)c3[pl] * (double)P;
            }
        }
        {
            const int b0 = (nbx * sl) / 8, b1 = (nbx * (sl + 1)) / 8;
            for (int f = 0; f < nf; ++f) {
                const float* src = wpart + (size_t)(n0 + f) * nbx * SW + (size_t)k * CIP + c;
                double s = 0.0;
                int b = b0;
                for (; b + 8 <= b1; b += 8) {   // 8 independent loads in flight, summed in a fixed order
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(b + j) * SW];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += (double)v[j];
                }
                for (; b < b1; ++b) s += (double)src[(size_t)b * SW];
                comb[f][sl][e] = s;
            }
        }
        __syncthreads();
        if (sl < nf) {      // slice index doubles as the frame of the round
            const int f = sl, n = n0 + f;
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) r += comb[f][q][e];
            const double R = (double)(float)r, S = sS[f][kk];
            // xmu: the products were taken on x - m (m = xmu[n, c], the plane's pivot): R holds Rc = sum du1n*(x - m), and
            // A0*(Rc + m*S) + B0*S = A0*Rc + (A0*m + B0)*S has no cancellation between two separately rounded sums when the
            // plane sits many standard deviations from zero (or is constant: InstanceNorm over a padded date); the statistics'
            // second component is then the centred sum da*(x - m)
            const double m = xmu ? (double)xmu[n * C + c] : 0.0;
            dwt[f][e] = (double)A0[n * C + c] * R + ((double)A0[n * C + c] * m + (double)B0[n * C + c]) * S;
            red[f][0][e] = w * S;
            red[f][1][e] = w * R;
        }
        __syncthreads();
        if (tid < 32 * nf) {
            const int f = tid >> 5, n = n0 + f;
            double a = 0.0, b = 0.0;
            for (int q = 0; q < 4; ++q) { a += red[f][0][q * 32 + cl]; b += red[f][1][q * 32 + cl]; }
            // each fp64 tile sum leaves as two fp32 slots (value and remainder): the finalize kernel adds the slots in fp64, so the
            // cancellation in sum_k W1[k,c]*S[n,k] (exactly zero behind a BatchNorm) survives as it did when one block held all k
            const float ah = (float)a, bh = (float)b;
            float2* dst = part0 + ((size_t)n * C + c) * NP0 + 2 * blockIdx.y;
            dst[0] = make_float2(ah, bh);
            dst[1] = make_float2((float)(a - (double)ah), (float)(b - (double)bh));
        }
        if (sl == 0)
            for (int f = 0; f < nf; ++f) dw += dwt[f][e];      // frame order
        __syncthreads();    // red / dwt / comb are rewritten by the next round
    }
    if (sl == 0) dW1[(size_t)k * C + c] = (float)dw;
}

extern "C" int uncr_prenorm_bwd_finish(const float* wpart, int nbx, int COP, int CIP, const float* W1, const float* part_b,
                                       int NPB, const float* part_f, int NPF, const float* c1, const float* c2,
                                       const float* c3, const float* cmu, const float* A0, const float* B0, float* part0,
                                       float* dW1, int N, int Ch, int C, int P, const float* xmu, hipStream_t stream) {
    if (N <= 0 || Ch <= 0 || (Ch & 3) || C <= 0 || (C & 31) || P <= 0 || nbx <= 0 || COP < Ch || CIP < C) return UNCR_ESHAPE;
    if (!wpart || !W1 || !part_b || NPB <= 0 || !c1 || !c2 || !c3 || !A0 || !B0 || !part0 || !dW1) return UNCR_EINVAL;
    if (part_f && NPF <= 0) return UNCR_EINVAL;
    if (cmu && !part_f) return UNCR_EINVAL;      // the centred form needs sum h1
    hipLaunchKernelGGL(prenorm_bwd_finish_kernel, dim3(C / 32, Ch / 4), dim3(1024), 0, stream, wpart, nbx, COP, CIP, W1,
                       (const float2*)part_b, NPB, (const float2*)part_f, NPF, c1, c2, c3, cmu, A0, B0, (float2*)part0, dW1, N, Ch,
                       C, P, xmu);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---------------------------------------------------------------------------------------------
// Synchronised BatchNorm under data parallelism ("offer sync_bn").  The per-channel sums leave the
// device-local partials as fp64 pairs, are all-reduced by the host layer (RCCL), and the coefficients are derived from
// the GLOBAL sums; d gamma / d beta stay LOCAL sums (the gradient all-reduce averages them like every other gradient).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_channel_sums_kernel(const float2* __restrict__ part, int NP, int N, int C,
                                                              double* __restrict__ sums) {
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < N * NP; i += 256) {
        const int n = i / NP, j = i - n * NP;
        const float2 v = part[((size_t)n * C + c) * NP + j];
        a += (double)v.x;
        b += (double)v.y;
    }
    __shared__ double red[8];
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[2 * c] = (red[0] + red[2]) + (red[4] + red[6]);
        sums[2 * c + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

// grid = ceil(C/256): one thread per channel (the work is a handful of flops; N coefficient copies per channel)
// synchronised BatchNorm: the LOCAL centred sums (S1, S2) = (sum (h - m0), sum (h - m0)^2) of every channel whose GLOBAL raw moments put
// it 8 sigma or more from zero (m0 = the global raw mean rounded to fp32: every rank decides and pivots alike); zeros for the others.
// The host all-reduces them like the raw sums; bn_finalize_fwd_sums_kernel then takes the statistics of those channels from them.
template <typename T>
__global__ __launch_bounds__(256) void bn_centred_sums_kernel(const double* __restrict__ sums, double M, const T* __restrict__ src,
                                                              int N, int C, int P, size_t pstride, double* __restrict__ out) {
    const int c = blockIdx.x;
    const double mean = sums[2 * c] / M;
    const double var = fmax(sums[2 * c + 1] / M - mean * mean, 0.0);
    if (!(mean != 0.0 && var <= ldexp(mean * mean, UNCR_REPAIR_SHIFT))) {       // block-uniform
        if (threadIdx.x == 0) { out[2 * c] = 0.0; out[2 * c + 1] = 0.0; }
        return;
    }
    double S1, S2;
    centred_partials<T>(src + (size_t)c * pstride, N, (size_t)C * pstride, P, (float)mean, threadIdx.x, 256, S1, S2);
    S1 = wave_sum_d(S1);
    S2 = wave_sum_d(S2);
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = S1; red[2 * w + 1] = S2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        S1 = S2 = 0.0;
        for (int i = 0; i < 4; ++i) { S1 += red[2 * i]; S2 += red[2 * i + 1]; }
        out[2 * c] = S1;
        out[2 * c + 1] = S2;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_fwd_sums_kernel(
    const double* __restrict__ sums, double M, int N, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ save_mean,
    float* __restrict__ save_rstd, const float2* __restrict__ part, int NP, float* __restrict__ ub, float* __restrict__ hb,
    const double* __restrict__ csums /* nullable: the all-reduced bn_centred_sums_kernel output */) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double mean = sums[2 * c] / M;
    double var = sums[2 * c + 1] / M - mean * mean;
    if (var < 0) var = 0;
    if (csums && mean != 0.0 && var <= ldexp(mean * mean, UNCR_REPAIR_SHIFT)) {     // the condition of bn_centred_sums_kernel
        const double e1 = csums[2 * c] / M;
        var = fmax(csums[2 * c + 1] / M - e1 * e1, 0.0);
        mean = (double)(float)mean + e1;
    }
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mf;
    save_rstd[c] = rf;
    if (running_mean) {
        const double unb = var * (M / (M > 1 ? M - 1 : 1));
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    const float a = gamma[c] * rf, b = beta[c] - mf * a;
    for (int n = 0; n < N; ++n) {
        coefA[n * C + c] = a; coefB[n * C + c] = b;
        if (ub) {      // bound on |a*h + b| over the LOCAL plane from its partial sums of squares (bn_finalize_fwd_kernel)
            float m = 0.f;
            for (int j = 0; j < NP; ++j) { const float v = part[((size_t)n * C + c) * NP + j].y; m = v > m || !(v == v) ? v : m; }
            ub[n * C + c] = fmaf(fabsf(a), sqrtf(m), fabsf(b));
            if (hb) hb[n * C + c] = sqrtf(m);
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_bwd_sums_kernel(
    const double* __restrict__ loc, const double* __restrict__ glob, double M, int N, int C,
    const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
    float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3, float* __restrict__ cmu,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int centered) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mu = (double)save_mean[c], r = (double)save_rstd[c], gm = (double)gamma[c];
    dgamma[c] = (float)(r * (centered ? loc[2 * c + 1] : loc[2 * c + 1] - mu * loc[2 * c]));
    dbeta[c] = (float)loc[2 * c];
    const double S1 = glob[2 * c], dg = r * (centered ? glob[2 * c + 1] : glob[2 * c + 1] - mu * S1);
    const double m1 = gm * S1 / M, m2 = gm * dg / M;
    const float k1 = (float)(r * gm), k2 = (float)(-r * r * m2);
    const float k3 = cmu ? (float)(-r * m1) : (float)(r * (-m1 + r * mu * m2));      // centred / raw form
    for (int n = 0; n < N; ++n) {
        c1[n * C + c] = k1; c2[n * C + c] = k2; c3[n * C + c] = k3;
        if (cmu) cmu[n * C + c] = (float)mu;
    }
}

extern "C" int uncr_bn_channel_sums(const float* part, int NP, int N, int C, double* sums, hipStream_t stream) {
    if (!part || !sums || NP <= 0 || N <= 0 || C <= 0) return UNCR_EINVAL;
    hipLaunchKernelGGL(bn_channel_sums_kernel, dim3(C), dim3(256), 0, stream, (const float2*)part, NP, N, C, sums);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_bn_finalize_fwd_sums(const double* sums, double count, int N, int C, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum,
                                         float eps, float* coefA, float* coefB, float* save_mean, float* save_rstd,
                                         const float* part, int NP, float* ub, float* hb, const double* csums,
                                         hipStream_t stream) {
    if (!sums || count <= 0 || N <= 0 || C <= 0 || !gamma || !beta || !coefA || !coefB || !save_mean || !save_rstd)
        return UNCR_EINVAL;
    if (ub && (!part || NP <= 0)) return UNCR_EINVAL;
    if (hb && !ub) return UNCR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_fwd_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, count, N, C,
                       gamma, beta, running_mean, running_var, momentum, eps, coefA, coefB, save_mean, save_rstd,
                       (const float2*)part, NP, ub, hb, csums);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_bn_centred_sums(const double* sums, double count, const void* src, int N, int C, int P, long long stride,
                                    int act, double* out, hipStream_t stream) {
    if (!sums || !src || !out || count <= 0) return UNCR_EINVAL;
    if (N <= 0 || C <= 0 || P <= 0 || stride < P) return UNCR_ESHAPE;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(bn_centred_sums_kernel<T>, dim3(C), dim3(256), 0, stream, sums, count, (const T*)src, N,
                                                  C, P, (size_t)stride, out));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_bn_finalize_bwd_sums(const double* sums_local, const double* sums_global, double count, int N,
                                         int C, const float* gamma, const float* save_mean, const float* save_rstd,
                                         float* c1, float* c2, float* c3, float* cmu, float* dgamma, float* dbeta,
                                         int centered, hipStream_t stream) {
    if (!sums_local || !sums_global || count <= 0 || N <= 0 || C <= 0 || !gamma || !save_mean || !save_rstd) return UNCR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_bwd_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums_local,
                       sums_global, count, N, C, gamma, save_mean, save_rstd, c1, c2, c3, cmu, dgamma, dbeta, centered);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_norm_finalize_fwd(const float* part, int NP, int N, int C, int groups, int P, int kind,
                                      const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float momentum, float eps, float* coefA, float* coefB,
                                      float* save_mean, float* save_rstd, float* ub, float* hb, const void* src,
                                      long long src_stride, int src_act, hipStream_t stream) {
    if (N <= 0 || C <= 0 || P <= 0) return UNCR_ESHAPE;
    if (src && (src_stride < P || (src_act != UNCR_F32 && src_act != UNCR_BF16))) return UNCR_EINVAL;
    if (!src) src_act = UNCR_F32;
    if (ub && (!part || NP <= 0)) return UNCR_EINVAL;      // the bound is taken from the partial sums of squares
    if (hb && !ub) return UNCR_EINVAL;
    if (kind == NORM_GROUP) {
        if (groups <= 0 || C % groups || !part) return UNCR_EINVAL;
        if (ub && C / groups > 256) return UNCR_ESHAPE;
        UNCR_DISPATCH_ACT(src_act, T, hipLaunchKernelGGL(gn_finalize_fwd_kernel<T>, dim3(N * groups), dim3(256), 0, stream,
                                                          (const float2*)part, NP, C, groups, P, gamma, beta, eps, coefA, coefB,
                                                          save_mean, save_rstd, ub, hb, (const T*)src, (size_t)src_stride));
    } else if (kind == NORM_BATCH_TRAIN || kind == NORM_BATCH_EVAL) {
        const int train = kind == NORM_BATCH_TRAIN;
        if (train && !part) return UNCR_EINVAL;
        if (!train && (!running_mean || !running_var)) return UNCR_EINVAL;
        UNCR_DISPATCH_ACT(src_act, T, hipLaunchKernelGGL(bn_finalize_fwd_kernel<T>, dim3(C), dim3(256), 0, stream,
                                                          (const float2*)part, NP, N, C, P, train, gamma, beta, running_mean,
                                                          running_var, momentum, eps, coefA, coefB, save_mean, save_rstd, ub, hb,
                                                          (const T*)src, (size_t)src_stride));
    } else {
        return UNCR_EINVAL;
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- InstanceNorm statistics of an ill-conditioned plane, recomputed about its mean --------------------------------------------
// (sum h, sum h^2) from fp32 slot sums resolve a plane's variance to ~1e-7 mean^2: enough while |mean| is a few sigma, not behind an
// encoder whose eval-mode BatchNorm leaves planes 100 sigma from zero (tools/fuzz_configs.py cases 542 / 743 / 526: the decoder's first
// PreNorm under decoder_norm='instance', eval output at 1.0-1.7e-4 where the CPU path sits at 1e-5).  One block per plane: a plane
// whose raw-moment variance is above 2^-6 mean^2 keeps its statistics (the block reads two floats and leaves); otherwise the block
// re-reads the plane and takes sum (h - m0), sum (h - m0)^2 about the raw mean m0 (no cancellation) with min and max, and rewrites
// mean, rstd, A, B (and the bounds, now tight: A*h + B is monotone in h).  The flat-plane rule of gn_finalize_fwd_kernel applies to
// the recomputed variance with the threshold of THAT resolution: a constant plane gives exactly 0, anything above 2^-34 mean^2 is data.
template <typename T>
__global__ __launch_bounds__(256) void instance_repair_kernel(const T* __restrict__ x, int C, int P, size_t stride,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float* __restrict__ coefA, float* __restrict__ coefB,
                                                              float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                              float* __restrict__ ub, float* __restrict__ hb) {
    const int pl = blockIdx.x;
    const float m0 = save_mean[pl], r0 = save_rstd[pl];
    const double var0 = 1.0 / ((double)r0 * (double)r0) - (double)eps;
    if (var0 > ldexp((double)m0 * (double)m0, -6)) return;             // block-uniform: well-conditioned, nothing to do
    const T* src = x + (size_t)pl * stride;
    float s1 = 0.f, s2 = 0.f, mn = INFINITY, mx = -INFINITY;
    const int P4 = P & ~3;
    for (int i = threadIdx.x * 4; i < P4; i += 1024) {
        const float4 v = widen4(ld4raw<T>(src + i));
        const float d0 = v.x - m0, d1 = v.y - m0, d2 = v.z - m0, d3 = v.w - m0;
        s1 += (d0 + d1) + (d2 + d3);
        s2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, s2))));
        mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int i = P4 + threadIdx.x; i < P; i += 256) {
        const float v = ld1<T>(src + i), d = v - m0;
        s1 += d;
        s2 = fmaf(d, d, s2);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    double S1 = wave_sum_d((double)s1), S2 = wave_sum_d((double)s2);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, m, 64));
        mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    }
    __shared__ double red[8];
    __shared__ float redm[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[2 * w] = S1; red[2 * w + 1] = S2; redm[2 * w] = mn; redm[2 * w + 1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        S1 = S2 = 0.0;
        for (int i = 0; i < 4; ++i) {
            S1 += red[2 * i]; S2 += red[2 * i + 1];
            mn = fminf(mn, redm[2 * i]); mx = fmaxf(mx, redm[2 * i + 1]);
        }
        const double e1 = S1 / (double)P;
        const double mean = (double)m0 + e1;
        double var = S2 / (double)P - e1 * e1;
        if (var < 0) var = 0;
        const bool flat = var <= ldexp(mean * mean, -34);
        if (flat) var = 0;
        const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[pl] = fm;
        save_rstd[pl] = fr;
        const int ch = pl % C;
        const float a = flat ? 0.f : gamma[ch] * fr;
        const float b = flat ? beta[ch] : beta[ch] - fm * a;
        coefA[pl] = a;
        coefB[pl] = b;
        const float hmax = fmaxf(fabsf(mn), fabsf(mx));
        // |A*h + B| over [mn, mx] peaks at an end point; the slack covers a consumer that rounds A*h before adding B
        if (ub) ub[pl] = fmaxf(fabsf(fmaf(a, mn, b)), fabsf(fmaf(a, mx, b))) + ldexpf(fmaf(fabsf(a), hmax, fabsf(b)), -21);
        if (hb) hb[pl] = hmax;
    }
}

/* InstanceNorm2d behind uncr_norm_finalize_fwd (groups == C): planes whose raw-moment variance is below 2^-6 mean^2 get their
 * statistics, coefficients and bounds recomputed from the tensor itself, centred on the mean (see instance_repair_kernel) */
extern "C" int uncr_instance_repair(const void* x, int N, int C, int P, long long stride, const float* gamma, const float* beta,
                                    float eps, float* coefA, float* coefB, float* save_mean, float* save_rstd, float* ub,
                                    float* hb, int act, hipStream_t stream) {
    if (N <= 0 || C <= 0 || P <= 0 || stride < P || (stride & 3)) return UNCR_ESHAPE;
    if (!x || !gamma || !beta || !coefA || !coefB || !save_mean || !save_rstd || (hb && !ub)) return UNCR_EINVAL;
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(instance_repair_kernel<T>, dim3(N * C), dim3(256), 0, stream, (const T*)x, C, P,
                                                  (size_t)stride, gamma, beta, eps, coefA, coefB, save_mean, save_rstd, ub, hb));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_norm_finalize_bwd(const float* part, int NP, int N, int C, int groups, int P, int kind,
                                      const float* gamma, const float* save_mean, const float* save_rstd, float* c1,
                                      float* c2, float* c3, float* cmu, float* dgamma, float* dbeta, float* scratch,
                                      int centered, hipStream_t stream) {
    if (N <= 0 || C <= 0 || P <= 0 || !part) return UNCR_ESHAPE;
    if (kind == NORM_GROUP) {
        if (groups <= 0 || C % groups || C / groups > 256 || !scratch) return UNCR_EINVAL;
        hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3(N * groups), dim3(256), 0, stream, (const float2*)part, NP, N,
                           C, groups, P, gamma, save_mean, save_rstd, c1, c2, c3, cmu, scratch, centered);
        UNCR_LAUNCH_CHECK();
        hipLaunchKernelGGL(gn_dgb_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, scratch, N, C, dgamma,
                           dbeta);
    } else if (kind == NORM_BATCH_TRAIN || kind == NORM_BATCH_EVAL) {
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(256), 0, stream, (const float2*)part, NP, N, C, P,
                           kind == NORM_BATCH_TRAIN, gamma, save_mean, save_rstd, c1, c2, c3, cmu, dgamma, dbeta, centered);
    } else {
        return UNCR_EINVAL;
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// out[n*C + c] = scale * mean of the statistics set plane (n, c) belongs to: groups > 0: mean [N*groups] (GroupNorm; groups == C:
// InstanceNorm), groups == 0: mean [C] (BatchNorm).  The per-plane pivots of the centred backward statistics and of the centred
// weight-gradient products where no per-plane array exists yet.
__global__ __launch_bounds__(256) void plane_means_kernel(const float* __restrict__ mean, int N, int C, int groups, float scale,
                                                          float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    out[i] = scale * (groups > 0 ? mean[n * groups + c / (C / groups)] : mean[c]);
}
extern "C" int uncr_plane_means(const float* mean, int N, int C, int groups, float scale, float* out, hipStream_t stream) {
    if (N <= 0 || C <= 0 || groups < 0 || (groups > 0 && C % groups)) return UNCR_ESHAPE;
    if (!mean || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(plane_means_kernel, dim3((N * C + 255) / 256), dim3(256), 0, stream, mean, N, C, groups, scale, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
