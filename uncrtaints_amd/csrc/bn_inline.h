// Consumer-side BatchNorm finalisation (train mode, per-replica statistics).
//
// A BatchNorm on the path is applied by the kernel that CONSUMES the normalised tensor (u = A*h + B per plane); the producer of h
// leaves (sum h, sum h^2) partials per (frame, channel) and a small finalize launch (norm.hip::bn_finalize_fwd_kernel, ~5 us on a
// dependent chain, 40 of them per training step) used to turn them into A / B.  Where the consumer works on whole (frame, channel)
// planes -- the depthwise kernels, the SE pooling pass, the closing residual -- every WAVE can do that reduction for its own channel
// itself: N * NP partial pairs (a few KB, L2 hits), fp64, fixed order, a few hundred cycles once per wave.  The separate launch and its
// dependency edge disappear from the step.  One designated wave per plane stores A / B (and the magnitude bounds ub / hb of
// uncr_norm_finalize_fwd) for the backward, one per channel stores mean / rstd and updates the running statistics.
//
// Same arithmetic as bn_finalize_fwd_kernel (torch BatchNorm2d train mode, utae.py:470-473 / uncrtaints.py:16-22): fp64 sums,
// biased variance for the normalisation, unbiased for running_var; the summation order over the partials differs (64 lanes instead of
// 256 threads), i.e. last-bit differences of an fp64 sum before it is rounded to fp32.
#pragma once
#include "common.h"

struct BnFin {
    const float2* part;      // [N*C][NP] (sum h, sum h^2) of the tensor being normalised; null = no consumer-side finalisation
    int NP, N;
    const float* gamma;      // [C]
    const float* beta;       // [C]
    float* running_mean;     // [C] or null
    float* running_var;
    float momentum, eps;
    float* coefA;            // [N*C] out
    float* coefB;            // [N*C] out
    float* save_mean;        // [C] out
    float* save_rstd;        // [C] out
    float* ub;               // [N*C] out or null: bound on |A*h + B| per plane
    float* hb;               // [N*C] out or null (needs ub): bound on |h| per plane
};

// Statistics sets far from zero.  (sum h, sum h^2) of fp32 slot sums lose a set's statistics as |mean| / sigma grows: a hidden channel
// 20 sigma from zero (tools/fuzz_configs.py case 1146: a BatchNorm-1 gamma of -0.02 leaves gelu(gamma*h + beta) ~ constant, so the
// depthwise output of that channel is its mean + 5 %) left the gamma gradient in front of it at 3e-4 where the CPU path has 1e-6.  A
// set with var <= 2^-6 mean^2 (|mean| >= 8 sigma) is therefore re-read once by the finalisation kernels: per-thread partial sums of
// (h - m0), (h - m0)^2 about the raw mean m0, `planes` planes of P valid elements, `pstep` elements apart, strided over `nthr` threads.
// Normalised tensors sit within a few sigma of zero: a set that needs the pass costs one read of its own elements by one block
// (~1 MB for a BatchNorm channel at N = 4, 256 x 256).  The consumer-side finalisation below (one WAVE per plane tile, every wave of
// the channel repeating the pass) keeps it for sets 32 sigma or more from zero, where the raw moments are unusable.
#ifndef UNCR_REPAIR_SHIFT
#define UNCR_REPAIR_SHIFT (-6)
#endif
#define UNCR_REPAIR_SHIFT_WAVE (-10)
template <typename T>
__device__ __forceinline__ void centred_partials(const T* base, int planes, size_t pstep, int P, float m0, int tid, int nthr,
                                                 double& S1, double& S2) {
    S1 = 0.0;
    S2 = 0.0;
    const bool vec = (pstep & 3) == 0 && (((size_t)base) & (4 * sizeof(T) - 1)) == 0;
    const int P4 = vec ? (P & ~3) : 0;
    for (int pl = 0; pl < planes; ++pl) {
        const T* src = base + (size_t)pl * pstep;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int i = tid * 4; i < P4; i += nthr * 4) {
            const float4 v = widen4(ld4raw<T>(src + i));
            const float d0 = v.x - m0, d1 = v.y - m0, d2 = v.z - m0, d3 = v.w - m0;
            s1 += (d0 + d1) + (d2 + d3);
            s2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, s2))));
        }
        for (int i = P4 + tid; i < P; i += nthr) {
            const float d = ld1<T>(src + i) - m0;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        S1 += (double)s1;
        S2 += (double)s2;
    }
}

// Every lane of the calling wave returns the same (A, B) of channel c.  P = pixels per plane.  store_plane: this wave stores the
// plane's A / B / ub / hb; store_channel: it also stores mean / rstd and updates the running statistics (exactly one wave per channel).
// src: the tensor being normalised ([N][C] planes of P elements), for the re-read above.
template <typename T>
__device__ __forceinline__ void bn_fin_wave(const BnFin& f, const T* src, int n, int c, int C, int P, bool store_plane,
                                            bool store_channel, float& A, float& B) {
    const int lane = threadIdx.x & 63;
    const int cnt = f.N * f.NP;
    double s = 0.0, ss = 0.0;
    unsigned mx = 0u;        // largest partial sum of squares of the wave's OWN plane (non-negative floats order like their bits)
    for (int i = lane; i < cnt; i += 64) {
        const int nn = i / f.NP, j = i - nn * f.NP;
        const float2 v = f.part[((size_t)nn * C + c) * f.NP + j];
        s += (double)v.x;
        ss += (double)v.y;
        if (nn == n) mx = max(mx, __float_as_uint(v.y));
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    const double M = (double)f.N * (double)P;
    double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0) var = 0;
    if (var <= ldexp(mean * mean, UNCR_REPAIR_SHIFT_WAVE) && mean != 0.0) {      // wave-uniform; every wave of the channel takes the same pass
        const float m0 = (float)mean;
        double S1, S2;
        centred_partials<T>(src + (size_t)c * P, f.N, (size_t)C * P, P, m0, lane, 64, S1, S2);
        S1 = wave_sum_d(S1);
        S2 = wave_sum_d(S2);
        const double e1 = S1 / M;
        mean = (double)m0 + e1;
        var = S2 / M - e1 * e1;
        if (var < 0) var = 0;
    }
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)f.eps));
    A = f.gamma[c] * rf;
    B = f.beta[c] - mf * A;
    if (store_plane) {
        if (f.ub) {
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, sft, 64));
        }
        if (lane == 0) {
            f.coefA[n * C + c] = A;
            f.coefB[n * C + c] = B;
            if (f.ub) {
                const float hm = sqrtf(__uint_as_float(mx));
                f.ub[n * C + c] = fmaf(fabsf(A), hm, fabsf(B));
                if (f.hb) f.hb[n * C + c] = hm;
            }
            if (store_channel) {
                f.save_mean[c] = mf;
                f.save_rstd[c] = rf;
                if (f.running_mean) {
                    const double unb = var * (M / (M > 1 ? M - 1 : 1));
                    f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
                    f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
                }
            }
        }
    }
}
