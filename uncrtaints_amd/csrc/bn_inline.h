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

// Every lane of the calling wave returns the same (A, B) of channel c.  P = pixels per plane.  store_plane: this wave stores the
// plane's A / B / ub / hb; store_channel: it also stores mean / rstd and updates the running statistics (exactly one wave per channel).
__device__ __forceinline__ void bn_fin_wave(const BnFin& f, int n, int c, int C, int P, bool store_plane, bool store_channel,
                                            float& A, float& B) {
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
    const double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0) var = 0;
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
