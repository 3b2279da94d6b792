// Squeeze-and-excitation MLP (uncrtaints.py:82-97) on the pooled [N,C] vectors: tiny, latency-bound.
//   fwd : pooled = mean_p gelu(norm(h2));  s = sigmoid(W2 * gelu(W1 * pooled))
//   bwd : from the per-frame products G[n] = sum_p dh3[n,:,p] (x) g2[n,:,p]  (weight-gradient GEMM,
//         un-scaled operand) both  ds[n,c] = sum_co Wpw[co,c] G[n,co,c]  and
//         dWpw[co,c] = sum_n s[n,c] G[n,co,c]  follow without another pass over the activations.
#include "common.h"

// grid = N, block = 1024 (latency-bound: 16 waves for every phase).  C <= 256, R <= 64 (wider: the *_any kernels further down).  Dynamic LDS: W2 staged with coalesced reads
// ([C][R + 1] floats) so that thread c's sequential dot product -- same order as a plain row walk -- reads LDS, not 64 scattered
// cache lines per step.
__global__ __launch_bounds__(1024) void se_mlp_fwd_kernel(const float2* __restrict__ pool_part, int NP, int C, int R,
                                                          int P, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ pooled,
                                                          float* __restrict__ hid_pre, float* __restrict__ s) {
    const int n = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float w2s[];          // [C][R + 1]
    __shared__ float sp[256], sh[64];
    __shared__ double comb[4][256];
    for (int i = tid; i < C * R; i += 1024) w2s[(i / R) * (R + 1) + (i % R)] = W2[i];      // consumed after two barriers
    {
        // partial sums of the pooling pass: 16 consecutive lanes walk one channel's slots (128 contiguous bytes per step), four
        // slices of the slot range per channel as before; lanes combined in a fixed order, everything in fp64
        const int l16 = tid & 15, grp = tid >> 4;              // 64 groups of 16 lanes
        for (int c = grp; c < C; c += 64) {
            const float2* src = pool_part + ((size_t)n * C + c) * NP;
            double a = 0.0;
            for (int j = l16; j < NP; j += 16) a += (double)src[j].x;
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) a += __shfl_xor(a, m, 16);
            if (l16 == 0) comb[0][c] = a;
        }
    }
    __syncthreads();
    if (tid < C) {
        const float m = (float)(comb[0][tid] / (double)P);
        sp[tid] = m;
        pooled[n * C + tid] = m;
    }
    __syncthreads();
    // hidden: one wave handles rows round-robin
    const int lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < R; j += 16) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(W1[j * C + c], sp[c], a);
        a = wave_sum(a);
        if (lane == 0) {
            hid_pre[n * R + j] = a;
            sh[j] = gelu_f(a);
        }
    }
    __syncthreads();
    if (tid < C) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(w2s[tid * (R + 1) + j], sh[j], a);
        s[n * C + tid] = sigmoid_f(a);
    }
}

// grid = N.  Produces per-frame ds_pre[n,c], dhid_pre[n,j] and the pass-B coefficient dpool_px[n,c].
__global__ __launch_bounds__(1024) void se_mlp_bwd_frame_kernel(
    const float* __restrict__ G, const float* __restrict__ Wpw, int Co, int C, int R, int P,
    const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ s,
    const float* __restrict__ hid_pre, float* __restrict__ ds_pre, float* __restrict__ dhid_pre,
    float* __restrict__ dpool_px) {
    const int n = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float w2s[];          // [C][R + 1]: W2 staged with coalesced reads for the column walks below
    __shared__ float sds[256], sdh[64];
    __shared__ double comb[4][256];
    for (int i = tid; i < C * R; i += 1024) w2s[(i / R) * (R + 1) + (i % R)] = W2[i];
    {
        // block = 1024: 4 slices of the co range per channel, combined in a fixed order
        const int c = tid & 255, sl = tid >> 8;
        double a = 0.0;
        if (c < C) {
            const float* g = G + (size_t)n * Co * C + c;
            const int co1 = (Co * (sl + 1)) / 4;
            int co = (Co * sl) / 4;
            for (; co + 8 <= co1; co += 8) {       // 16 independent loads in flight, fixed summation order
                float wv[8], gv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { wv[j] = Wpw[(co + j) * C + c]; gv[j] = g[(size_t)(co + j) * C]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) a += (double)wv[j] * (double)gv[j];
            }
            for (; co < co1; ++co) a += (double)Wpw[co * C + c] * (double)g[(size_t)co * C];
        }
        comb[sl][c] = a;
    }
    __syncthreads();
    if (tid < C) {
        const double a = comb[0][tid] + comb[1][tid] + comb[2][tid] + comb[3][tid];
        const float sv = s[n * C + tid];
        const float d = (float)a * sv * (1.f - sv);
        sds[tid] = d;
        ds_pre[n * C + tid] = d;
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < R; j += 16) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(w2s[c * (R + 1) + j], sds[c], a);
        a = wave_sum(a);
        if (lane == 0) {
            const float d = a * gelu_grad_f(hid_pre[n * R + j]);
            sdh[j] = d;
            dhid_pre[n * R + j] = d;
        }
    }
    __syncthreads();
    if (tid < C) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(W1[j * C + tid], sdh[j], a);
        dpool_px[n * C + tid] = a / (float)P;
    }
}

// grid = Co + R blocks, block = 256 (thread = channel c).
//   blocks [0,Co)   : dWpw[co][c] = sum_n s[n,c] G[n,co,c]
//   blocks [Co,Co+R): j = b-Co:  dW1[j][c] = sum_n dhid_pre[n,j] pooled[n,c];  dW2[c][j] = sum_n ds_pre[n,c] gelu(hid_pre[n,j])
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ G, int N, int Co, int C, int R,
                                                       const float* __restrict__ s, const float* __restrict__ pooled,
                                                       const float* __restrict__ hid_pre,
                                                       const float* __restrict__ ds_pre,
                                                       const float* __restrict__ dhid_pre, float* __restrict__ dWpw,
                                                       float* __restrict__ dW1, float* __restrict__ dW2) {
    const int c = blockIdx.y * 256 + threadIdx.x;      // grid.y = ceil(C / 256)
    if (c >= C) return;
    if ((int)blockIdx.x < Co) {
        const int co = blockIdx.x;
        double a = 0.0;
        for (int n = 0; n < N; ++n) a += (double)s[n * C + c] * (double)G[((size_t)n * Co + co) * C + c];
        dWpw[co * C + c] = (float)a;
    } else {
        const int j = blockIdx.x - Co;
        double a = 0.0, b = 0.0;
        for (int n = 0; n < N; ++n) {
            a += (double)dhid_pre[n * R + j] * (double)pooled[n * C + c];
            b += (double)ds_pre[n * C + c] * (double)gelu_f(hid_pre[n * R + j]);
        }
        dW1[j * C + c] = (float)a;
        dW2[c * R + j] = (float)b;
    }
}

// ---- any width (C > 256 or R > 64: MBConv blocks wider than 128 channels) ----
// The same arithmetic without the staging that the 256-channel kernels are tuned around: grid = N, block = 256, channels and hidden
// units walked in strides of the block, weights read from global memory, fp64 where the kernels above use it.  Dynamic LDS:
// [C] + [R] floats.
__global__ __launch_bounds__(256) void se_mlp_fwd_any_kernel(const float2* __restrict__ pool_part, int NP, int C, int R, int P,
                                                             const float* __restrict__ W1, const float* __restrict__ W2,
                                                             float* __restrict__ pooled, float* __restrict__ hid_pre,
                                                             float* __restrict__ s) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    extern __shared__ float sm[];
    float* sp = sm;          // [C]
    float* sh = sm + C;      // [R]
    for (int c = tid; c < C; c += 256) {
        const float2* src = pool_part + ((size_t)n * C + c) * NP;
        double a = 0.0;
        for (int j = 0; j < NP; ++j) a += (double)src[j].x;
        const float m = (float)(a / (double)P);
        sp[c] = m;
        pooled[n * C + c] = m;
    }
    __syncthreads();
    for (int j = wv; j < R; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(W1[j * C + c], sp[c], a);
        a = wave_sum(a);
        if (lane == 0) {
            hid_pre[n * R + j] = a;
            sh[j] = gelu_f(a);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(W2[c * R + j], sh[j], a);
        s[n * C + c] = sigmoid_f(a);
    }
}

__global__ __launch_bounds__(256) void se_mlp_bwd_frame_any_kernel(
    const float* __restrict__ G, const float* __restrict__ Wpw, int Co, int C, int R, int P, const float* __restrict__ W1,
    const float* __restrict__ W2, const float* __restrict__ s, const float* __restrict__ hid_pre, float* __restrict__ ds_pre,
    float* __restrict__ dhid_pre, float* __restrict__ dpool_px) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    extern __shared__ float sm[];
    float* sds = sm;         // [C]
    float* sdh = sm + C;     // [R]
    for (int c = tid; c < C; c += 256) {
        const float* g = G + (size_t)n * Co * C + c;
        double a = 0.0;
        for (int co = 0; co < Co; ++co) a += (double)Wpw[co * C + c] * (double)g[(size_t)co * C];
        const float sv = s[n * C + c];
        const float d = (float)a * sv * (1.f - sv);
        sds[c] = d;
        ds_pre[n * C + c] = d;
    }
    __syncthreads();
    for (int j = wv; j < R; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(W2[c * R + j], sds[c], a);
        a = wave_sum(a);
        if (lane == 0) {
            const float d = a * gelu_grad_f(hid_pre[n * R + j]);
            sdh[j] = d;
            dhid_pre[n * R + j] = d;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(W1[j * C + c], sdh[j], a);
        dpool_px[n * C + c] = a / (float)P;
    }
}

// dynamic LDS above 64 KB (C = 256 with R = 64: 66.6 KB) needs the opt-in; done once per process
static void se_lds_optin() {
    static bool done = false;
    if (done) return;
    hipFuncSetAttribute((const void*)se_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)se_mlp_bwd_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    done = true;
}

extern "C" int uncr_se_mlp_fwd(const float* pool_part, int NP, int N, int C, int R, int P, const float* W1,
                               const float* W2, float* pooled, float* hid_pre, float* s, hipStream_t stream) {
    if (C <= 0 || R <= 0 || N <= 0 || (size_t)(C + R) * sizeof(float) > 64 * 1024) return UNCR_ESHAPE;
    if (C > 256 || R > 64) {
        hipLaunchKernelGGL(se_mlp_fwd_any_kernel, dim3(N), dim3(256), (size_t)(C + R) * sizeof(float), stream, (const float2*)pool_part,
                           NP, C, R, P, W1, W2, pooled, hid_pre, s);
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
    se_lds_optin();
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(N), dim3(1024), (size_t)C * (R + 1) * sizeof(float), stream, (const float2*)pool_part,
                       NP, C, R, P, W1, W2, pooled, hid_pre, s);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_se_mlp_bwd(const float* G, const float* Wpw, int N, int Co, int C, int R, int P,
                               const float* W1, const float* W2, const float* s, const float* pooled,
                               const float* hid_pre, float* ds_pre, float* dhid_pre, float* dpool_px, float* dWpw,
                               float* dW1, float* dW2, hipStream_t stream) {
    if (C <= 0 || R <= 0 || N <= 0 || (size_t)(C + R) * sizeof(float) > 64 * 1024) return UNCR_ESHAPE;
    if (C > 256 || R > 64) {
        hipLaunchKernelGGL(se_mlp_bwd_frame_any_kernel, dim3(N), dim3(256), (size_t)(C + R) * sizeof(float), stream, G, Wpw, Co, C, R,
                           P, W1, W2, s, hid_pre, ds_pre, dhid_pre, dpool_px);
    } else {
        se_lds_optin();
        hipLaunchKernelGGL(se_mlp_bwd_frame_kernel, dim3(N), dim3(1024), (size_t)C * (R + 1) * sizeof(float), stream, G, Wpw, Co, C, R,
                           P, W1, W2, s, hid_pre, ds_pre, dhid_pre, dpool_px);
    }
    UNCR_LAUNCH_CHECK();
    if (!dWpw && !dW1 && !dW2) return UNCR_OK;      // the weight gradients follow later, in uncr_mbconv_param_grads
    hipLaunchKernelGGL(se_wgrad_kernel, dim3(Co + R, (C + 255) / 256), dim3(256), 0, stream, G, N, Co, C, R, s, pooled, hid_pre,
                       ds_pre, dhid_pre, dWpw, dW1, dW2);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// The parameter-gradient reductions of one MBConv backward that sit on no critical path, in ONE launch (round 4: were two, 5 us each
// plus a launch boundary, six times per step): blocks [0, Co + R) = se_wgrad_kernel's (dW of pw2 from the per-frame products, the two
// SE weight gradients), blocks [Co + R, Co + R + Cdw) = the depthwise weight gradient dWdw[c][tap] from dw_bwd's per-tile partials
// (dwconv.hip::dw_wgrad_reduce_kernel's arithmetic: fp64, the same 7-slice fixed order).
__global__ __launch_bounds__(256) void mbconv_param_grads_kernel(const float* __restrict__ G, int N, int Co, int C, int R,
                                                                 const float* __restrict__ s, const float* __restrict__ pooled,
                                                                 const float* __restrict__ hid_pre, const float* __restrict__ ds_pre,
                                                                 const float* __restrict__ dhid_pre, float* __restrict__ dWpw,
                                                                 float* __restrict__ dW1, float* __restrict__ dW2,
                                                                 const float* __restrict__ dw_part, int Cdw, int NPT,
                                                                 float* __restrict__ dwdw) {
    const int b = blockIdx.x;
    if (b < Co + R) {           // block-uniform
        const int c = blockIdx.y * 256 + threadIdx.x;  // grid.y = ceil(C / 256)
        if (c >= C) return;
        if (b < Co) {
            double a = 0.0;
            for (int n = 0; n < N; ++n) a += (double)s[n * C + c] * (double)G[((size_t)n * Co + b) * C + c];
            dWpw[b * C + c] = (float)a;
        } else {
            const int j = b - Co;
            double a = 0.0, bb = 0.0;
            for (int n = 0; n < N; ++n) {
                a += (double)dhid_pre[n * R + j] * (double)pooled[n * C + c];
                bb += (double)ds_pre[n * C + c] * (double)gelu_f(hid_pre[n * R + j]);
            }
            dW1[j * C + c] = (float)a;
            dW2[c * R + j] = (float)bb;
        }
        return;
    }
    if (blockIdx.y) return;     // (the depthwise blocks serve every channel from row 0 of the grid)
    const int c = b - (Co + R), lane = threadIdx.x;
    const int tap = lane % 9, sl = lane / 9;           // 7 slices x 9 taps = 63 lanes
    __shared__ double comb[7][9];
    if (lane < 63) {
        const int cnt = N * NPT;
        const int i1 = (cnt * (sl + 1)) / 7;
        int i = (cnt * sl) / 7;
        double acc = 0.0;
        auto at = [&](int q) {
            const int n = q / NPT, j = q - n * NPT;
            return dw_part[(((size_t)n * Cdw + c) * NPT + j) * 9 + tap];
        };
        for (; i + 4 <= i1; i += 4) {
            const float v0 = at(i), v1 = at(i + 1), v2 = at(i + 2), v3 = at(i + 3);
            acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
        }
        for (; i < i1; ++i) acc += (double)at(i);
        comb[sl][tap] = acc;
    }
    __syncthreads();
    if (lane < 9) {
        double acc = 0.0;
        for (int q = 0; q < 7; ++q) acc += comb[q][lane];
        dwdw[c * 9 + lane] = (float)acc;
    }
}

extern "C" int uncr_mbconv_param_grads(const float* G, int N, int Co, int C, int R, const float* s, const float* pooled,
                                       const float* hid_pre, const float* ds_pre, const float* dhid_pre, float* dWpw, float* dW1,
                                       float* dW2, const float* dw_part, int Cdw, int NPT, float* dwdw, hipStream_t stream) {
    if (C <= 0 || R <= 0 || N <= 0 || Co <= 0 || Cdw <= 0 || NPT <= 0) return UNCR_ESHAPE;
    if (!G || !s || !pooled || !hid_pre || !ds_pre || !dhid_pre || !dWpw || !dW1 || !dW2 || !dw_part || !dwdw) return UNCR_EINVAL;
    hipLaunchKernelGGL(mbconv_param_grads_kernel, dim3(Co + R + Cdw, (C + 255) / 256), dim3(256), 0, stream, G, N, Co, C, R, s, pooled, hid_pre, ds_pre,
                       dhid_pre, dWpw, dW1, dW2, dw_part, Cdw, NPT, dwdw);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
