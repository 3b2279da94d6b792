// Adam update of the whole parameter set in ONE launch (BaseModel's optimizer, base_model.py:48: torch.optim.Adam(params, lr), default
// betas / eps, no weight decay, no amsgrad).  torch's own multi-tensor kernel takes three launches of ~26 us for the path's 91 tensors
// (0.57 M parameters, 16 MB of traffic: microseconds at HBM speed); here every block owns one chunk of one tensor, found through two small
// device tables the caller builds from the tensors' addresses:
//   desc   [n_tensors][5] int64 : param, exp_avg, exp_avg_sq (addresses), element count, address of the tensor's OWN step counter
//                                 (a float device scalar, torch.optim.Adam's state["step"])  -- fixed for the life of the optimizer state
//   chunks [n_chunks][2]  int32 : tensor index, first element
// The gradients' addresses change from step to step (autograd allocates them) and, inside a graph capture, are only known at capture
// time; they travel BY VALUE in the kernel arguments (up to ADAM_MAXT pointers, 2 KB of the 4 KB argument segment), so neither an eager
// step nor a captured one needs a host-to-device copy for them.
// Arithmetic as torch's (torch/optim/adam.py, ATen fused_adam_utils.cuh), fp32 with the bias corrections taken in fp64:
//   g += wd * p (if wd);  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) * g * g
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t is read PER TENSOR from the device scalar the caller has already incremented (so a captured graph advances it on every replay, and
// parameters that start to receive gradients later -- unfrozen layers, train_reconstruct.py:657-660 -- get their own bias corrections,
// exactly as torch.optim.Adam keeps one step count per parameter); lr is a float
// argument or, if lr_dev is given, a device scalar (a learning-rate schedule then reaches a captured graph).
#include "common.h"

#define ADAM_CHUNK 2048      // elements per block: 256 threads x 2 float4
#define ADAM_MAXT 256        // tensors per launch

struct AdamGrads { const float* g[ADAM_MAXT]; };

__global__ __launch_bounds__(256) void adam_multi_kernel(const long long* __restrict__ desc, const AdamGrads grads,
                                                         const int* __restrict__ chunks, float lr,
                                                         const float* __restrict__ lr_dev, double beta1, double beta2, float eps, float wd) {
    const int t = chunks[2 * blockIdx.x], start = chunks[2 * blockIdx.x + 1];
    float* __restrict__ p = (float*)desc[5 * t];
    const float* __restrict__ g = grads.g[t];
    float* __restrict__ m = (float*)desc[5 * t + 1];
    float* __restrict__ v = (float*)desc[5 * t + 2];
    const long long n = desc[5 * t + 3];
    const double step = (double)*(const float*)desc[5 * t + 4];
    const float bc1 = (float)(1.0 - pow(beta1, step));
    const float bc2s = (float)sqrt(1.0 - pow(beta2, step));
    const float step_size = (lr_dev ? lr_dev[0] : lr) / bc1;
    // 1 - beta in double first, as the Python side of torch.optim.Adam forms it (1 - 0.999f is 1.3e-5 away from 0.001f)
    const float b2 = (float)beta2, w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        if (wd != 0.f) gg = fmaf(wd, pp, gg);
        mm = mm + w1 * (gg - mm);
        vv = b2 * vv + w2 * gg * gg;
        const float denom = sqrtf(vv) / bc2s + eps;
        pp -= step_size * mm / denom;
    };
    const long long end = start + ADAM_CHUNK < n ? start + ADAM_CHUNK : n;
    const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (al) {
        for (long long i = start + 4 * threadIdx.x; i + 3 < end; i += 1024) {
            float4 pp = *(float4*)(p + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
            const float4 gg = *(const float4*)(g + i);
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
        }
        const long long tail = start + ((end - start) & ~3LL);
        for (long long i = tail + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (long long i = start + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

extern "C" int uncr_adam_chunk(void) { return ADAM_CHUNK; }
extern "C" int uncr_adam_max_tensors(void) { return ADAM_MAXT; }

extern "C" int uncr_adam_step(const long long* desc, const long long* grads_host, int n_tensors, const int* chunks, int n_chunks,
                              float lr, const float* lr_dev, double beta1, double beta2, float eps, float weight_decay,
                              hipStream_t stream) {
    if (!desc || !grads_host || !chunks || n_chunks <= 0) return UNCR_EINVAL;
    if (n_tensors <= 0 || n_tensors > ADAM_MAXT) return UNCR_ESHAPE;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.f)) return UNCR_EINVAL;
    AdamGrads gs;
    for (int i = 0; i < ADAM_MAXT; ++i) gs.g[i] = i < n_tensors ? (const float*)grads_host[i] : nullptr;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, stream, desc, gs, chunks, lr, lr_dev, beta1, beta2, eps,
                       weight_decay);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
