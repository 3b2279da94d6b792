// Evaluation metrics right behind the path: img_metrics of model/src/learning/metrics.py:20-63
// (RMSE, MAE, PSNR, spectral angle, SSIM with the 11x11 Gaussian window of util/pytorch_ssim/__init__.py:17-73,
// nan-aware error / variance statistics incl. the pixel-wise maps) as four small kernels on tensors that are
// already in HBM; every cross-block combination is done in fp64 in a fixed order.
#include "common.h"

#define MT_NPART 9   // per block: sum se, sum ae, sum sam, nan-aware sum err, sum ae, sum se, count, sum var, count var

// one thread per (b, pixel): loops over the channels (spectral angle needs the channel sums of the pixel)
__global__ __launch_bounds__(256) void metrics_point_kernel(const float* __restrict__ targ, const float* __restrict__ pred,
                                                            const float* __restrict__ var, float* __restrict__ part,
                                                            int C, int P) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    float v[MT_NPART];
#pragma unroll
    for (int i = 0; i < MT_NPART; ++i) v[i] = 0.f;
    if (p < P) {
        float tp = 0.f, tt = 0.f, pp = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t o = ((size_t)b * C + c) * P + p;
            const float t = targ[o], q = pred[o], e = t - q;
            v[0] += e * e;
            v[1] += fabsf(e);
            tp = fmaf(t, q, tp); tt = fmaf(t, t, tt); pp = fmaf(q, q, pp);
            if (e == e) { v[3] += e; v[4] += fabsf(e); v[5] += e * e; v[6] += 1.f; }
            if (var) {
                const float s = var[o];
                if (s == s) { v[7] += s; v[8] += 1.f; }
            }
        }
        float m = tp / sqrtf(tt);
        m = m / sqrtf(pp);
        // torch.clamp propagates NaN; fminf/fmaxf would not
        const float cl = m != m ? m : fminf(fmaxf(m, -1.f), 1.f);
        v[2] = acosf(cl) * 180.f / 3.14159265358979323846f;
    }
    __shared__ float red[4][MT_NPART];
#pragma unroll
    for (int i = 0; i < MT_NPART; ++i) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < MT_NPART; ++i) red[w][i] = v[i];
    __syncthreads();
    if (threadIdx.x < MT_NPART)
        part[((size_t)b * gridDim.x + blockIdx.x) * MT_NPART + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// SSIM map of one 16x16 tile of one (b, c) plane: zero padding 5, window w[11][11] (the reference's outer product)
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                   const float* __restrict__ win, float* __restrict__ part, int H,
                                                   int W) {
    __shared__ float a[26][27], b[26][27], wk[121];
    const int plane = blockIdx.z, y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    const float* p1 = img1 + (size_t)plane * H * W;
    const float* p2 = img2 + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int r = i / 26, c = i % 26, y = y0 + r - 5, x = x0 + c - 5;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        a[r][c] = in ? p1[(size_t)y * W + x] : 0.f;
        b[r][c] = in ? p2[(size_t)y * W + x] : 0.f;
    }
    if (threadIdx.x < 121) wk[threadIdx.x] = win[threadIdx.x];
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float val = 0.f;
    if (y0 + ty < H && x0 + tx < W) {
        float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
        for (int i = 0; i < 11; ++i)
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const float w = wk[i * 11 + j], u = a[ty + i][tx + j], v = b[ty + i][tx + j];
                mu1 = fmaf(w, u, mu1); mu2 = fmaf(w, v, mu2);
                s11 = fmaf(w, u * u, s11); s22 = fmaf(w, v * v, s22); s12 = fmaf(w, u * v, s12);
            }
        const float m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        val = ((2.f * m12 + C1) * (2.f * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
    }
    __shared__ float red[8];
    float dummy = 0.f;
    block_sum2<256>(val, dummy, red);
    if (threadIdx.x == 0) part[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = val;
}

// pixel-wise maps: x.nanmean(0).nanmean(0) of [B][C][P] tensors -> [P];  out = [4][P]: error, ae, se, var
__global__ __launch_bounds__(256) void metrics_pixelwise_kernel(const float* __restrict__ targ, const float* __restrict__ pred,
                                                                const float* __restrict__ var, float* __restrict__ out,
                                                                int B, int C, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float nanv = __uint_as_float(0x7fc00000u);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, cnt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        float s[4] = {0.f, 0.f, 0.f, 0.f}, n[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const size_t o = ((size_t)b * C + c) * P + p;
            const float e = targ[o] - pred[o];
            if (e == e) { s[0] += e; s[1] += fabsf(e); s[2] += e * e; n[0] += 1.f; n[1] += 1.f; n[2] += 1.f; }
            if (var) { const float v = var[o]; if (v == v) { s[3] += v; n[3] += 1.f; } }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n[k] > 0.f) { acc[k] += s[k] / n[k]; cnt[k] += 1.f; }     // a channel whose B values are all NaN is skipped
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(size_t)k * P + p] = (k < 3 || var) ? (cnt[k] > 0.f ? acc[k] / cnt[k] : nanv) : nanv;
}

// out[0..8] = RMSE, MAE, PSNR, SAM, SSIM, error, mean ae, mean se, mean var;  out[16 + b] = SSIM of batch item b
__global__ __launch_bounds__(256) void metrics_final_kernel(const float* __restrict__ ppart, int npp,
                                                            const float* __restrict__ spart, int nsp_per_b, int B,
                                                            double n_elem, double n_pix, int has_var,
                                                            float* __restrict__ out) {
    __shared__ double sh[MT_NPART + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // wave w handles quantities w, w+4, ...; fixed order -> deterministic
    for (int q = w; q < MT_NPART; q += 4) {
        double s = 0.0;
        for (int i = lane; i < npp; i += 64) s += (double)ppart[(size_t)i * MT_NPART + q];
        s = wave_sum_d(s);
        if (lane == 0) sh[q] = s;
    }
    __syncthreads();
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {     // per-item SSIM sums, then the overall mean
        double s = 0.0;
        for (int i = threadIdx.x; i < nsp_per_b; i += 256) s += (double)spart[(size_t)b * nsp_per_b + i];
        s = wave_sum_d(s);
        __shared__ double r4[4];
        if (lane == 0) r4[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double sb = r4[0] + r4[1] + r4[2] + r4[3];
            out[16 + b] = (float)(sb / (n_elem / B));
            tot += sb;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nanv = __longlong_as_double(0x7ff8000000000000ll);
        const double rmse = sqrt(sh[0] / n_elem);
        out[0] = (float)rmse;
        out[1] = (float)(sh[1] / n_elem);
        out[2] = (float)(20.0 * log10(1.0 / rmse));
        out[3] = (float)(sh[2] / n_pix);
        out[4] = (float)(tot / n_elem);
        out[5] = (float)(sh[6] > 0 ? sh[3] / sh[6] : nanv);
        out[6] = (float)(sh[6] > 0 ? sh[4] / sh[6] : nanv);
        out[7] = (float)(sh[6] > 0 ? sh[5] / sh[6] : nanv);
        out[8] = (float)((has_var && sh[8] > 0) ? sh[7] / sh[8] : nanv);
    }
}

extern "C" int uncr_img_metrics_work(int B, int C, int H, int W) {   // floats of workspace
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return -1;
    const int P = H * W;
    return B * ((P + 255) / 256) * MT_NPART + B * C * ((H + 15) / 16) * ((W + 15) / 16);
}

// target / pred / var: [B][C][H][W];  win: the 11x11 window (121 floats);  out: >= 16 + B floats;
// pixelwise: null or [4][H*W];  work: uncr_img_metrics_work floats
extern "C" int uncr_img_metrics(const float* target, const float* pred, const float* var, const float* win, float* out,
                                float* pixelwise, float* work, int B, int C, int H, int W, hipStream_t stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return UNCR_ESHAPE;
    if (!target || !pred || !win || !out || !work) return UNCR_EINVAL;
    const int P = H * W, nbx = (P + 255) / 256;
    float* ppart = work;
    float* spart = work + (size_t)B * nbx * MT_NPART;
    hipLaunchKernelGGL(metrics_point_kernel, dim3(nbx, B), dim3(256), 0, stream, target, pred, var, ppart, C, P);
    UNCR_LAUNCH_CHECK();
    const int ty = (H + 15) / 16, tx = (W + 15) / 16;
    hipLaunchKernelGGL(ssim_kernel, dim3(tx, ty, B * C), dim3(256), 0, stream, target, pred, win, spart, H, W);
    UNCR_LAUNCH_CHECK();
    if (pixelwise) {
        hipLaunchKernelGGL(metrics_pixelwise_kernel, dim3(nbx), dim3(256), 0, stream, target, pred, var, pixelwise, B, C, P);
        UNCR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(256), 0, stream, ppart, B * nbx, spart, C * ty * tx, B,
                       (double)B * C * P, (double)B * P, var ? 1 : 0, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
