// Stand-alone calls of the attention classes (ltae.py:388-458 ScaledDotProductAttention / ScaledDotProductAttentionSmall,
// ltae.py:244-307 / 312-385 MultiHeadAttention / MultiHeadAttentionSmall): pixel-major rows [m][T][d] with a per-row pad mask,
// the layout those classes are defined on.  Inside UNCRTAINTS the same arithmetic runs fused into the plane-tiled L-TAE stage
// kernels (ltae_fused.hip, aggregate.hip); these kernels serve the classes called on their own: one thread per row, the row's T
// scores in its own output row (any T), everything else streamed.  fp32 throughout; masked_fill(-1e3) and softmax as in the reference.
//   score[t] = (q . k[t]) / temperature ; pad -> -1e3 ; attn = softmax_t(score) ; (dropout) ; out = attn @ v
// plus a 2-D transpose with zero padding, the bridge between the pixel-major rows of these classes and the channel-major planes of
// the MFMA GEMM (nn.Linear on rows = a 1x1 convolution on the transposed tensor).
#include "common.h"


// q [m][dk] (q_rows == m) or q [q_rows][dk] shared by consecutive groups of m / q_rows rows (MultiHeadAttention: one query per head)
__global__ __launch_bounds__(256) void sdpa_rows_fwd_kernel(
    const float* __restrict__ q, int q_rows, const float* __restrict__ k, const float* __restrict__ v,
    const int* __restrict__ pad, float inv_temp, float* __restrict__ attn_sm, float* __restrict__ attn_out,
    float* __restrict__ out, float* __restrict__ comp, int m, int T, int dk, int dv, float p_drop,
    unsigned long long seed, const long long* __restrict__ seed_dev) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const float* qr = q + (size_t)(q_rows == m ? r : r / (m / q_rows)) * dk;
    const float* kr = k + (size_t)r * T * dk;
    // the row's scores live in its attn_sm row (L1 / L2 resident: T floats), so T is not limited by a register array
    float* sc = attn_sm + (size_t)r * T;
    float mx = -3.0e38f;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        float a = 0.f;
        for (int j = 0; j < dk; ++j) a = fmaf(qr[j], kr[(size_t)t * dk + j], a);
        a *= inv_temp;
        if (pad && pad[(size_t)r * T + t]) a = -1.0e3f;
        if (comp) comp[(size_t)r * T + t] = a;
        sc[t] = a;
        mx = fmaxf(mx, a);
    }
    float den = 0.f;
    for (int t = 0; t < T; ++t) { const float e = expf(sc[t] - mx); sc[t] = e; den += e; }
    const float inv = 1.f / den;
    const unsigned long long sd = seed + (seed_dev ? (unsigned long long)seed_dev[0] : 0ull);
    const float keep = 1.f / (1.f - p_drop);
    auto dropped = [&](int t, float a) { return p_drop > 0.f ? (hash_uniform(sd, (uint64_t)r * T + t) < p_drop ? 0.f : a * keep) : a; };
    for (int t = 0; t < T; ++t) {
        const float a = sc[t] * inv;
        sc[t] = a;                                     // attn_sm: the softmax itself (the backward reads it)
        if (attn_out) attn_out[(size_t)r * T + t] = dropped(t, a);
    }
    if (out) {
        const float* vr = v + (size_t)r * T * dv;
        for (int j = 0; j < dv; ++j) {
            float a = 0.f;
            for (int t = 0; t < T; ++t) a = fmaf(dropped(t, sc[t]), vr[(size_t)t * dv + j], a);
            out[(size_t)r * dv + j] = a;
        }
    }
}

// d(attn) [m][T] (gradient of the returned, post-dropout attention; nullable), d(out) [m][dv] (nullable), d(comp) [m][T] (nullable)
// -> dq_rows [m][dk] (per row; shared queries are summed by the caller with uncr_colsum), dk [m][T][dk], dv [m][T][dv]
__global__ __launch_bounds__(256) void sdpa_rows_bwd_kernel(
    const float* __restrict__ dattn, const float* __restrict__ dout, const float* __restrict__ dcomp,
    const float* __restrict__ q, int q_rows, const float* __restrict__ k, const float* __restrict__ v,
    const int* __restrict__ pad, const float* __restrict__ attn_sm, float inv_temp, float* __restrict__ dq_rows,
    float* __restrict__ dkk, float* __restrict__ dvv, int m, int T, int dk, int dv, float p_drop, unsigned long long seed,
    const long long* __restrict__ seed_dev) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const float* qr = q + (size_t)(q_rows == m ? r : r / (m / q_rows)) * dk;
    const unsigned long long sd = seed + (seed_dev ? (unsigned long long)seed_dev[0] : 0ull);
    const float keep = 1.f / (1.f - p_drop);
    // gradient w.r.t. the softmax output of date t (evaluated twice: once for the row's dot product, once for the score gradients --
    // no per-row array, so T is not limited); the first evaluation also writes d(v)
    auto grad_t = [&](int t, float a, bool write_dv) {
        float g = dattn ? dattn[(size_t)r * T + t] : 0.f;
        float mask = 1.f;
        if (p_drop > 0.f) mask = hash_uniform(sd, (uint64_t)r * T + t) < p_drop ? 0.f : keep;
        if (dout) {
            const float* vr = v + ((size_t)r * T + t) * dv;
            float gv = 0.f;
            for (int j = 0; j < dv; ++j) {
                const float d = dout[(size_t)r * dv + j];
                gv = fmaf(d, vr[j], gv);
                if (write_dv && dvv) dvv[((size_t)r * T + t) * dv + j] = a * mask * d;
            }
            g += gv;
        } else if (write_dv && dvv) {
            for (int j = 0; j < dv; ++j) dvv[((size_t)r * T + t) * dv + j] = 0.f;
        }
        return g * mask;
    };
    float dot = 0.f;
    for (int t = 0; t < T; ++t) {
        const float a = attn_sm[(size_t)r * T + t];
        dot = fmaf(grad_t(t, a, true), a, dot);
    }
    for (int j = 0; j < dk; ++j) if (dq_rows) dq_rows[(size_t)r * dk + j] = 0.f;
    for (int t = 0; t < T; ++t) {
        const float a = attn_sm[(size_t)r * T + t];
        float ds = a * (grad_t(t, a, false) - dot);           // softmax backward
        if (dcomp) ds += dcomp[(size_t)r * T + t];            // comp is the masked, scaled score itself
        if (pad && pad[(size_t)r * T + t]) ds = 0.f;          // masked_fill: no gradient into a filled score
        ds *= inv_temp;
        const float* kr = k + ((size_t)r * T + t) * dk;
        for (int j = 0; j < dk; ++j) {
            if (dkk) dkk[((size_t)r * T + t) * dk + j] = ds * qr[j];
            if (dq_rows) dq_rows[(size_t)r * dk + j] = fmaf(ds, kr[j], dq_rows[(size_t)r * dk + j]);
        }
    }
}

extern "C" int uncr_sdpa_rows_fwd(const float* q, int q_rows, const float* k, const float* v, const int* pad,
                                  float temperature, float* attn_sm, float* attn_out, float* out, float* comp, int m, int T,
                                  int dk, int dv, float p_drop, unsigned long long seed, const long long* seed_dev,
                                  hipStream_t stream) {
    if (m <= 0 || T <= 0 || dk <= 0 || q_rows <= 0 || m % q_rows || !(temperature > 0.f)) return UNCR_ESHAPE;
    if (!q || !k || !attn_sm || (out && (!v || dv <= 0)) || p_drop < 0.f || p_drop >= 1.f) return UNCR_EINVAL;
    hipLaunchKernelGGL(sdpa_rows_fwd_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, q, q_rows, k, v, pad,
                       1.f / temperature, attn_sm, attn_out, out, comp, m, T, dk, dv, p_drop, seed, seed_dev);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_sdpa_rows_bwd(const float* dattn, const float* dout, const float* dcomp, const float* q, int q_rows,
                                  const float* k, const float* v, const int* pad, const float* attn_sm, float temperature,
                                  float* dq_rows, float* dk_out, float* dv_out, int m, int T, int dk, int dv, float p_drop,
                                  unsigned long long seed, const long long* seed_dev, hipStream_t stream) {
    if (m <= 0 || T <= 0 || dk <= 0 || q_rows <= 0 || m % q_rows || !(temperature > 0.f)) return UNCR_ESHAPE;
    if (!q || !k || !attn_sm || (dout && (!v || dv <= 0)) || (dv_out && dv <= 0)) return UNCR_EINVAL;
    hipLaunchKernelGGL(sdpa_rows_bwd_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, dattn, dout, dcomp, q, q_rows, k, v,
                       pad, attn_sm, 1.f / temperature, dq_rows, dk_out, dv_out, m, T, dk, dv, p_drop, seed, seed_dev);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// dst [cols][dst_ld] = src [rows][cols]^T, columns rows .. dst_ld-1 of dst zero-filled (LDS tile transpose, both sides coalesced)
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                          int cols, int dst_ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < dst_ld) dst[(size_t)c * dst_ld + r] = tile[tx][ty + i];
    }
}

extern "C" int uncr_transpose2d(const float* src, float* dst, int rows, int cols, int dst_ld, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || dst_ld < rows || !src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 31) / 32, (dst_ld + 31) / 32), dim3(256), 0, stream, src, dst, rows, cols,
                       dst_ld);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
