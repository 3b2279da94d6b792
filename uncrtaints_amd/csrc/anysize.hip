// Any H x W (the reference takes any spatial size, uncrtaints.py:391-447; the streaming kernels of this library tile a plane in
// 1024-pixel / 128-pixel pieces of float4 lanes).  For sizes those tilings do not fit, the engine keeps every full-resolution tensor as
// DENSE planes of H*W pixels followed by a ZERO TAIL up to the next multiple of 1024 (plane stride Pc):
//   * the flat kernels (pointwise GEMMs, weight gradients, element-wise passes, SE pooling) run over the whole stride unchanged.  With
//     zero inputs every tail pixel of a plane holds the same value f(0) after a point-wise kernel, so `uncr_fix_tail` behind the
//     producer reads that value, takes n_tail * f(0) (and n_tail * f(0)^2) out of the plane's statistics slot and zeroes the tail;
//     `uncr_fix_sepool_tail`, `uncr_fix_wgrad_tail` and `uncr_fix_rowsum_tail` are the analytic corrections of the three reductions
//     whose tail term is not a plain statistic (SE pooling of gelu(B), the dW2 products, in_conv's bias gradient);
//   * the 2-D kernels get scalar any-width variants that read and write valid pixels only: depthwise 3x3 forward / backward here, the
//     adaptive max-pool with a plane stride in ltae.hip, the temporal aggregation in aggregate.hip;
//   * `uncr_embed_tail` / `uncr_extract_tail` convert between the caller's dense tensors and the padded planes.
// fp32 storage.  These are functional kernels for sizes outside the tuned tilings, not tuned ones; the tuned path is untouched.
#include "common.h"

#define ANY_NB 8      // blocks (= statistics slots) per plane of the scalar 2-D kernels

extern "C" int uncr_any_slots(void) { return ANY_NB; }
// plane stride of an H x W image: the next multiple of 1024 pixels (0: the tuned tilings take the size as it is)
extern "C" int uncr_any_plane_stride(int H, int W) {
    const long long P = (long long)H * W;
    if (H <= 0 || W <= 0 || P > (1ll << 30)) return -1;
    if (P % 1024 == 0 && W % 4 == 0) return 0;
    return (int)((P + 1023) / 1024 * 1024);
}

__global__ __launch_bounds__(256) void embed_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int Pc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Pc) dst[(size_t)blockIdx.y * Pc + i] = i < P ? src[(size_t)blockIdx.y * P + i] : 0.f;
}
__global__ __launch_bounds__(256) void extract_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int Pc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) dst[(size_t)blockIdx.y * P + i] = src[(size_t)blockIdx.y * Pc + i];
}
extern "C" int uncr_embed_tail(const float* src, float* dst, int planes, int P, int Pc, hipStream_t stream) {
    if (planes <= 0 || P <= 0 || Pc < P) return UNCR_ESHAPE;
    if (!src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(embed_tail_kernel, dim3((Pc + 255) / 256, planes), dim3(256), 0, stream, src, dst, P, Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_extract_tail(const float* src, float* dst, int planes, int P, int Pc, hipStream_t stream) {
    if (planes <= 0 || P <= 0 || Pc < P) return UNCR_ESHAPE;
    if (!src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(extract_tail_kernel, dim3((P + 255) / 256, planes), dim3(256), 0, stream, src, dst, P, Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// mode 0: part = (sum t, sum t^2): both components lose the tail's share; mode 1: part = (sum t, sum t*aux) with aux zero on the tail:
// the first component only; mode 2 (or part == null): zero the tail, no statistics.  One block per plane.
__global__ __launch_bounds__(256) void fix_tail_kernel(float* __restrict__ t, float2* __restrict__ part, int slots, int P, int Pc,
                                                       int mode) {
    float* p = t + (size_t)blockIdx.x * Pc;
    if (threadIdx.x == 0 && part && mode < 2 && P < Pc) {
        const float v = p[P];                      // every tail pixel of the plane holds this value
        const float n = (float)(Pc - P);
        float2 s = part[(size_t)blockIdx.x * slots];
        s.x -= n * v;
        if (mode == 0) s.y -= n * v * v;
        part[(size_t)blockIdx.x * slots] = s;
    }
    __syncthreads();
    for (int i = P + threadIdx.x; i < Pc; i += 256) p[i] = 0.f;
}
extern "C" int uncr_fix_tail(float* t, float* part, int slots, int planes, int P, int Pc, int mode, hipStream_t stream) {
    if (planes <= 0 || P <= 0 || Pc < P || (part && slots <= 0) || mode < 0 || mode > 2) return UNCR_ESHAPE;
    if (!t) return UNCR_EINVAL;
    if (Pc == P) return UNCR_OK;
    hipLaunchKernelGGL(fix_tail_kernel, dim3(planes), dim3(256), 0, stream, t, (float2*)part, slots, P, Pc, mode);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// SE pooling partials (sum_p gelu(A*h2 + B), .) over a plane whose tail holds h2 = 0: take n_tail * gelu(B) out again
__global__ __launch_bounds__(256) void fix_sepool_tail_kernel(float2* __restrict__ part, int slots, const float* __restrict__ cB,
                                                              int planes, float ntail) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < planes) part[(size_t)i * slots].x -= ntail * gelu_f(cB[i]);      // the pass evaluated gelu(fma(A, 0, B)) = gelu(B) there
}
extern "C" int uncr_fix_sepool_tail(float* part, int slots, const float* cB, int planes, int ntail, hipStream_t stream) {
    if (planes <= 0 || slots <= 0 || ntail < 0) return UNCR_ESHAPE;
    if (!part || !cB) return UNCR_EINVAL;
    if (ntail == 0) return UNCR_OK;
    hipLaunchKernelGGL(fix_sepool_tail_kernel, dim3((planes + 255) / 256), dim3(256), 0, stream, (float2*)part, slots, cB, planes,
                       (float)ntail);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// per-frame products G[n][co][ci] = sum_p dh[n,co,p] * z[n,ci,p] with dh = c1*dy + c2*(h3 - mu) + c3 and z = gelu(A*h2 + B), taken over a
// stride whose tail holds dy = h3 = h2 = 0: the tail contributed n_tail * (c3 - c2*mu)[n,co] * gelu(B)[n,ci]
__global__ __launch_bounds__(256) void fix_wgrad_tail_kernel(float* __restrict__ G, int Cd, int Cx, const float* __restrict__ c2,
                                                             const float* __restrict__ c3, const float* __restrict__ mu,
                                                             const float* __restrict__ cB, float ntail) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cd * Cx) return;
    const int co = i / Cx, ci = i - co * Cx;
    const float kap = fmaf(-c2[n * Cd + co], mu ? mu[n * Cd + co] : 0.f, c3[n * Cd + co]);
    G[((size_t)n * Cd + co) * Cx + ci] -= ntail * kap * gelu_f(cB[n * Cx + ci]);
}
extern "C" int uncr_fix_wgrad_tail(float* G, int N, int Cd, int Cx, const float* c2, const float* c3, const float* mu, const float* cB,
                                   int ntail, hipStream_t stream) {
    if (N <= 0 || Cd <= 0 || Cx <= 0 || ntail < 0) return UNCR_ESHAPE;
    if (!G || !c2 || !c3 || !cB) return UNCR_EINVAL;
    if (ntail == 0) return UNCR_OK;
    hipLaunchKernelGGL(fix_wgrad_tail_kernel, dim3((Cd * Cx + 255) / 256, N), dim3(256), 0, stream, G, Cd, Cx, c2, c3, mu, cB,
                       (float)ntail);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// row sums db[co] = sum_{n,p} (c1*d + c2*(d2 - mu) + c3) over a stride whose tail holds d = d2 = 0
__global__ __launch_bounds__(256) void fix_rowsum_tail_kernel(float* __restrict__ rs, int N, int C, const float* __restrict__ c2,
                                                              const float* __restrict__ c3, const float* __restrict__ mu, float ntail) {
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= C) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += fmaf(-c2[n * C + co], mu ? mu[n * C + co] : 0.f, c3[n * C + co]);
    rs[co] -= ntail * s;
}
extern "C" int uncr_fix_rowsum_tail(float* rs, int N, int C, const float* c2, const float* c3, const float* mu, int ntail,
                                    hipStream_t stream) {
    if (N <= 0 || C <= 0 || ntail < 0) return UNCR_ESHAPE;
    if (!rs || !c2 || !c3) return UNCR_EINVAL;
    if (ntail == 0) return UNCR_OK;
    hipLaunchKernelGGL(fix_rowsum_tail_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, rs, N, C, c2, c3, mu, (float)ntail);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- depthwise 3x3, reflect padding, any H x W (uncrtaints.py:130-131): h2 = dw(gelu(A*h1 + B)), (sum h2, sum h2^2) per block ----
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// g1 = gelu(A*h1 + B) over the whole stride (a flat float4 pass; the stencil then reads nine plain neighbours instead of evaluating nine
// GELUs per output pixel: 474 -> ~100 us at 4 x 256 x 250 x 250)
__global__ __launch_bounds__(256) void affine_gelu_any_kernel(const float* __restrict__ in, const float* __restrict__ cA,
                                                              const float* __restrict__ cB, float* __restrict__ out, int Pc) {
    const int plane = blockIdx.y;
    const float A = cA[plane], B = cB[plane];
    const size_t o = (size_t)plane * Pc + (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    const float4 v = *(const float4*)(in + o);
    *(float4*)(out + o) = make_float4(gelu_f(fmaf(A, v.x, B)), gelu_f(fmaf(A, v.y, B)), gelu_f(fmaf(A, v.z, B)), gelu_f(fmaf(A, v.w, B)));
}

__global__ __launch_bounds__(256) void dw_fwd_any_kernel(const float* __restrict__ g1, const float* __restrict__ w,
                                                         float* __restrict__ out, float2* __restrict__ part, int C, int H, int W, int Pc) {
    const int plane = blockIdx.y, c = plane % C;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
    const float* ip = g1 + (size_t)plane * Pc;
    float* op = out + (size_t)plane * Pc;
    float s0 = 0.f, s1 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += ANY_NB * 256) {
        const int y = i / W, x = i - y * W;
        const int ym = refl(y - 1, H) * W, y0 = y * W, yp = refl(y + 1, H) * W, xm = refl(x - 1, W), xp = refl(x + 1, W);
        float acc = wk[0] * ip[ym + xm];
        acc = fmaf(wk[1], ip[ym + x], acc); acc = fmaf(wk[2], ip[ym + xp], acc);
        acc = fmaf(wk[3], ip[y0 + xm], acc); acc = fmaf(wk[4], ip[y0 + x], acc); acc = fmaf(wk[5], ip[y0 + xp], acc);
        acc = fmaf(wk[6], ip[yp + xm], acc); acc = fmaf(wk[7], ip[yp + x], acc); acc = fmaf(wk[8], ip[yp + xp], acc);
        op[i] = acc;
        s0 += acc;
        s1 = fmaf(acc, acc, s1);
    }
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * ANY_NB + blockIdx.x] = make_float2(s0, s1);
    }
}
extern "C" int uncr_dw_fwd_any(const float* in, const float* cA, const float* cB, const float* w, float* out, float* part,
                               float* scratch /* [N*C][Pc]: gelu(A*in + B) */, int N, int C, int H, int W, int Pc, hipStream_t stream) {
    if (N <= 0 || C <= 0 || H < 2 || W < 2 || Pc < H * W || Pc % 1024) return UNCR_ESHAPE;
    if (!in || !cA || !cB || !w || !out || !scratch) return UNCR_EINVAL;
    hipLaunchKernelGGL(affine_gelu_any_kernel, dim3(Pc / 1024, N * C), dim3(256), 0, stream, in, cA, cB, scratch, Pc);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_fwd_any_kernel, dim3(ANY_NB, N * C), dim3(256), 0, stream, scratch, w, out, (float2*)part, C, H, W, Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// backward: dh2 = k1*du2 + k2*(h2 - kmu) + k3 (norm-2 backward), then per INPUT pixel q and tap k the gathered sum
//   t_k[q] = sum of dh2 over the outputs that read q through tap k (one output in the interior; up to four next to a reflecting border),
// from which both results follow with ONE erf per pixel:  dg1[q] = sum_k w_k t_k[q],  du1 = gelu'(u1) * dg1,  and the depthwise weight
// gradient  dW_k = sum_q gelu(u1)[q] * t_k[q];  statistics (sum du1, sum du1*(h1 - mean1)).
// The outputs y' with refl(y' + d) == y for a row offset d = ky - 1:  y' = y - d;  y' = -y - d (reflection at row 0);
// y' = 2H - 2 - y - d (reflection at row H-1) -- each only if it lies in [0, H) and really maps to y.
__device__ __forceinline__ int refl_sources(int y, int d, int n, int (&src)[3]) {
    int cnt = 0;
    const int a = y - d, b = -y - d, c = 2 * n - 2 - y - d;
    if (a >= 0 && a < n) src[cnt++] = a;
    if (b >= 0 && b < n && b != a && refl(b + d, n) == y) src[cnt++] = b;
    if (c >= 0 && c < n && c != a && c != b && refl(c + d, n) == y) src[cnt++] = c;
    return cnt;
}
// dh2 = k1*du2 + k2*(h2 - kmu) + k3 over the whole stride (flat float4 pass; the gather below then reads one value per source)
__global__ __launch_bounds__(256) void normbwd_any_kernel(const float* __restrict__ du2, const float* __restrict__ h2,
                                                          const float* __restrict__ k1, const float* __restrict__ k2,
                                                          const float* __restrict__ k3, const float* __restrict__ kmu,
                                                          float* __restrict__ out, int Pc) {
    const int plane = blockIdx.y;
    const float K1 = k1[plane], K2 = k2[plane], K3 = k3[plane], KM = kmu ? kmu[plane] : 0.f;
    const size_t o = (size_t)plane * Pc + (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    const float4 a = *(const float4*)(du2 + o), b = *(const float4*)(h2 + o);
    *(float4*)(out + o) = make_float4(fmaf(K1, a.x, fmaf(K2, b.x - KM, K3)), fmaf(K1, a.y, fmaf(K2, b.y - KM, K3)),
                                      fmaf(K1, a.z, fmaf(K2, b.z - KM, K3)), fmaf(K1, a.w, fmaf(K2, b.w - KM, K3)));
}
__global__ __launch_bounds__(256) void dw_bwd_any_kernel(const float* __restrict__ dh2, const float* __restrict__ h1,
                                                         const float* __restrict__ cA1, const float* __restrict__ cB1,
                                                         const float* __restrict__ w, float* __restrict__ du1, float2* __restrict__ part,
                                                         float* __restrict__ dw_part, const float* __restrict__ mean1, int mean_groups,
                                                         int C, int H, int W, int Pc) {
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    const float A1 = cA1[plane], B1 = cB1[plane];
    const float m1 = mean1 ? (mean_groups > 0 ? mean1[n * mean_groups + c / (C / mean_groups)] : mean1[c]) : 0.f;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
    const size_t base = (size_t)plane * Pc;
    const float* dp = dh2 + base;
    float s0 = 0.f, s1 = 0.f, gw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gw[k] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += ANY_NB * 256) {
        const int y = i / W, x = i - y * W;
        const float hv = h1[base + i];
        const float u = fmaf(A1, hv, B1);
        const float gq = gelu_f(u);
        float dg = 0.f;
        if (y >= 2 && y < H - 2 && x >= 2 && x < W - 2) {
            // interior: tap (ky, kx) of exactly one output, (y - ky + 1, x - kx + 1), reads this pixel
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float t = dp[(y - ky + 1) * W + (x - kx + 1)];
                    dg = fmaf(wk[ky * 3 + kx], t, dg);
                    gw[ky * 3 + kx] = fmaf(gq, t, gw[ky * 3 + kx]);
                }
        } else {
            int ys[3][3], xs[3][3], ny[3], nx[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { ny[k] = refl_sources(y, k - 1, H, ys[k]); nx[k] = refl_sources(x, k - 1, W, xs[k]); }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float t = 0.f;
                    for (int a = 0; a < ny[ky]; ++a)
                        for (int b = 0; b < nx[kx]; ++b) t += dp[ys[ky][a] * W + xs[kx][b]];
                    dg = fmaf(wk[ky * 3 + kx], t, dg);
                    gw[ky * 3 + kx] = fmaf(gq, t, gw[ky * 3 + kx]);
                }
        }
        const float v = gelu_grad_f(u) * dg;
        du1[base + i] = v;
        s0 += v;
        s1 = fmaf(v, hv - m1, s1);
    }
    __shared__ float red[8];
    block_sum2<256>(s0, s1, red);
    if (threadIdx.x == 0) part[(size_t)plane * ANY_NB + blockIdx.x] = make_float2(s0, s1);
#pragma unroll
    for (int k = 0; k < 9; k += 2) {
        __syncthreads();
        float a = gw[k], b = k + 1 < 9 ? gw[k + 1] : 0.f;
        block_sum2<256>(a, b, red);
        if (threadIdx.x == 0) {
            dw_part[((size_t)plane * ANY_NB + blockIdx.x) * 9 + k] = a;
            if (k + 1 < 9) dw_part[((size_t)plane * ANY_NB + blockIdx.x) * 9 + k + 1] = b;
        }
    }
}
extern "C" int uncr_dw_bwd_any(const float* du2, const float* h2, const float* h1, const float* k1, const float* k2, const float* k3,
                               const float* kmu, const float* cA1, const float* cB1, const float* w, float* du1, float* part,
                               float* dw_part, const float* mean1, int mean_groups, float* scratch /* [N*C][Pc]: the norm-2 backward dh2 */,
                               int N, int C, int H, int W, int Pc, hipStream_t stream) {
    if (N <= 0 || C <= 0 || H < 2 || W < 2 || Pc < H * W || Pc % 1024 || (mean1 && mean_groups > 0 && C % mean_groups)) return UNCR_ESHAPE;
    if (!du2 || !h2 || !h1 || !k1 || !k2 || !k3 || !cA1 || !cB1 || !w || !du1 || !part || !dw_part || !scratch) return UNCR_EINVAL;
    hipLaunchKernelGGL(normbwd_any_kernel, dim3(Pc / 1024, N * C), dim3(256), 0, stream, du2, h2, k1, k2, k3, kmu, scratch, Pc);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dw_bwd_any_kernel, dim3(ANY_NB, N * C), dim3(256), 0, stream, scratch, h1, cA1, cB1, w, du1, (float2*)part, dw_part,
                       mean1, mean_groups, C, H, W, Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
