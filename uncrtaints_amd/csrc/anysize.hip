// Any H x W (the reference takes any spatial size, uncrtaints.py:391-447; the streaming kernels of this library tile a plane in
// 1024-pixel / 128-pixel pieces of float4 lanes).  For sizes those tilings do not fit, the engine keeps every full-resolution tensor as
// DENSE planes of H*W pixels followed by a ZERO TAIL up to the next multiple of 1024 (plane stride Pc):
//   * the flat kernels (pointwise GEMMs, weight gradients, element-wise passes, SE pooling) take the valid pixel count next to the
//     stride and keep the tail OUT of every reduction: the element-wise family writes its tail as zeros and takes its statistics over
//     valid pixels (ew.hip), the weight-gradient kernels sum whole 32-pixel chunks below the count and `uncr_wgrad_boundary` adds the
//     rest, the pointwise GEMMs leave tiles that reach into the tail out of their statistics and `uncr_fix_tail` behind them adds
//     the boundary tile's valid pixels and zeroes the tail (which holds f(0) until then).  Nothing is subtracted after the fact
//     (rounds 5's analytic corrections took a coherent n_tail * f(0) term out of an fp32 sum whose valid terms cancel);
//   * the 2-D kernels get scalar any-width variants that read and write valid pixels only: depthwise 3x3 forward / backward here, the
//     adaptive max-pool with a plane stride in ltae.hip, the temporal aggregation in aggregate.hip;
//   * `uncr_embed_tail` / `uncr_extract_tail` convert between the caller's dense tensors and the padded planes.
// fp32 storage.  These are functional kernels for sizes outside the tuned tilings, not tuned ones; the tuned path is untouched.
#include "common.h"


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

// Behind a pointwise GEMM on padded planes (uncr_pw_gemm / uncr_pw_gemm_dx with Pv < P): the GEMM's statistics epilogue left out every
// `unit`-pixel tile that reaches into the tail, so the BOUNDARY tile's valid pixels [P / unit * unit, P) are summed here -- from the
// stored values, at most unit - 1 terms per plane, fp64 -- and added to the slot of the block that owned that tile; then the tail
// (which holds the GEMM's f(0)) is zeroed.  mode 0: (sum t, sum t^2); mode 1: (sum t, sum t*aux); mode 2 (or part == null): zero the
// tail only.  One block per plane.  Nothing is ever subtracted: a plane's statistics see its valid pixels only.
__global__ __launch_bounds__(256) void fix_tail_kernel(float* __restrict__ t, const float* __restrict__ aux, float2* __restrict__ part,
                                                       int slots, int P, int Pc, int mode, int unit, const float* __restrict__ pivot) {
    float* p = t + (size_t)blockIdx.x * Pc;
    if (part && mode < 2) {
        const int u0 = P / unit * unit;
        const float* q = (mode == 1 && aux) ? aux + (size_t)blockIdx.x * Pc : p;
        const double piv = (mode == 1 && pivot) ? (double)pivot[blockIdx.x] : 0.0;      // centred cross statistics: sum t*(aux - pivot)
        double s0 = 0.0, s1 = 0.0;
        for (int i = u0 + threadIdx.x; i < P; i += 256) {
            const double v = (double)p[i];
            s0 += v;
            s1 += v * ((double)q[i] - piv);
        }
        __shared__ double red[2][256];
        red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if (threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
            __syncthreads();
        }
        if (threadIdx.x == 0 && u0 < P) {
            const size_t slot = (size_t)blockIdx.x * slots + (size_t)((u0 / unit) % slots);
            float2 s = part[slot];
            s.x = (float)((double)s.x + red[0][0]);
            s.y = (float)((double)s.y + red[1][0]);
            part[slot] = s;
        }
    }
    for (int i = P + threadIdx.x; i < Pc; i += 256) p[i] = 0.f;
}
extern "C" int uncr_fix_tail(float* t, const float* aux, float* part, int slots, int planes, int P, int Pc, int mode, int unit,
                             const float* pivot, hipStream_t stream) {
    if (planes <= 0 || P <= 0 || Pc < P || (part && (slots <= 0 || unit <= 0)) || mode < 0 || mode > 2) return UNCR_ESHAPE;
    if (!t) return UNCR_EINVAL;
    if (Pc == P) return UNCR_OK;
    hipLaunchKernelGGL(fix_tail_kernel, dim3(planes), dim3(256), 0, stream, t, aux, (float2*)part, slots, P, Pc, mode, unit > 0 ? unit : 1, pivot);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- depthwise 3x3, reflect padding, any H x W (uncrtaints.py:130-131) as ROW-BAND kernels: one 256-thread block owns TR consecutive
// rows of one plane and stages them with their halo into an LDS tile ON THE PADDED GRID (the point-wise prologue applied on the way: one
// GELU / one norm-backward per element, not nine), so the stencil loop reads LDS at constant offsets -- no reflection arithmetic, no
// alignment assumptions, any width, no scratch tensor.  What counts is instructions per
// pixel and blocks per CU (the first versions gathered nine reflected neighbours from global memory, then from LDS with a per-pixel border branch whose
// 81 predicated reads every wave crossing a row end had to walk: 356 / 852 us, then 235 / 654 us at 4 x 256 x 250 x 250).
// rows per band: the forward tile is (TR + 2) x (W + 2) (reflected halo), the backward tile (TR + 5) x (W + 4) (zero-extended, see below).
// A 20 KB tile (eight blocks = 32 waves per CU) beats a 40 KB one by 25 % at W = 250 (425 -> 320 us backward): the two phases of a block
// (stage, stencil) overlap only across blocks.  Wider images take a larger tile to keep at least eight rows per band, up to 62.5 KB.
static int dw_band_rows(int W, int bwd) {
    if (W < 2) return 0;
    const int budgets[3] = {5120, 10240, 16000};          // floats; 16000 < 2^14 keeps row_of exact
    int tr = 0;
    for (int i = 0; i < 3; ++i) {
        tr = bwd ? budgets[i] / (W + 4) - 5 : budgets[i] / (W + 2) - 2;
        if (tr >= 8) break;
    }
    return tr < 2 ? 0 : (tr > 32 ? 32 : tr);
}
// statistics slots (= row bands) per plane of uncr_dw_fwd_any (bwd = 0) / uncr_dw_bwd_any (bwd = 1); -1: width not supported
extern "C" int uncr_dw_any_slots(int H, int W, int bwd) {
    const int tr = dw_band_rows(W, bwd);
    return (tr && H >= 2) ? (H + tr - 1) / tr : -1;
}
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
// row of the flat index i in a grid of width 1 / inv: (int)((i + 0.5) * inv) is exact for i < 2^14 -- the quotient is at least
// 0.5 / width from an integer and the fp32 rounding error of the product is below 2^14 * 2^-23
__device__ __forceinline__ int row_of(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// forward: h2 = dw(gelu(A*h1 + B)), (sum h2, sum h2^2) per band; the last band also writes the plane's zero tail.
// tile (r, c) = gelu(u1) at image (refl(y0 - 1 + r), refl(c - 1)), pitch W + 2
__global__ __launch_bounds__(256) void dw_fwd_band_kernel(const float* __restrict__ in, const float* __restrict__ cA,
                                                          const float* __restrict__ cB, const float* __restrict__ w,
                                                          float* __restrict__ out, float2* __restrict__ part, int C, int H, int W, int Pc,
                                                          int TR, float invW) {
    extern __shared__ __attribute__((aligned(16))) float t[];
    const int plane = blockIdx.y, c = plane % C;
    const int y0 = blockIdx.x * TR, rows = min(TR, H - y0), nrow = rows + 2, pitch = W + 2;
    const float A = cA[plane], B = cB[plane];
    const float* src = in + (size_t)plane * Pc;
    const int n = nrow * W;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = row_of(i, invW), x = i - r * W;
        t[r * pitch + 1 + x] = gelu_f(fmaf(A, src[refl(y0 - 1 + r, H) * W + x], B));
    }
    for (int i = threadIdx.x; i < 2 * nrow; i += 256) {                 // the two halo columns: column -1 = column 1, column W = W - 2
        const int r = i >> 1, side = i & 1;
        t[r * pitch + (side ? W + 1 : 0)] = gelu_f(fmaf(A, src[refl(y0 - 1 + r, H) * W + (side ? W - 2 : 1)], B));
    }
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
    __syncthreads();
    float* dst = out + (size_t)plane * Pc + (size_t)y0 * W;
    const int m = rows * W;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 2
    for (int j = threadIdx.x; j < m; j += 256) {
        const int r = row_of(j, invW);
        const float* tc = t + r * pitch + (j - r * W);                  // tile (r, x): the stencil's top-left corner
        float acc = wk[0] * tc[0];
        acc = fmaf(wk[1], tc[1], acc); acc = fmaf(wk[2], tc[2], acc);
        acc = fmaf(wk[3], tc[pitch], acc); acc = fmaf(wk[4], tc[pitch + 1], acc); acc = fmaf(wk[5], tc[pitch + 2], acc);
        acc = fmaf(wk[6], tc[2 * pitch], acc); acc = fmaf(wk[7], tc[2 * pitch + 1], acc); acc = fmaf(wk[8], tc[2 * pitch + 2], acc);
        dst[j] = acc;
        s0 += acc;
        s1 = fmaf(acc, acc, s1);
    }
    if (blockIdx.x == gridDim.x - 1)
        for (int i = H * W + threadIdx.x; i < Pc; i += 256) out[(size_t)plane * Pc + i] = 0.f;
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
    }
}
extern "C" int uncr_dw_fwd_any(const float* in, const float* cA, const float* cB, const float* w, float* out,
                               float* part /* [N*C][uncr_dw_any_slots(H, W, 0)][2] or null */, int N, int C, int H, int W, int Pc,
                               hipStream_t stream) {
    const int tr = dw_band_rows(W, 0);
    if (N <= 0 || C <= 0 || H < 2 || !tr || Pc < H * W || Pc % 1024) return UNCR_ESHAPE;
    if (!in || !cA || !cB || !w || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(dw_fwd_band_kernel, dim3((H + tr - 1) / tr, N * C), dim3(256), (size_t)(tr + 2) * (W + 2) * sizeof(float), stream,
                       in, cA, cB, w, out, (float2*)part, C, H, W, Pc, tr, 1.0f / (float)W);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// backward: with gp = the reflect-padded gelu(u1) on [-1, H] x [-1, W], the forward is a plain correlation  h2[q] = sum_k w_k gp[q + k - 1],
// so the gradient on the padded grid is  dgp[p~] = sum_k w_k t_k[p~],  t_k[p~] = dh2z[p~ - k + 1]  (dh2z = the norm-2 backward
// dh2 = k1*du2 + k2*(h2 - kmu) + k3 inside the image, ZERO outside), and the reflection folds the ring back:
//   dg1[p] = sum of dgp over the padded positions that mirror onto p = {y, and -1 if y == 1, and H if y == H-2} x {x, -1 if x == 1, W if x == W-2}
// -- one position in the interior, two along rows / columns 1 and H-2 / W-2, four at their crossings.  The same t_k give the depthwise
// weight gradient  dW_k = sum_p gelu(u1)[p] * (sum over p's padded positions of t_k)  with ONE erf per pixel;  du1 = gelu'(u1) * dg1,
// statistics (sum du1, sum du1*(h1 - mean1)).  tile (r, c) = dh2z at image (y0 - 2 + r, c - 2), rows y0 - 2 ... y0 + rows + 2 (row H - 2 may be a
// band's last row, and its mirrored position H reads rows H - 1 ... H + 1), pitch W + 4: every read of the loop is in range without a test.
__global__ __launch_bounds__(256) void dw_bwd_band_kernel(const float* __restrict__ du2, const float* __restrict__ h2,
                                                          const float* __restrict__ h1, const float* __restrict__ k1,
                                                          const float* __restrict__ k2, const float* __restrict__ k3,
                                                          const float* __restrict__ kmu, const float* __restrict__ cA1,
                                                          const float* __restrict__ cB1, const float* __restrict__ w,
                                                          float* __restrict__ du1, float2* __restrict__ part, float* __restrict__ dw_part,
                                                          const float* __restrict__ mean1, int mean_groups, int C, int H, int W, int Pc,
                                                          int TR, float invW, float invP) {
    extern __shared__ __attribute__((aligned(16))) float t[];
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    const int y0 = blockIdx.x * TR, rows = min(TR, H - y0), nrow = rows + 5, pitch = W + 4;
    const size_t base = (size_t)plane * Pc;
    {
        const float K1 = k1[plane], K2 = k2[plane], K3 = k3[plane], KM = kmu ? kmu[plane] : 0.f;
        const float* pa = du2 + base;
        const float* pb = h2 + base;
        const int cnt = nrow * pitch;
#pragma unroll 4
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const int r = row_of(i, invP), yy = y0 - 2 + r, xx = i - r * pitch - 2;
            const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int o = min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);          // (unconditional loads)
            const float v = fmaf(K1, pa[o], fmaf(K2, pb[o] - KM, K3));
            t[i] = inside ? v : 0.f;
        }
    }
    const float A1 = cA1[plane], B1 = cB1[plane];
    const float m1 = mean1 ? (mean_groups > 0 ? mean1[n * mean_groups + c / (C / mean_groups)] : mean1[c]) : 0.f;
    float wk[9], gw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { wk[k] = w[c * 9 + k]; gw[k] = 0.f; }
    __syncthreads();
    const float* hp = h1 + base + (size_t)y0 * W;
    float* dst = du1 + base + (size_t)y0 * W;
    const int m = rows * W;
    float s0 = 0.f, s1 = 0.f;
    for (int j = threadIdx.x; j < m; j += 256) {
        const int r = row_of(j, invW), x = j - r * W, y = y0 + r;
        const float hv = hp[j];
        const float u = fmaf(A1, hv, B1);
        const float gq = gelu_f(u);
        // the padded positions that mirror onto (y, x): rows {y, e1y, e2y}[0 .. ny), columns likewise
        const int ny = 1 + (y == 1) + (y == H - 2), nx = 1 + (x == 1) + (x == W - 2);
        const int e1y = (y == 1) ? -1 : H, e1x = (x == 1) ? -1 : W;
        float tk[9];
        {
            const float* tc = t + (r + 2) * pitch + x + 2;              // the pixel's own padded position: all an interior lane needs
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) tk[ky * 3 + kx] = tc[(1 - ky) * pitch + (1 - kx)];
        }
        if (ny + nx > 2) {                                              // the mirrored positions (rows / columns 1 and H-2 / W-2 only)
            for (int a = 0; a < ny; ++a) {
                const int ry = (a == 0 ? y : (a == 1 ? e1y : H)) - (y0 - 2);
                for (int b = (a == 0 ? 1 : 0); b < nx; ++b) {
                    const int cx = (b == 0 ? x : (b == 1 ? e1x : W)) + 2;
                    const float* tc = t + ry * pitch + cx;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) tk[ky * 3 + kx] += tc[(1 - ky) * pitch + (1 - kx)];
                }
            }
        }
        float dg = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            dg = fmaf(wk[k], tk[k], dg);
            gw[k] = fmaf(gq, tk[k], gw[k]);
        }
        const float v = gelu_grad_f(u) * dg;
        dst[j] = v;
        s0 += v;
        s1 = fmaf(v, hv - m1, s1);
    }
    if (blockIdx.x == gridDim.x - 1)
        for (int i = H * W + threadIdx.x; i < Pc; i += 256) du1[base + i] = 0.f;
    __shared__ float red[8];
    const size_t slot = (size_t)plane * gridDim.x + blockIdx.x;
    block_sum2<256>(s0, s1, red);
    if (threadIdx.x == 0) part[slot] = make_float2(s0, s1);
#pragma unroll
    for (int k = 0; k < 9; k += 2) {
        __syncthreads();
        float a = gw[k], b = k + 1 < 9 ? gw[k + 1] : 0.f;
        block_sum2<256>(a, b, red);
        if (threadIdx.x == 0) {
            dw_part[slot * 9 + k] = a;
            if (k + 1 < 9) dw_part[slot * 9 + k + 1] = b;
        }
    }
}
extern "C" int uncr_dw_bwd_any(const float* du2, const float* h2, const float* h1, const float* k1, const float* k2, const float* k3,
                               const float* kmu, const float* cA1, const float* cB1, const float* w, float* du1,
                               float* part /* [N*C][slots][2] */, float* dw_part /* [N*C][slots][9], slots = uncr_dw_any_slots(H, W, 1) */,
                               const float* mean1, int mean_groups, int N, int C, int H, int W, int Pc, hipStream_t stream) {
    const int tr = dw_band_rows(W, 1);
    if (N <= 0 || C <= 0 || H < 2 || !tr || Pc < H * W || Pc % 1024 || (mean1 && mean_groups > 0 && C % mean_groups)) return UNCR_ESHAPE;
    if (!du2 || !h2 || !h1 || !k1 || !k2 || !k3 || !cA1 || !cB1 || !w || !du1 || !part || !dw_part) return UNCR_EINVAL;
    hipLaunchKernelGGL(dw_bwd_band_kernel, dim3((H + tr - 1) / tr, N * C), dim3(256), (size_t)(tr + 5) * (W + 4) * sizeof(float), stream,
                       du2, h2, h1, k1, k2, k3, kmu, cA1, cB1, w, du1, (float2*)part, dw_part, mean1, mean_groups, C, H, W, Pc, tr,
                       1.0f / (float)W, 1.0f / (float)(W + 4));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
