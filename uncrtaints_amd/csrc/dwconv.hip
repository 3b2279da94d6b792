// Depthwise 3x3, padding 1 'reflect' (uncrtaints.py:130-131) on NCHW planes: im2col-free LDS-tiled
// stencil.  The producing norm + GELU is applied while the tile (with halo) is staged into LDS, so
// erf is evaluated once per element; the epilogue emits the partial statistics the next norm needs.
// No MFMA here: 9 FMAs per output, HBM-bound (read 4 B + write 4 B per element).
//
// forward : h2 = dw(gelu(A1*h1+B1));                     stats (sum h2, sum h2^2)
// backward: dh2 = C1*du2 + C2*h2 + C3   (norm-2 backward, fused prologue)
//           dg1 = dw^T(dh2)  (exact adjoint of reflect padding)
//           du1 = gelu'(A1*h1+B1) * dg1;                 stats (sum du1, sum du1*h1)
//           dWdw[c,tap] partials = sum_p dh2[p] * g1[reflect(p+tap)]
#include "common.h"
#include "bn_inline.h"

#define DW_TR_FWD 32
#define DW_TR_BWD 16

// LDS tile layout: row pitch = W + 8 floats; image column x lives at offset 4 + x, so the interior is 16-byte
// aligned (one ds_write_b128 per staged float4, one ds_read_b128 + two ds_read_b32 per stencil row); the
// reflect / zero halo columns sit at offsets 3 and 4 + W.
// Tile loops are division-free: thread -> (row r0 = tid / W4, column quad c4 = tid % W4) once, then r += 256 / W4.
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ in, const float* __restrict__ cA,
                                                     const float* __restrict__ cB, const float* __restrict__ w,
                                                     T* __restrict__ out, float2* __restrict__ part, int C,
                                                     int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float t[];   // [(TR+2)][W+8]
    constexpr int TR = DW_TR_FWD;
    const int plane = blockIdx.y, c = plane % C;
    const int y0 = blockIdx.x * TR;
    const int W4 = W >> 2, pitch = W + 8;
    const int rpp = 256 / W4;                       // rows per pass
    const int r0 = threadIdx.x / W4, c4 = threadIdx.x - r0 * W4;
    const bool active = r0 < rpp;
    const float A = cA[plane], B = cB[plane];
    const T* src = in + (size_t)plane * H * W;
    const int rows = min(TR, H - y0) + 2;
    if (active) {
        for (int r = r0; r < rows; r += rpp) {
            const int gy = reflect1(y0 - 1 + r, H);
            float4 v = ld4<T>(src + (size_t)gy * W + 4 * c4);
            v.x = gelu_f(fmaf(A, v.x, B));
            v.y = gelu_f(fmaf(A, v.y, B));
            v.z = gelu_f(fmaf(A, v.z, B));
            v.w = gelu_f(fmaf(A, v.w, B));
            *(float4*)(t + r * pitch + 4 + 4 * c4) = v;
            if (c4 == 0) t[r * pitch + 3] = v.y;                    // col -1 -> col 1
            if (c4 == W4 - 1) t[r * pitch + 4 + W] = v.z;           // col W  -> col W-2
        }
    }
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    T* dst = out + (size_t)plane * H * W;
    const int orows = rows - 2;
    if (active) {
        for (int r = r0; r < orows; r += rpp) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* row = t + (r + dy) * pitch + 4 + 4 * c4;
                const float4 m = *(const float4*)row;
                const float v[6] = {row[-1], m.x, m.y, m.z, m.w, row[4]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = fmaf(wk[dy * 3 + 0], v[j], fmaf(wk[dy * 3 + 1], v[j + 1], fmaf(wk[dy * 3 + 2], v[j + 2], o[j])));
            }
            const float4 ov = rnd4<T>(make_float4(o[0], o[1], o[2], o[3]));      // statistics of the values as stored
            st4<T>(dst + (size_t)(y0 + r) * W + 4 * c4, ov);
            s0 += (ov.x + ov.y) + (ov.z + ov.w);
            s1 += ov.x * ov.x + ov.y * ov.y + ov.z * ov.z + ov.w * ov.w;
        }
    }
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
    }
}

template <int NRMAX, typename T>   // row passes per thread: ceil((TR + 2) / (256 / (W/4)))
__global__ __launch_bounds__(256) void dw_bwd_kernel(
    const T* __restrict__ du2, const T* __restrict__ h2, const T* __restrict__ h1,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3,
    const float* __restrict__ kmu, const float* __restrict__ cA1, const float* __restrict__ cB1,
    const float* __restrict__ w, T* __restrict__ du1, float2* __restrict__ part, float* __restrict__ dw_part,
    const float* __restrict__ mean1, int mean_groups, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TR = DW_TR_BWD;
    const int plane = blockIdx.y, c = plane % C;
    const int y0 = blockIdx.x * TR;
    const int W4 = W >> 2, pitch = W + 8;
    const int rpp = 256 / W4;
    const int r0 = threadIdx.x / W4, c4 = threadIdx.x - r0 * W4;
    const bool active = r0 < rpp;
    float* Dt = sm;                          // zero-padded dh2 tile   (image col x at offset 4 + x)
    float* Gt = sm + (TR + 2) * pitch;       // reflect-padded g1 tile
    const float C1 = k1[plane], C2 = k2[plane], C3 = k3[plane];
    const float M2 = kmu ? kmu[plane] : 0.f;       // centred norm-2 backward: dh2 = C1*du2 + C2*(h2 - M2) + C3
    const float A1 = cA1[plane], B1 = cB1[plane];
    // second statistic sum du1*(h1 - M1): with M1 = the norm's mean it is free of the |mean|/std cancellation
    const float M1 = mean1 ? mean1[mean_groups > 0 ? (plane / C) * mean_groups + c / (C / mean_groups) : c] : 0.f;
    const size_t pbase = (size_t)plane * H * W;
    const int rows = min(TR, H - y0) + 2;
    // Staging: every load of the block is issued before the first use (one HBM round trip instead of one per row
    // pass), branch-free: rows outside the tile / image re-read a clamped row and are masked afterwards.  A thread's
    // rows are r0 + i*rpp; the raw h1 values stay in registers for the compute phase (same rows are its output rows).
    float4 ra[NRMAX], rb[NRMAX], rh[NRMAX];
#pragma unroll
    for (int i = 0; i < NRMAX; ++i) {
        const int r = r0 + i * rpp;
        const int y = y0 - 1 + r;
        const int yc = min(max(y, 0), H - 1);
        const int gy = min(max(reflect1(y, H), 0), H - 1);
        const int cc = active ? c4 : 0;
        ra[i] = ld4<T>(du2 + pbase + (size_t)yc * W + 4 * cc);
        rb[i] = ld4<T>(h2 + pbase + (size_t)yc * W + 4 * cc);
        rh[i] = ld4<T>(h1 + pbase + (size_t)gy * W + 4 * cc);
    }
#pragma unroll
    for (int i = 0; i < NRMAX; ++i) {
        const int r = r0 + i * rpp;
        const int y = y0 - 1 + r;
        if (active && r < rows) {                 // LDS writes only: no memory loads under this branch
            const float m = (y >= 0 && y < H) ? 1.f : 0.f;
            float4 d;
            d.x = m * fmaf(C1, ra[i].x, fmaf(C2, rb[i].x - M2, C3));
            d.y = m * fmaf(C1, ra[i].y, fmaf(C2, rb[i].y - M2, C3));
            d.z = m * fmaf(C1, ra[i].z, fmaf(C2, rb[i].z - M2, C3));
            d.w = m * fmaf(C1, ra[i].w, fmaf(C2, rb[i].w - M2, C3));
            *(float4*)(Dt + r * pitch + 4 + 4 * c4) = d;
            if (c4 == 0) Dt[r * pitch + 3] = 0.f;
            if (c4 == W4 - 1) Dt[r * pitch + 4 + W] = 0.f;
            float4 v;
            v.x = gelu_f(fmaf(A1, rh[i].x, B1));
            v.y = gelu_f(fmaf(A1, rh[i].y, B1));
            v.z = gelu_f(fmaf(A1, rh[i].z, B1));
            v.w = gelu_f(fmaf(A1, rh[i].w, B1));
            *(float4*)(Gt + r * pitch + 4 + 4 * c4) = v;
            if (c4 == 0) Gt[r * pitch + 3] = v.y;
            if (c4 == W4 - 1) Gt[r * pitch + 4 + W] = v.z;
        }
    }
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    __syncthreads();

    float s0 = 0.f, s1 = 0.f;
    float gw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gw[i] = 0.f;
    const int orows = rows - 2;
    // tile rows of image rows 0 and H-1 (only meaningful when the block holds them)
    const int er0 = 0 - (y0 - 1), er1 = (H - 1) - (y0 - 1);
#pragma unroll
    for (int i = 0; i < NRMAX; ++i) {
        const int r = r0 + i * rpp - 1;          // output row of the tile (staged tile row r + 1)
        if (active && r >= 0 && r < orows) {     // LDS reads and global stores only under this branch
            const int y = y0 + r;
            // adjoint of reflect padding: the plain transposed stencil on the zero-padded tile, plus the
            // contributions that forward reflection folded back onto rows/cols 1 and H-2 / W-2.
            const bool ry0 = (y == 1), ry1 = (y == H - 2);
            const size_t o = pbase + (size_t)y * W + 4 * c4;
            const float* ph = (const float*)&rh[i];          // raw h1 of this row, kept from the staging loads
            // rows r..r+2 of both tiles, columns x-1 .. x+4 (tile offsets 3+4c4 .. 8+4c4)
            float dt[3][6], gt[3][6];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* drow = Dt + (r + dy) * pitch + 4 + 4 * c4;
                const float4 dm = *(const float4*)drow;
                dt[dy][0] = drow[-1]; dt[dy][1] = dm.x; dt[dy][2] = dm.y; dt[dy][3] = dm.z; dt[dy][4] = dm.w; dt[dy][5] = drow[4];
                const float* grow = Gt + (r + dy) * pitch + 4 + 4 * c4;
                const float4 gm = *(const float4*)grow;
                gt[dy][0] = grow[-1]; gt[dy][1] = gm.x; gt[dy][2] = gm.y; gt[dy][3] = gm.z; gt[dy][4] = gm.w; gt[dy][5] = grow[4];
            }
            float res[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = 4 * c4 + j;
                float acc = 0.f;
                // transposed stencil: image (y - ty + 1, x - tx + 1) = tile row r + 2 - ty, local column j + 2 - tx
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) acc = fmaf(wk[ty * 3 + tx], dt[2 - ty][j + 2 - tx], acc);
                const bool cx0 = (x == 1), cx1 = (x == W - 2);
                if (ry0 || ry1 || cx0 || cx1) {     // rare: image border fix-ups straight from LDS
                    if (ry0) {
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) acc = fmaf(wk[tx], Dt[er0 * pitch + 4 + x + 1 - tx], acc);
                    }
                    if (ry1) {
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) acc = fmaf(wk[6 + tx], Dt[er1 * pitch + 4 + x + 1 - tx], acc);
                    }
                    if (cx0) {
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) acc = fmaf(wk[ty * 3], Dt[(r + 2 - ty) * pitch + 4], acc);
                        if (ry0) acc = fmaf(wk[0], Dt[er0 * pitch + 4], acc);
                        if (ry1) acc = fmaf(wk[6], Dt[er1 * pitch + 4], acc);
                    }
                    if (cx1) {
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) acc = fmaf(wk[ty * 3 + 2], Dt[(r + 2 - ty) * pitch + 4 + W - 1], acc);
                        if (ry0) acc = fmaf(wk[2], Dt[er0 * pitch + 4 + W - 1], acc);
                        if (ry1) acc = fmaf(wk[8], Dt[er1 * pitch + 4 + W - 1], acc);
                    }
                }
                const float u = fmaf(A1, ph[j], B1);
                float dv = gelu_grad_f(u) * acc;
                if constexpr (sizeof(T) == 2) dv = bf16_round(dv);      // statistics of the value as stored
                res[j] = dv;
                s0 += dv;
                s1 += dv * (ph[j] - M1);
                // depthwise weight gradient: dh2 at (y,x) times g1 at the reflect-padded neighbours
                const float dc = dt[1][j + 1];
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) gw[ty * 3 + tx] = fmaf(dc, gt[ty][j + tx], gw[ty * 3 + tx]);
            }
            st4<T>(du1 + o, make_float4(res[0], res[1], res[2], res[3]));
        }
    }
    __shared__ float red[4][12];
    s0 = wave_sum_dpp(s0);
    s1 = wave_sum_dpp(s1);
#pragma unroll
    for (int i = 0; i < 9; ++i) gw[i] = wave_sum_dpp(gw[i]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) {
        red[wv][0] = s0; red[wv][1] = s1;
#pragma unroll
        for (int i = 0; i < 9; ++i) red[wv][2 + i] = gw[i];
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t slot = (size_t)plane * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0) part[slot].x = v;
        else if (threadIdx.x == 1) part[slot].y = v;
        else dw_part[slot * 9 + (threadIdx.x - 2)] = v;
    }
}

// dWdw[c][tap] = sum over frames n and row tiles of dw_part[(n*C+c)][tile][tap]   (fp64, fixed order)
// grid = C, block = one wave: lanes = (slice, tap).  Latency-bound, so the N*NPT
// partials of a (channel, tap) are split over 7 lanes and read 4 at a time; fixed combination order.
__global__ __launch_bounds__(64) void dw_wgrad_reduce_kernel(const float* __restrict__ dw_part, int N, int C, int NPT,
                                                             float* __restrict__ dw) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int tap = lane % 9, sl = lane / 9;           // 7 slices x 9 taps = 63 lanes
    __shared__ double comb[7][9];
    if (lane < 63) {
        const int cnt = N * NPT;
        const int i1 = (cnt * (sl + 1)) / 7;
        int i = (cnt * sl) / 7;
        double s = 0.0;
        auto at = [&](int q) {
            const int n = q / NPT, j = q - n * NPT;
            return dw_part[(((size_t)n * C + c) * NPT + j) * 9 + tap];
        };
        for (; i + 4 <= i1; i += 4) {
            const float v0 = at(i), v1 = at(i + 1), v2 = at(i + 2), v3 = at(i + 3);
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; i < i1; ++i) s += (double)at(i);
        comb[sl][tap] = s;
    }
    __syncthreads();
    if (lane < 9) {
        double s = 0.0;
        for (int q = 0; q < 7; ++q) s += comb[q][lane];
        dw[c * 9 + lane] = (float)s;
    }
}

extern "C" int uncr_dw_slots_fwd(int H) { return (H + DW_TR_FWD - 1) / DW_TR_FWD; }
extern "C" int uncr_dw_slots_bwd(int H) { return (H + DW_TR_BWD - 1) / DW_TR_BWD; }

// dwconv_row.hip: streaming kernels for W == 256
int dw_fwd_row_launch(const void* in, const float* cA, const float* cB, const float* w, void* out, float* part, int N,
                      int C, int H, int slots, int act, const BnFin* fin, hipStream_t stream);
int dw_bwd_row_launch(const void* du2, const void* h2, const void* h1, const float* k1, const float* k2,
                      const float* k3, const float* kmu, const float* cA1, const float* cB1, const float* w, void* du1, float* part,
                      float* dw_part, const float* mean1, int mean_groups, int N, int C, int H, int slots, int act,
                      float* amax_out, hipStream_t stream);

template <typename T>
static int dw_fwd_tiled(const void* in, const float* cA, const float* cB, const float* w, void* out, float* part, int N, int C,
                        int H, int W, size_t lds, hipStream_t stream) {
    static size_t lds_attr = 0;
    if (lds > 60 * 1024 && lds > lds_attr) {   // > 64 KiB of dynamic LDS needs the opt-in attribute (W > 440)
        if (hipFuncSetAttribute((const void*)dw_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return UNCR_EINVAL;
        lds_attr = lds;
    }
    hipLaunchKernelGGL(dw_fwd_kernel<T>, dim3(uncr_dw_slots_fwd(H), N * C), dim3(256), lds, stream, (const T*)in, cA, cB, w,
                       (T*)out, (float2*)part, C, H, W);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_dw_fwd(const void* in, const float* cA, const float* cB, const float* w, void* out,
                           float* part, int N, int C, int H, int W, int act, int variant, hipStream_t stream) {
    if (N <= 0 || C <= 0 || H < 2 || W < 4 || (W & 3) || W > 1024) return UNCR_ESHAPE;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    if (variant == 0 && W == 256 && H >= 4 && (H & 3) == 0) return dw_fwd_row_launch(in, cA, cB, w, out, part, N, C, H, uncr_dw_slots_fwd(H), act, nullptr, stream);
    const size_t lds = (size_t)(DW_TR_FWD + 2) * (W + 8) * sizeof(float);
    if (lds > 150 * 1024) return UNCR_ESHAPE;
    if (act == UNCR_BF16) return dw_fwd_tiled<bf16_t>(in, cA, cB, w, out, part, N, C, H, W, lds, stream);
    return dw_fwd_tiled<float>(in, cA, cB, w, out, part, N, C, H, W, lds, stream);
}

// The depthwise forward with the train-mode BatchNorm of its input finalised by the kernel itself (bn_inline.h): fin_part
// [N*C][fin_NP] = the (sum, sum^2) partials the producer of `in` left; cA / cB / save_mean / save_rstd (/ ub / hb) are OUTPUTS
// here, written for the backward; running statistics updated as by uncr_norm_finalize_fwd(kind = BATCH_TRAIN).  Only where the
// row-streaming kernel runs (uncr_dw_fwd_bn_supported); other shapes: uncr_norm_finalize_fwd + uncr_dw_fwd.
extern "C" int uncr_dw_fwd_bn_supported(int H, int W) { return (W == 256 && H >= 4 && (H & 3) == 0) ? 1 : 0; }
extern "C" int uncr_dw_fwd_bn(const void* in, const float* fin_part, int fin_NP, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* cA, float* cB,
                              float* save_mean, float* save_rstd, float* ub, float* hb, const float* w, void* out, float* part,
                              int N, int C, int H, int W, int act, hipStream_t stream) {
    if (N <= 0 || C <= 0 || !uncr_dw_fwd_bn_supported(H, W)) return UNCR_ESHAPE;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    if (!in || !fin_part || fin_NP <= 0 || !gamma || !beta || !cA || !cB || !save_mean || !save_rstd || !w || !out) return UNCR_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr) || (hb && !ub)) return UNCR_EINVAL;
    const BnFin f{(const float2*)fin_part, fin_NP, N, gamma, beta, running_mean, running_var, momentum, eps, cA, cB, save_mean,
                  save_rstd, ub, hb};
    return dw_fwd_row_launch(in, cA, cB, w, out, part, N, C, H, uncr_dw_slots_fwd(H), act, &f, stream);
}

template <typename T>
static int dw_bwd_tiled(const void* du2, const void* h2, const void* h1, const float* k1, const float* k2, const float* k3,
                        const float* kmu, const float* cA1, const float* cB1, const float* w, void* du1, float* part, float* dw_part,
                        const float* mean1, int mean_groups, int N, int C, int H, int W, size_t lds, hipStream_t stream) {
    auto kern = W <= 256 ? dw_bwd_kernel<5, T> : (W <= 512 ? dw_bwd_kernel<9, T> : dw_bwd_kernel<18, T>);
    static size_t lds_attr[3] = {0, 0, 0};
    const int ki = W <= 256 ? 0 : (W <= 512 ? 1 : 2);
    if (lds > lds_attr[ki]) {   // > 64 KiB of dynamic LDS needs the opt-in attribute (gfx950 has 160 KiB per CU)
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return UNCR_EINVAL;
        lds_attr[ki] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(uncr_dw_slots_bwd(H), N * C), dim3(256), lds, stream, (const T*)du2, (const T*)h2,
                       (const T*)h1, k1, k2, k3, kmu, cA1, cB1, w, (T*)du1, (float2*)part, dw_part, mean1, mean_groups, C, H, W);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// the row-streaming kernel (fp32 storage) can leave max |du1| per statistics slot; the LDS-tiled kernels do not
extern "C" int uncr_dw_bwd_emits_amax(int H, int W, int act, int variant) {
    return (variant == 0 && W == 256 && (H & 3) == 0 && act == UNCR_F32) ? 1 : 0;
}

extern "C" int uncr_dw_bwd(const void* du2, const void* h2, const void* h1, const float* k1, const float* k2,
                           const float* k3, const float* kmu, const float* cA1, const float* cB1, const float* w, void* du1,
                           float* part, float* dw_part, const float* mean1, int mean_groups, int N, int C, int H,
                           int W, int act, int variant, float* amax_out, hipStream_t stream) {
    if (mean1 && mean_groups > 0 && C % mean_groups) return UNCR_ESHAPE;
    if (N <= 0 || C <= 0 || H < 4 || W < 4 || (W & 3) || W > 1024) return UNCR_ESHAPE;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    if (amax_out && !uncr_dw_bwd_emits_amax(H, W, act, variant)) return UNCR_EINVAL;
    if (variant == 0 && W == 256 && (H & 3) == 0)
        return dw_bwd_row_launch(du2, h2, h1, k1, k2, k3, kmu, cA1, cB1, w, du1, part, dw_part, mean1, mean_groups, N, C, H,
                                 uncr_dw_slots_bwd(H), act, amax_out, stream);
    const size_t lds = (size_t)2 * (DW_TR_BWD + 2) * (W + 8) * sizeof(float);
    if (lds > 150 * 1024) return UNCR_ESHAPE;
    if (act == UNCR_BF16)
        return dw_bwd_tiled<bf16_t>(du2, h2, h1, k1, k2, k3, kmu, cA1, cB1, w, du1, part, dw_part, mean1, mean_groups, N, C, H, W, lds, stream);
    return dw_bwd_tiled<float>(du2, h2, h1, k1, k2, k3, kmu, cA1, cB1, w, du1, part, dw_part, mean1, mean_groups, N, C, H, W, lds, stream);
}

extern "C" int uncr_dw_wgrad_reduce(const float* dw_part, int N, int C, int NPT, float* dw, hipStream_t stream) {
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(C), dim3(64), 0, stream, dw_part, N, C, NPT, dw);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
