// Dense 3x3 convolution, padding 1 'reflect' (ResidualConvBlock, uncrtaints.py:24-69; utae.py:478-487), built from the
// pointwise GEMMs: on the PADDED grid a tap (dy, dx) is a pure pointer offset dy*(W+2)+dx, so
//     out_p[co][q] = sum_taps W_tap[co][ci] * x_p[ci][q + off_tap]
// is nine accumulating 1x1 GEMMs (pw_gemm epi 4) on padded planes, the data gradient nine more with the offsets
// negated on a zero-padded gradient, and the weight gradient nine 128x128 contractions with an offset operand.
// This file holds the glue: padding (with the producing norm + ReLU or norm-backward fused in), un-padding with the
// statistics of the next norm, and the adjoint of the reflect padding.  Padded plane: (H+2)*(W+2) values, pixel (y,x)
// at (y+1)*(W+2) + (x+1), zero tail up to the plane stride S_p (a multiple of 1024).
#include "common.h"

// mode 0 reflect, 1 zero.  pro: PRO_NONE | PRO_AFFINE_RELU (k0, k1) | PRO_NORMBWD (k0*src + k1*src2 + k2)
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ src, const float* __restrict__ src2,
                                                    float* __restrict__ dst, const float* __restrict__ k0,
                                                    const float* __restrict__ k1, const float* __restrict__ k2,
                                                    const float* __restrict__ kmu, int pro, int mode, int H, int W, int Sp,
                                                    int Ps /* source plane stride >= H*W */) {
    const int plane = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Sp) return;
    const int Wp = W + 2;
    float v = 0.f;
    if (q < (H + 2) * Wp) {
        int y = q / Wp - 1, x = q % Wp - 1;
        const bool ring = y < 0 || y >= H || x < 0 || x >= W;
        if (!(ring && mode == 1)) {
            y = reflect1(y, H); x = reflect1(x, W);
            const size_t o = (size_t)plane * Ps + (size_t)y * W + x;
            v = src[o];
            if (pro == PRO_AFFINE_RELU) v = fmaxf(fmaf(k0[plane], v, k1[plane]), 0.f);
            else if (pro == PRO_NORMBWD) v = fmaf(k0[plane], v, fmaf(k1[plane], src2[o] - (kmu ? kmu[plane] : 0.f), k2[plane]));
            else if (pro == PRO_AFFINE) v = fmaf(k0[plane], v, k1[plane]);
        }
    }
    dst[(size_t)plane * Sp + q] = v;
}

// interior of a padded plane -> [planes][H][W] (plane stride Ps >= H*W, the tail written as zeros), with (sum, sum^2) partials per
// 1024-pixel chunk
__global__ __launch_bounds__(256) void unpad2d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      float2* __restrict__ part, int H, int W, int Sp, int Ps) {
    const int plane = blockIdx.y;
    const int P = H * W, Wp = W + 2;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * 1024 + i * 256 + threadIdx.x;
        if (p < P) {
            const int y = p / W, x = p % W;
            const float v = src[(size_t)plane * Sp + (size_t)(y + 1) * Wp + x + 1];
            dst[(size_t)plane * Ps + p] = v;
            s0 += v;
            s1 = fmaf(v, v, s1);
        } else if (p < Ps) {
            dst[(size_t)plane * Ps + p] = 0.f;
        }
    }
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
    }
}

// adjoint of the reflect padding: gradient on the padded grid -> gradient of the un-padded image.  Pixel (y, x)
// collects its own padded position plus the ring positions that mirror onto it (rows -1 -> 1, H -> H-2; same for
// columns; corners through both).
__global__ __launch_bounds__(256) void unpad2d_reflect_adjoint_kernel(const float* __restrict__ src,
                                                                      float* __restrict__ dst, int H, int W, int Sp, int Ps) {
    const int plane = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) {
        if (p < Ps) dst[(size_t)plane * Ps + p] = 0.f;
        return;
    }
    const int y = p / W, x = p % W, Wp = W + 2;
    const float* s = src + (size_t)plane * Sp;
    int ys[2] = {y + 1, -1}, xs[2] = {x + 1, -1};
    if (y == 1) ys[1] = 0;
    if (y == H - 2) ys[1] = H + 1;          // H >= 4: rows 1 and H-2 are distinct
    if (x == 1) xs[1] = 0;
    if (x == W - 2) xs[1] = W + 1;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (ys[i] >= 0 && xs[j] >= 0) a += s[(size_t)ys[i] * Wp + xs[j]];
    dst[(size_t)plane * Ps + p] = a;
}

extern "C" int uncr_conv3_plane_stride(int H, int W) {          // S_p: padded plane stride (floats)
    if (H < 4 || W < 4) return -1;
    // the zero tail must cover the largest tap offset (W + 3): ring positions of the NEXT plane read backwards into it
    return (((H + 2) * (W + 2)) + (W + 3) + 1023) / 1024 * 1024;
}
extern "C" int uncr_conv3_margin(int W) { return W + 3; }       // slack (floats) the caller keeps before and after the tensor

static int pad2d_launch(const float* src, const float* src2, float* dst, const float* k0, const float* k1, const float* k2,
                        const float* kmu, int pro, int mode, int planes, int H, int W, int Ps, hipStream_t stream) {
    const int Sp = uncr_conv3_plane_stride(H, W);
    if (planes <= 0 || Sp <= 0 || mode < 0 || mode > 1 || Ps < H * W) return UNCR_ESHAPE;
    if (!src || !dst || (pro == PRO_NORMBWD && (!src2 || !k0 || !k1 || !k2)) ||
        ((pro == PRO_AFFINE_RELU || pro == PRO_AFFINE) && (!k0 || !k1)))
        return UNCR_EINVAL;
    hipLaunchKernelGGL(pad2d_kernel, dim3(Sp / 256, planes), dim3(256), 0, stream, src, src2, dst, k0, k1, k2, kmu, pro,
                       mode, H, W, Sp, Ps);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_pad2d(const float* src, const float* src2, float* dst, const float* k0, const float* k1,
                          const float* k2, const float* kmu, int pro, int mode, int planes, int H, int W, hipStream_t stream) {
    return pad2d_launch(src, src2, dst, k0, k1, k2, kmu, pro, mode, planes, H, W, H * W, stream);
}
extern "C" int uncr_unpad2d(const float* src, float* dst, float* part, int planes, int H, int W, hipStream_t stream) {
    const int Sp = uncr_conv3_plane_stride(H, W);
    if (planes <= 0 || Sp <= 0 || ((H * W) % 1024)) return UNCR_ESHAPE;
    hipLaunchKernelGGL(unpad2d_kernel, dim3(H * W / 1024, planes), dim3(256), 0, stream, src, dst, (float2*)part, H, W, Sp, H * W);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_unpad2d_reflect_adjoint(const float* src, float* dst, int planes, int H, int W, hipStream_t stream) {
    const int Sp = uncr_conv3_plane_stride(H, W);
    if (planes <= 0 || Sp <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(unpad2d_reflect_adjoint_kernel, dim3((H * W + 255) / 256, planes), dim3(256), 0, stream, src,
                       dst, H, W, Sp, H * W);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// ---- the same three on the dense planes of an any-size image (csrc/anysize.hip): plane stride Ps = a multiple of 1024 >= H*W; the
// un-padded side's tail [H*W, Ps) is never read and is written as zeros; part [planes][Ps/1024][2] ----
extern "C" int uncr_pad2d_strided(const float* src, const float* src2, float* dst, const float* k0, const float* k1,
                                  const float* k2, const float* kmu, int pro, int mode, int planes, int H, int W, int Ps,
                                  hipStream_t stream) {
    return pad2d_launch(src, src2, dst, k0, k1, k2, kmu, pro, mode, planes, H, W, Ps, stream);
}
extern "C" int uncr_unpad2d_strided(const float* src, float* dst, float* part, int planes, int H, int W, int Ps, hipStream_t stream) {
    const int Sp = uncr_conv3_plane_stride(H, W);
    if (planes <= 0 || Sp <= 0 || Ps < H * W || (Ps % 1024)) return UNCR_ESHAPE;
    if (!src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(unpad2d_kernel, dim3(Ps / 1024, planes), dim3(256), 0, stream, src, dst, (float2*)part, H, W, Sp, Ps);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_unpad2d_reflect_adjoint_strided(const float* src, float* dst, int planes, int H, int W, int Ps,
                                                    hipStream_t stream) {
    const int Sp = uncr_conv3_plane_stride(H, W);
    if (planes <= 0 || Sp <= 0 || Ps < H * W) return UNCR_ESHAPE;
    if (!src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(unpad2d_reflect_adjoint_kernel, dim3((Ps + 255) / 256, planes), dim3(256), 0, stream, src, dst, H, W, Sp, Ps);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
