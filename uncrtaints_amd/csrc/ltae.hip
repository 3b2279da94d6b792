// L-TAE (LTAE2dtiny, ltae.py:145-239) low-resolution branch.  Runs at 32x32 (uncrtaints.py:403), <1 % of
// the bytes of the path; kernels are simple, coalesced along the pixel axis.  The two linear maps
// (Conv1d 128->256, Linear 256->64) reuse the MFMA pointwise GEMM on [frames, C, 1024] planes.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// adaptive max-pool [planes][H][W] -> [planes][OH][OW] (+ argmax as flat in-plane index)
// (nn.AdaptiveMaxPool2d window: start = floor(i*H/OH), end = ceil((i+1)*H/OH); first max wins)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                          int* __restrict__ idx, int H, int W, int OH, int OW, int pstride) {
    const int plane = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= OH * OW) return;
    const int oy = o / OW, ox = o % OW;
    const int ys = (oy * H) / OH, ye = ((oy + 1) * H + OH - 1) / OH;
    const int xs = (ox * W) / OW, xe = ((ox + 1) * W + OW - 1) / OW;
    const T* p = in + (size_t)plane * pstride;      // (pstride = H*W, or the padded plane stride of csrc/anysize.hip)
    float best = -INFINITY;
    int bi = ys * W + xs;
    if (((xe - xs) & 3) == 0 && (xs & 3) == 0 && (W & 3) == 0) {
        // 16-byte loads along the window rows (the 256->32 case: two float4 per row); same scan order
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; x += 4) {
                const float4 q = ld4<T>(p + y * W + x);
                const float vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vv[j] > best || vv[j] != vv[j]) { best = vv[j]; bi = y * W + x + j; }
            }
    } else {
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) {
                const float v = ld1<T>(p + y * W + x);
                if (v > best || v != v) { best = v; bi = y * W + x; }   // NaN propagates like ATen
            }
    }
    out[(size_t)plane * OH * OW + o] = best;
    idx[(size_t)plane * OH * OW + o] = bi;
}

// ---------------------------------------------------------------------------------------------
// Last encoder block: MBConv's closing residual  y = x + A*h3 + B  (uncrtaints.py:146) with the 8x8 max-pool of the
// L-TAE stage (uncrtaints.py:403-404) taken while y is in registers -- the pool's own pass over e [B,T,C,H,W]
// (403 MB at the bench shape) disappears.  Built for W == 256 and 8x8 windows (256^2 -> 32^2): one wave = one strip of
// 8 image rows, lane = 4 pixels; per-lane running (max, first index) over the 8 rows, then the two lanes of a window
// are merged.  Scan-order semantics of the stand-alone kernel: first maximum in row-major order wins, NaN propagates.
// grid = (H/32, planes), block = 256 = 4 strips.  part: [planes][H/8] (sum y, sum y^2) or null.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void residual_pool_kernel(const T* __restrict__ x, const T* __restrict__ h3,
                                                            const float* __restrict__ cA, const float* __restrict__ cB,
                                                            T* __restrict__ out, float2* __restrict__ part,
                                                            float* __restrict__ down, int* __restrict__ idx, int H) {
    constexpr int W = 256;
    const int plane = blockIdx.y, lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip * 8 >= H) return;
    const float A = cA[plane], B = cB[plane];
    const size_t base = (size_t)plane * H * W + (size_t)strip * 8 * W + 4 * lane;
    float4 xv[8], hv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        xv[r] = ld_nt4t(x + base + (size_t)r * W);
        hv[r] = ld_nt4t(h3 + base + (size_t)r * W);
    }
    float best = -INFINITY, s0 = 0.f, s1 = 0.f;
    int bi = strip * 8 * W + 4 * lane;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float4 o;
        o.x = xv[r].x + fmaf(A, hv[r].x, B);
        o.y = xv[r].y + fmaf(A, hv[r].y, B);
        o.z = xv[r].z + fmaf(A, hv[r].z, B);
        o.w = xv[r].w + fmaf(A, hv[r].w, B);
        o = rnd4<T>(o);      // statistics and the pooled maximum of the values as stored
        st_nt4t(out + base + (size_t)r * W, o);
        const float vv[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0 += vv[j];
            s1 = fmaf(vv[j], vv[j], s1);
            if (vv[j] > best || vv[j] != vv[j]) { best = vv[j]; bi = (strip * 8 + r) * W + 4 * lane + j; }
        }
    }
    // merge the two half-windows (lanes 2k, 2k+1).  Row-major scan order inside a window == order of the flat index,
    // so on equal maxima the smaller index wins; a NaN anywhere wins (the later one in scan order, like the serial scan).
    const float ob = __shfl_xor(best, 1);
    const int oi = __shfl_xor(bi, 1);
    const bool me_nan = best != best, ot_nan = ob != ob;
    bool take;
    if (me_nan || ot_nan) take = ot_nan && (!me_nan || oi > bi);
    else take = ob > best || (ob == best && oi < bi);
    if (take) { best = ob; bi = oi; }
    if ((lane & 1) == 0) {
        const size_t o = ((size_t)plane * (H / 8) + strip) * 32 + (lane >> 1);
        down[o] = best;
        idx[o] = bi;
    }
    if (part) {
        s0 = wave_sum_dpp(s0);
        s1 = wave_sum_dpp(s1);
        if (lane == 63) part[(size_t)plane * (H / 8) + strip] = make_float2(s0, s1);
    }
}

// de[plane][idx] += dpooled   (windows are disjoint when H % OH == 0; otherwise atomics)
// bf16 storage: two neighbouring elements share a dword and may belong to different windows (threads): the update is a
// compare-and-swap on the containing dword (the value is rounded once per contribution).
__device__ __forceinline__ void atomic_add_bf16(bf16_t* dst, float v) {
    unsigned* w = (unsigned*)((size_t)dst & ~(size_t)3);
    const bool hi = ((size_t)dst & 2) != 0;
    unsigned old = *w, assumed;
    do {
        assumed = old;
        const float cur = hi ? bf16_hi(assumed) : bf16_lo(assumed);
        const unsigned nv = cvt_pk_bf16(cur + v, 0.f) & 0xFFFFu;
        const unsigned repl = hi ? ((assumed & 0x0000FFFFu) | (nv << 16)) : ((assumed & 0xFFFF0000u) | nv);
        old = atomicCAS(w, assumed, repl);
    } while (old != assumed);
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx,
                                                          T* __restrict__ din, int HW, int OHW, int disjoint) {
    const int plane = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= OHW) return;
    const size_t q = (size_t)plane * OHW + o;
    T* dst = din + (size_t)plane * HW + idx[q];
    if constexpr (sizeof(T) == 2) atomic_add_bf16(dst, dout[q]);
    else {
        if (disjoint) *dst += dout[q];
        else atomicAdd(dst, dout[q]);
    }
}

// The pooled-gradient scatter fused with the statistics pass the last encoder block's backward needs anyway:
//   de[plane][argmax] += dpooled   and   part = (sum de, sum de*h3) of the UPDATED tensor, in one pass over de and h3
// (the stand-alone scatter is a sparse read-modify-write of its own, 52 us at the bench shape).  Disjoint windows whose width is a
// multiple of 4 (a thread's four pixels share a window): H % OH == 0, W % OW == 0, (W / OW) % 4 == 0.  grid = (P/1024, planes).
template <typename T>
__global__ __launch_bounds__(256) void pool_scatter_stats_kernel(const float* __restrict__ dpool, const int* __restrict__ idx,
                                                                 T* __restrict__ de, const T* __restrict__ h3,
                                                                 float2* __restrict__ part, int H, int W, int OH, int OW,
                                                                 float* __restrict__ amax_out) {
    const int plane = blockIdx.y;
    const int p = blockIdx.x * 1024 + threadIdx.x * 4;
    const size_t off = (size_t)plane * H * W + p;
    float4 v = ld4<T>(de + off);
    const float4 hv = ld4<T>(h3 + off);
    const int y = p / W, x0 = p - y * W;
    const size_t q = (size_t)plane * OH * OW + (size_t)(y / (H / OH)) * OW + x0 / (W / OW);
    const int k = idx[q] - p;                 // position of the window's arg-max relative to this thread's pixels
    if (k >= 0 && k < 4) {
        ((float*)&v)[k] += dpool[q];
        v = rnd4<T>(v);
        st4<T>(de + off, v);
    }
    float s0 = (v.x + v.y) + (v.z + v.w);
    float s1 = fmaf(v.x, hv.x, fmaf(v.y, hv.y, fmaf(v.z, hv.z, v.w * hv.w)));
    __shared__ float red[8];
    block_sum2<256>(s0, s1, red);
    if (threadIdx.x == 0) part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
    if (amax_out) {      // kernel-uniform: this block's max |de| (the consumer reduces a frame's entries; an atomic per frame would serialise
                         // ~8000 blocks on one address)
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
        __shared__ float mr[4];
        __syncthreads();
        if ((threadIdx.x & 63) == 0) mr[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) amax_out[(size_t)plane * gridDim.x + blockIdx.x] = fmaxf(fmaxf(mr[0], mr[1]), fmaxf(mr[2], mr[3]));
    }
}
// The same pass with CH chunks of a plane per block: every lane has its 2 * CH operand loads (and the CH window indices) in flight
// before the first dependent instruction.  One 16-byte load per lane and stream in short-lived blocks reads at 4.0-5.5 TB/s on this
// part, four at 6.5-7.1 (tools/probe_oneshot_blocks.py: the stream probe at 1 / 4 float4 per lane); the one-chunk kernel ran its
// 805 MB at 4.0 TB/s.  The per-chunk arithmetic, the wave sums (DPP) and the in-order sum of the four waves are those of the
// one-chunk kernel and block_sum2, the partial / maximum slots are the same ones: bit-identical results.  grid = (P / (CH * 1024), planes).
// Measured at the step's shape (tools/bench_pool_scatter.py, de rewritten by a copy kernel before the call as the aggregation backward
// leaves it): bf16 storage 153 -> 96 us with four chunks (8-byte loads: the one-chunk kernel is short of bytes in flight), fp32 storage
// 200 -> 181 us with two, 191 us with four, 202 us with eight -- fp32 gains little, whatever bounds it there is not the loads in flight.
#ifndef PSS_CH_F32
#define PSS_CH_F32 2      // chunks per block, fp32 storage (0: the one-chunk kernel)
#endif
#ifndef PSS_CH_BF16
#define PSS_CH_BF16 4     // chunks per block, bf16 storage
#endif
#ifndef PSS_NT
#define PSS_NT 1      // non-temporal operand loads in the multi-chunk kernel
#endif
template <typename T, int CH>
__global__ __launch_bounds__(256) void pool_scatter_stats_multi_kernel(const float* __restrict__ dpool, const int* __restrict__ idx,
                                                                       T* __restrict__ de, const T* __restrict__ h3,
                                                                       float2* __restrict__ part, int H, int W, int OH, int OW,
                                                                       float* __restrict__ amax_out) {
    const int plane = blockIdx.y;
    const int p0 = blockIdx.x * (CH * 1024) + threadIdx.x * 4;
    const size_t off0 = (size_t)plane * H * W + p0;
    float4 v[CH], hv[CH];
    int k[CH];
    size_t q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        v[c] = ld4<T, PSS_NT != 0>(de + off0 + c * 1024);
        hv[c] = ld4<T, PSS_NT != 0>(h3 + off0 + c * 1024);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int p = p0 + c * 1024;
        const int y = p / W, x0 = p - y * W;
        q[c] = (size_t)plane * OH * OW + (size_t)(y / (H / OH)) * OW + x0 / (W / OW);
        k[c] = idx[q[c]] - p;
    }
    __shared__ float red[CH][4][2], mr[CH][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (k[c] >= 0 && k[c] < 4) {
            const float d = dpool[q[c]];      // (selects, not a dynamic register index: that would put the operands in LDS)
            v[c].x = k[c] == 0 ? v[c].x + d : v[c].x; v[c].y = k[c] == 1 ? v[c].y + d : v[c].y;
            v[c].z = k[c] == 2 ? v[c].z + d : v[c].z; v[c].w = k[c] == 3 ? v[c].w + d : v[c].w;
            v[c] = rnd4<T>(v[c]);
            st4<T>(de + off0 + c * 1024, v[c]);
        }
        float s0 = (v[c].x + v[c].y) + (v[c].z + v[c].w);
        float s1 = fmaf(v[c].x, hv[c].x, fmaf(v[c].y, hv[c].y, fmaf(v[c].z, hv[c].z, v[c].w * hv[c].w)));
        s0 = wave_sum_dpp(s0);
        s1 = wave_sum_dpp(s1);
        if (lane == 63) { red[c][w][0] = s0; red[c][w][1] = s1; }
        if (amax_out) {
            float m = fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].y)), fmaxf(fabsf(v[c].z), fabsf(v[c].w)));
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
            if (lane == 0) mr[c][w] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x < CH) {
        const int c = threadIdx.x;
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sa += red[c][i][0]; sb += red[c][i][1]; }
        const size_t slot = (size_t)plane * (gridDim.x * CH) + blockIdx.x * CH + c;
        part[slot] = make_float2(sa, sb);
        if (amax_out) amax_out[slot] = fmaxf(fmaxf(mr[c][0], mr[c][1]), fmaxf(mr[c][2], mr[c][3]));
    }
}
extern "C" int uncr_pool_scatter_stats_supported(int H, int W, int OH, int OW) {
    return (OH > 0 && OW > 0 && H % OH == 0 && W % OW == 0 && ((W / OW) & 3) == 0 && ((H * W) % 1024) == 0) ? 1 : 0;
}
extern "C" int uncr_pool_scatter_stats(const float* dpool, const int* idx, void* de, const void* h3, float* part, int planes,
                                       int H, int W, int OH, int OW, int act, float* amax_out, hipStream_t stream) {
    if (planes <= 0 || !uncr_pool_scatter_stats_supported(H, W, OH, OW)) return UNCR_ESHAPE;
    if (!dpool || !idx || !de || !h3 || !part || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
#if PSS_CH_F32 > 0
    if (act == UNCR_F32 && (H * W) % (PSS_CH_F32 * 1024) == 0) {
        hipLaunchKernelGGL((pool_scatter_stats_multi_kernel<float, PSS_CH_F32>), dim3(H * W / (PSS_CH_F32 * 1024), planes), dim3(256), 0,
                           stream, dpool, idx, (float*)de, (const float*)h3, (float2*)part, H, W, OH, OW, amax_out);
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
#endif
#if PSS_CH_BF16 > 0
    if (act == UNCR_BF16 && (H * W) % (PSS_CH_BF16 * 1024) == 0) {
        hipLaunchKernelGGL((pool_scatter_stats_multi_kernel<bf16_t, PSS_CH_BF16>), dim3(H * W / (PSS_CH_BF16 * 1024), planes), dim3(256), 0,
                           stream, dpool, idx, (bf16_t*)de, (const bf16_t*)h3, (float2*)part, H, W, OH, OW, amax_out);
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
#endif
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(pool_scatter_stats_kernel<T>, dim3(H * W / 1024, planes), dim3(256), 0, stream,
                                                 dpool, idx, (T*)de, (const T*)h3, (float2*)part, H, W, OH, OW, amax_out));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---------------------------------------------------------------------------------------------
// per-pixel GroupNorm over (Cg channels x T dates)  (ltae.py:191-194,211; n_head groups)
// x [B][T][C][S] (S = low-res pixels), thread = (b, g, s)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ltae_gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd, int T, int C, int G, int S) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    if (s >= S) return;
    const int Cg = C / G;
    float sum = 0.f, sq = 0.f;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < Cg; ++j) {
            const float v = x[(((size_t)b * T + t) * C + g * Cg + j) * S + s];
            sum += v;
            sq = fmaf(v, v, sq);
        }
    const float M = (float)(T * Cg);
    const float mu = sum / M;
    // second pass for the variance (numerically safer than E[x^2]-mu^2 at 24 samples)
    float var = 0.f;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < Cg; ++j) {
            const float d = x[(((size_t)b * T + t) * C + g * Cg + j) * S + s] - mu;
            var = fmaf(d, d, var);
        }
    var /= M;
    const float r = 1.0f / sqrtf(var + eps);
    mean[((size_t)b * G + g) * S + s] = mu;
    rstd[((size_t)b * G + g) * S + s] = r;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < Cg; ++j) {
            const int c = g * Cg + j;
            const size_t o = (((size_t)b * T + t) * C + c) * S + s;
            y[o] = (x[o] - mu) * r * gamma[c] + beta[c];
        }
    (void)sq;
}

// backward; gb_part[(b*nchunk+chunk)][C][2] = per-block (d gamma, d beta) partials.  Cg <= CGMAX (16, or 32 for four heads at 128 channels).
template <int CGMAX>
__global__ __launch_bounds__(256) void ltae_gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          float* __restrict__ gb_part, int T, int C, int G, int S) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    const int Cg = C / G;
    const bool act = s < S;
    float mu = 0.f, r = 0.f, m1 = 0.f, m2 = 0.f;
    float dgam[CGMAX], dbet[CGMAX];
#pragma unroll
    for (int j = 0; j < CGMAX; ++j) { dgam[j] = 0.f; dbet[j] = 0.f; }
    if (act) {
        mu = mean[((size_t)b * G + g) * S + s];
        r = rstd[((size_t)b * G + g) * S + s];
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < CGMAX; ++j)
                if (j < Cg) {
                    const int c = g * Cg + j;
                    const size_t o = (((size_t)b * T + t) * C + c) * S + s;
                    const float d = dy[o], xh = (x[o] - mu) * r;
                    m1 = fmaf(gamma[c], d, m1);
                    m2 = fmaf(gamma[c] * d, xh, m2);
                    dgam[j] = fmaf(d, xh, dgam[j]);
                    dbet[j] += d;
                }
        const float M = (float)(T * Cg);
        m1 /= M;
        m2 /= M;
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < Cg; ++j) {
                const int c = g * Cg + j;
                const size_t o = (((size_t)b * T + t) * C + c) * S + s;
                const float xh = (x[o] - mu) * r;
                dx[o] = r * (gamma[c] * dy[o] - m1 - xh * m2);
            }
    }
    __shared__ float red[4][2 * CGMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CGMAX; ++j) {
        const float a = wave_sum(dgam[j]), bb = wave_sum(dbet[j]);
        if (lane == 0) { red[wv][2 * j] = a; red[wv][2 * j + 1] = bb; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * Cg) {
        const int j = threadIdx.x >> 1, which = threadIdx.x & 1;
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        gb_part[(blk * C + g * Cg + j) * 2 + which] = v;
    }
}

// out[r][k] summed over r (fp64, fixed order): generic column sum for small partial arrays
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int R, int K,
                                                     float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    part += (size_t)blockIdx.y * R * K;       // batched form: [batches][R][K] -> [batches][K]
    out += (size_t)blockIdx.y * K;
    double s = 0.0;
    int r = 0;
    for (; r + 8 <= R; r += 8) {          // eight loads in flight, summed in a fixed order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(r + q) * K + k];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
    for (; r < R; ++r) s += (double)part[(size_t)r * K + k];
    out[k] = (float)s;
}

// ---------------------------------------------------------------------------------------------
// per-frame bias of the first linear map:  bias[n][c] = b_in[c] + PE(date[n], c)
// PositionalEncoder (positional_encoding.py:5-31): channel c uses sinusoid i = c mod d, even -> sin,
// odd -> cos, argument date / denom[i].  denom is computed on the host exactly like the reference
// __init__ does (torch.pow), so the only device transcendental is sinf/cosf with full range reduction.
// ---------------------------------------------------------------------------------------------
__global__ void ltae_posbias_kernel(const float* __restrict__ dates, const float* __restrict__ denom, int d,
                                    const float* __restrict__ bin, float* __restrict__ out, int NF, int D,
                                    int use_pe) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NF * D) return;
    const int n = idx / D, c = idx % D;
    float v = bin ? bin[c] : 0.f;
    if (use_pe) {
        const int i = c % d;
        const float a = dates[n] / denom[i];
        v += (i & 1) ? cosf(a) : sinf(a);
    }
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// temporal softmax (ltae.py:341-385, 431-458): k [B][T][NH*DK][S], Q [NH][DK] -> att [NH][B][T][S]
//   score = (Q[h] . k[b,t,h*DK:(h+1)*DK, s]) / sqrt(DK); pad -> -1e3; softmax over t.
// thread = (b, h, s)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ltae_score(const float* __restrict__ k, const float* q, int DK, size_t base,
                                            int S, float inv_temp) {
    float a = 0.f;
    for (int d = 0; d < DK; ++d) a = fmaf(q[d], k[base + (size_t)d * S], a);
    return a * inv_temp;
}

__global__ __launch_bounds__(256) void ltae_softmax_fwd_kernel(const float* __restrict__ k,
                                                               const float* __restrict__ Q,
                                                               const int* __restrict__ pad, float* __restrict__ att,
                                                               int B, int T, int NH, int DK, int S, float inv_temp) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    if (s >= S) return;
    float q[8];
    for (int d = 0; d < DK; ++d) q[d] = Q[h * DK + d];
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) {
        float sc = ltae_score(k, q, DK, (((size_t)b * T + t) * NH * DK + h * DK) * S + s, S, inv_temp);
        if (pad && pad[b * T + t]) sc = -1e3f;
        mx = fmaxf(mx, sc);
    }
    float den = 0.f;
    for (int t = 0; t < T; ++t) {
        float sc = ltae_score(k, q, DK, (((size_t)b * T + t) * NH * DK + h * DK) * S + s, S, inv_temp);
        if (pad && pad[b * T + t]) sc = -1e3f;
        den += expf(sc - mx);
    }
    const float inv = 1.0f / den;
    for (int t = 0; t < T; ++t) {
        float sc = ltae_score(k, q, DK, (((size_t)b * T + t) * NH * DK + h * DK) * S + s, S, inv_temp);
        if (pad && pad[b * T + t]) sc = -1e3f;
        att[(((size_t)h * B + b) * T + t) * S + s] = expf(sc - mx) * inv;
    }
}

// backward: datt, att -> dk [B][T][NH*DK][S]; dq_part[(b*nchunk+chunk)][NH*DK]
__global__ __launch_bounds__(256) void ltae_softmax_bwd_kernel(const float* __restrict__ datt,
                                                               const float* __restrict__ att,
                                                               const float* __restrict__ k,
                                                               const float* __restrict__ Q,
                                                               const int* __restrict__ pad, float* __restrict__ dk,
                                                               float* __restrict__ dq_part, int B, int T, int NH,
                                                               int DK, int S, float inv_temp) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const bool act = s < S;
    float dq[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = 0.f;
    if (act) {
        float dot = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t o = (((size_t)h * B + b) * T + t) * S + s;
            dot = fmaf(att[o], datt[o], dot);
        }
        for (int t = 0; t < T; ++t) {
            const size_t o = (((size_t)h * B + b) * T + t) * S + s;
            float ds = att[o] * (datt[o] - dot);
            if (pad && pad[b * T + t]) ds = 0.f;      // masked_fill: no gradient to the replaced score
            ds *= inv_temp;
            const size_t kb = (((size_t)b * T + t) * NH * DK + h * DK) * S + s;
#pragma unroll
            for (int d = 0; d < 8; ++d)
                if (d < DK) {
                    dk[kb + (size_t)d * S] = ds * Q[h * DK + d];
                    dq[d] = fmaf(ds, k[kb + (size_t)d * S], dq[d]);
                }
        }
    }
    __shared__ float red[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float a = wave_sum(dq[d]);
        if (lane == 0) red[wv][d] = a;
    }
    __syncthreads();
    if (threadIdx.x < DK) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        dq_part[blk * NH * DK + h * DK + threadIdx.x] = v;
    }
}

// pad_mask[n] = 1 iff every element of frame n equals pad_value (uncrtaints.py:392-394).  One block per
// frame with a block-wide early exit: a real frame is rejected after its first chunk.
__global__ __launch_bounds__(256) void pad_mask_kernel(const float* __restrict__ x, long long frame_elems,
                                                       float pad_value, int* __restrict__ mask) {
    const float* src = x + (size_t)blockIdx.x * frame_elems;
    int found = 0;
    for (long long base = 0; base < frame_elems; base += 256 * 4) {
        int local = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long i = base + j * 256 + threadIdx.x;
            if (i < frame_elems && src[i] != pad_value) local = 1;
        }
        if (__syncthreads_or(local)) { found = 1; break; }
    }
    if (threadIdx.x == 0) mask[blockIdx.x] = found ? 0 : 1;
}

extern "C" int uncr_pad_mask(const float* x, int NF, long long frame_elems, float pad_value, int* mask,
                             hipStream_t stream) {
    if (NF <= 0 || frame_elems <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(pad_mask_kernel, dim3(NF), dim3(256), 0, stream, x, frame_elems, pad_value, mask);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_maxpool_fwd(const void* in, float* out, int* idx, int planes, int H, int W, int OH, int OW, int act,
                                hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return UNCR_ESHAPE;      // (H < OH pools UP: every window holds >= 1 element)
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((OH * OW + 255) / 256, planes), dim3(256), 0,
                                                 stream, (const T*)in, out, idx, H, W, OH, OW, H * W));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// Row-wise form for windows of several pixels (any-size images: 250 x 250 -> 32 x 32, overlapping 8-9 pixel windows): one block per
// (output row, plane) reads the window's rows COALESCED and keeps per column the maximum and the row of its first occurrence; a thread
// per output cell then combines its columns.  ATen's scan order (row-major, first maximum wins) is reproduced by comparing (value,
// row, column); a NaN wins over everything (the last one in (row, column) order).  The cell-per-thread kernel above read the same
// bytes as 64 scattered scalar loads per thread: 198 us for 384 MB at 12 x 128 x 250 x 250, this one streams.
__global__ __launch_bounds__(256) void maxpool_fwd_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ idx,
                                                               int H, int W, int OH, int OW, int pstride) {
    extern __shared__ float mp_s[];
    float* cmax = mp_s;                     // [W]
    int* crow = (int*)(mp_s + W);           // [W]
    const int oy = blockIdx.x, plane = blockIdx.y;
    const int ys = (oy * H) / OH, ye = ((oy + 1) * H + OH - 1) / OH;
    const float* p = in + (size_t)plane * pstride;
    for (int x = threadIdx.x; x < W; x += 256) {
        float best = -INFINITY;
        int br = ys;
        for (int y = ys; y < ye; y += 4) {          // four rows in flight (a load per iteration waited for its own latency: 8 round trips)
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[(size_t)min(y + u, ye - 1) * W + x];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (y + u < ye && (v[u] > best || v[u] != v[u])) { best = v[u]; br = y + u; }
        }
        cmax[x] = best;
        crow[x] = br;
    }
    __syncthreads();
    for (int ox = threadIdx.x; ox < OW; ox += 256) {
        const int xs = (ox * W) / OW, xe = ((ox + 1) * W + OW - 1) / OW;
        float best = cmax[xs];
        int br = crow[xs], bc = xs;
        for (int x = xs + 1; x < xe; ++x) {
            const float v = cmax[x];
            const int r = crow[x];
            const bool vn = v != v, bn = best != best;
            // a NaN beats a number; two NaNs: the later one in scan order; numbers: the larger, ties to the earlier (row, column)
            const bool take = vn ? (!bn || r > br || (r == br && x > bc)) : (!bn && (v > best || (v == best && r < br)));
            if (take) { best = v; br = r; bc = x; }
        }
        out[((size_t)plane * OH + oy) * OW + ox] = best;
        idx[((size_t)plane * OH + oy) * OW + ox] = br * W + bc;
    }
}

// the same on planes with a padded stride (csrc/anysize.hip: dense H*W pixels + a zero tail); idx stays the flat index inside the H x W image
extern "C" int uncr_maxpool_fwd_strided(const float* in, float* out, int* idx, int planes, int H, int W, int pstride, int OH, int OW,
                                        hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || pstride < H * W) return UNCR_ESHAPE;
    if (!in || !out || !idx) return UNCR_EINVAL;
    if (H >= 2 * OH && W >= 2 * OW && W <= 6144) {        // windows of several pixels: the streaming row-wise kernel
        hipLaunchKernelGGL(maxpool_fwd_rows_kernel, dim3(OH, planes), dim3(256), (size_t)W * 8, stream, in, out, idx, H, W, OH, OW, pstride);
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3((OH * OW + 255) / 256, planes), dim3(256), 0, stream, in, out, idx, H, W, OH, OW,
                       pstride);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_residual_pool_supported(int H, int W, int OH, int OW) {
    return (W == 256 && OW == 32 && H >= 8 && (H & 7) == 0 && OH * 8 == H) ? 1 : 0;
}
extern "C" int uncr_residual_pool_slots(int H) { return H / 8; }
extern "C" int uncr_residual_pool(const void* x, const void* h3, const float* cA, const float* cB, void* out,
                                  float* part, float* down, int* idx, int planes, int H, int W, int OH, int OW, int act,
                                  hipStream_t stream) {
    if (planes <= 0 || !uncr_residual_pool_supported(H, W, OH, OW)) return UNCR_ESHAPE;
    if (!x || !h3 || !cA || !cB || !out || !down || !idx || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(residual_pool_kernel<T>, dim3((H / 8 + 3) / 4, planes), dim3(256), 0, stream,
                                                 (const T*)x, (const T*)h3, cA, cB, (T*)out, (float2*)part, down, idx, H));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_maxpool_bwd(const float* dout, const int* idx, void* din, int planes, int H, int W, int OH,
                                int OW, int act, hipStream_t stream) {
    if (planes <= 0) return UNCR_ESHAPE;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    const int disjoint = (H % OH == 0) && (W % OW == 0);
    UNCR_DISPATCH_ACT(act, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((OH * OW + 255) / 256, planes), dim3(256), 0,
                                                 stream, dout, idx, (T*)din, H * W, OH * OW, disjoint));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_maxpool_bwd_strided(const float* dout, const int* idx, float* din, int planes, int H, int W, int pstride, int OH,
                                        int OW, hipStream_t stream) {
    if (planes <= 0 || pstride < H * W) return UNCR_ESHAPE;
    if (!dout || !idx || !din) return UNCR_EINVAL;
    const int disjoint = (H % OH == 0) && (W % OW == 0);
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3((OH * OW + 255) / 256, planes), dim3(256), 0, stream, dout, idx, din, pstride,
                       OH * OW, disjoint);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_gn_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y,
                                float* mean, float* rstd, int B, int T, int C, int G, int S, hipStream_t stream) {
    if (B <= 0 || T <= 0 || C % G) return UNCR_ESHAPE;
    hipLaunchKernelGGL(ltae_gn_fwd_kernel, dim3((S + 255) / 256, G, B), dim3(256), 0, stream, x, gamma, beta, eps, y,
                       mean, rstd, T, C, G, S);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_gn_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                const float* rstd, float* dx, float* gb_part, int B, int T, int C, int G, int S,
                                hipStream_t stream) {
    if (B <= 0 || T <= 0 || C % G || C / G > 32) return UNCR_ESHAPE;
    if (C / G > 16)
        hipLaunchKernelGGL(ltae_gn_bwd_kernel<32>, dim3((S + 255) / 256, G, B), dim3(256), 0, stream, dy, x, gamma, mean, rstd, dx,
                           gb_part, T, C, G, S);
    else
        hipLaunchKernelGGL(ltae_gn_bwd_kernel<16>, dim3((S + 255) / 256, G, B), dim3(256), 0, stream, dy, x, gamma, mean, rstd, dx,
                           gb_part, T, C, G, S);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// dst[r][i] = scale * src[i] for r < R   (head broadcast of the averaged attention / its gradient)
__global__ __launch_bounds__(256) void bcast_scale_kernel(const float* __restrict__ src, int R, long long n,
                                                          float scale, float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = scale * src[i];
    for (int r = 0; r < R; ++r) dst[(size_t)r * n + i] = v;
}

// 'mean' aggregation (uncrtaints.py:189-192,220-221): weight[h,b,t,s] = (1 - pad[b,t]) / #non-padded dates of b
__global__ __launch_bounds__(256) void mean_weights_kernel(const int* __restrict__ pad, int NH, int B, int T, int S,
                                                           float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)NH * B * T * S;
    if (i >= n) return;
    const int t = (int)((i / S) % T), b = (int)((i / ((long long)S * T)) % B);
    int cnt = 0;
    for (int tt = 0; tt < T; ++tt) cnt += (pad && pad[b * T + tt]) ? 0 : 1;
    const bool p = pad && pad[b * T + t];
    out[i] = (p || cnt == 0) ? 0.f : 1.f / (float)cnt;
}

extern "C" int uncr_bcast_scale(const float* src, int R, long long n, float scale, float* dst, hipStream_t stream) {
    if (R <= 0 || n <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(bcast_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, R, n, scale,
                       dst);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_mean_weights(const int* pad, int NH, int B, int T, int S, float* out, hipStream_t stream) {
    if (NH <= 0 || B <= 0 || T <= 0 || S <= 0) return UNCR_ESHAPE;
    const long long n = (long long)NH * B * T * S;
    hipLaunchKernelGGL(mean_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pad, NH, B, T, S,
                       out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_colsum(const float* part, int R, int K, float* out, hipStream_t stream) {
    if (R <= 0 || K <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(colsum_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, part, R, K, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_colsum_batched(const float* part, int batches, int R, int K, float* out, hipStream_t stream) {
    if (batches <= 0 || batches > 65535 || R <= 0 || K <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(colsum_kernel, dim3((K + 255) / 256, batches), dim3(256), 0, stream, part, R, K, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_posbias(const float* dates, const float* denom, int d, const float* bin, float* out,
                                 int NF, int D, int use_pe, hipStream_t stream) {
    if (NF <= 0 || D <= 0 || (use_pe && (!dates || !denom || d <= 0))) return UNCR_EINVAL;
    hipLaunchKernelGGL(ltae_posbias_kernel, dim3((NF * D + 255) / 256), dim3(256), 0, stream, dates, denom, d, bin,
                       out, NF, D, use_pe);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_softmax_fwd(const float* k, const float* Q, const int* pad, float* att, int B, int T,
                                     int NH, int DK, int S, hipStream_t stream) {
    if (DK > 8 || B <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(ltae_softmax_fwd_kernel, dim3((S + 255) / 256, NH, B), dim3(256), 0, stream, k, Q, pad, att, B,
                       T, NH, DK, S, 1.0f / sqrtf((float)DK));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_softmax_bwd(const float* datt, const float* att, const float* k, const float* Q,
                                     const int* pad, float* dk, float* dq_part, int B, int T, int NH, int DK, int S,
                                     hipStream_t stream) {
    if (DK > 8 || B <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(ltae_softmax_bwd_kernel, dim3((S + 255) / 256, NH, B), dim3(256), 0, stream, datt, att, k, Q,
                       pad, dk, dq_part, B, T, NH, DK, S, 1.0f / sqrtf((float)DK));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}


// ---- input assembly in front of the path (model/train_reconstruct.py:161-179 prepare_data_multi, fused with
//      data/dataLoader.py:38-61 process_MS / process_SAR when raw intensities are given) ----
// desc (device, int64): per (t, group) {src pointer [B][Cg][P], Cg, channel offset in x, kind}; kind 0 copy,
// 1 MS 'default', 2 MS 'resnet', 3 SAR 'default', 4 SAR 'resnet'.  x is [B][T][C][P].  grid = (P/chunk, B*C, T*ngroups)
__device__ __forceinline__ float prep_value(float v, int kind, int ch) {
    if (kind == 0) return v;
    float r;
    if (kind == 1 || kind == 2) {
        r = fminf(fmaxf(v, 0.f), 10000.f) * (kind == 1 ? 1e-4f : 5e-4f);
    } else if (kind == 3) {
        r = (fminf(fmaxf(v, -25.f), 0.f) + 25.f) * 0.04f;
    } else {
        const float lo = ch == 0 ? -25.f : -32.5f;
        r = 2.f * (fminf(fmaxf(v, lo), 0.f) - lo) / (0.f - lo);
    }
    // np.clip propagates NaN, np.nan_to_num maps it to 0 (infinities were clipped to the range bounds)
    return v != v ? 0.f : r;
}
__global__ __launch_bounds__(256) void assemble_input_kernel(const long long* __restrict__ desc, float* __restrict__ x,
                                                             int B, int T, int C, int P, int ngroups) {
    const int tg = blockIdx.z, t = tg / ngroups;
    const long long* d = desc + (size_t)tg * 4;
    const float* src = (const float*)d[0];
    const int Cg = (int)d[1], c0 = (int)d[2], kind = (int)d[3];
    const int b = blockIdx.y / C, cl = blockIdx.y % C;      // cl: channel inside the group (blocks beyond Cg exit)
    if (cl >= Cg || !src) return;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= P) return;
    const float* s = src + ((size_t)b * Cg + cl) * P + p;
    float* o = x + (((size_t)b * T + t) * C + c0 + cl) * P + p;
    if (p + 3 < P && (((size_t)s | (size_t)o) & 15) == 0) {
        float4 v = *(const float4*)s;
        v.x = prep_value(v.x, kind, cl); v.y = prep_value(v.y, kind, cl);
        v.z = prep_value(v.z, kind, cl); v.w = prep_value(v.w, kind, cl);
        *(float4*)o = v;
    } else {
        for (int q = 0; q < 4 && p + q < P; ++q) o[q] = prep_value(s[q], kind, cl);
    }
}
extern "C" int uncr_assemble_input(const long long* desc, float* x, int B, int T, int C, int P, int ngroups,
                                   hipStream_t stream) {
    if (!desc || !x || B <= 0 || T <= 0 || C <= 0 || P <= 0 || ngroups <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(assemble_input_kernel, dim3((P + 1023) / 1024, B * C, T * ngroups), dim3(256), 0, stream, desc,
                       x, B, T, C, P, ngroups);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
