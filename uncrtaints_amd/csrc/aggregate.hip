// Compact_Temporal_Aggregator, mode 'att_group' (uncrtaints.py:156-221): the only full-resolution piece
// of the L-TAE stage.  The low-res attention [NH,B,T,32,32] is bilinearly up-sampled ON THE FLY
// (align_corners=False, uncrtaints.py:197-200) -- the reference materialises a [NH*B,T,H,W] tensor --
// multiplied by the (train-only) dropout mask and the pad mask, and contracted over T against the
// encoder features:  g[b, c, p] = sum_t a[c / (C/NH), b, t, p] * e[b, t, c, p].
//
// Pure HBM streaming: fwd reads e once ((T) planes) and writes g (1 plane) per channel; every access is
// 16 B per lane, 1 KiB contiguous per wave.  The epilogue emits (sum g, sum g^2) partials for the first
// decoder BatchNorm.  Algorithmic bytes: (T+1)*C*P*4 forward, (2T+1)*C*P*4 backward.
#include "common.h"

#define AGG_PX 1024   // pixels per block
#ifndef AGG_ABL
#define AGG_ABL 0      // development ablations: 1 no dropout hash, 2 no attention gathers
#endif

struct AggArgs {
    const void* e;         // [B][T][C][P], storage type T (fp32 | bf16) like out / dg / de
    const float* att;      // [NH][B][T][AH][AW]
    const int* pad;        // [B][T] or null
    const float* dmask;    // explicit dropout mask [NH*B][T][P] (values 0 or 1/(1-p)) or null
    void* out;             // fwd: g [B][C][P]
    const void* dg;        // bwd: [B][C][P]
    void* de;              // bwd: [B][T][C][P]
    float* datt_up;        // bwd: [NH][B][T][P]
    float2* part;          // fwd: [B*C][NP] or null
    unsigned long long seed;
    const long long* seed_dev;   // optional device-resident step counter added to `seed` (HIP-graph replays)
    float p_drop;          // > 0 with dmask == null: hash dropout
    int shared_mask;       // 1: one dropout mask per (b,t,pixel) shared by all heads ('att_mean' mode)
    int B, T, C, NH, H, W, AH, AW;
    // backward pass 2 (BMODE 2): the pooled gradient and the arg-max of the 8x8 (OH x OW) max-pool that was taken from e, the h3 of the block
    // that produced e, and that block's statistics / magnitude slots
    const float* dpool = nullptr;   // [B*T*C][OH*OW] or null (no scatter)
    const int* pidx = nullptr;      // [B*T*C][OH*OW] flat in-plane arg-max
    const void* h3 = nullptr;       // [B][T][C][P] or null (no statistics)
    float2* bpart = nullptr;        // [B*T*C][P / 1024] (sum de, sum de*h3)
    float* amax = nullptr;          // [B*T*C][P / 1024] max |de| or null
    int OH = 0, OW = 0;
};

struct Bilin {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Bilin bilin_src(int dst, float scale, int in_size) {
    // area_pixel_compute_source_index, align_corners=False, clamped at 0 (bilinear)
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Bilin r;
    r.i0 = (int)src;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// a * b rounded on its own: the multiply carries no `contract` flag, so it cannot be fused into a following add
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

__device__ __forceinline__ float agg_keep(const AggArgs& g, int h, int b, int t, size_t p) {
    float m = 1.f;
    const size_t P = (size_t)g.H * g.W;
    if (g.shared_mask) h = 0;
    if (g.dmask) m = g.dmask[(((size_t)h * g.B + b) * g.T + t) * P + p];
    else if (g.p_drop > 0.f && !(AGG_ABL & 1)) {
        const unsigned long long sd = g.seed + (g.seed_dev ? (unsigned long long)g.seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull);
        const float u = hash_uniform(sd, (((size_t)h * g.B + b) * g.T + t) * P + p);
        m = u < g.p_drop ? 0.f : 1.f / (1.f - g.p_drop);
    }
    if (g.pad && g.pad[b * g.T + t]) m = 0.f;   // attn * (~pad_mask), uncrtaints.py:172
    return m;
}

// STAGE: the block's low-resolution attention rows (all its heads and dates) are copied to LDS once; the 16 taps per
// (head, date) of a thread's 4 pixels then come from LDS instead of 16 scattered global loads (-20 us of 127 at the
// bench shape).  Dynamic LDS = heads_per_block * T * agg_rows * AW floats; larger problems use the gather path.
// FOLD (backward, W == 256, H == 8 AH, W == 8 AW): the gradient of the up-sampled attention is reduced to the low-resolution grid
// inside the kernel instead of being written at full resolution (NH*T planes, 50 MB at the bench shape) for a separate adjoint pass.
// A wave holds one image row (64 lanes x 4 pixels); a lane's four pixels interpolate between the same two low-res columns, so its
// contributions fall on cells f-1, f, f+1 (f = lane / 2) and a cell collects from four neighbouring lanes by shuffles in a fixed
// order; the block's four rows interpolate between the same two low-res rows, whose 2 x 32 cells the waves combine through LDS, again
// in a fixed order.  Per (head, date) the block leaves 2 x AW partial cells in datt_up (re-used as [NH*B*T][tiles][2][AW]);
// datt_fold_reduce_kernel adds the (at most four) tiles that touch a low-res row.  Deterministic; no atomics.
// BMODE (backward): 0 = de and datt in one pass (reads dg, e; writes de): the general path.
// Two-pass backward (round 5), for the case where the last encoder block's statistics pass follows anyway:
//   1 = datt ONLY (reads dg, e; writes nothing at full resolution): a read-only stream;
//   2 = de = a * dg + scatter(pooled gradient at the arg-max) written ONCE, with the (sum de, sum de*h3) partials and the per-block
//       maxima the encoder block's backward needs, in the same pass (reads dg, h3; e is not touched).
// One-pass + uncr_pool_scatter_stats moved (2T+1) + 2T planes per channel; the two passes move (T+1) + (2T+1): de is never re-read.
// The statistics are those of uncr_pool_scatter_stats (per 1024-pixel chunk: (x+y)+(z+w), the fma chain, DPP wave sums, the four waves
// in order), so with fp32 storage the partials are bit-identical to the old pair of launches.
template <bool BWD, int CH, bool STAGE, typename T, bool FOLD = false, int BMODE = 0>
__global__ __launch_bounds__(256) void aggregate_kernel(AggArgs g, int nrows) {
    extern __shared__ float att_s[];
    const int b = blockIdx.y;
    const int P = g.H * g.W;
    const int p0 = blockIdx.x * AGG_PX + threadIdx.x * 4;
    const int y = p0 / g.W, x0 = p0 % g.W;
    const float sy = (float)g.AH / (float)g.H, sx = (float)g.AW / (float)g.W;
    const Bilin by = bilin_src(y, sy, g.AH);
    Bilin bx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bx[j] = bilin_src(x0 + j, sx, g.AW);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    __shared__ float wred[BMODE == 2 ? 1 : 4][BMODE == 2 ? 1 : 256][2];   // per-wave statistics partials (forward), C <= 256
    __shared__ float sred[2][BMODE == 2 ? 4 : 1][BMODE == 2 ? CH : 1][3];    // pass 2: per-wave (sum, sum*h3, max) of the CH planes of one (head, date)
    __shared__ float rowc[FOLD ? 2 : 1][FOLD ? 2 : 1][4][32];      // FOLD: [pair parity][low-res row][wave][cell]: the four rows' cells, weighted for the two low-res rows
    // heads are split over blockIdx.z (more blocks in flight: the kernel is pure latency/bandwidth bound)
    const int hpb = (g.NH + gridDim.z - 1) / gridDim.z;
    const int h_beg = blockIdx.z * hpb, h_end = min(g.NH, h_beg + hpb);
    int rlo = 0;
    if constexpr (STAGE) {
        const int pb = blockIdx.x * AGG_PX;
        rlo = bilin_src(pb / g.W, sy, g.AH).i0;                     // first low-res row any pixel of the block touches
        const int per_h = g.T * nrows * g.AW;
        for (int i = threadIdx.x; i < (h_end - h_beg) * per_h; i += 256) {
            const int hl = i / per_h, rem = i - hl * per_h;
            const int t = rem / (nrows * g.AW), rr = rem - t * nrows * g.AW;
            const int r = min(rlo + rr / g.AW, g.AH - 1), ax = rr % g.AW;
            att_s[i] = g.att[((((size_t)(h_beg + hl) * g.B + b) * g.T + t) * g.AH + r) * g.AW + ax];
        }
        __syncthreads();
    }
    for (int h = h_beg; h < h_end; ++h) {
        float4 acc[CH];                 // fwd: output accumulators; bwd: dg of the head's channels
#pragma unroll
        for (int jc = 0; jc < CH; ++jc) {
            if constexpr (BWD) acc[jc] = ld_nt4t((const T*)g.dg + ((size_t)b * g.C + h * CH + jc) * P + p0);
            else acc[jc] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int t = 0; t < g.T; ++t) {
            // up-sampled attention (x dropout x pad) for this thread's 4 pixels
            const float* ap = STAGE ? att_s + (((h - h_beg) * g.T + t) * nrows - rlo) * g.AW
                                    : g.att + (((size_t)h * g.B + b) * g.T + t) * g.AH * g.AW;
            float a[4], keep[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float top = bx[j].l0 * ap[by.i0 * g.AW + bx[j].i0] + bx[j].l1 * ap[by.i0 * g.AW + bx[j].i1];
                float bot = bx[j].l0 * ap[by.i1 * g.AW + bx[j].i0] + bx[j].l1 * ap[by.i1 * g.AW + bx[j].i1];
                if (AGG_ABL & 2) { top = bx[j].l0 * 0.3f; bot = bx[j].l1 * 0.2f + (float)t; }
                keep[j] = agg_keep(g, h, b, t, (size_t)p0 + j);
                a[j] = (by.l0 * top + by.l1 * bot) * keep[j];
            }
            if constexpr (BWD && BMODE == 2) {
                // statistics buffer of this (head, date): the parity of a counter that runs over the block's (head, date) pairs -- the date's
                // parity alone meets itself between the last date of a head and date 0 of the next when T is odd and a block owns
                // several heads (n_head > 64): waves 1-3 then rewrote the buffer wave 0 was still reducing
                const int sb_ = ((h - h_beg) * g.T + t) & 1;
                // pass 2: every load of this (head, date) -- the CH rows of h3 and the CH window indices -- is requested before the first
                // dependent instruction (groups of at most 8 channels: 32 operand registers)
                // Branch-free: a load under a per-lane branch costs an s_waitcnt vmcnt(0) at the join, and on gfx9 the stores share that counter --
                // the first version (the pooled gradient read only by the lanes that hold an arg-max) waited for every earlier store of the
                // block at each channel: 378 us for 940 MB, whatever the storage type.  The pooled value is read by every lane (the index
                // and value tables are 6 MB, cache resident) and selected.
                constexpr int GB = CH < 4 ? CH : 4;      // (eight rows in flight with bf16 storage: 169 VGPRs, 213 -> 265 us)
                const unsigned cell = (unsigned)(y / (g.H / (g.OH > 0 ? g.OH : 1))) * (unsigned)g.OW + (unsigned)(x0 / (g.W / (g.OW > 0 ? g.OW : 1)));
                const unsigned ncell = (unsigned)(g.OH * g.OW);
                const int* pix = g.dpool ? g.pidx : (const int*)g.att;         // dummy readable locations when there is no scatter
                const float* dpl = g.dpool ? g.dpool : g.att;
                const bool scat = g.dpool != nullptr;
#pragma unroll
                for (int j0 = 0; j0 < CH; j0 += GB) {
                    float4 hv[GB];
                    int kk[GB];
                    float dpv[GB];
#pragma unroll
                    for (int u = 0; u < GB; ++u) {
                        const unsigned plane = (unsigned)((b * g.T + t) * g.C + h * CH + j0 + u);
                        hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (g.h3) hv[u] = ld_nt4t((const T*)g.h3 + (size_t)plane * P + p0);     // (kernel-uniform)
                        const unsigned q = scat ? plane * ncell + cell : 0u;
                        kk[u] = pix[q] - p0;            // the window's arg-max relative to this thread's four pixels
                        dpv[u] = dpl[q];
                    }
#pragma unroll
                    for (int u = 0; u < GB; ++u) {
                        const int jc = j0 + u;
                        const size_t eo = (((size_t)b * g.T + t) * g.C + h * CH + jc) * P + p0;
                        const float4 dgv = acc[jc];
                        // (products rounded on their own, not contracted with the pooled gradient's add: de then equals the one-pass kernel's
                        // product + the scatter kernel's add bit for bit)
                        float4 v = make_float4(mul_rounded(a[0], dgv.x), mul_rounded(a[1], dgv.y), mul_rounded(a[2], dgv.z), mul_rounded(a[3], dgv.w));
                        const float dd = scat ? dpv[u] : 0.f;
                        const int k = scat ? kk[u] : -1;
                        v.x = k == 0 ? v.x + dd : v.x; v.y = k == 1 ? v.y + dd : v.y;
                        v.z = k == 2 ? v.z + dd : v.z; v.w = k == 3 ? v.w + dd : v.w;
                        v = rnd4<T>(v);
                        st_nt4t((T*)g.de + eo, v);
                        if (g.bpart) {
                            float s0 = (v.x + v.y) + (v.z + v.w);
                            float s1 = fmaf(v.x, hv[u].x, fmaf(v.y, hv[u].y, fmaf(v.z, hv[u].z, v.w * hv[u].w)));
                            s0 = wave_sum_dpp(s0);
                            s1 = wave_sum_dpp(s1);
                            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                            if (g.amax) m = wave_max_dpp(m);
                            if (lane == 63) { sred[sb_][wv][jc][0] = s0; sred[sb_][wv][jc][1] = s1; sred[sb_][wv][jc][2] = m; }
                        }
                    }
                }
                if (g.bpart) {            // kernel-uniform: the CH planes of this (head, date), the four waves in order.  Two buffers by the
                                          // pair counter's parity: one barrier per date (a buffer is rewritten two barriers after it was read)
                    __syncthreads();
                    if (threadIdx.x < CH) {
                        const int jc = threadIdx.x;
                        const size_t slot = (((size_t)b * g.T + t) * g.C + h * CH + jc) * gridDim.x + blockIdx.x;
                        float sa = 0.f, sb = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { sa += sred[sb_][i][jc][0]; sb += sred[sb_][i][jc][1]; }
                        g.bpart[slot] = make_float2(sa, sb);
                        if (g.amax) g.amax[slot] = fmaxf(fmaxf(sred[sb_][0][jc][2], sred[sb_][1][jc][2]),
                                                         fmaxf(sred[sb_][2][jc][2], sred[sb_][3][jc][2]));
                    }
                }
                continue;
            }
            float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jc = 0; jc < CH; ++jc) {
                const size_t eo = (((size_t)b * g.T + t) * g.C + h * CH + jc) * P + p0;
                const float4 ev = ld_nt4t((const T*)g.e + eo);
                if constexpr (!BWD) {
                    acc[jc].x = fmaf(a[0], ev.x, acc[jc].x);
                    acc[jc].y = fmaf(a[1], ev.y, acc[jc].y);
                    acc[jc].z = fmaf(a[2], ev.z, acc[jc].z);
                    acc[jc].w = fmaf(a[3], ev.w, acc[jc].w);
                } else {
                    const float4 dgv = acc[jc];
                    if constexpr (BMODE == 0) st_nt4t((T*)g.de + eo, make_float4(a[0] * dgv.x, a[1] * dgv.y, a[2] * dgv.z, a[3] * dgv.w));
                    d[0] = fmaf(dgv.x, ev.x, d[0]);
                    d[1] = fmaf(dgv.y, ev.y, d[1]);
                    d[2] = fmaf(dgv.z, ev.z, d[2]);
                    d[3] = fmaf(dgv.w, ev.w, d[3]);
                }
            }
            if constexpr (BWD && !FOLD) {
                // gradient w.r.t. the up-sampled attention: keep * sum_j dg * e
                *(float4*)(g.datt_up + (((size_t)h * g.B + b) * g.T + t) * P + p0) =
                    make_float4(d[0] * keep[0], d[1] * keep[1], d[2] * keep[2], d[3] * keep[3]);
            }
            if constexpr (BWD && FOLD) {
                const int f = lane >> 1;
                float vm = 0.f, v0 = 0.f, vp = 0.f;      // this lane's contributions to the cells f-1, f, f+1 of its row
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dv = d[j] * keep[j];
                    const float c0 = bx[j].l0 * dv, c1 = bx[j].l1 * dv;
                    vm += (bx[j].i0 == f - 1 ? c0 : 0.f) + (bx[j].i1 == f - 1 ? c1 : 0.f);
                    v0 += (bx[j].i0 == f ? c0 : 0.f) + (bx[j].i1 == f ? c1 : 0.f);
                    vp += (bx[j].i0 == f + 1 ? c0 : 0.f) + (bx[j].i1 == f + 1 ? c1 : 0.f);
                }
                // cell a (at lane 2a) = vp[2a-1] + v0[2a] + v0[2a+1] + vm[2a+2]
                const float from_left = __shfl(vp, (lane + 63) & 63, 64), from_r1 = __shfl(v0, (lane + 1) & 63, 64),
                            from_r2 = __shfl(vm, (lane + 2) & 63, 64);
                const float cell = (((lane > 0 ? from_left : 0.f) + v0) + from_r1) + (lane < 62 ? from_r2 : 0.f);
                // two buffers by the parity of the block's (head, date) counter: ONE barrier per pair (a buffer is rewritten two barriers
                // after it was read; round 5 had a second barrier in front of the writes: 96 instead of 48 per block at 16 heads x 3 dates)
                const int rp = ((h - h_beg) * g.T + t) & 1;
                if ((lane & 1) == 0) {
                    rowc[rp][0][wv][f] = by.l0 * cell;
                    rowc[rp][1][wv][f] = by.l1 * cell;
                }
                __syncthreads();
                if (threadIdx.x < 64) {
                    const int r = threadIdx.x >> 5, a = threadIdx.x & 31;
                    g.datt_up[((((size_t)h * g.B + b) * g.T + t) * gridDim.x + blockIdx.x) * 2 * g.AW + r * g.AW + a] =
                        ((rowc[rp][r][0][a] + rowc[rp][r][1][a]) + rowc[rp][r][2][a]) + rowc[rp][r][3][a];
                }
            }
        }
        if constexpr (!BWD) {
#pragma unroll
            for (int jc = 0; jc < CH; ++jc) {
                const int c = h * CH + jc;
                const float4 o = rnd4<T>(acc[jc]);      // statistics of the values as stored
                st_nt4t((T*)g.out + ((size_t)b * g.C + c) * P + p0, o);
                if (g.part) {
                    const float s0 = wave_sum_dpp(o.x + o.y + o.z + o.w);
                    const float s1 = wave_sum_dpp(o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w);
                    if (lane == 63) { wred[wv][c][0] = s0; wred[wv][c][1] = s1; }
                }
            }
        }
    }
    if constexpr (!BWD) {
        if (g.part) {
            __syncthreads();
            for (int c = h_beg * CH + threadIdx.x; c < h_end * CH; c += 256)
                g.part[((size_t)b * g.C + c) * gridDim.x + blockIdx.x] =
                    make_float2(wred[0][c][0] + wred[1][c][0] + wred[2][c][0] + wred[3][c][0],
                                wred[0][c][1] + wred[1][c][1] + wred[2][c][1] + wred[3][c][1]);
        }
    }
}

// adjoint of the bilinear up-sampling: datt[q, ay, ax] = sum_{y,x} wy(y,ay) wx(x,ax) dup[q, y, x]
// grid = (AH, planes q = NH*B*T), block = 256.  Separable: the rows that feed `ay` are read as coalesced float4
// columns, weighted by wy and summed over rows (4 row lanes -> LDS), then every ax folds its x range of the row sums.
#define BADJ_MAXW 1024
__global__ __launch_bounds__(256) void bilinear_adjoint_kernel(const float* __restrict__ dup,
                                                               float* __restrict__ datt, int H, int W, int AH,
                                                               int AW) {
    const int ay = blockIdx.x, q = blockIdx.y;
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const float sy = (float)AH / (float)H, sx = (float)AW / (float)W;
    // rows / columns whose source coordinate sy*(y+0.5)-0.5 lies in [a-1, a+1) can carry weight for cell a; the bounds are
    // conservative (the weight itself is evaluated exactly per row / column) and hold for non-integer ratios H/AH too
    const int ylo = max(0, (int)floorf(((float)ay - 0.5f) / sy - 0.5f) - 1);
    const int yhi = min(H, (int)ceilf(((float)ay + 1.5f) / sy - 0.5f) + 2);
    const float* src = dup + (size_t)q * H * W;
    __shared__ float colsum[4][BADJ_MAXW];
    const int W4 = W >> 2;
    for (int c4 = cl; c4 < W4; c4 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int yy = ylo + rl; yy < yhi; yy += 4) {
            const Bilin by = bilin_src(yy, sy, AH);
            const float wy = (by.i0 == ay ? by.l0 : 0.f) + (by.i1 == ay ? by.l1 : 0.f);
            if (wy == 0.f) continue;
            const float4 v = *(const float4*)(src + (size_t)yy * W + 4 * c4);
            acc.x = fmaf(wy, v.x, acc.x); acc.y = fmaf(wy, v.y, acc.y);
            acc.z = fmaf(wy, v.z, acc.z); acc.w = fmaf(wy, v.w, acc.w);
        }
        *(float4*)&colsum[rl][4 * c4] = acc;
    }
    __syncthreads();
    for (int ax = tid; ax < AW; ax += 256) {
        const int xlo = max(0, (int)floorf(((float)ax - 0.5f) / sx - 0.5f) - 1);
        const int xhi = min(W, (int)ceilf(((float)ax + 1.5f) / sx - 0.5f) + 2);
        float s = 0.f;
        for (int xx = xlo; xx < xhi; ++xx) {
            const Bilin bx = bilin_src(xx, sx, AW);
            const float wx = (bx.i0 == ax ? bx.l0 : 0.f) + (bx.i1 == ax ? bx.l1 : 0.f);
            if (wx != 0.f) s = fmaf(wx, (colsum[0][xx] + colsum[1][xx]) + (colsum[2][xx] + colsum[3][xx]), s);
        }
        datt[((size_t)q * AH + ay) * AW + ax] = s;
    }
}

// datt[q][r][a] = sum over the 1024-pixel tiles whose rows interpolate from low-res row r (tile order, fixed)
__global__ __launch_bounds__(256) void datt_fold_reduce_kernel(const float* __restrict__ dpart, float* __restrict__ datt, int H, int W,
                                                               int AH, int AW, int ntile, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = (int)(i % AW), r = (int)((i / AW) % AH);
    const long long q = i / ((long long)AW * AH);
    const float sy = (float)AH / (float)H;
    const int rows_per_tile = AGG_PX / W;
    // rows 8r-8 .. 8r+15 can interpolate from low-res row r: the tiles that hold them
    const int t0 = max(0, (8 * r - 8) / rows_per_tile), t1 = min(ntile - 1, (8 * r + 15) / rows_per_tile);
    float s = 0.f;
    for (int k = t0; k <= t1; ++k) {
        const Bilin by = bilin_src(k * rows_per_tile, sy, AH);
        const float* src = dpart + ((size_t)q * ntile + k) * 2 * AW + a;
        if (by.i0 == r) s += src[0];
        if (by.i1 == r) s += src[AW];
    }
    datt[i] = s;
}

extern "C" int uncr_agg_slots(int P) { return P / AGG_PX; }

static int agg_check(int B, int T, int C, int NH, int H, int W, int AH, int AW) {
    if (B <= 0 || T <= 0 || C % NH || C > 256) return UNCR_ESHAPE;
    { const int ch = C / NH; if (ch != 2 && ch != 4 && ch != 6 && ch != 8 && ch != 16 && ch != 32) return UNCR_ESHAPE; }   // channels per head
    if ((W & 3) || W > BADJ_MAXW || ((H * W) % AGG_PX)) return UNCR_ESHAPE;
    if (H < AH || W < AW) return UNCR_ESHAPE;     // avg-pool branch (uncrtaints.py:204) not built; H == AH is the
                                                  // identity up-sampling (LTAE2d's attention-weighted values)
    return UNCR_OK;
}

// low-res rows a 1024-pixel block can touch: its (at most 1024/W + 1) image rows map to a span of that many * AH/H
// source rows, plus the two interpolation partners
static int agg_rows(int H, int W, int AH) {
    const int img_rows = (AGG_PX + W - 1) / W + 1;      // a block may straddle one more image row
    int r = (int)((double)(img_rows - 1) * AH / H) + 3;
    return r > AH ? AH : r;
}

// the in-kernel adjoint of the up-sampling (aggregate_kernel FOLD) applies to the 8x up-sampling of 256-pixel rows
static bool agg_fold(const AggArgs& g) { return g.W == 256 && g.W == 8 * g.AW && g.H == 8 * g.AH && g.AW == 32; }

template <bool BWD, typename T, int BMODE = 0>
static void agg_launch(const AggArgs& g, hipStream_t stream) {
    const int zs = g.NH <= 64 ? g.NH : (g.NH % 4 == 0 ? 4 : 1);      // one head per block: measured 4 % faster than four (more blocks in flight)
    const dim3 grid(g.H * g.W / AGG_PX, g.B, zs);
    const int nrows = agg_rows(g.H, g.W, g.AH);
    const size_t lds = (size_t)((g.NH + zs - 1) / zs) * g.T * nrows * g.AW * sizeof(float);
    const bool stage = lds <= 48 * 1024;
    const bool fold = BWD && BMODE != 2 && agg_fold(g);      // (pass 2 takes no attention gradient)
#define AGG_GO(CHV)                                                                                              \
    do {                                                                                                         \
        if (fold && stage) hipLaunchKernelGGL((aggregate_kernel<BWD, CHV, true, T, BWD && BMODE != 2, BMODE>), grid, dim3(256), lds, stream, g, nrows); \
        else if (fold) hipLaunchKernelGGL((aggregate_kernel<BWD, CHV, false, T, BWD && BMODE != 2, BMODE>), grid, dim3(256), 0, stream, g, nrows);      \
        else if (stage) hipLaunchKernelGGL((aggregate_kernel<BWD, CHV, true, T, false, BMODE>), grid, dim3(256), lds, stream, g, nrows); \
        else hipLaunchKernelGGL((aggregate_kernel<BWD, CHV, false, T, false, BMODE>), grid, dim3(256), 0, stream, g, nrows);        \
    } while (0)
    switch (g.C / g.NH) {
        case 2: AGG_GO(2); break;
        case 4: AGG_GO(4); break;
        case 6: AGG_GO(6); break;
        case 8: AGG_GO(8); break;
        case 16: AGG_GO(16); break;
        default: AGG_GO(32); break;
    }
#undef AGG_GO
}

extern "C" int uncr_aggregate_fwd(const void* e, const float* att, const int* pad, const float* dmask,
                                  unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask,
                                  void* out, float* part, int B, int T, int C, int NH, int H, int W, int AH, int AW,
                                  int act, hipStream_t stream) {
    const int rc = agg_check(B, T, C, NH, H, W, AH, AW);
    if (rc) return rc;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    AggArgs g{e, att, pad, dmask, out, nullptr, nullptr, nullptr, (float2*)part, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    UNCR_DISPATCH_ACT(act, T, (agg_launch<false, T>(g, stream)));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_aggregate_bwd(const void* dg, const void* e, const float* att, const int* pad,
                                  const float* dmask, unsigned long long seed, const long long* seed_dev,
                                  float p_drop, int shared_mask, void* de, float* datt_up, float* datt, int B, int T,
                                  int C, int NH, int H, int W, int AH, int AW, int act, hipStream_t stream) {
    const int rc = agg_check(B, T, C, NH, H, W, AH, AW);
    if (rc) return rc;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    AggArgs g{e, att, pad, dmask, nullptr, dg, de, datt_up, nullptr, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    UNCR_DISPATCH_ACT(act, T, (agg_launch<true, T>(g, stream)));
    UNCR_LAUNCH_CHECK();
    if (agg_fold(g)) {       // the kernel left per-tile partial cells in datt_up
        const long long n = (long long)NH * B * T * AH * AW;
        hipLaunchKernelGGL(datt_fold_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, datt_up, datt, H, W, AH,
                           AW, H * W / AGG_PX, n);
    } else {
        hipLaunchKernelGGL(bilinear_adjoint_kernel, dim3(AH, NH * B * T), dim3(256), 0, stream, datt_up, datt, H, W, AH, AW);
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// Two-pass backward (see aggregate_kernel): pass 1 = the attention's gradient alone ...
extern "C" int uncr_aggregate_bwd_datt(const void* dg, const void* e, const float* att, const int* pad, const float* dmask,
                                       unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask,
                                       float* datt_up, float* datt, int B, int T, int C, int NH, int H, int W, int AH, int AW, int act,
                                       hipStream_t stream) {
    const int rc = agg_check(B, T, C, NH, H, W, AH, AW);
    if (rc) return rc;
    if (!dg || !e || !att || !datt_up || !datt || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    AggArgs g{e, att, pad, dmask, nullptr, dg, nullptr, datt_up, nullptr, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    UNCR_DISPATCH_ACT(act, T, (agg_launch<true, T, 1>(g, stream)));
    UNCR_LAUNCH_CHECK();
    if (agg_fold(g)) {
        const long long n = (long long)NH * B * T * AH * AW;
        hipLaunchKernelGGL(datt_fold_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, datt_up, datt, H, W, AH,
                           AW, H * W / AGG_PX, n);
    } else {
        hipLaunchKernelGGL(bilinear_adjoint_kernel, dim3(AH, NH * B * T), dim3(256), 0, stream, datt_up, datt, H, W, AH, AW);
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// ... pass 2 = de = a * dg (+ the pooled gradient at the arg-max of the OH x OW max-pool of e) with the producing block's statistics.
// Windows as in uncr_pool_scatter_stats: disjoint, width a multiple of 4 (H % OH == 0, W % OW == 0, (W / OW) % 4 == 0).
extern "C" int uncr_aggregate_bwd_de_supported(int H, int W, int OH, int OW) {
    return (OH > 0 && OW > 0 && H % OH == 0 && W % OW == 0 && ((W / OW) & 3) == 0 && ((H * W) % AGG_PX) == 0) ? 1 : 0;
}
extern "C" int uncr_aggregate_bwd_de(const void* dg, const float* att, const int* pad, const float* dmask, unsigned long long seed,
                                     const long long* seed_dev, float p_drop, int shared_mask, void* de, const float* dpool,
                                     const int* pidx, const void* h3, float* part, float* amax, int B, int T, int C, int NH, int H, int W,
                                     int AH, int AW, int OH, int OW, int act, hipStream_t stream) {
    const int rc = agg_check(B, T, C, NH, H, W, AH, AW);
    if (rc) return rc;
    if (!dg || !att || !de || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    if ((dpool != nullptr) != (pidx != nullptr) || (part != nullptr) != (h3 != nullptr) || (amax && !part)) return UNCR_EINVAL;
    if (dpool && !uncr_aggregate_bwd_de_supported(H, W, OH, OW)) return UNCR_ESHAPE;
    if (dpool && (unsigned long long)B * T * C * OH * OW >= (1ull << 31)) return UNCR_ESHAPE;      // 32-bit cell offsets
    AggArgs g{nullptr, att, pad, dmask, nullptr, dg, de, nullptr, nullptr, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    g.dpool = dpool; g.pidx = pidx; g.h3 = h3; g.bpart = (float2*)part; g.amax = amax; g.OH = OH; g.OW = OW;
    UNCR_DISPATCH_ACT(act, T, (agg_launch<true, T, 2>(g, stream)));
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- any H x W (csrc/anysize.hip): scalar variants on dense planes with a padded stride Pc -- one thread per pixel, valid pixels only,
// no float4 rows, so any width and any up-sampling ratio; fp32 storage.  Same arithmetic as aggregate_kernel (bilin_src, agg_keep).
#define AGGA_NB 8
__global__ __launch_bounds__(256) void aggregate_any_fwd_kernel(AggArgs g, int Pc) {
    const int plane = blockIdx.y, b = plane / g.C, c = plane - b * g.C, h = c / (g.C / g.NH);
    const int P = g.H * g.W;
    const float sy = (float)g.AH / (float)g.H, sx = (float)g.AW / (float)g.W;
    float s0 = 0.f, s1 = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += AGGA_NB * 256) {
        const int y = p / g.W, x = p - y * g.W;
        const Bilin by = bilin_src(y, sy, g.AH), bx = bilin_src(x, sx, g.AW);
        float acc = 0.f;
        for (int t = 0; t < g.T; ++t) {
            const float* ap = g.att + (((size_t)h * g.B + b) * g.T + t) * g.AH * g.AW;
            const float top = bx.l0 * ap[by.i0 * g.AW + bx.i0] + bx.l1 * ap[by.i0 * g.AW + bx.i1];
            const float bot = bx.l0 * ap[by.i1 * g.AW + bx.i0] + bx.l1 * ap[by.i1 * g.AW + bx.i1];
            const float a = (by.l0 * top + by.l1 * bot) * agg_keep(g, h, b, t, (size_t)p);
            acc = fmaf(a, ((const float*)g.e)[(((size_t)b * g.T + t) * g.C + c) * Pc + p], acc);
        }
        ((float*)g.out)[(size_t)plane * Pc + p] = acc;
        s0 += acc;
        s1 = fmaf(acc, acc, s1);
    }
    if (g.part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) g.part[(size_t)plane * AGGA_NB + blockIdx.x] = make_float2(s0, s1);
    }
}
// de[b,t,c,p] = a * dg[b,c,p]
__global__ __launch_bounds__(256) void aggregate_any_de_kernel(AggArgs g, int Pc) {
    const int plane = blockIdx.y, b = plane / g.C, c = plane - b * g.C, h = c / (g.C / g.NH);
    const int P = g.H * g.W;
    const float sy = (float)g.AH / (float)g.H, sx = (float)g.AW / (float)g.W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += AGGA_NB * 256) {
        const int y = p / g.W, x = p - y * g.W;
        const Bilin by = bilin_src(y, sy, g.AH), bx = bilin_src(x, sx, g.AW);
        const float dgv = ((const float*)g.dg)[(size_t)plane * Pc + p];
        for (int t = 0; t < g.T; ++t) {
            const float* ap = g.att + (((size_t)h * g.B + b) * g.T + t) * g.AH * g.AW;
            const float top = bx.l0 * ap[by.i0 * g.AW + bx.i0] + bx.l1 * ap[by.i0 * g.AW + bx.i1];
            const float bot = bx.l0 * ap[by.i1 * g.AW + bx.i0] + bx.l1 * ap[by.i1 * g.AW + bx.i1];
            const float a = (by.l0 * top + by.l1 * bot) * agg_keep(g, h, b, t, (size_t)p);
            ((float*)g.de)[(((size_t)b * g.T + t) * g.C + c) * Pc + p] = a * dgv;
        }
    }
}
// gradient w.r.t. the up-sampled attention, dense [NH*B*T][P]: keep * sum_{c in head} dg * e
__global__ __launch_bounds__(256) void aggregate_any_dup_kernel(AggArgs g, int Pc) {
    const int q = blockIdx.y, t = q % g.T, b = (q / g.T) % g.B, h = q / (g.T * g.B);
    const int P = g.H * g.W, CH = g.C / g.NH;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += AGGA_NB * 256) {
        float d = 0.f;
        for (int jc = 0; jc < CH; ++jc) {
            const int c = h * CH + jc;
            d = fmaf(((const float*)g.dg)[((size_t)b * g.C + c) * Pc + p], ((const float*)g.e)[(((size_t)b * g.T + t) * g.C + c) * Pc + p], d);
        }
        g.datt_up[(size_t)q * Pc + p] = d * agg_keep(g, h, b, t, (size_t)p);
    }
}
// adjoint of the bilinear up-sampling, any ratio: one thread per low-resolution cell gathers its footprint
__global__ __launch_bounds__(256) void bilinear_adjoint_any_kernel(const float* __restrict__ dup, float* __restrict__ datt, int H, int W,
                                                                   int AH, int AW, size_t stride /* of a source plane, >= H*W */) {
    const int q = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= AH * AW) return;
    const int ay = o / AW, ax = o - ay * AW;
    const float sy = (float)AH / (float)H, sx = (float)AW / (float)W;
    const int ylo = max(0, (int)floorf(((float)ay - 0.5f) / sy - 0.5f) - 1), yhi = min(H, (int)ceilf(((float)ay + 1.5f) / sy - 0.5f) + 2);
    const int xlo = max(0, (int)floorf(((float)ax - 0.5f) / sx - 0.5f) - 1), xhi = min(W, (int)ceilf(((float)ax + 1.5f) / sx - 0.5f) + 2);
    const float* src = dup + (size_t)q * stride;
    float s = 0.f;
    for (int yy = ylo; yy < yhi; ++yy) {
        const Bilin by = bilin_src(yy, sy, AH);
        const float wy = (by.i0 == ay ? by.l0 : 0.f) + (by.i1 == ay ? by.l1 : 0.f);
        if (wy == 0.f) continue;
        float r = 0.f;
        for (int xx = xlo; xx < xhi; ++xx) {
            const Bilin bx = bilin_src(xx, sx, AW);
            const float wx = (bx.i0 == ax ? bx.l0 : 0.f) + (bx.i1 == ax ? bx.l1 : 0.f);
            if (wx != 0.f) r = fmaf(wx, src[(size_t)yy * W + xx], r);
        }
        s = fmaf(wy, r, s);
    }
    datt[(size_t)q * AH * AW + o] = s;
}
// ---- float4 form of the same (round 6): a plane is dense -- H*W contiguous pixels, 16-byte aligned at every multiple of four whatever
// the width -- so a thread takes four consecutive FLAT pixels (they may sit on two image rows: the bilinear source is evaluated per
// pixel) of all CH channels of one head, like aggregate_kernel; the attention taps are gathered (L1 / L2 resident: 4 KB per (head,
// date)).  Tail pixels of the padded stride read zeros and write zeros.  The scalar kernels above recomputed the up-sampled attention
// once per CHANNEL and moved one float per lane: 256 / 474 us at 4 x 3 x 128 x 250 x 250 against 109 / 292 us of the tuned kernels at
// 256 x 256.  Backward: de = a*dg and the dense gradient of the up-sampled attention (stride Pc) in one pass.
static bool agg_anyv_ok(int C, int NH, int Pc) {
    if (NH <= 0 || C % NH || Pc % AGG_PX) return false;
    const int ch = C / NH;
    return ch == 2 || ch == 4 || ch == 6 || ch == 8 || ch == 16 || ch == 32;
}
template <bool BWD, int CH>
__global__ __launch_bounds__(256) void aggregate_anyv_kernel(AggArgs g, int Pc) {
    const int b = blockIdx.y, h = blockIdx.z;
    const int P = g.H * g.W;
    const int p0 = blockIdx.x * AGG_PX + threadIdx.x * 4;
    const float sy = (float)g.AH / (float)g.H, sx = (float)g.AW / (float)g.W;
    Bilin by[4], bx[4];
    bool in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = min(p0 + j, P - 1);            // (tail pixels: any valid source; their operands are zero)
        const int y = p / g.W;
        in[j] = p0 + j < P;
        by[j] = bilin_src(y, sy, g.AH);
        bx[j] = bilin_src(p - y * g.W, sx, g.AW);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ float wred[4][32][2];
    float4 acc[CH];
#pragma unroll
    for (int jc = 0; jc < CH; ++jc) {
        if constexpr (BWD) acc[jc] = *(const float4*)((const float*)g.dg + ((size_t)b * g.C + h * CH + jc) * Pc + p0);
        else acc[jc] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = 0; t < g.T; ++t) {
        const float* ap = g.att + (((size_t)h * g.B + b) * g.T + t) * g.AH * g.AW;
        float a[4], keep[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float top = bx[j].l0 * ap[by[j].i0 * g.AW + bx[j].i0] + bx[j].l1 * ap[by[j].i0 * g.AW + bx[j].i1];
            const float bot = bx[j].l0 * ap[by[j].i1 * g.AW + bx[j].i0] + bx[j].l1 * ap[by[j].i1 * g.AW + bx[j].i1];
            keep[j] = in[j] ? agg_keep(g, h, b, t, (size_t)p0 + j) : 0.f;
            a[j] = (by[j].l0 * top + by[j].l1 * bot) * keep[j];
        }
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jc = 0; jc < CH; ++jc) {
            const size_t eo = (((size_t)b * g.T + t) * g.C + h * CH + jc) * Pc + p0;
            const float4 ev = *(const float4*)((const float*)g.e + eo);
            if constexpr (!BWD) {
                acc[jc].x = fmaf(a[0], ev.x, acc[jc].x); acc[jc].y = fmaf(a[1], ev.y, acc[jc].y);
                acc[jc].z = fmaf(a[2], ev.z, acc[jc].z); acc[jc].w = fmaf(a[3], ev.w, acc[jc].w);
            } else {
                const float4 dgv = acc[jc];
                *(float4*)((float*)g.de + eo) = make_float4(a[0] * dgv.x, a[1] * dgv.y, a[2] * dgv.z, a[3] * dgv.w);
                d[0] = fmaf(dgv.x, ev.x, d[0]); d[1] = fmaf(dgv.y, ev.y, d[1]);
                d[2] = fmaf(dgv.z, ev.z, d[2]); d[3] = fmaf(dgv.w, ev.w, d[3]);
            }
        }
        if constexpr (BWD)
            *(float4*)(g.datt_up + (((size_t)h * g.B + b) * g.T + t) * Pc + p0) =
                make_float4(d[0] * keep[0], d[1] * keep[1], d[2] * keep[2], d[3] * keep[3]);
    }
    if constexpr (!BWD) {
#pragma unroll
        for (int jc = 0; jc < CH; ++jc) {
            const float4 o = acc[jc];
            *(float4*)((float*)g.out + ((size_t)b * g.C + h * CH + jc) * Pc + p0) = o;
            if (g.part) {
                const float s0 = wave_sum_dpp(o.x + o.y + o.z + o.w);
                const float s1 = wave_sum_dpp(o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w);
                if (lane == 63) { wred[wv][jc][0] = s0; wred[wv][jc][1] = s1; }
            }
        }
        if (g.part) {
            __syncthreads();
            if (threadIdx.x < CH) {
                const int jc = threadIdx.x;
                g.part[((size_t)b * g.C + h * CH + jc) * gridDim.x + blockIdx.x] =
                    make_float2((wred[0][jc][0] + wred[1][jc][0]) + (wred[2][jc][0] + wred[3][jc][0]),
                                (wred[0][jc][1] + wred[1][jc][1]) + (wred[2][jc][1] + wred[3][jc][1]));
            }
        }
    }
}
template <bool BWD>
static void agg_anyv_launch(const AggArgs& g, int Pc, hipStream_t stream) {
    const dim3 grid(Pc / AGG_PX, g.B, g.NH);
    switch (g.C / g.NH) {
        case 2: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 2>), grid, dim3(256), 0, stream, g, Pc); break;
        case 4: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 4>), grid, dim3(256), 0, stream, g, Pc); break;
        case 6: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 6>), grid, dim3(256), 0, stream, g, Pc); break;
        case 8: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 8>), grid, dim3(256), 0, stream, g, Pc); break;
        case 16: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 16>), grid, dim3(256), 0, stream, g, Pc); break;
        default: hipLaunchKernelGGL((aggregate_anyv_kernel<BWD, 32>), grid, dim3(256), 0, stream, g, Pc); break;
    }
}

static int agg_any_check(int B, int T, int C, int NH, int H, int W, int AH, int AW, int Pc) {
    if (B <= 0 || T <= 0 || NH <= 0 || C % NH || H < AH || W < AW || Pc < H * W) return UNCR_ESHAPE;
    return UNCR_OK;
}
// statistics slots per plane of uncr_aggregate_any_fwd: one per 1024-pixel block on the float4 kernels, AGGA_NB on the scalar ones
extern "C" int uncr_agg_any_slots(int Pc, int C, int NH) { return agg_anyv_ok(C, NH, Pc) ? Pc / AGG_PX : AGGA_NB; }
extern "C" int uncr_aggregate_any_fwd(const float* e, const float* att, const int* pad, const float* dmask, unsigned long long seed,
                                      const long long* seed_dev, float p_drop, int shared_mask, float* out, float* part, int B, int T,
                                      int C, int NH, int H, int W, int Pc, int AH, int AW, hipStream_t stream) {
    const int rc = agg_any_check(B, T, C, NH, H, W, AH, AW, Pc);
    if (rc) return rc;
    if (!e || !att || !out) return UNCR_EINVAL;
    AggArgs g{e, att, pad, dmask, out, nullptr, nullptr, nullptr, (float2*)part, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    if (agg_anyv_ok(C, NH, Pc)) agg_anyv_launch<false>(g, Pc, stream);
    else hipLaunchKernelGGL(aggregate_any_fwd_kernel, dim3(AGGA_NB, B * C), dim3(256), 0, stream, g, Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_aggregate_any_bwd(const float* dg, const float* e, const float* att, const int* pad, const float* dmask,
                                      unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask, float* de,
                                      float* datt_up /* [NH*B*T][H*W] scratch */, float* datt, int B, int T, int C, int NH, int H, int W,
                                      int Pc, int AH, int AW, hipStream_t stream) {
    const int rc = agg_any_check(B, T, C, NH, H, W, AH, AW, Pc);
    if (rc) return rc;
    if (!dg || !e || !att || !de || !datt_up || !datt) return UNCR_EINVAL;
    AggArgs g{e, att, pad, dmask, nullptr, dg, de, datt_up, nullptr, seed, seed_dev, p_drop, shared_mask, B, T, C, NH, H, W, AH, AW};
    if (agg_anyv_ok(C, NH, Pc)) {
        agg_anyv_launch<true>(g, Pc, stream);
    } else {
        hipLaunchKernelGGL(aggregate_any_de_kernel, dim3(AGGA_NB, B * C), dim3(256), 0, stream, g, Pc);
        UNCR_LAUNCH_CHECK();
        hipLaunchKernelGGL(aggregate_any_dup_kernel, dim3(AGGA_NB, NH * B * T), dim3(256), 0, stream, g, Pc);
    }
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(bilinear_adjoint_any_kernel, dim3((AH * AW + 255) / 256, NH * B * T), dim3(256), 0, stream, datt_up, datt, H, W,
                       AH, AW, (size_t)Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- the aggregator's AvgPool branch (uncrtaints.py:197-204): feature maps NOT larger than the attention map.  The attention is
// average-pooled with kernel = stride = k = AW / H (nn.AvgPool2d(kernel_size=w // x.shape[-2])) down to the feature map's size and
// no dropout is applied.  Planes here are smaller than the 1024-pixel tiles of the streaming kernels above (at most 32 x 32), so
// these are plain one-thread-per-element kernels: the whole problem is a few hundred KB.
//   fwd : out[b,c,p] = sum_t pool(att)[c/CH,b,t,p] * [not pad] * e[b,t,c,p]
//   bwd : de[b,t,c,p] = pool(att) * [not pad] * dg[b,c,p];   datt[h,b,t,ay,ax] = [not pad] / k^2 * sum_{c in h} dg[b,c,p] e[b,t,c,p]
//         with p = (ay / k, ax / k) for ay < k*H, ax < k*W, else 0 (cells the pooling never reads)
__device__ __forceinline__ float agg_pool_att(const float* __restrict__ att, int AW, int k, int y, int x) {
    float s = 0.f;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) s += att[(y * k + i) * AW + x * k + j];
    return s / (float)(k * k);
}
__global__ __launch_bounds__(256) void aggregate_pool_fwd_kernel(const float* __restrict__ e, const float* __restrict__ att,
                                                                 const int* __restrict__ pad, float* __restrict__ out, int B,
                                                                 int T, int C, int NH, int H, int W, int AH, int AW, int k) {
    const int P = H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C * P) return;
    const int p = (int)(i % P), c = (int)((i / P) % C), b = (int)(i / ((long long)P * C));
    const int h = c / (C / NH), y = p / W, x = p % W;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        if (pad && pad[b * T + t]) continue;
        const float a = agg_pool_att(att + (((size_t)h * B + b) * T + t) * AH * AW, AW, k, y, x);
        acc = fmaf(a, e[(((size_t)b * T + t) * C + c) * P + p], acc);
    }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void aggregate_pool_bwd_e_kernel(const float* __restrict__ dg, const float* __restrict__ att,
                                                                   const int* __restrict__ pad, float* __restrict__ de, int B,
                                                                   int T, int C, int NH, int H, int W, int AH, int AW, int k) {
    const int P = H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * T * C * P) return;
    const int p = (int)(i % P), c = (int)((i / P) % C), t = (int)((i / ((long long)P * C)) % T), b = (int)(i / ((long long)P * C * T));
    const int h = c / (C / NH);
    float a = 0.f;
    if (!(pad && pad[b * T + t])) a = agg_pool_att(att + (((size_t)h * B + b) * T + t) * AH * AW, AW, k, p / W, p % W);
    de[i] = a * dg[((size_t)b * C + c) * P + p];
}
__global__ __launch_bounds__(256) void aggregate_pool_bwd_att_kernel(const float* __restrict__ dg, const float* __restrict__ e,
                                                                     const int* __restrict__ pad, float* __restrict__ datt,
                                                                     int B, int T, int C, int NH, int H, int W, int AH, int AW,
                                                                     int k) {
    const int P = H * W, S = AH * AW, CH = C / NH;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)NH * B * T * S) return;
    const int a = (int)(i % S), t = (int)((i / S) % T), b = (int)((i / ((long long)S * T)) % B), h = (int)(i / ((long long)S * T * B));
    const int ay = a / AW, ax = a % AW;
    float s = 0.f;
    if (ay < k * H && ax < k * W && !(pad && pad[b * T + t])) {
        const int p = (ay / k) * W + ax / k;
        for (int j = 0; j < CH; ++j) {
            const int c = h * CH + j;
            s = fmaf(dg[((size_t)b * C + c) * P + p], e[(((size_t)b * T + t) * C + c) * P + p], s);
        }
        s /= (float)(k * k);
    }
    datt[i] = s;
}
static int agg_pool_check(int B, int T, int C, int NH, int H, int W, int AH, int AW, int k) {
    if (B <= 0 || T <= 0 || NH <= 0 || C % NH || H <= 0 || W <= 0 || k <= 0) return UNCR_ESHAPE;
    if (AH / k != H || AW / k != W) return UNCR_ESHAPE;       // the pooled attention must have the feature map's size
    return UNCR_OK;
}
extern "C" int uncr_aggregate_pool_fwd(const float* e, const float* att, const int* pad, float* out, int B, int T, int C,
                                       int NH, int H, int W, int AH, int AW, int k, hipStream_t stream) {
    const int rc = agg_pool_check(B, T, C, NH, H, W, AH, AW, k);
    if (rc) return rc;
    if (!e || !att || !out) return UNCR_EINVAL;
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(aggregate_pool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, e, att, pad, out, B, T, C,
                       NH, H, W, AH, AW, k);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_aggregate_pool_bwd(const float* dg, const float* e, const float* att, const int* pad, float* de,
                                       float* datt, int B, int T, int C, int NH, int H, int W, int AH, int AW, int k,
                                       hipStream_t stream) {
    const int rc = agg_pool_check(B, T, C, NH, H, W, AH, AW, k);
    if (rc) return rc;
    if (!dg || !e || !att || !de || !datt) return UNCR_EINVAL;
    const long long n = (long long)B * T * C * H * W, na = (long long)NH * B * T * AH * AW;
    hipLaunchKernelGGL(aggregate_pool_bwd_e_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dg, att, pad, de, B, T,
                       C, NH, H, W, AH, AW, k);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(aggregate_pool_bwd_att_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, dg, e, pad, datt, B,
                       T, C, NH, H, W, AH, AW, k);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- use_v (uncrtaints.py:414-417): out = a + bilinear_up(z) with (sum, sum^2) partials for the next PreNorm.
// include_v(cat(g, up(v))) = Wa*g + up(Wv*v + b): the 1x1 convolution commutes with the up-sampling, so the value
// branch is convolved at 32x32 and only this add touches full resolution.  grid = (P/1024, B*C), block 256 x float4.
__global__ __launch_bounds__(256) void add_upsampled_kernel(const float* __restrict__ a, const float* __restrict__ z,
                                                            float* __restrict__ out, float2* __restrict__ part, int H,
                                                            int W, int AH, int AW) {
    const int plane = blockIdx.y;
    const int p = blockIdx.x * AGG_PX + threadIdx.x * 4;
    const float sy = (float)AH / (float)H, sx = (float)AW / (float)W;
    const int y = p / W, x0 = p % W;              // W % 4 == 0: the four pixels share a row
    const Bilin by = bilin_src(y, sy, AH);
    const float* zp = z + (size_t)plane * AH * AW;
    const float4 va = *(const float4*)(a + (size_t)plane * H * W + p);
    const float* pa = (const float*)&va;
    float4 vo;
    float* o = (float*)&vo;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const Bilin bx = bilin_src(x0 + i, sx, AW);
        const float up = by.l0 * (bx.l0 * zp[by.i0 * AW + bx.i0] + bx.l1 * zp[by.i0 * AW + bx.i1]) +
                         by.l1 * (bx.l0 * zp[by.i1 * AW + bx.i0] + bx.l1 * zp[by.i1 * AW + bx.i1]);
        o[i] = pa[i] + up;
        s0 += o[i];
        s1 = fmaf(o[i], o[i], s1);
    }
    *(float4*)(out + (size_t)plane * H * W + p) = vo;
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * gridDim.x + blockIdx.x] = make_float2(s0, s1);
    }
}
extern "C" int uncr_add_upsampled(const float* a, const float* z, float* out, float* part, int planes, int H, int W,
                                  int AH, int AW, hipStream_t stream) {
    if (planes <= 0 || (W & 3) || ((H * W) % AGG_PX) || H < AH || W < AW) return UNCR_ESHAPE;
    if (!a || !z || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(add_upsampled_kernel, dim3(H * W / AGG_PX, planes), dim3(256), 0, stream, a, z, out,
                       (float2*)part, H, W, AH, AW);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// use_v on any-size planes (plane stride Pc >= H*W, csrc/anysize.hip): the same sum with scalar accesses and AGGA_NB statistics
// slots per plane; the tail [H*W, Pc) of `out` is written as zeros, as every consumer of a padded plane expects.
__global__ __launch_bounds__(256) void add_upsampled_any_kernel(const float* __restrict__ a, const float* __restrict__ z,
                                                                float* __restrict__ out, float2* __restrict__ part, int H, int W,
                                                                int Pc, int AH, int AW) {
    const int plane = blockIdx.y, P = H * W;
    const float sy = (float)AH / (float)H, sx = (float)AW / (float)W;
    const float* zp = z + (size_t)plane * AH * AW;
    float s0 = 0.f, s1 = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < Pc; p += AGGA_NB * 256) {
        float o = 0.f;
        if (p < P) {
            const int y = p / W, x = p - y * W;
            const Bilin by = bilin_src(y, sy, AH), bx = bilin_src(x, sx, AW);
            const float up = by.l0 * (bx.l0 * zp[by.i0 * AW + bx.i0] + bx.l1 * zp[by.i0 * AW + bx.i1]) +
                             by.l1 * (bx.l0 * zp[by.i1 * AW + bx.i0] + bx.l1 * zp[by.i1 * AW + bx.i1]);
            o = a[(size_t)plane * Pc + p] + up;
        }
        out[(size_t)plane * Pc + p] = o;
        s0 += o;
        s1 = fmaf(o, o, s1);
    }
    if (part) {
        __shared__ float red[8];
        block_sum2<256>(s0, s1, red);
        if (threadIdx.x == 0) part[(size_t)plane * AGGA_NB + blockIdx.x] = make_float2(s0, s1);
    }
}
extern "C" int uncr_add_upsampled_any(const float* a, const float* z, float* out, float* part, int planes, int H, int W, int Pc,
                                      int AH, int AW, hipStream_t stream) {
    if (planes <= 0 || H < AH || W < AW || Pc < H * W) return UNCR_ESHAPE;
    if (!a || !z || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(add_upsampled_any_kernel, dim3(AGGA_NB, planes), dim3(256), 0, stream, a, z, out, (float2*)part, H, W, Pc, AH,
                       AW);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_bilinear_adjoint_any(const float* src, float* dst, int planes, int H, int W, int Pc, int AH, int AW,
                                         hipStream_t stream) {
    if (planes <= 0 || H < AH || W < AW || Pc < H * W) return UNCR_ESHAPE;
    if (!src || !dst) return UNCR_EINVAL;
    hipLaunchKernelGGL(bilinear_adjoint_any_kernel, dim3((AH * AW + 255) / 256, planes), dim3(256), 0, stream, src, dst, H, W, AH, AW,
                       (size_t)Pc);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// adjoint of the bilinear up-sampling on its own: dst[q, ay, ax] = sum_{y,x} wy wx src[q, y, x]
extern "C" int uncr_bilinear_adjoint(const float* src, float* dst, int planes, int H, int W, int AH, int AW,
                                     hipStream_t stream) {
    if (planes <= 0 || H < AH || W < AW || (W & 3) || W > BADJ_MAXW) return UNCR_ESHAPE;
    hipLaunchKernelGGL(bilinear_adjoint_kernel, dim3(AH, planes), dim3(256), 0, stream, src, dst, H, W, AH, AW);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
// element-wise helpers of the value branch: out = a + b, and out = a * keep(seed, index) (nn.Dropout with the same
// counter-based stream as the aggregator; the backward applies the identical mask to the gradient)
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
extern "C" int uncr_add(const float* a, const float* b, float* out, long long n, hipStream_t stream) {
    if (n <= 0 || !a || !b || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ a, float* __restrict__ out, long long n,
                                                      unsigned long long seed, const long long* __restrict__ seed_dev,
                                                      float p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long sd = seed + (seed_dev ? (unsigned long long)seed_dev[0] * 0x9E3779B97F4A7C15ull : 0ull);
    out[i] = hash_uniform(sd, (unsigned long long)i) < p ? 0.f : a[i] / (1.f - p);
}
extern "C" int uncr_dropout(const float* a, float* out, long long n, unsigned long long seed, const long long* seed_dev,
                            float p, hipStream_t stream) {
    if (n <= 0 || !a || !out || p < 0.f || p >= 1.f) return UNCR_EINVAL;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, out, n, seed,
                       seed_dev, p);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
