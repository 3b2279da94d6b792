// Development probes (NOT part of the product library): MFMA peak probes, accuracy probes of the device erf / GELU and of
// the exact 3-way bf16 operand split, ds_read_b64_tr_b16 semantics.  Built into lib/libuncr_dev.so, declared in
// include/uncr_dev.h; used by tools/ and by tests/test_gpu_kernels.py::test_bf16_split_accuracy.
#include "pw_gemm.h"

// ---- debug: pure fp32-MFMA throughput probe (no memory traffic), used by tools/bench_kernels.py ----
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) out[0] = s;   // keep the chain alive
}
__global__ __launch_bounds__(256, 2) void mfma_probe_bf16_kernel(float* out, int iters, unsigned a0, unsigned b0) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 a = {a0 + threadIdx.x, a0 * 3u + threadIdx.x * 7u, a0 ^ 0x3f803f80u, a0 + 0x3c003c00u};
    u4 b = {b0 + threadIdx.x * 5u, b0 * 5u, b0 ^ 0x3f003f00u, b0 + 0x3d003d00u};
    a = (a & 0x3fff3fffu) | 0x30003000u; b = (b & 0x3fff3fffu) | 0x30003000u;   // random finite bf16 pairs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) out[0] = s;
}
extern "C" int uncr_debug_mfma_probe_bf16(float* out, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_probe_bf16_kernel, dim3(blocks), dim3(256), 0, stream, out, iters, 0x12345678u, 0x9abcdef1u);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
extern "C" int uncr_debug_mfma_probe(float* out, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, stream, out, iters, 1.0f, 0.5f);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- debug: numerics probe for the 3-way bf16 split (x = hi + mid + lo exactly, truncation split) fed to
// v_mfma_f32_32x32x16_bf16.  terms = 6 keeps hh, hm, mh, mm, hl, lh (drops <= 3*2^-24 |a||b|); 9 keeps all.
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
// the PRODUCT's split (pw_gemm.h::split3_bf16), so that the accuracy test exercises what the GEMM kernels use
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    unsigned hb, mb, lb;
    split3_bf16(x, hb, mb, lb);
    h = hb >> 16; m = mb >> 16; l = lb >> 16;
}
__global__ __launch_bounds__(64) void bf16split_probe_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             float* __restrict__ out, int K, int terms) {
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        u16x8_t ah, am, al, bh, bm, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, l;
            split3(A[i * K + k0 + 8 * kg + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
            split3(B[(k0 + 8 * kg + e) * 32 + i], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
        }
#define MF(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0)
        if (terms >= 9) { MF(al, bl); MF(am, bl); MF(al, bm); }
        if (terms >= 6) { MF(ah, bl); MF(al, bh); MF(am, bm); }
        if (terms >= 3) { MF(ah, bm); MF(am, bh); }
        MF(ah, bh);
#undef MF
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = acc[r];
}
extern "C" int uncr_debug_bf16split_probe(const float* A, const float* B, float* out, int K, int terms,
                                          hipStream_t stream) {
    if (K <= 0 || K % 16) return UNCR_ESHAPE;
    hipLaunchKernelGGL(bf16split_probe_kernel, dim3(1), dim3(64), 0, stream, A, B, out, K, terms);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ---- debug: the device erf_f / gelu_f / gelu_grad_f on an array (accuracy measurements, tools/probe_erf.py) ----
__global__ __launch_bounds__(256) void erf_probe_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int what) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = what == 0 ? erf_f(v) : (what == 1 ? gelu_f(v) : (what == 2 ? gelu_grad_f(v) : __builtin_amdgcn_exp2f(v)));
}
extern "C" int uncr_debug_erf(const float* x, float* y, int n, int what, hipStream_t stream) {
    if (n <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(erf_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, y, n, what);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read) semantics probe: LDS holds lds[i] = i (16-bit), lane l reads at element
// offset offs[l]; out[l*4 + j] = the j-th 16-bit value the lane receives.  Groundwork for feeding the same LDS tile to MFMA
// in both operand orientations (data GEMM + weight-gradient GEMM from one staged tile).
typedef short uncr_s4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void tr_b16_probe_kernel(const int* __restrict__ offs, int* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const uncr_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (uncr_s4 __attribute__((address_space(3)))*)(lds + offs[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
extern "C" int uncr_debug_tr_b16_probe(const int* offs, int* out, hipStream_t stream) {
    if (!offs || !out) return UNCR_EINVAL;
    hipLaunchKernelGGL(tr_b16_probe_kernel, dim3(1), dim3(64), 0, stream, offs, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}


// ---- stream roofs (round 4): what a pure HBM stream reaches on this part for a given read : write mix ----
// mode 0 read-only (R reads per element group, one tiny guarded store), 1 write-only, 2 copy (1 : 1), 3 two reads : one write,
// 4 one read : two writes, 5 three reads : one write.  float4 per lane, each block walks a CONTIGUOUS slab of every stream with
// UNR groups in flight (the access pattern of the plane-tiled kernels of the step); nt = non-temporal loads AND stores.
// Streams are separate buffers (a, b, c read; x, y written), n4 float4 elements each.
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void stream_probe_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           const float4* __restrict__ c, float4* __restrict__ x,
                                                           float4* __restrict__ y, size_t n4) {
    constexpr int UNR = 4;
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](const float4* p) { return ld4<float, NT>((const float*)p); };
    auto st = [&](float4* p, const float4& v) { st4<float, NT>((float*)p, v); };
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * UNR) {
        float4 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t k = i + (size_t)u * 256 < hi ? i + (size_t)u * 256 : i;
            if constexpr (MODE != 1) va[u] = ld(a + k);
            if constexpr (MODE == 3 || MODE == 5) vb[u] = ld(b + k);
            if constexpr (MODE == 5) vc[u] = ld(c + k);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t k = i + (size_t)u * 256;
            float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
            if constexpr (MODE != 1) v = va[u];
            if constexpr (MODE == 3 || MODE == 5) { v.x += vb[u].x; v.y += vb[u].y; v.z += vb[u].z; v.w += vb[u].w; }
            if constexpr (MODE == 5) { v.x += vc[u].x; v.y += vc[u].y; v.z += vc[u].z; v.w += vc[u].w; }
            if constexpr (MODE == 0) { keep.x += v.x; keep.y += v.y; keep.z += v.z; keep.w += v.w; }
            else if (k < hi) {
                st(x + k, v);
                if constexpr (MODE == 4) st(y + k, v);
            }
        }
    }
    if constexpr (MODE == 0)
        if (keep.x + keep.y + keep.z + keep.w == 1.2345e-30f) x[0] = keep;
}
// the same traffic in the access pattern of the plane-tiled GEMMs: every stream is [rows][65536] fp32 (one row = one (frame, channel)
// plane), a job = 256 rows x one 128-pixel tile, i.e. 512-BYTE pieces 256 KB apart; persistent blocks walk jobs b, b + G, ...
// MODE 6: read two streams, write one (the dz GEMM's 2 : 1), MODE 7: read two, write none, MODE 8: 7 reads : 1 write (the dx GEMM's mix:
// seven row groups read, one written)
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void stream_tile_probe_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                float* __restrict__ x, int rows_total) {
    constexpr int PP = 65536, TP = 128, RG = 256;
    const int ntile = PP / TP, njob = (rows_total / RG) * ntile;
    const int rsub = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int job = blockIdx.x; job < njob; job += gridDim.x) {
        const int g = job / ntile, t = job % ntile;
        const size_t base = (size_t)g * RG * PP + (size_t)t * TP + c4;
#pragma unroll 1
        for (int rr = 0; rr < RG / 8; rr += 4) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = base + (size_t)((rr + u) * 8 + rsub) * PP;
                va[u] = ld4<float, NT>(a + o);
                vb[u] = ld4<float, NT>(b + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = base + (size_t)((rr + u) * 8 + rsub) * PP;
                float4 v = va[u];
                v.x += vb[u].x; v.y += vb[u].y; v.z += vb[u].z; v.w += vb[u].w;
                if (MODE == 6 || (MODE == 8 && ((rr + u) & 3) == 0 && (rsub & 1) == 0)) st4<float, NT>(x + o, v);      // mode 8: every 8th row
                else { keep.x += v.x; keep.y += v.y; keep.z += v.z; keep.w += v.w; }
            }
        }
    }
    if (keep.x + keep.y + keep.z + keep.w == 1.2345e-30f) x[0] = keep.x;
}

// the weight-gradient kernels' pattern, read only: one 512-thread block per CU owns a contiguous pixel range of a 512-row frame and
// walks it in visits of PW pixels; a visit reads PW * 4 BYTES of every row (rows 256 KB apart), eight float4 per thread in flight.
// PW = 32 is what pw_wgrad_split.hip does (128-byte pieces); 64 / 128 ask what wider pieces would buy.
template <int PW, bool NT>
__global__ __launch_bounds__(512, 1) void stream_rows_probe_kernel(const float* __restrict__ a, float* __restrict__ x, int frames) {
    constexpr int PP = 65536, R = 512, TPR = PW / 4, RPI = 512 / TPR, NPIECE = R / RPI;
    const int nbx = gridDim.x / frames, n = blockIdx.x / nbx, bx = blockIdx.x % nbx;
    const int p0 = (int)((long long)bx * PP / nbx), p1 = (int)((long long)(bx + 1) * PP / nbx);
    const int lrow = threadIdx.x / TPR, c4 = (threadIdx.x % TPR) * 4;
    const float* base = a + ((size_t)n * R + lrow) * PP + c4;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = p0; p < p1; p += PW) {
#pragma unroll 1
        for (int i0 = 0; i0 < NPIECE; i0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4<float, NT>(base + (size_t)((i0 + u) * RPI) * PP + p);
#pragma unroll
            for (int u = 0; u < 8; ++u) { keep.x += v[u].x; keep.y += v[u].y; keep.z += v[u].z; keep.w += v[u].w; }
        }
    }
    if (keep.x + keep.y + keep.z + keep.w == 1.2345e-30f) x[0] = keep.x;
}

extern "C" int uncr_debug_stream_probe(const float* a, const float* b, const float* c, float* x, float* y, long long n_floats,
                                       int mode, int nt, int blocks, hipStream_t stream) {
    if (n_floats <= 0 || n_floats % 4 || blocks <= 0 || mode < 0 || mode > 11) return UNCR_ESHAPE;
    if (mode >= 9) {      // rows pattern: frames of 512 rows x 65536, `blocks` a multiple of the frame count
        const int frames = (int)(n_floats / (65536LL * 512));
        if (frames < 1 || blocks % frames) return UNCR_ESHAPE;
#define SR_LAUNCH(M, PW)                                                                                                      \
        case M:                                                                                                               \
            if (nt) hipLaunchKernelGGL((stream_rows_probe_kernel<PW, true>), dim3(blocks), dim3(512), 0, stream, a, x, frames);  \
            else hipLaunchKernelGGL((stream_rows_probe_kernel<PW, false>), dim3(blocks), dim3(512), 0, stream, a, x, frames);   \
            break;
        switch (mode) { SR_LAUNCH(9, 32) SR_LAUNCH(10, 64) SR_LAUNCH(11, 128) }
#undef SR_LAUNCH
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
    if (mode >= 6) {
        if (n_floats % (65536LL * 256)) return UNCR_ESHAPE;
        const int rows = (int)(n_floats / 65536);
#define ST_LAUNCH(M)                                                                                                            \
        case M:                                                                                                                 \
            if (nt) hipLaunchKernelGGL((stream_tile_probe_kernel<M, true>), dim3(blocks), dim3(256), 0, stream, a, b, x, rows);   \
            else hipLaunchKernelGGL((stream_tile_probe_kernel<M, false>), dim3(blocks), dim3(256), 0, stream, a, b, x, rows);    \
            break;
        switch (mode) { ST_LAUNCH(6) ST_LAUNCH(7) ST_LAUNCH(8) }
#undef ST_LAUNCH
        UNCR_LAUNCH_CHECK();
        return UNCR_OK;
    }
    const size_t n4 = (size_t)n_floats / 4;
#define SP_LAUNCH(M)                                                                                                          \
    case M:                                                                                                                   \
        if (nt) hipLaunchKernelGGL((stream_probe_kernel<M, true>), dim3(blocks), dim3(256), 0, stream, (const float4*)a,      \
                                   (const float4*)b, (const float4*)c, (float4*)x, (float4*)y, n4);                           \
        else hipLaunchKernelGGL((stream_probe_kernel<M, false>), dim3(blocks), dim3(256), 0, stream, (const float4*)a,        \
                                (const float4*)b, (const float4*)c, (float4*)x, (float4*)y, n4);                              \
        break;
    switch (mode) { SP_LAUNCH(0) SP_LAUNCH(1) SP_LAUNCH(2) SP_LAUNCH(3) SP_LAUNCH(4) SP_LAUNCH(5) }
#undef SP_LAUNCH
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
