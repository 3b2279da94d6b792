// EXPERIMENT (round 4), kept under tools/ for the record -- NOT part of the product library and no longer hooked into it (round 4
// linked it behind `if (pw_kres_dz_applies(g, in_dt)) return pw_kres_dz_launch(g, N, stream);` in uncr_pw_gemm's PRO_NORMBWD case;
// it includes "pw_gemm.h" from uncrtaints_amd/csrc).  Result: bit-identical to the
// chunk-pipelined kernel and EXACTLY as fast (isolated 201-206 us vs 204-206 us, in the step 12.01 vs 12.02 ms, interleaved A/B) --
// two structurally different kernels landing on the same time is the evidence that the dz GEMM is bound by its traffic (805 MB
// with one third written: ablating all compute leaves 165 us = 4.9 TB/s, the plain-store rate of this access pattern) and not by the
// chunk pipeline's depth or its epilogue latency.  Ablations of THIS kernel: no GELU' 175 us, no MFMA 175, neither 165-171, no stores 157.
//
// The dz GEMM of an MBConv backward (uncrtaints.py:121-146 backward of the pw2 / SE / GELU tail) with its fused pass-B epilogue,
//     du2[n, co, p] = GELU'(A2 h2 + B2) * (s * (W2^T dh3)[co, p] + dpool),   dh3 = C1 dy + C2 (h3 - mu) + C3,
// for the model's shape 128 -> 256 channels on fp32 storage, as a K-RESIDENT kernel: the whole contraction axis of a 128-pixel
// tile (Cin = 128: four 32-channel chunks, two scaled fp16 parts each = 64 KB) is staged into LDS once, and the 256 output
// channels are then computed in TWO passes of 128 (one 32-channel tile per wave and pass) that read nothing but LDS and the packed
// weights.  Against the chunk-pipelined kernel (pw_gemm_split.hip <2, NORMBWD, 3, 1, float, H2>):
//   * a wave keeps 64 accumulator registers instead of 128, which pays for two raw chunks in flight (that kernel has one: 254
//     VGPRs) and for double-buffered operand rows in the epilogue -- per-phase s_memtime stamps showed 62 % of its tile time in the
//     pass-B epilogue and its k-steps waiting on requests issued one chunk earlier;
//   * requests are ordered for gfx9's single in-order vmcnt queue: the next tile's first two raw chunks and the next pass's first
//     weight fragments are requested right behind a pass's last MFMA group and BEFORE that pass's epilogue stores, so nothing that
//     is waited for sits behind a fresh HBM request or an unacknowledged store;
//   * the MFMA passes touch no global activation data at all.
// Arithmetic is the chunk-pipelined kernel's: same prologue expression, same two-part fp16 split and per-frame power-of-two scale
// (pw_gemm.h, pw_gemm_split.hip), same three products per k-step in the same order, same epilogue expressions and statistics order
// -- the results are bit-identical (checked on MI355X in round 4 with the test that is now a comment at the end of this file).
// LDS bytes of a staged chunk c (16 KB): part * 8192 + ((ks * 2 + kg) * 4 + e) * 512 + j * 16 + half * 8, pixel = 4 j + e, as in the
// chunked kernel; weights Wp[ks][co tile][slot][lane] (slots 3, 4 = the scaled fp16 parts; tail = 1 / scale per output channel).
#include "../../uncrtaints_amd/csrc/pw_gemm.h"
#include <type_traits>

#ifndef KR_NT_LD
#define KR_NT_LD 1          // non-temporal activation loads (as the chunked kernel: measured -0.09 ms / step there)
#endif
#define KR_TP 128
#define KR_NK 4             // 32-channel chunks of the contraction axis
#define KR_CHUNK 16384      // staged bytes per chunk: two fp16 parts x 8 KB
#define KR_NCT 8            // 32-channel output tiles
#ifndef KR_ABL
#define KR_ABL 0    // development ablations: 1 no GELU' in the epilogue, 2 no MFMA, 4 no staging VALU (raw chunks are still requested), 8 no stores
#endif
#ifndef KR_RB
#define KR_RB 4             // operand rows per epilogue batch (two register sets)
#endif

__global__ __launch_bounds__(256, 2) void pw_dz_kres_kernel(PwArgs g) {
    constexpr int NT = 256, Cin = 128, Cout = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];      // [KR_NK][KR_CHUNK]
    __shared__ float cf[4][Cin];          // C1, C2, C3 (scaled), mean
    __shared__ float ecf[4][Cout];        // pass-B coefficients A, B, S, D
    __shared__ float red[Cout][2];
    __shared__ float hsc[Cout];           // 1 / (weight scale * frame scale) per output channel
    __shared__ float bred[4][6];
    __shared__ float bscale;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int n = blockIdx.y, P = g.P;
    const int ntile = P / KR_TP, G = gridDim.x, bx = blockIdx.x;
    const int nt = (ntile - bx + G - 1) / G;

    {
        const float* p0 = g.k0 ? g.k0 + (size_t)n * Cin : g.Wt;      // optional pointers are read branch-free (dummy location + select)
        const float* p1 = g.k1 ? g.k1 + (size_t)n * Cin : g.Wt;
        const float* p2 = g.k2 ? g.k2 + (size_t)n * Cin : g.Wt;
        const float* p3 = g.k3 ? g.k3 + (size_t)n * Cin : g.Wt;
        for (int i = tid; i < Cin; i += NT) {
            const float a = p0[i], b = p1[i], c = p2[i], m = p3[i];
            cf[0][i] = g.k0 ? a : 1.f;
            cf[1][i] = g.k1 ? b : 0.f;
            cf[2][i] = g.k2 ? c : 0.f;
            cf[3][i] = g.k3 ? m : 0.f;
        }
        for (int c = tid; c < Cout; c += NT) {
            const int ci = n * Cout + c;
            red[c][0] = 0.f; red[c][1] = 0.f;
            ecf[0][c] = g.e0[ci]; ecf[1][c] = g.e1[ci]; ecf[2][c] = g.e2[ci]; ecf[3][c] = g.e3[ci];
        }
    }

    // staging ownership: rows 4*cig .. 4*cig+3 of a chunk, pixels 4*sj .. 4*sj+3 of the tile
    const int sj = tid & 31, cig = tid >> 5;
    const float* inb = (const float*)g.in + (size_t)n * Cin * P + 4 * sj;
    const float* in2b = (const float*)g.in2 + (size_t)n * Cin * P + 4 * sj;
    const int st_off = ((cig >> 2) * 2 + ((cig >> 1) & 1)) * 2048 + sj * 16 + (cig & 1) * 8;

    // the block's raw chunk stream: chunk c of its ti-th tile; positions past the end re-read the last tile (never consumed)
    struct Pos { int c, ti; };
    auto advance = [&](Pos& p) { const bool wrap = p.c + 1 == KR_NK; p.c = wrap ? 0 : p.c + 1; p.ti += wrap ? 1 : 0; };
    auto tile_px = [&](int ti) { return (bx + (ti < nt ? ti : nt - 1) * G) * KR_TP; };
    float4 pre[2][4], pre2[2][4];
    auto load_chunk = [&](const Pos& p, auto slot) {
        constexpr int S = decltype(slot)::value;
        const int px = tile_px(p.ti);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = p.c * PWS_KC + 4 * cig + r;
            pre[S][r] = ld4<float, KR_NT_LD != 0>(inb + (size_t)k * P + px);
            pre2[S][r] = ld4<float, KR_NT_LD != 0>(in2b + (size_t)k * P + px);
        }
    };
    auto stage_chunk = [&](int kc, auto slot) {
        constexpr int S = decltype(slot)::value;
        float c0[4], c1[4], c2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kc * PWS_KC + 4 * cig + r;
            c0[r] = cf[0][k]; c1[r] = cf[1][k]; c2[r] = cf[2][k];
            const float c3 = cf[3][k];
            // centre the second operand in place (the chunked kernel's order of operations)
            pre2[S][r].x -= c3; pre2[S][r].y -= c3; pre2[S][r].z -= c3; pre2[S][r].w -= c3;
        }
        unsigned char* b = xs + kc * KR_CHUNK + st_off;
#pragma unroll
        for (int e = 0; e < ((KR_ABL & 4) ? 1 : 4); ++e) {
            float vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vv[r] = fmaf(c0[r], ((const float*)&pre[S][r])[e], fmaf(c1[r], ((const float*)&pre2[S][r])[e], c2[r]));
            if (KR_ABL & 4) vv[0] += ((const float*)&pre[S][1])[1] + ((const float*)&pre[S][2])[2] + ((const float*)&pre[S][3])[3]
                                   + ((const float*)&pre2[S][0])[1] + ((const float*)&pre2[S][1])[2] + ((const float*)&pre2[S][2])[3] + ((const float*)&pre2[S][3])[0];
            unsigned h01, l01, h23, l23;
            split2_f16_pair(vv[0], vv[1], h01, l01);
            split2_f16_pair(vv[2], vv[3], h23, l23);
            *(u32x2_t*)(b + e * 512) = u32x2_t{h01, h23};
            *(u32x2_t*)(b + 8192 + e * 512) = u32x2_t{l01, l23};
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    Pos lp{0, 0};
    load_chunk(lp, S0{}); advance(lp);
    load_chunk(lp, S1{}); advance(lp);
    __syncthreads();      // cf / ecf visible

    // per-frame power-of-two scale of the staged operand (rigorous bound -> 2^14), as the chunked kernel derives it
    {
        float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < Cin; i += NT) {
            m[0] = fmaxf(m[0], fabsf(cf[0][i])); m[1] = fmaxf(m[1], fabsf(cf[1][i]));
            m[2] = fmaxf(m[2], fabsf(cf[2][i])); m[3] = fmaxf(m[3], fabsf(cf[3][i]));
        }
        for (int i = tid; i < g.in2_amax_n; i += NT) m[5] = fmaxf(m[5], g.in2_amax[(size_t)n * g.in2_amax_n + i]);
        for (int i = tid; i < g.in_amax_n; i += NT) {
            const float v = g.in_amax[(size_t)n * g.in_amax_n + i];
            m[4] = v > m[4] || !(v == v) ? v : m[4];         // a NaN bound stays (no scaling below)
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) { const float o = __shfl_xor(m[q], sft, 64); m[q] = o > m[q] || !(o == o) ? o : m[q]; }
            if (lane == 0) bred[wn][q] = m[q];
        }
        __syncthreads();
        if (tid == 0) {
            float mm[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                mm[q] = bred[0][q];
#pragma unroll
                for (int w = 1; w < 4; ++w) mm[q] = bred[w][q] > mm[q] || !(bred[w][q] == bred[w][q]) ? bred[w][q] : mm[q];
            }
            const float bound = mm[0] * mm[4] + mm[1] * (mm[5] + mm[3]) + mm[2];
            float sc = 1.f;
            if (bound > 0.f && bound < 3.0e38f) {
                int e;
                (void)frexpf(bound, &e);
                e = 14 - e;
                e = e > 100 ? 100 : (e < -100 ? -100 : e);
                sc = ldexpf(1.f, e);
            }
            bscale = sc;
        }
        __syncthreads();
        const float sc = bscale;
        for (int i = tid; i < Cin; i += NT) { cf[0][i] *= sc; cf[1][i] *= sc; cf[2][i] *= sc; }
        const float* wtail = g.Wt + (size_t)pws_nks(Cin) * KR_NCT * PWS_NSLOT * 64 * 4;
        const float isc = 1.f / sc;
        for (int c = tid; c < Cout; c += NT) hsc[c] = wtail[c] * isc;
        __syncthreads();
    }

    const int rd_off = kg * 2048 + j * 16;
    const u32x4_t* wp = (const u32x4_t*)g.Wt + lane;
    auto lda = [&](int ks, int cot, int part) { return wp[((size_t)(ks * KR_NCT + cot) * PWS_NSLOT + 3 + part) * 64]; };
    auto ldb = [&](const unsigned char* p0, int part, u32x4_t (&b)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = *(const u32x4_t*)(p0 + part * 8192 + e * 512);
    };
#define KR_SB() __builtin_amdgcn_sched_barrier(0)
#define KR_MF(A, B)                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                              \
        if (!(KR_ABL & 2)) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, A), __builtin_bit_cast(f16x8_t, B[e]), acc[e], 0, 0, 0)

    // weight fragments of two k-steps ahead (set = k-step parity): requested before a pass's epilogue stores, waited for after
    u32x4_t ah[2], al[2];
    {
        const int cot = wn * 2;
        ah[0] = lda(0, cot, 0); al[0] = lda(0, cot, 1);
        ah[1] = lda(1, cot, 0); al[1] = lda(1, cot, 1);
    }

    for (int ti = 0; ti < nt; ++ti) {
        // ---- stage the tile's four chunks (chunks 0, 1 were requested during the previous tile's first pass) ----
        stage_chunk(0, S0{}); load_chunk(lp, S0{}); advance(lp);      // -> this tile's chunk 2
        stage_chunk(1, S1{}); load_chunk(lp, S1{}); advance(lp);      // -> chunk 3
        stage_chunk(2, S0{});
        stage_chunk(3, S1{});
        __syncthreads();

        const int tile = bx + ti * G;
        const int loff = 4 * kg * P + tile * KR_TP + 4 * j;
        int nco = n * Cout;
        asm volatile("" : "+s"(nco));
        const float* auxp = (const float*)g.aux;
        float* outp = (float*)g.out;

#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int cot = wn * 2 + pass;
            auto row_of = [&](int r) { return cot * 32 + (r & 3) + 8 * (r >> 2); };     // + 4*kg per lane
            // operand rows of the epilogue's first batch: in flight under the MFMA pass
            float4 xa[2][KR_RB];
#pragma unroll
            for (int q = 0; q < KR_RB; ++q) xa[0][q] = ld4<float, false>(auxp + (size_t)(nco + row_of(q)) * P + loff);

            f32x16 acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
            u32x4_t bh[4], bl[4];
            ldb(xs + rd_off, 0, bh);
#pragma unroll
            for (int ks = 0; ks < 2 * KR_NK; ++ks) {
                const unsigned char* cb = xs + (ks >> 1) * KR_CHUNK + (ks & 1) * 4096 + rd_off;
                const unsigned char* nb = xs + ((ks + 1) >> 1) * KR_CHUNK + ((ks + 1) & 1) * 4096 + rd_off;
                const int q = ks & 1;
                // fragments of k-step ks + 2; past the pass's end: k-steps 0 / 1 of the NEXT pass (the other co tile of this wave,
                // or -- same fragments for every tile -- the first pass of the next tile)
                const int ksn = ks + 2 < 2 * KR_NK ? ks + 2 : ks + 2 - 2 * KR_NK;
                const int cotn = ks + 2 < 2 * KR_NK ? cot : wn * 2 + (pass ^ 1);
                ldb(cb, 1, bl);
                KR_SB();
                KR_MF(ah[q], bh); KR_SB();
                KR_MF(al[q], bh); KR_SB();
                al[q] = lda(ksn, cotn, 1);
                KR_SB();
                KR_MF(ah[q], bl); KR_SB();
                ah[q] = lda(ksn, cotn, 0);
                if (ks + 1 < 2 * KR_NK) ldb(nb, 0, bh);
                KR_SB();
            }
            if (pass == 0) {
                // the next tile's first two raw chunks: behind this pass's MFMA groups (every weight request of the pass is older),
                // ahead of both epilogues' stores
                load_chunk(lp, S0{}); advance(lp);
                load_chunk(lp, S1{}); advance(lp);
            }
            // ---- epilogue of the pass: 16 rows of this wave's co tile in batches of KR_RB, operand rows double-buffered ----
#pragma unroll
            for (int bi = 0; bi < 16 / KR_RB; ++bi) {
                if (bi + 1 < 16 / KR_RB) {
#pragma unroll
                    for (int q = 0; q < KR_RB; ++q)
                        xa[(bi + 1) & 1][q] = ld4<float, false>(auxp + (size_t)(nco + row_of(KR_RB * (bi + 1) + q)) * P + loff);
                }
                KR_SB();
#pragma unroll
                for (int q = 0; q < KR_RB; ++q) {
                    const int r = KR_RB * bi + q;
                    const int rw = row_of(r);
                    const int col = rw + 4 * kg;
                    const float hinv = hsc[col];
                    float4 v = make_float4(fmaf(acc[0][r], hinv, 0.f), fmaf(acc[1][r], hinv, 0.f), fmaf(acc[2][r], hinv, 0.f),
                                           fmaf(acc[3][r], hinv, 0.f));
                    const float4 x = xa[bi & 1][q];
                    const float eA = ecf[0][col], eB = ecf[1][col], eS = ecf[2][col], eD = ecf[3][col];
                    if (KR_ABL & 1) {
                        v.x = fmaf(eA, x.x, eB) * fmaf(eS, v.x, eD); v.y = fmaf(eA, x.y, eB) * fmaf(eS, v.y, eD);
                        v.z = fmaf(eA, x.z, eB) * fmaf(eS, v.z, eD); v.w = fmaf(eA, x.w, eB) * fmaf(eS, v.w, eD);
                    } else {
                    v.x = gelu_grad_f(fmaf(eA, x.x, eB)) * fmaf(eS, v.x, eD);
                    v.y = gelu_grad_f(fmaf(eA, x.y, eB)) * fmaf(eS, v.y, eD);
                    v.z = gelu_grad_f(fmaf(eA, x.z, eB)) * fmaf(eS, v.z, eD);
                    v.w = gelu_grad_f(fmaf(eA, x.w, eB)) * fmaf(eS, v.w, eD);
                    }
                    float s0 = v.x + v.y + v.z + v.w;
                    float s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                    s0 = half_wave_sum_dpp(s0);
                    s1 = half_wave_sum_dpp(s1);
                    if (j == 31) { red[col][0] += s0; red[col][1] += s1; }
                    if (!(KR_ABL & 8) || v.x == 1.2345e-30f) st4<float, false>(outp + (size_t)(nco + rw) * P + loff, v);
                }
                KR_SB();
            }
        }
        __syncthreads();      // every wave is done reading the staged tile
    }
#undef KR_MF
#undef KR_SB
    __syncthreads();
    for (int c = tid; c < Cout; c += NT) g.part[((size_t)n * Cout + c) * G + bx] = make_float2(red[c][0], red[c][1]);
}

// the launches pw_gemm's dispatcher hands over: fp32 storage, 128 -> 256, norm-backward prologue, pass-B epilogue, both magnitude
// bounds given (the two-part fp16 route) -- exactly the dz GEMM of the model's MBConv blocks
bool pw_kres_dz_applies(const PwArgs& g, int act) {
    return act == UNCR_F32 && g.pro == PRO_NORMBWD && g.epi == 3 && g.Cin == 128 && g.Cout == 256 && g.h2 && g.in_amax && g.in2_amax &&
           g.in_amax_n > 0 && g.in2_amax_n > 0 && !g.bias && g.part && g.P % KR_TP == 0;
}

int pw_kres_dz_launch(const PwArgs& g, int N, hipStream_t stream) {
    static bool attr_set = false;
    constexpr size_t lds = (size_t)KR_NK * KR_CHUNK;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)pw_dz_kres_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return UNCR_ESHAPE;
        attr_set = true;
    }
    dim3 grid(pw_split_blocks_per_frame(N, g.P), N);
    hipLaunchKernelGGL(pw_dz_kres_kernel, grid, dim3(256), lds, stream, g);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

/* The check that was run (tests/test_gpu_kernels.py, round 4; 3 shapes, outputs and statistics partials torch.equal):
 *   out_k, part_k = E.pw_gemm(dy, Wk, N, 128, 256, P, pro=3, k=kk, x2=h3, epi=3, aux=h2, ek=ek, in_amax=a1, in2_amax=a2)           # this kernel
 *   out_c, part_c = E.pw_gemm(... same ..., bias=torch.zeros(256))        # zero bias: the dispatcher keeps the chunked kernel
 *   assert torch.equal(out_k, out_c) and torch.equal(part_k.buf, part_c.buf)
 */
