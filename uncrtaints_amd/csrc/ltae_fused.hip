// LTAE2dtiny (ltae.py:145-239) as ONE fused kernel per direction: per low-resolution pixel the GroupNorm over (8 channels x T
// dates), the two linear maps, the positional encoding, the query product and the masked temporal softmax.
//
// Nothing between the GroupNorm and the softmax is non-linear (ltae.py:211-224 inconv -> + PE -> fc1_k -> q.k / sqrt(d_k)), so the
// attention score is ONE linear functional of the normalised input per head:
//     score[h,b,t,s] = sum_c A'[h,c] * xhat[b,t,c,s] + B'[h,b,t],          xhat = (x - mu[b,g,s]) * rstd[b,g,s]
//     A [h,c]  = 1/sqrt(d_k) * sum_d Q[h,d] * (Wk Wi)[h*d_k+d, c],         A' = A * gamma[c]
//     B'[h,bt] = 1/sqrt(d_k) * sum_d Q[h,d] * (Wk (b_i + PE[bt]) + b_k)[h*d_k+d] + sum_c A[h,c] * beta[c]
// `ltae_compose` builds A' (NH x C = 8 KB) and B' in fp64 from the parameters (a few microseconds); the fused forward then reads
// the pooled features once per pass, keeps A' in LDS and the T x 4-head score tile of a thread in registers, and writes the
// attention; the fused backward stages the score gradients of its 64 pixels in LDS ([NH][T][64]), applies the composed map
// transposed, the GroupNorm backward, and reduces the gradients of A' and B' with wave shuffles.  `ltae_compose_bwd` turns those
// two small gradients into the gradients of Q, fc1_k, inconv and in_norm by the chain rule (dense algebra on <= 64 x 256 matrices).
// Replaces 8 small kernels + 2 narrow GEMMs per direction; UNCRTAINTS(use_v=True) needs the projected features themselves and
// keeps the unfused path.
#include "common.h"


struct LfArgs {
    const float* x;        // [B][T][C][S] pooled features
    const float* Ap;       // [NH][C]
    const float* Bp;       // [NH][B*T]
    const int* pad;        // [B][T] or null
    float* att;            // [NH][B][T][S]
    float* mean;           // [B][G][S]
    float* rstd;
    const float* datt;     // bwd
    float* dx;             // bwd [B][T][C][S]
    float* partA;          // bwd [nblk][NH][C], nblk = B*S / (256/NH) blocks, sample-major
    float* partB;          // bwd [nblk][NH][T]
    int B, T, C, NH, S;
    float eps;
};

// ---- parameter composition ---------------------------------------------------------------------------------------------
// One launch, grid = NH, block = 1024 (everything here is latency: sixteen waves per head).  Block h:
//   pe[n][i]   = sin / cos(date[n] / denom[i])                      PositionalEncoder (positional_encoding.py:5-31): channel j of the
//   bias1[n][j] = b_i[j] + pe[n][j mod d]                           d_model-wide bias uses sinusoid j mod d; block 0 stores bias1
//   M[hd][c] = sum_j Wk[hd][j] Wi[j][c]                             for the head's DK rows hd = h*DK + d
//   U[hd][n] = sum_j Wk[hd][j] bias1[n][j] + bk[hd]
//   Ap[h][c] = gamma[c] * A[h][c], A = sc * sum_d Q[h][d] M[hd][c];  Bp[h][n] = sc * sum_d Q[h][d] U[hd][n] + sum_c A[h][c] beta[c]
// dynamic LDS: pe [NF*dpe] f32 | Mf [DK*C] f32 | Uf [DK*NF] f32 | comb [2][DK*C] f64 (re-used as the 1024-entry reduction buffer)
__global__ __launch_bounds__(1024) void ltae_compose_kernel(
    const float* __restrict__ Q, const float* __restrict__ Wk, const float* __restrict__ bk, const float* __restrict__ Wi,
    const float* __restrict__ bin, const float* __restrict__ dates, const float* __restrict__ denom, int dpe, int use_pe,
    const float* __restrict__ gamma, const float* __restrict__ beta, int DK, int D, int C, int NF, float* __restrict__ bias1,
    float* __restrict__ Ap, float* __restrict__ Bp, float* __restrict__ M, float* __restrict__ U) {
    extern __shared__ double lds_d[];
    const int h = blockIdx.x, tid = threadIdx.x;
    const int ncomb = max(2 * DK * C, 1024);
    double* comb = lds_d;                                  // [2][DK*C] / red[1024]
    float* spe = (float*)(lds_d + ncomb);                  // [NF][dpe]
    float* sM = spe + (size_t)NF * dpe;                    // [DK][C]
    float* sU = sM + (size_t)DK * C;                       // [DK][NF]
    for (int i = tid; i < NF * dpe; i += 1024) {
        float v = 0.f;
        if (use_pe) {
            const float a = dates[i / dpe] / denom[i % dpe];
            v = ((i % dpe) & 1) ? cosf(a) : sinf(a);
        }
        spe[i] = v;
    }
    __syncthreads();
    if (h == 0)
        for (int i = tid; i < NF * D; i += 1024) {
            const int n = i / D, j = i % D;
            float v = bin ? bin[j] : 0.f;
            if (use_pe) v += spe[n * dpe + j % dpe];
            bias1[i] = v;
        }
    // M: two halves of the j range per output, the loads of 16 steps issued together (a plain loop serialises 256 L2 round trips)
    {
        const int jh = tid >> 9;
        const int j0 = jh ? D / 2 : 0, j1 = jh ? D : D / 2;
        for (int o = tid & 511; o < DK * C; o += 512) {
            const int d = o / C, c = o % C;
            const float* wk = Wk + (size_t)(h * DK + d) * D;
            double sacc = 0.0;
            int j = j0;
            for (; j + 16 <= j1; j += 16) {
                float w[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = Wi[(size_t)(j + q) * C + c];
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc += (double)wk[j + q] * (double)w[q];
            }
            for (; j < j1; ++j) sacc += (double)wk[j] * (double)Wi[(size_t)j * C + c];
            comb[(size_t)jh * DK * C + o] = sacc;
        }
    }
    // U: sixteen lanes per (row, frame) pair walk j with stride 16, combined by shuffles in a fixed order
    {
        const int l16 = tid & 15;
        for (int pr = tid >> 4; pr < DK * NF; pr += 64) {
            const int d = pr / NF, n = pr % NF;
            const float* wk = Wk + (size_t)(h * DK + d) * D;
            double sacc = 0.0;
            for (int j = l16; j < D; j += 16) {
                float v = bin ? bin[j] : 0.f;
                if (use_pe) v += spe[n * dpe + j % dpe];
                sacc += (double)wk[j] * (double)v;
            }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m, 16);
            if (l16 == 0) {
                const float u = (float)(sacc + (double)bk[h * DK + d]);
                sU[d * NF + n] = u;
                U[(size_t)(h * DK + d) * NF + n] = u;
            }
        }
    }
    __syncthreads();
    for (int o = tid; o < DK * C; o += 1024) {
        const float m = (float)(comb[o] + comb[(size_t)DK * C + o]);
        sM[o] = m;
        M[(size_t)h * DK * C + o] = m;
    }
    __syncthreads();
    const double sc = 1.0 / sqrt((double)DK);
    double ab = 0.0;
    for (int c = tid; c < C; c += 1024) {
        double a = 0.0;
        for (int d = 0; d < DK; ++d) a += (double)Q[h * DK + d] * (double)sM[d * C + c];
        a *= sc;
        Ap[(size_t)h * C + c] = (float)(a * (double)gamma[c]);
        ab += a * (double)beta[c];
    }
    double* red = comb;
    red[tid] = ab;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double abt = red[0];
    for (int n = tid; n < NF; n += 1024) {
        double sacc = 0.0;
        for (int d = 0; d < DK; ++d) sacc += (double)Q[h * DK + d] * (double)sU[d * NF + n];
        Bp[(size_t)h * NF + n] = (float)(sacc * sc + abt);
    }
}

// Cross-lane sums of the fused kernels without the LDS crossbar (ds_bpermute: ~2 issue slots + an LDS round trip per __shfl_xor; the
// forward kernel makes 192 of them per thread, the backward 300).  For the model's 16 heads the NH lanes of a pixel are exactly one
// DPP row: four full-rate v_add_f32_dpp give every lane the row sum; the sum over the wave's four pixels (lanes i, i+16, i+32, i+48)
// takes gfx950's v_permlane16_swap / v_permlane32_swap (rows 1 <-> 0 and 3 <-> 2 of a copy, then the wave's halves) and two adds.
// Other head counts keep the shuffle butterfly.  (The summation order differs from the butterfly's: last-bit differences.)
__device__ __forceinline__ float lf_pixel_sum(float p, int NH) {            // over the pixel's NH lanes; every lane gets the sum
    if (NH == 16) {                                                        // wave-uniform
        p += dpp_mov<0xB1>(p);                 // quad_perm [1,0,3,2]
        p += dpp_mov<0x4E>(p);                 // quad_perm [2,3,0,1]
        p += dpp_mov<0x124>(p);                // row_ror:4
        p += dpp_mov<0x128>(p);                // row_ror:8
        return p;
    }
    for (int m = NH >> 1; m >= 1; m >>= 1) p += __shfl_xor(p, m, 64);
    return p;
}
__device__ __forceinline__ float lf_wave_pixels_sum(float v, int NH) {      // over the wave's 64 / NH pixels (same lane within the pixel)
    if (NH == 16) {
        // Inline asm, not __builtin_amdgcn_permlane16_swap: with ROCm 7.2's hipcc the sum of the builtin's TWO results came out as
        // v_add v, vdst, vdst (the second result replaced by the first; seen in the ISA, 100 % wrong weight gradients) -- the asm
        // statement names both registers as read-write.  s_nop 1 = the two wait states the swap needs behind a VALU write of its
        // operands (cdna_hip_programming.md T21).
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a = (r0, r0, r2, r2), b = (r1, r1, r3, r3)
        float c = a + b, d = c;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));      // c = (lower, lower), d = (upper, upper)
        return c + d;
    }
    for (int m = NH; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- fused forward -----------------------------------------------------------------------------------------------------
// thread = (pixel, GroupNorm group g) -- the number of groups equals the number of heads (ltae.py:191-194), so the NH lanes of a
// pixel hold its NH groups in the first half of the kernel and its NH heads in the second.  Wave = 64 / NH pixels, block = 4 waves,
// grid = B * S / (256 / NH) blocks (256 blocks at the model's shape: the problem is tiny, parallelism and latency are everything).
//   1. the thread loads its group's T x CG values once, computes mean / rstd in registers and normalises in place;
//   2. for every head h the group's partial score  sum_{c in g} A'[h][c] * xhat[t][c]  is summed over the pixel's NH lanes with
//      wave shuffles; lane h keeps head h's T scores;
//   3. masked softmax over T, one attention row per lane.
template <int TMAX, int CG>
__global__ __launch_bounds__(256) void ltae_fused_fwd_kernel(LfArgs g) {
    extern __shared__ float lds[];          // Ap [NH][C]
    const int T = g.T, C = g.C, NH = g.NH, S = g.S;
    const int ppb = 256 / NH;                                   // pixels per block
    const int lane_g = threadIdx.x % NH, pl = threadIdx.x / NH;  // group / head of this lane, pixel within the block
    const long long pix = (long long)blockIdx.x * ppb + pl;     // over B*S
    const int b = (int)(pix / S), s = (int)(pix % S);
    for (int i = threadIdx.x; i < NH * C; i += 256) lds[i] = g.Ap[i];
    __syncthreads();
    float xv[TMAX][CG];
    const float* xb = g.x + ((size_t)b * T * C + (size_t)lane_g * CG) * S + s;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) {
            xv[t][j] = t < T ? xb[((size_t)t * C + j) * S] : 0.f;
            sum += xv[t][j];
        }
    const float Mn = (float)(T * CG);
    const float mu = sum / Mn;
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j)
            if (t < T) { const float d = xv[t][j] - mu; var = fmaf(d, d, var); }
    const float r = 1.0f / sqrtf(var / Mn + g.eps);
    g.mean[((size_t)b * NH + lane_g) * S + s] = mu;
    g.rstd[((size_t)b * NH + lane_g) * S + s] = r;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) xv[t][j] = (xv[t][j] - mu) * r;
    float sc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sc[t] = 0.f;
    for (int h = 0; h < NH; ++h) {
        const float* ap = lds + h * C + lane_g * CG;
        float a[CG];
#pragma unroll
        for (int j = 0; j < CG; ++j) a[j] = ap[j];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {                      // wave-uniform
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < CG; ++j) p = fmaf(a[j], xv[t][j], p);
                p = lf_pixel_sum(p, NH);                                              // over the pixel's NH lanes
                if (lane_g == h) sc[t] = p;
            }
        }
    }
    const int h = lane_g;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) {
            float v = sc[t] + g.Bp[(size_t)h * g.B * T + b * T + t];
            if (g.pad && g.pad[b * T + t]) v = -1e3f;          // masked_fill(pad, -1e3), ltae.py:435
            sc[t] = v;
            mx = fmaxf(mx, v);
        }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) { sc[t] = expf(sc[t] - mx); den += sc[t]; }
    const float inv = 1.0f / den;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) g.att[(((size_t)h * g.B + b) * T + t) * S + s] = sc[t] * inv;
}

// ---- fused backward ----------------------------------------------------------------------------------------------------
// Same thread mapping.  1. lane (pixel, h): softmax backward -> ds[t]; staged in LDS per pixel ([ppb][NH][T]); block partial of
// d B'.  2. lane (pixel, g): d xhat[t][c] = sum_h A'[h][c] ds[h][t] for its own channels, GroupNorm backward in registers -> dx;
// d A'[h][c] contributions (sum_t ds[h][t] xhat[t][c]) reduced over the block's pixels (wave shuffles, then LDS) -> block partial.
template <int TMAX, int CG>
__global__ __launch_bounds__(256) void ltae_fused_bwd_kernel(LfArgs g) {
    extern __shared__ float lds[];
    const int T = g.T, C = g.C, NH = g.NH, S = g.S;
    const int ppb = 256 / NH, ppw = 64 / NH;
    const int lane_g = threadIdx.x % NH, pl = threadIdx.x / NH, wv = threadIdx.x >> 6;
    const long long pix = (long long)blockIdx.x * ppb + pl;
    const int b = (int)(pix / S), s = (int)(pix % S);
    float* Ap = lds;                        // [NH][C]
    float* dsl = Ap + NH * C;               // [ppb][NH][T]
    float* accA = dsl + ppb * NH * T;       // [4 waves][NH][C]  (d A' partials of the waves)
    float* accB = accA + 4 * NH * C;        // [4 waves][NH][T]
    for (int i = threadIdx.x; i < NH * C; i += 256) Ap[i] = g.Ap[i];
    // 1. softmax backward (lane = head)
    {
        const int h = lane_g;
        float a[TMAX], d[TMAX], dot = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            a[t] = d[t] = 0.f;
            if (t < T) {
                const size_t o = (((size_t)h * g.B + b) * T + t) * S + s;
                a[t] = g.att[o];
                d[t] = g.datt[o];
                dot = fmaf(a[t], d[t], dot);
            }
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                float v = a[t] * (d[t] - dot);
                if (g.pad && g.pad[b * T + t]) v = 0.f;          // a padded date's score is the constant -1e3
                dsl[(pl * NH + h) * T + t] = v;
                v = lf_wave_pixels_sum(v, NH);                                    // over the wave's pixels (same head)
                if ((threadIdx.x & 63) == lane_g) accB[(wv * NH + h) * T + t] = v;
            }
    }
    __syncthreads();
    // 2. (lane = group)
    float xv[TMAX][CG];
    const float* xb = g.x + ((size_t)b * T * C + (size_t)lane_g * CG) * S + s;
    const float mu = g.mean[((size_t)b * NH + lane_g) * S + s], r = g.rstd[((size_t)b * NH + lane_g) * S + s];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) xv[t][j] = t < T ? (xb[((size_t)t * C + j) * S] - mu) * r : 0.f;
    float dxh[TMAX][CG];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) dxh[t][j] = 0.f;
    const float* dsp = dsl + pl * NH * T;
    for (int h = 0; h < NH; ++h) {
        const float* ap = Ap + h * C + lane_g * CG;
        float a[CG], da[CG];
#pragma unroll
        for (int j = 0; j < CG; ++j) { a[j] = ap[j]; da[j] = 0.f; }
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                const float dv = dsp[h * T + t];
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    dxh[t][j] = fmaf(a[j], dv, dxh[t][j]);
                    da[j] = fmaf(dv, xv[t][j], da[j]);
                }
            }
#pragma unroll
        for (int j = 0; j < CG; ++j) {
            const float v = lf_wave_pixels_sum(da[j], NH);                       // over the wave's pixels (same group)
            if ((threadIdx.x & 63) == lane_g) accA[(wv * NH + h) * C + lane_g * CG + j] = v;
        }
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) { m1 += dxh[t][j]; m2 = fmaf(dxh[t][j], xv[t][j], m2); }
    const float Mn = (float)(T * CG);
    m1 /= Mn;
    m2 /= Mn;
    float* dxb = g.dx + ((size_t)b * T * C + (size_t)lane_g * CG) * S + s;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j)
            if (t < T) dxb[((size_t)t * C + j) * S] = r * (dxh[t][j] - m1 - xv[t][j] * m2);
    __syncthreads();
    for (int i = threadIdx.x; i < NH * C; i += 256)
        g.partA[(size_t)blockIdx.x * NH * C + i] = (accA[i] + accA[NH * C + i]) + (accA[2 * NH * C + i] + accA[3 * NH * C + i]);
    for (int i = threadIdx.x; i < NH * T; i += 256)
        g.partB[(size_t)blockIdx.x * NH * T + i] = (accB[i] + accB[NH * T + i]) + (accB[2 * NH * T + i] + accB[3 * NH * T + i]);
    (void)ppw;
}

// ---- gradients of the parameters from d A' [NH][C] and d B' ([B][NH][T], as the block partials reduce) --------------------
#define LF_DB(h, n) dBp[((size_t)((n) / T) * NH + (h)) * T + (n) % T]
// grid = NH, block = 1024: the head's d A' [C] and d B' [NF] from the fused backward's block partials (what two column-sum launches
// did before: fp64, fixed order), then d A[h][:] (gamma / beta folded back), d Q[h][:] and the per-head contributions to d gamma /
// d beta (summed over h by the weight-gradient kernel's last block).  dBp [B][NH][T] is stored for that kernel.
__global__ __launch_bounds__(1024) void ltae_compose_bwd_a_kernel(const float* __restrict__ Q, const float* __restrict__ M,
                                                                  const float* __restrict__ U, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ partA,
                                                                  const float* __restrict__ partB, int nblk, int DK, int C, int NF,
                                                                  int T, float* __restrict__ dBp, float* __restrict__ dA,
                                                                  float* __restrict__ dQ, float* __restrict__ dgb /* [NH][2][C] */,
                                                                  float* __restrict__ sBout /* [NH] = sum_n dB'[h][n] */) {
    const int h = blockIdx.x, NH = gridDim.x, tid = threadIdx.x;
    const double sc = 1.0 / sqrt((double)DK);
    __shared__ double red[1024];
    __shared__ float sdAp[256];         // C <= 256
    __shared__ double sB;
    {
        // d A'[h][c] = sum over the B*nblk block partials: 4 slices of the row range per channel, combined in a fixed order
        const int c = tid & 255, sl = tid >> 8, R = (NF / T) * nblk;
        double acc = 0.0;
        if (c < C) {
            const float* src = partA + (size_t)h * C + c;
            const size_t ld = (size_t)NH * C;
            const int r1 = (R * (sl + 1)) / 4;
            int r = (R * sl) / 4;
            for (; r + 8 <= r1; r += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(r + q) * ld];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += (double)v[q];
            }
            for (; r < r1; ++r) acc += (double)src[(size_t)r * ld];
        }
        red[tid] = acc;
    }
    __syncthreads();
    if (tid < C) sdAp[tid] = (float)(((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid]);
    // d B'[b][h][t] = sum over the nblk blocks of sample b: one wave per output
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int n = wv; n < NF; n += 16) {
            const int b = n / T, t = n % T;
            double acc = 0.0;
            for (int blk = lane; blk < nblk; blk += 64) acc += (double)partB[(((size_t)b * nblk + blk) * NH + h) * T + t];
            acc = wave_sum_d(acc);
            if (lane == 0) LF_DB(h, n) = (float)acc;
        }
    }
    __syncthreads();        // dBp of this head is read back below by other threads of the block (global, same block: visible)
    double v = 0.0;
    for (int n = tid; n < NF; n += 1024) v += (double)LF_DB(h, n);
    red[tid] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) { sB = red[0]; sBout[h] = (float)red[0]; }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        double a = 0.0;
        for (int d = 0; d < DK; ++d) a += (double)Q[h * DK + d] * (double)M[(size_t)(h * DK + d) * C + c];
        a *= sc;                                                        // A[h][c]
        const double dap = (double)sdAp[c];
        dA[(size_t)h * C + c] = (float)(dap * (double)gamma[c] + sB * (double)beta[c]);
        dgb[((size_t)h * 2 + 0) * C + c] = (float)(dap * a);            // d gamma contribution
        dgb[((size_t)h * 2 + 1) * C + c] = (float)(a * sB);             // d beta contribution
    }
    __syncthreads();
    for (int d = 0; d < DK; ++d) {
        double q = 0.0;
        for (int c = tid; c < C; c += 1024)
            q += ((double)sdAp[c] * (double)gamma[c] + sB * (double)beta[c]) * (double)M[(size_t)(h * DK + d) * C + c];
        for (int n = tid; n < NF; n += 1024) q += (double)LF_DB(h, n) * (double)U[(size_t)(h * DK + d) * NF + n];
        red[tid] = q;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) dQ[h * DK + d] = (float)(red[0] * sc);
        __syncthreads();
    }
}
// block hd of HK: d Wk[hd][:], d bk[hd]          (d M[hd][c] = sc Q[h][d] dA[h][c],  d U[hd][n] = sc Q[h][d] dB'[h][n])
__device__ __forceinline__ void ltae_compose_bwd_wk_body(int hd, int NH, const float* __restrict__ Q, const float* __restrict__ Wi,
                                                         const float* __restrict__ bias1, const float* __restrict__ dA,
                                                         const float* __restrict__ dBp, const float* __restrict__ sBv,
                                                         int DK, int D, int C, int NF, int T, float* __restrict__ dWk,
                                                         float* __restrict__ dbk) {
    const int h = hd / DK;
    const double q = (double)Q[hd] / sqrt((double)DK);
    for (int j = threadIdx.x; j < D; j += 256) {
        double s = 0.0;
        int c = 0;
        for (; c + 16 <= C; c += 16) {          // loads of 16 steps issued together (see ltae_compose_kernel)
            float w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = Wi[(size_t)j * C + c + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += (double)dA[(size_t)h * C + c + q] * (double)w[q];
        }
        for (; c < C; ++c) s += (double)dA[(size_t)h * C + c] * (double)Wi[(size_t)j * C + c];
        for (int n = 0; n < NF; ++n) s += (double)LF_DB(h, n) * (double)bias1[(size_t)n * D + j];
        dWk[(size_t)hd * D + j] = (float)(s * q);
    }
    if (threadIdx.x == 0) dbk[hd] = (float)((double)sBv[h] * q);
}
// block j of D: d Wi[j][:], d bi[j]
__device__ __forceinline__ void ltae_compose_bwd_wi_body(int j, const float* __restrict__ Q, const float* __restrict__ Wk,
                                                         const float* __restrict__ dA, const float* __restrict__ sBv,
                                                         int NH, int DK, int D, int C, int NF, int T,
                                                         float* __restrict__ dWi, float* __restrict__ dbi) {
    const double sc = 1.0 / sqrt((double)DK);
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        int hd = 0;
        for (; hd + 16 <= NH * DK; hd += 16) {
            float w[16], a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { w[q] = Wk[(size_t)(hd + q) * D + j] * Q[hd + q]; a[q] = dA[(size_t)((hd + q) / DK) * C + c]; }
#pragma unroll
            for (int q = 0; q < 16; ++q) s += (double)w[q] * (double)a[q];
        }
        for (; hd < NH * DK; ++hd)
            s += (double)Wk[(size_t)hd * D + j] * (double)Q[hd] * (double)dA[(size_t)(hd / DK) * C + c];
        dWi[(size_t)j * C + c] = (float)(s * sc);
    }
    // d b_i[j] = sc * sum_hd Wk[hd][j] Q[hd] sB[h]: the block's threads take one hd each, fixed-order tree sum
    __shared__ double red[256];
    double v = 0.0;
    for (int hd = threadIdx.x; hd < NH * DK; hd += 256) v += (double)Wk[(size_t)hd * D + j] * (double)Q[hd] * (double)sBv[hd / DK];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbi[j] = (float)(red[0] * sc);
    (void)NF; (void)T;
}
// the two weight gradients depend on dA only: one launch, grid = HK + D + 1 (blocks [0, HK): d Wk / d bk, the next D: d Wi / d bi,
// the last one: d gamma / d beta = the heads' contributions summed in head order)
__global__ __launch_bounds__(256) void ltae_compose_bwd_w_kernel(const float* __restrict__ Q, const float* __restrict__ Wk,
                                                                 const float* __restrict__ Wi, const float* __restrict__ bias1,
                                                                 const float* __restrict__ dA, const float* __restrict__ dBp,
                                                                 const float* __restrict__ sBv, const float* __restrict__ dgb,
                                                                 int NH, int DK, int D, int C, int NF, int T,
                                                                 float* __restrict__ dWk, float* __restrict__ dbk,
                                                                 float* __restrict__ dWi, float* __restrict__ dbi,
                                                                 float* __restrict__ gb /* [2][C] */) {
    const int b = blockIdx.x, HK = NH * DK;       // block-uniform branch
    if (b < HK) ltae_compose_bwd_wk_body(b, NH, Q, Wi, bias1, dA, dBp, sBv, DK, D, C, NF, T, dWk, dbk);
    else if (b < HK + D) ltae_compose_bwd_wi_body(b - HK, Q, Wk, dA, sBv, NH, DK, D, C, NF, T, dWi, dbi);
    else
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            double acc = 0.0;
            for (int hh = 0; hh < NH; ++hh) acc += (double)dgb[(size_t)hh * 2 * C + i];
            gb[i] = (float)acc;
        }
}

extern "C" int uncr_ltae_fused_supported(int T, int C, int NH, int S) {
    if (!(NH == 4 || NH == 8 || NH == 16 || NH == 32) || C % NH || C > 256 || T < 1 || T > 16) return 0;
    const int cg = C / NH;
    if (!(cg == 2 || cg == 4 || cg == 8 || cg == 16) || T * cg > 128) return 0;      // T x CG values live in registers
    if (S % (256 / NH)) return 0;
    // backward LDS: A' + the pixels' score gradients + the four waves' partials
    const size_t lds = ((size_t)NH * C * 5 + (size_t)(256 / NH) * NH * T + 4 * (size_t)NH * T) * sizeof(float);
    return lds <= 64 * 1024 ? 1 : 0;
}

extern "C" int uncr_ltae_compose(const float* Q, const float* Wk, const float* bk, const float* Wi, const float* bin,
                                 const float* dates, const float* denom, int dpe, int use_pe, const float* gamma,
                                 const float* beta, int NH, int DK, int D, int C, int NF, float* bias1, float* Ap, float* Bp,
                                 float* M, float* U, hipStream_t stream) {
    if (NH <= 0 || DK <= 0 || D <= 0 || (D & 1) || C <= 0 || NF <= 0) return UNCR_ESHAPE;
    if (!Q || !Wk || !bk || !Wi || !gamma || !beta || !bias1 || !Ap || !Bp || !M || !U) return UNCR_EINVAL;
    if (use_pe && (!dates || !denom || dpe <= 0)) return UNCR_EINVAL;
    if (!use_pe) dpe = 1;
    const size_t ncomb = (size_t)(2 * DK * C > 1024 ? 2 * DK * C : 1024);
    const size_t lds = ncomb * sizeof(double) + ((size_t)NF * dpe + (size_t)DK * C + (size_t)DK * NF) * sizeof(float);
    if (lds > 150 * 1024) return UNCR_ESHAPE;       // B*T beyond ~1800 frames
    if (lds > 48 * 1024) {      // dynamic LDS above the default limit needs the opt-in (large batches: 80 B per frame)
        static size_t allowed = 0;
        if (lds > allowed) {
            if (hipFuncSetAttribute((const void*)ltae_compose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                return UNCR_ESHAPE;
            allowed = 150 * 1024;
        }
    }
    hipLaunchKernelGGL(ltae_compose_kernel, dim3(NH), dim3(1024), lds, stream, Q, Wk, bk, Wi, bin, dates, denom, dpe, use_pe, gamma,
                       beta, DK, D, C, NF, bias1, Ap, Bp, M, U);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

static size_t lf_lds(const LfArgs& g, bool bwd) {
    const int ppb = 256 / g.NH;
    size_t f = (size_t)g.NH * g.C;
    if (bwd) f += (size_t)ppb * g.NH * g.T + 4 * (size_t)g.NH * g.C + 4 * (size_t)g.NH * g.T;
    return f * sizeof(float);
}
template <int TMAX>
static int lf_launch(const LfArgs& g, bool bwd, hipStream_t stream) {
    const int ppb = 256 / g.NH;
    const dim3 grid((unsigned)((long long)g.B * g.S / ppb));
    const size_t lds = lf_lds(g, bwd);
#define LF_GO(CGV)                                                                                                   \
    do {                                                                                                             \
        if (bwd) hipLaunchKernelGGL((ltae_fused_bwd_kernel<TMAX, CGV>), grid, dim3(256), lds, stream, g);            \
        else hipLaunchKernelGGL((ltae_fused_fwd_kernel<TMAX, CGV>), grid, dim3(256), lds, stream, g);                \
    } while (0)
    switch (g.C / g.NH) {
        case 2: LF_GO(2); break;
        case 4: LF_GO(4); break;
        case 8: LF_GO(8); break;
        case 16: LF_GO(16); break;
        default: return UNCR_ESHAPE;
    }
#undef LF_GO
    return UNCR_OK;
}

extern "C" int uncr_ltae_fused_fwd(const float* x, const float* Ap, const float* Bp, const int* pad, float eps, float* att,
                                   float* mean, float* rstd, int B, int T, int C, int NH, int S, hipStream_t stream) {
    if (!uncr_ltae_fused_supported(T, C, NH, S) || B <= 0 || false) return UNCR_ESHAPE;
    if (!x || !Ap || !Bp || !att || !mean || !rstd) return UNCR_EINVAL;
    LfArgs g{x, Ap, Bp, pad, att, mean, rstd, nullptr, nullptr, nullptr, nullptr, B, T, C, NH, S, eps};
    const int rc = T <= 4 ? lf_launch<4>(g, false, stream) : (T <= 8 ? lf_launch<8>(g, false, stream) : lf_launch<16>(g, false, stream));
    if (rc) return rc;
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_fused_bwd(const float* datt, const float* att, const float* x, const float* Ap, const int* pad,
                                   const float* mean, const float* rstd, float* dx, float* partA, float* partB, int B, int T,
                                   int C, int NH, int S, hipStream_t stream) {
    if (!uncr_ltae_fused_supported(T, C, NH, S) || B <= 0 || false) return UNCR_ESHAPE;
    if (!datt || !att || !x || !Ap || !mean || !rstd || !dx || !partA || !partB) return UNCR_EINVAL;
    LfArgs g{x, Ap, nullptr, pad, const_cast<float*>(att), const_cast<float*>(mean), const_cast<float*>(rstd), datt, dx, partA,
             partB, B, T, C, NH, S, 0.f};
    const int rc = T <= 4 ? lf_launch<4>(g, true, stream) : (T <= 8 ? lf_launch<8>(g, true, stream) : lf_launch<16>(g, true, stream));
    if (rc) return rc;
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// partA [B*nblk][NH][C], partB [B*nblk][NH][T] (the fused backward's block partials of d A' and d B'; NF = B*T) -> dQ [NH][DK],
// dWk [HK][D], dbk [HK], dWi [D][C], dbi [D], gb [2][C] (d gamma, d beta); scratch: NH*C + NH + NF*NH + NH*2*C floats
extern "C" int uncr_ltae_compose_bwd(const float* Q, const float* Wk, const float* Wi, const float* bias1, const float* gamma,
                                     const float* beta, const float* M, const float* U, const float* partA, const float* partB,
                                     int nblk, int NH, int DK, int D, int C, int NF, int T, float* scratch, float* dQ, float* dWk,
                                     float* dbk, float* dWi, float* dbi, float* gb, hipStream_t stream) {
    if (NH <= 0 || DK <= 0 || D <= 0 || C <= 0 || C > 256 || NF <= 0 || T <= 0 || NF % T || nblk <= 0) return UNCR_ESHAPE;
    if (!Q || !Wk || !Wi || !bias1 || !gamma || !beta || !M || !U || !partA || !partB || !scratch || !dQ || !dWk || !dbk || !dWi ||
        !dbi || !gb)
        return UNCR_EINVAL;
    float* dA = scratch;                            // [NH][C]
    float* sB = dA + (size_t)NH * C;                // [NH]
    float* dBp = sB + NH;                           // [B][NH][T]
    float* dgb = dBp + (size_t)NF * NH;             // [NH][2][C]
    hipLaunchKernelGGL(ltae_compose_bwd_a_kernel, dim3(NH), dim3(1024), 0, stream, Q, M, U, gamma, beta, partA, partB, nblk, DK, C,
                       NF, T, dBp, dA, dQ, dgb, sB);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ltae_compose_bwd_w_kernel, dim3(NH * DK + D + 1), dim3(256), 0, stream, Q, Wk, Wi, bias1, dA, dBp, sB, dgb, NH,
                       DK, D, C, NF, T, dWk, dbk, dWi, dbi, gb);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
