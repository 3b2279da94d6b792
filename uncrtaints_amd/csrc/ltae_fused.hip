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

#define LF_PX 64       // pixels per block (one wave = the 64 pixels of one head group / channel-group quarter)

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
    float* partA;          // bwd [B*nblk][NH][C]
    float* partB;          // bwd [B*nblk][NH][T]
    int B, T, C, NH, S;
    float eps;
};

// ---- parameter composition ---------------------------------------------------------------------------------------------
// M[hd][c] = sum_j Wk[hd][j] Wi[j][c];  U[hd][n] = sum_j Wk[hd][j] bias1[n][j] + bk[hd]       grid = HK, block 256
__global__ __launch_bounds__(256) void ltae_compose_mu_kernel(const float* __restrict__ Wk, const float* __restrict__ bk,
                                                              const float* __restrict__ Wi, const float* __restrict__ bias1,
                                                              int D, int C, int NF, float* __restrict__ M,
                                                              float* __restrict__ U) {
    const int hd = blockIdx.x;
    const float* wk = Wk + (size_t)hd * D;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += (double)wk[j] * (double)Wi[(size_t)j * C + c];
        M[(size_t)hd * C + c] = (float)s;
    }
    for (int n = threadIdx.x; n < NF; n += 256) {
        double s = (double)bk[hd];
        for (int j = 0; j < D; ++j) s += (double)wk[j] * (double)bias1[(size_t)n * D + j];
        U[(size_t)hd * NF + n] = (float)s;
    }
}
// Ap[h][c] = gamma[c] * A[h][c], A = sc * sum_d Q[h][d] M[hd][c];  Bp[h][n] = sc * sum_d Q[h][d] U[hd][n] + sum_c A[h][c] beta[c]
__global__ __launch_bounds__(256) void ltae_compose_ab_kernel(const float* __restrict__ Q, const float* __restrict__ M,
                                                              const float* __restrict__ U, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int DK, int C, int NF,
                                                              float* __restrict__ Ap, float* __restrict__ Bp) {
    const int h = blockIdx.x;
    const double sc = 1.0 / sqrt((double)DK);
    __shared__ double red[256];
    double ab = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0.0;
        for (int d = 0; d < DK; ++d) a += (double)Q[h * DK + d] * (double)M[(size_t)(h * DK + d) * C + c];
        a *= sc;
        Ap[(size_t)h * C + c] = (float)(a * (double)gamma[c]);
        ab += a * (double)beta[c];
    }
    red[threadIdx.x] = ab;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double abt = red[0];
    for (int n = threadIdx.x; n < NF; n += 256) {
        double s = 0.0;
        for (int d = 0; d < DK; ++d) s += (double)Q[h * DK + d] * (double)U[(size_t)(h * DK + d) * NF + n];
        Bp[(size_t)h * NF + n] = (float)(s * sc + abt);
    }
}

// ---- fused forward -----------------------------------------------------------------------------------------------------
// grid = (S / 64, B), block 256: lane = pixel, wave w = heads [w*NH/4, (w+1)*NH/4)
template <int TMAX, int HPW>     // dates bound, heads per wave
__global__ __launch_bounds__(256) void ltae_fused_fwd_kernel(LfArgs g) {
    extern __shared__ float lds[];          // Ap [NH][C]
    const int sp = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, s = blockIdx.x * LF_PX + sp;
    const int T = g.T, C = g.C, NH = g.NH, S = g.S, Cg = C / NH;
    for (int i = threadIdx.x; i < NH * C; i += 256) lds[i] = g.Ap[i];
    __syncthreads();
    const int h0 = wv * HPW;
    float acc[HPW][TMAX];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[hh][t] = 0.f;
    const float* xb = g.x + (size_t)b * T * C * S + s;
    const float Mn = (float)(T * Cg);
    for (int grp = 0; grp < NH; ++grp) {            // GroupNorm groups = heads (ltae.py:191-194)
        float sum = 0.f;
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < Cg; ++j) sum += xb[((size_t)t * C + grp * Cg + j) * S];
        const float mu = sum / Mn;
        float var = 0.f;
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < Cg; ++j) {
                const float d = xb[((size_t)t * C + grp * Cg + j) * S] - mu;
                var = fmaf(d, d, var);
            }
        const float r = 1.0f / sqrtf(var / Mn + g.eps);
        if (wv == 0) {
            g.mean[((size_t)b * NH + grp) * S + s] = mu;
            g.rstd[((size_t)b * NH + grp) * S + s] = r;
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {
                for (int j = 0; j < Cg; ++j) {
                    const int c = grp * Cg + j;
                    const float xh = (xb[((size_t)t * C + c) * S] - mu) * r;
#pragma unroll
                    for (int hh = 0; hh < HPW; ++hh) acc[hh][t] = fmaf(lds[(h0 + hh) * C + c], xh, acc[hh][t]);
                }
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        const int h = h0 + hh;
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                float sc = acc[hh][t] + g.Bp[(size_t)h * g.B * T + b * T + t];
                if (g.pad && g.pad[b * T + t]) sc = -1e3f;          // masked_fill(pad, -1e3), ltae.py:435
                acc[hh][t] = sc;
                mx = fmaxf(mx, sc);
            }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) { acc[hh][t] = expf(acc[hh][t] - mx); den += acc[hh][t]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) g.att[(((size_t)h * g.B + b) * T + t) * S + s] = acc[hh][t] * inv;
    }
}

// ---- fused backward ----------------------------------------------------------------------------------------------------
// grid = (S / 64, B), block 256.  Phase 1 (wave = head quarter): softmax backward -> ds[h][t][px] in LDS + partial d B'.
// Phase 2 (wave = GroupNorm-group quarter): d xhat = A'^T ds, GroupNorm backward -> dx, partial d A' (wave sums over the pixels).
template <int TMAX, int HPW>
__global__ __launch_bounds__(256) void ltae_fused_bwd_kernel(LfArgs g) {
    extern __shared__ float lds[];
    const int sp = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, s = blockIdx.x * LF_PX + sp;
    const int T = g.T, C = g.C, NH = g.NH, S = g.S, Cg = C / NH;
    float* Ap = lds;                       // [NH][C]
    float* ds = lds + NH * C;              // [NH][T][64]
    for (int i = threadIdx.x; i < NH * C; i += 256) Ap[i] = g.Ap[i];
    const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
    {
        const int h0 = wv * HPW;
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int h = h0 + hh;
            float a[TMAX], d[TMAX];
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
                    const size_t o = (((size_t)h * g.B + b) * T + t) * S + s;
                    a[t] = g.att[o];
                    d[t] = g.datt[o];
                    dot = fmaf(a[t], d[t], dot);
                }
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
                    float v = a[t] * (d[t] - dot);
                    if (g.pad && g.pad[b * T + t]) v = 0.f;          // a padded date's score is the constant -1e3
                    ds[(h * T + t) * LF_PX + sp] = v;
                    const float sv = wave_sum_dpp(v);                 // over the block's 64 pixels (lane 63 holds the sum)
                    if (sp == 63) g.partB[(blk * NH + h) * T + t] = sv;
                }
        }
    }
    __syncthreads();
    const float Mn = (float)(T * Cg);
    const float* xb = g.x + (size_t)b * T * C * S + s;
    float* dxb = g.dx + (size_t)b * T * C * S + s;
    const int gpw = NH / 4;                 // GroupNorm groups per wave (NH % 4 == 0)
    for (int gi = 0; gi < gpw; ++gi) {
        const int grp = wv * gpw + gi;
        const float mu = g.mean[((size_t)b * NH + grp) * S + s], r = g.rstd[((size_t)b * NH + grp) * S + s];
        float m1 = 0.f, m2 = 0.f;
        for (int j = 0; j < Cg; ++j) {
            const int c = grp * Cg + j;
            float da[32];                  // d A'[h][c] contributions of this pixel, NH <= 32
#pragma unroll
            for (int h = 0; h < 32; ++h) da[h] = 0.f;
            for (int t = 0; t < T; ++t) {
                const float xh = (xb[((size_t)t * C + c) * S] - mu) * r;
                float dxh = 0.f;
#pragma unroll
                for (int h = 0; h < 32; ++h)
                    if (h < NH) {
                        const float dv = ds[(h * T + t) * LF_PX + sp];
                        dxh = fmaf(Ap[h * C + c], dv, dxh);
                        da[h] = fmaf(dv, xh, da[h]);
                    }
                m1 += dxh;
                m2 = fmaf(dxh, xh, m2);
            }
#pragma unroll
            for (int h = 0; h < 32; ++h)
                if (h < NH) {
                    const float sv = wave_sum_dpp(da[h]);
                    if (sp == 63) g.partA[(blk * NH + h) * C + c] = sv;
                }
        }
        m1 /= Mn;
        m2 /= Mn;
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < Cg; ++j) {
                const int c = grp * Cg + j;
                const float xh = (xb[((size_t)t * C + c) * S] - mu) * r;
                float dxh = 0.f;
#pragma unroll
                for (int h = 0; h < 32; ++h)
                    if (h < NH) dxh = fmaf(Ap[h * C + c], ds[(h * T + t) * LF_PX + sp], dxh);
                dxb[((size_t)t * C + c) * S] = r * (dxh - m1 - xh * m2);
            }
    }
}

// ---- gradients of the parameters from d A' [NH][C] and d B' ([B][NH][T], as the block partials reduce) --------------------
#define LF_DB(h, n) dBp[((size_t)((n) / T) * NH + (h)) * T + (n) % T]
// grid = NH: d A[h][:] (gamma / beta folded back), d Q[h][:], per-head contributions to d gamma / d beta (colsum over h later)
__global__ __launch_bounds__(256) void ltae_compose_bwd_a_kernel(const float* __restrict__ Q, const float* __restrict__ M,
                                                                 const float* __restrict__ U, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ dAp,
                                                                 const float* __restrict__ dBp, int DK, int C, int NF, int T,
                                                                 float* __restrict__ dA, float* __restrict__ dQ,
                                                                 float* __restrict__ dgb /* [NH][2][C] */) {
    const int h = blockIdx.x, NH = gridDim.x;
    const double sc = 1.0 / sqrt((double)DK);
    __shared__ double red[256];
    __shared__ double sB;
    double v = 0.0;
    for (int n = threadIdx.x; n < NF; n += 256) v += (double)LF_DB(h, n);
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sB = red[0];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0.0;
        for (int d = 0; d < DK; ++d) a += (double)Q[h * DK + d] * (double)M[(size_t)(h * DK + d) * C + c];
        a *= sc;                                                        // A[h][c]
        const double dap = (double)dAp[(size_t)h * C + c];
        dA[(size_t)h * C + c] = (float)(dap * (double)gamma[c] + sB * (double)beta[c]);
        dgb[((size_t)h * 2 + 0) * C + c] = (float)(dap * a);            // d gamma contribution
        dgb[((size_t)h * 2 + 1) * C + c] = (float)(a * sB);             // d beta contribution
    }
    __syncthreads();
    for (int d = 0; d < DK; ++d) {
        double q = 0.0;
        for (int c = threadIdx.x; c < C; c += 256)
            q += ((double)dAp[(size_t)h * C + c] * (double)gamma[c] + sB * (double)beta[c]) * (double)M[(size_t)(h * DK + d) * C + c];
        for (int n = threadIdx.x; n < NF; n += 256) q += (double)LF_DB(h, n) * (double)U[(size_t)(h * DK + d) * NF + n];
        red[threadIdx.x] = q;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) dQ[h * DK + d] = (float)(red[0] * sc);
        __syncthreads();
    }
}
// grid = HK: d Wk[hd][:], d bk[hd]          (d M[hd][c] = sc Q[h][d] dA[h][c],  d U[hd][n] = sc Q[h][d] dB'[h][n])
__global__ __launch_bounds__(256) void ltae_compose_bwd_wk_kernel(const float* __restrict__ Q, const float* __restrict__ Wi,
                                                                  const float* __restrict__ bias1, const float* __restrict__ dA,
                                                                  const float* __restrict__ dBp, int DK, int D, int C, int NF,
                                                                  int T, float* __restrict__ dWk, float* __restrict__ dbk) {
    const int hd = blockIdx.x, h = hd / DK, NH = gridDim.x / DK;
    const double q = (double)Q[hd] / sqrt((double)DK);
    for (int j = threadIdx.x; j < D; j += 256) {
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += (double)dA[(size_t)h * C + c] * (double)Wi[(size_t)j * C + c];
        for (int n = 0; n < NF; ++n) s += (double)LF_DB(h, n) * (double)bias1[(size_t)n * D + j];
        dWk[(size_t)hd * D + j] = (float)(s * q);
    }
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int n = 0; n < NF; ++n) s += (double)LF_DB(h, n);
        dbk[hd] = (float)(s * q);
    }
}
// grid = D: d Wi[j][:], d bi[j]
__global__ __launch_bounds__(256) void ltae_compose_bwd_wi_kernel(const float* __restrict__ Q, const float* __restrict__ Wk,
                                                                  const float* __restrict__ dA, const float* __restrict__ dBp,
                                                                  int NH, int DK, int D, int C, int NF, int T,
                                                                  float* __restrict__ dWi, float* __restrict__ dbi) {
    const int j = blockIdx.x;
    const double sc = 1.0 / sqrt((double)DK);
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0;
        for (int hd = 0; hd < NH * DK; ++hd)
            s += (double)Wk[(size_t)hd * D + j] * (double)Q[hd] * (double)dA[(size_t)(hd / DK) * C + c];
        dWi[(size_t)j * C + c] = (float)(s * sc);
    }
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int h = 0; h < NH; ++h) {
            double sb = 0.0;
            for (int n = 0; n < NF; ++n) sb += (double)LF_DB(h, n);
            for (int d = 0; d < DK; ++d) s += (double)Wk[(size_t)(h * DK + d) * D + j] * (double)Q[h * DK + d] * sb;
        }
        dbi[j] = (float)(s * sc);
    }
}

extern "C" int uncr_ltae_fused_supported(int T, int C, int NH, int S) {
    return (T >= 1 && T <= 16 && NH >= 4 && NH <= 32 && NH % 4 == 0 && C % NH == 0 && C <= 256 && S % LF_PX == 0) ? 1 : 0;
}

extern "C" int uncr_ltae_compose(const float* Q, const float* Wk, const float* bk, const float* Wi, const float* bias1,
                                 const float* gamma, const float* beta, int NH, int DK, int D, int C, int NF, float* Ap,
                                 float* Bp, float* M, float* U, hipStream_t stream) {
    if (NH <= 0 || DK <= 0 || D <= 0 || C <= 0 || NF <= 0) return UNCR_ESHAPE;
    if (!Q || !Wk || !bk || !Wi || !bias1 || !gamma || !beta || !Ap || !Bp || !M || !U) return UNCR_EINVAL;
    hipLaunchKernelGGL(ltae_compose_mu_kernel, dim3(NH * DK), dim3(256), 0, stream, Wk, bk, Wi, bias1, D, C, NF, M, U);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ltae_compose_ab_kernel, dim3(NH), dim3(256), 0, stream, Q, M, U, gamma, beta, DK, C, NF, Ap, Bp);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

template <int TMAX>
static void lf_launch(const LfArgs& g, bool bwd, hipStream_t stream) {
    const dim3 grid(g.S / LF_PX, g.B);
    const size_t lds = (size_t)g.NH * g.C * sizeof(float) + (bwd ? (size_t)g.NH * g.T * LF_PX * sizeof(float) : 0);
#define LF_GO(HPW)                                                                                                   \
    do {                                                                                                             \
        if (bwd) hipLaunchKernelGGL((ltae_fused_bwd_kernel<TMAX, HPW>), grid, dim3(256), lds, stream, g);            \
        else hipLaunchKernelGGL((ltae_fused_fwd_kernel<TMAX, HPW>), grid, dim3(256), lds, stream, g);                \
    } while (0)
    switch (g.NH / 4) {
        case 1: LF_GO(1); break;
        case 2: LF_GO(2); break;
        case 4: LF_GO(4); break;
        default: LF_GO(8); break;
    }
#undef LF_GO
}

extern "C" int uncr_ltae_fused_fwd(const float* x, const float* Ap, const float* Bp, const int* pad, float eps, float* att,
                                   float* mean, float* rstd, int B, int T, int C, int NH, int S, hipStream_t stream) {
    if (!uncr_ltae_fused_supported(T, C, NH, S) || B <= 0 || (NH != 4 && NH != 8 && NH != 16 && NH != 32)) return UNCR_ESHAPE;
    if (!x || !Ap || !Bp || !att || !mean || !rstd) return UNCR_EINVAL;
    LfArgs g{x, Ap, Bp, pad, att, mean, rstd, nullptr, nullptr, nullptr, nullptr, B, T, C, NH, S, eps};
    if (T <= 4) lf_launch<4>(g, false, stream);
    else if (T <= 8) lf_launch<8>(g, false, stream);
    else lf_launch<16>(g, false, stream);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_ltae_fused_bwd(const float* datt, const float* att, const float* x, const float* Ap, const int* pad,
                                   const float* mean, const float* rstd, float* dx, float* partA, float* partB, int B, int T,
                                   int C, int NH, int S, hipStream_t stream) {
    if (!uncr_ltae_fused_supported(T, C, NH, S) || B <= 0 || (NH != 4 && NH != 8 && NH != 16 && NH != 32)) return UNCR_ESHAPE;
    if (!datt || !att || !x || !Ap || !mean || !rstd || !dx || !partA || !partB) return UNCR_EINVAL;
    if ((size_t)NH * C * 4 + (size_t)NH * T * LF_PX * 4 > 64 * 1024) return UNCR_ESHAPE;
    LfArgs g{x, Ap, nullptr, pad, const_cast<float*>(att), const_cast<float*>(mean), const_cast<float*>(rstd), datt, dx, partA,
             partB, B, T, C, NH, S, 0.f};
    if (T <= 4) lf_launch<4>(g, true, stream);
    else if (T <= 8) lf_launch<8>(g, true, stream);
    else lf_launch<16>(g, true, stream);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// dAp [NH][C], dBp [B][NH][T] (NF = B*T) -> dQ [NH][DK], dWk [HK][D], dbk [HK], dWi [D][C], dbi [D], dgb [NH][2][C] (per-head d gamma / d beta
// contributions: sum over h with uncr_colsum), scratch dA [NH][C]
extern "C" int uncr_ltae_compose_bwd(const float* Q, const float* Wk, const float* Wi, const float* bias1, const float* gamma,
                                     const float* beta, const float* M, const float* U, const float* dAp, const float* dBp,
                                     int NH, int DK, int D, int C, int NF, int T, float* dA, float* dQ, float* dWk,
                                     float* dbk, float* dWi, float* dbi, float* dgb, hipStream_t stream) {
    if (NH <= 0 || DK <= 0 || D <= 0 || C <= 0 || NF <= 0 || T <= 0 || NF % T) return UNCR_ESHAPE;
    if (!Q || !Wk || !Wi || !bias1 || !gamma || !beta || !M || !U || !dAp || !dBp || !dA || !dQ || !dWk || !dbk || !dWi || !dbi || !dgb)
        return UNCR_EINVAL;
    hipLaunchKernelGGL(ltae_compose_bwd_a_kernel, dim3(NH), dim3(256), 0, stream, Q, M, U, gamma, beta, dAp, dBp, DK, C, NF, T, dA,
                       dQ, dgb);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ltae_compose_bwd_wk_kernel, dim3(NH * DK), dim3(256), 0, stream, Q, Wi, bias1, dA, dBp, DK, D, C, NF, T, dWk,
                       dbk);
    UNCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ltae_compose_bwd_wi_kernel, dim3(D), dim3(256), 0, stream, Q, Wk, dA, dBp, NH, DK, D, C, NF, T, dWi, dbi);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
