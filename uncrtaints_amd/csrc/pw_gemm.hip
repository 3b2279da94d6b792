// 1x1 ("pointwise") convolutions over NCHW planes as fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: exact
// fp32 fma chain, 157 TF peak on gfx950).
//
//   out[n, co, p] = sum_ci Wt[ci, co] * f(in[n, ci, p])  (+ bias)
//
// where f is the fused prologue (norm-apply / GELU / SE scale / norm-backward affine) and the epilogue
// optionally emits per-(n,co) partial statistics for the NEXT normalisation (sum, sum^2) or for a
// norm backward (sum out, sum out*aux).  Layout decisions (MI355X):
//   * pixels are the contiguous axis (NCHW), so a wave's MFMA B operand (k x 32 px) is a 128-B row
//     segment; each lane carries FOUR consecutive pixels (float4) = four interleaved 32-px MFMA
//     column tiles, so every global/LDS access on the activation stream is 16 B per lane and the
//     accumulator epilogue stores float4 rows (no LDS transpose anywhere).
//   * the activation chunk [32 k][TP px] is staged ONCE per block in LDS with the prologue applied
//     (GELU/erf is evaluated once per element, not once per consuming wave); the next chunk is
//     prefetched into registers while the current one feeds the MFMAs.
//   * weights are tiny (<=128 KB, L2 resident): each wave reads its A fragments (Wt[k][co], co
//     contiguous => 128-B coalesced) straight from global.
#include "pw_gemm.h"
#include <cstdlib>


// PRE2 = false drops the second prefetch register set (PRO_NORMBWD unavailable): keeps the 32-wide
// variant (16 prefetch float4 per lane) free of spills.
// KCV: rows of the contraction axis per staged chunk (32; 16 halves the LDS image -- 32 KB per block at TP = 256 -- so that four blocks
// instead of two share a CU: the head's 128 -> 26 GEMM has one wave per SIMD otherwise and waits for HBM at every chunk)
template <int CT, int WN, int WM, bool PRE2, typename TI = float, typename TO = float, int KCV = 32>
__global__ __launch_bounds__(64 * WN * WM, KCV == 32 ? 2 : 4) void pw_gemm_kernel(PwArgs g) {
    constexpr int NT = 64 * WN * WM;
    constexpr int TP = 128 * WM;
    constexpr int COUTP = 32 * CT * WN;
    constexpr int KC = KCV;
    constexpr int KS = KC / 2;                    // MFMA k-steps per chunk (32x32x2)
    constexpr int NL = (KC * TP / 4) / NT;        // float4 loads per thread per chunk
    constexpr int ROWS_PER_I = NT / (TP / 4);     // rows covered per load index

    __shared__ __attribute__((aligned(16))) float xs[2][KC][TP];   // double-buffered activation chunk
    __shared__ float cf[4][256];      // [3]: the norm's mean (centred PRO_NORMBWD)
    __shared__ float red[WM][COUTP][2];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int n = blockIdx.y;
    const int px0 = blockIdx.x * TP;
    const int Cin = g.Cin, Cout = g.Cout, P = g.P;
    const int nk = (Cin + KC - 1) / KC;
    const int pro = g.pro;

    // prologue coefficients for this frame -> LDS
    for (int i = tid; i < Cin; i += NT) {
        cf[0][i] = g.k0 ? g.k0[n * Cin + i] : 1.f;
        cf[1][i] = g.k1 ? g.k1[n * Cin + i] : 0.f;
        cf[2][i] = g.k2 ? g.k2[n * Cin + i] : (pro == PRO_AFFINE_GELU ? 1.f : 0.f);
        cf[3][i] = g.k3 ? g.k3[n * Cin + i] : 0.f;
    }

    const int lrow = tid / (TP / 4), lc4 = tid % (TP / 4);
    const TI* inb = (const TI*)g.in + (size_t)n * Cin * P + px0 + 4 * lc4;
    const TI* in2b = g.in2 ? (const TI*)g.in2 + (size_t)n * Cin * P + px0 + 4 * lc4 : nullptr;

    // Software pipeline: while the MFMAs of chunk kc run from xs[kc&1], the register-prefetched chunk kc+1 is
    // transformed and written to xs[(kc+1)&1], chunk kc+2 is fetched into the registers just freed, and the
    // A fragments of chunk kc+1 are re-loaded into the registers the MFMAs have just consumed.  One barrier
    // per chunk; no global latency is exposed inside the loop.
    float4 pre[NL], pre2[PRE2 ? NL : 1];
    auto load_piece = [&](int i, int kc) {
        const int k = kc * KC + lrow + i * ROWS_PER_I;
        const int kk = k < Cin ? k : 0;                   // out-of-range rows re-read row 0 and are zeroed at staging
        pre[i] = ld4<TI>(inb + (size_t)kk * P);
        if constexpr (PRE2) pre2[i] = ld4<TI>((pro == PRO_NORMBWD ? in2b : inb) + (size_t)kk * P);
    };
    auto stage_piece = [&](int i, int kc, int buf) {
        const int r = lrow + i * ROWS_PER_I;
        const int k = kc * KC + r;
        const int kk = k < Cin ? k : 0;
        float4 v = pre[i];
        const float c0 = cf[0][kk], c1 = cf[1][kk], c2 = cf[2][kk];
        float* pv = (float*)&v;
        if (pro == PRO_AFFINE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = fmaf(c0, pv[j], c1);
        } else if (pro == PRO_AFFINE_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = c2 * gelu_f(fmaf(c0, pv[j], c1));
        } else if (pro == PRO_NORMBWD) {
            if constexpr (PRE2) {
                const float* p2 = (const float*)&pre2[i];
                const float c3 = cf[3][kk];
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[j] = fmaf(c0, pv[j], fmaf(c1, p2[j] - c3, c2));
            }
        } else if (pro == PRO_AFFINE_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = fmaxf(fmaf(c0, pv[j], c1), 0.f);
        }
        if (k >= Cin) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&xs[buf][r][4 * lc4] = v;
    };

    f32x16 acc[4][CT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][ct][r] = 0.f;

    const float* wbase = g.Wt + (size_t)(lane >> 5) * COUTP + wn * (CT * 32) + (lane & 31);
    float afr[KS][CT];
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(i, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) afr[s][ct] = wbase[(size_t)(2 * s) * COUTP + ct * 32];
    __syncthreads();   // cf visible
#pragma unroll
    for (int i = 0; i < NL; ++i) stage_piece(i, 0, 0);
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(i, nk > 1 ? 1 : 0);
    __syncthreads();

    constexpr int SLOT = KS / NL >= 1 ? KS / NL : 1;   // MFMA k-steps between two staging pieces
    static_assert(NL <= KS, "every staging piece of the next chunk needs its own k-step");
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        // successors are clamped instead of branched on: memory operations under (even uniform) branches make the
        // compiler lose count of the outstanding loads and fall back to s_waitcnt vmcnt(0), which drains the
        // just-issued HBM prefetch at every staging slot.  The redundant work at the tail is harmless.
        const int k1 = kc + 1 < nk ? kc + 1 : nk - 1, k2 = kc + 2 < nk ? kc + 2 : nk - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 b = *(const float4*)&xs[cur][2 * s + (lane >> 5)][wm * 128 + 4 * (lane & 31)];
            const float* pb = (const float*)&b;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[e][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s][ct], pb[e], acc[e][ct], 0, 0, 0);
            // rolling A prefetch into the registers this k-step has just consumed
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) afr[s][ct] = wbase[(size_t)(k1 * KC + 2 * s) * COUTP + ct * 32];
            // one staging piece of the next chunk every SLOT k-steps (NL pieces per chunk)
            if constexpr (NL <= KS) {
                if (s % SLOT == SLOT - 1 && s / SLOT < NL) {
                    const int i = s / SLOT;
                    stage_piece(i, k1, cur ^ 1);
                    load_piece(i, k2);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, store (float4 along pixels), statistics ----
    const int epi = g.epi;
    const int pxw = px0 + wm * 128 + 4 * (lane & 31);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = wn * (CT * 32) + ct * 32 + row;   // local output channel
            float4 v = make_float4(acc[0][ct][r], acc[1][ct][r], acc[2][ct][r], acc[3][ct][r]);
            float s0 = 0.f, s1 = 0.f;
            if (col < Cout) {
                if (g.bias) {
                    const float bb = g.bias[(size_t)n * g.bias_stride_n + col];
                    v.x += bb; v.y += bb; v.z += bb; v.w += bb;
                }
                const size_t o = ((size_t)n * Cout + col) * P + pxw;
                if (epi == 3) {
                    // du2 = gelu'(A*h2 + B) * (S*dz + D): the SE / GELU backward applied to the fresh accumulator
                    const float4 x = ld4<TO>((const TO*)g.aux + o);
                    const int ci = n * Cout + col;
                    const float A = g.e0[ci], B = g.e1[ci], S = g.e2[ci], D = g.e3[ci];
                    v.x = gelu_grad_f(fmaf(A, x.x, B)) * fmaf(S, v.x, D);
                    v.y = gelu_grad_f(fmaf(A, x.y, B)) * fmaf(S, v.y, D);
                    v.z = gelu_grad_f(fmaf(A, x.z, B)) * fmaf(S, v.z, D);
                    v.w = gelu_grad_f(fmaf(A, x.w, B)) * fmaf(S, v.w, D);
                    v = rnd4<TO>(v);
                    s0 = v.x + v.y + v.z + v.w;
                    s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                } else if (epi == 7) {
                    // out_conv's nonlinearities (uncrtaints.py:441-445): mean channels scale*sigmoid (or identity),
                    // variance channels softplus / elu+1 / identity (+ eps): the pre-activation never reaches HBM
                    const int nm = g.head_nm < 0 ? -g.head_nm : g.head_nm;
                    if (g.head_pre) *(float4*)(g.head_pre + o) = v;     // pre-activation for the backward (exact derivatives)
                    float* pv = (float*)&v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = pv[i];
                        if (col < nm) pv[i] = g.head_nm > 0 ? g.head_scale * sigmoid_f(x) : x;
                        else if (g.head_var == 0) pv[i] = (x > 20.f ? x : log1pf(expf(x))) + g.head_eps;
                        else if (g.head_var == 1) pv[i] = (x > 0.f ? x : expm1f(x)) + 1.f + g.head_eps;
                    }
                }
                v = rnd4<TO>(v);      // statistics of the values as stored
                st4<TO>((TO*)g.out + o, v);
                if (epi == 1) {
                    s0 = v.x + v.y + v.z + v.w;
                    s1 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                } else if (epi == 2) {
                    const float4 x = ld4<TO>((const TO*)g.aux + o);
                    s0 = v.x + v.y + v.z + v.w;
                    s1 = v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
                }
            }
            if (epi && epi != 7) {   // block-uniform
                s0 = half_wave_sum_dpp(s0);
                s1 = half_wave_sum_dpp(s1);
                if ((lane & 31) == 31) { red[wm][col][0] = s0; red[wm][col][1] = s1; }
            }
        }
    }
    if (epi && epi != 7) {
        __syncthreads();
        for (int c = tid; c < COUTP; c += NT) {
            if (c < Cout) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m) { s0 += red[m][c][0]; s1 += red[m][c][1]; }
                if (g.Pv > 0 && px0 + TP > g.Pv) s0 = s1 = 0.f;      // a tile that reaches into an any-size plane's tail: uncr_fix_tail
                g.part[((size_t)n * Cout + c) * gridDim.x + blockIdx.x] = make_float2(s0, s1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[co, ci] = sum_p fD(d[n,co,p]) * fX(x[n,ci,p])   (reduction over pixels)
// One block reduces PXB pixels of one frame into a full [COP][CIP] partial (accumulators live in
// the MFMA result registers across the whole pixel loop); a second kernel reduces the partials in
// fp64 in a fixed order.  Both operands are staged through LDS with their prologues applied.
// The reduction axis (pixels) is the contiguous one for BOTH operands, so a lane's float4 feeds
// four k-steps.
// ---------------------------------------------------------------------------------------------
struct WgArgs {
    const void* d;       // operands: fp32 or bf16 (kernel template TD / TX)
    const void* d2;
    const void* x;
    const void* x2;
    const float* dk0; const float* dk1; const float* dk2;   // [N*Cd]
    const float* xk0; const float* xk1; const float* xk2;   // [N*Cx]
    float* part;       // [N*NBX][COP][CIP]
    float* rs_part;    // [N*NBX][COP] row sums of fD(d) (bias gradient) or null
    int Cd, Cx, P, PXB;
    int pro_d, pro_x;
    const float* dk3 = nullptr;     // PRO_NORMBWD on d: the norm's mean per (n, co) (centred form), null: 0
    int Pv = 0;                     // > 0: padded planes of an any-size image -- only the whole 32-pixel chunks below Pv are summed here
                                    // (uncr_wgrad_boundary adds the last Pv % 32 pixels); 0: all P
};

__device__ __forceinline__ float4 apply_pro(int pro, float4 v, const float4& v2, float c0, float c1, float c2, float c3 = 0.f) {
    float* pv = (float*)&v;
    const float* p2 = (const float*)&v2;
    if (pro == PRO_AFFINE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = fmaf(c0, pv[j], c1);
    } else if (pro == PRO_AFFINE_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = c2 * gelu_f(fmaf(c0, pv[j], c1));
    } else if (pro == PRO_NORMBWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = fmaf(c0, pv[j], fmaf(c1, p2[j] - c3, c2));
    } else if (pro == PRO_AFFINE_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = fmaxf(fmaf(c0, pv[j], c1), 0.f);
    }
    return v;
}

// D2 = true keeps a second register set for the PRO_NORMBWD operand of D.
template <int MT, int NTL, int WCO, int WCI, bool D2, typename TD = float, typename TX = float>
__global__ __launch_bounds__(64 * WCO * WCI, 2) void pw_wgrad_kernel(WgArgs g) {
    constexpr int NT = 64 * WCO * WCI;
    constexpr int COP = 32 * MT * WCO;
    constexpr int CIP = 32 * NTL * WCI;
    constexpr int KP = 32, PITCH = 36;
    constexpr int ND = COP * 8 / NT;   // float4 per thread per chunk, D operand
    constexpr int NX = CIP * 8 / NT;   // X operand
    static_assert(COP * 8 % NT == 0 && CIP * 8 % NT == 0, "loader mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ds = smem;                   // [COP][PITCH]
    float* xs = smem + COP * PITCH;     // [CIP][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wco = wv / WCI, wci = wv % WCI;
    const int n = blockIdx.y;
    const int P = g.P, Cd = g.Cd, Cx = g.Cx;
    const int pbeg = blockIdx.x * g.PXB;
    const int pend = g.Pv > 0 ? min(pbeg + g.PXB, g.Pv / KP * KP) : pbeg + g.PXB;
    const int nchunks = pend > pbeg ? (pend - pbeg) / KP : 0;      // (0: a block that lies in the tail writes a zero partial)
    const int pro_d = g.pro_d, pro_x = g.pro_x;

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTL; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float rowsum = 0.f;

    // loader mapping: float4 index f = tid + i*NT -> (row = f >> 3, c4 = f & 7); rows advance by NT/8 per i
    const int lrow = tid >> 3, lc4 = tid & 7;
    constexpr int RSTEP = NT / 8;
    const TD* dbase = (const TD*)g.d + (size_t)n * Cd * P + 4 * lc4;
    const TD* d2base = g.d2 ? (const TD*)g.d2 + (size_t)n * Cd * P + 4 * lc4 : nullptr;
    const TX* xbase = (const TX*)g.x + (size_t)n * Cx * P + 4 * lc4;

    // per-row prologue coefficients of this frame -> LDS (chunk-invariant)
    float* cfd = smem + (COP + CIP) * PITCH;    // [4][COP]
    float* cfx = cfd + 4 * COP;                 // [3][CIP]
    for (int i = tid; i < COP; i += NT) {
        const int ci = n * Cd + (i < Cd ? i : 0);
        cfd[i] = g.dk0 ? g.dk0[ci] : 1.f;
        cfd[COP + i] = g.dk1 ? g.dk1[ci] : 0.f;
        cfd[2 * COP + i] = g.dk2 ? g.dk2[ci] : (pro_d == PRO_AFFINE_GELU ? 1.f : 0.f);
        cfd[3 * COP + i] = g.dk3 ? g.dk3[ci] : 0.f;
    }
    for (int i = tid; i < CIP; i += NT) {
        const int ci = n * Cx + (i < Cx ? i : 0);
        cfx[i] = g.xk0 ? g.xk0[ci] : 1.f;
        cfx[CIP + i] = g.xk1 ? g.xk1[ci] : 0.f;
        cfx[2 * CIP + i] = g.xk2 ? g.xk2[ci] : (pro_x == PRO_AFFINE_GELU ? 1.f : 0.f);
    }
    __syncthreads();

    // Register-level software pipeline: the raw tiles of chunk ch+1 are requested right after chunk ch has been staged, so their
    // HBM latency runs under chunk ch's MFMA phase (one exposed round trip per chunk before: the narrow in_conv shape, 128 x 15 at
    // N = 12, sat at 2.0 TB/s).  Shapes whose raw tiles exceed 12 float4 per lane keep the two-step order (register budget).
    constexpr bool SPLIT = (ND * (D2 ? 2 : 1) + NX) > 12;
    float4 dv[ND], dv2[D2 ? ND : 1], xv[NX];
    auto issue_d = [&](int ch) {
        const size_t p0 = (size_t)pbeg + (size_t)ch * KP;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int row = lrow + i * RSTEP;
            const int rr = row < Cd ? row : 0;            // rows past Cd re-read row 0 (branch-free) and are zeroed at staging
            dv[i] = ld4<TD>(dbase + (size_t)rr * P + p0);
            if constexpr (D2) {
                if (pro_d == PRO_NORMBWD) dv2[i] = ld4<TD>(d2base + (size_t)rr * P + p0);
            }
        }
    };
    auto issue_x = [&](int ch) {
        const size_t p0 = (size_t)pbeg + (size_t)ch * KP;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = lrow + i * RSTEP;
            xv[i] = ld4<TX>(xbase + (size_t)(row < Cx ? row : 0) * P + p0);
        }
    };
    auto stage_d = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int row = lrow + i * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Cd) v = apply_pro(pro_d, dv[i], D2 ? dv2[D2 ? i : 0] : dv[i], cfd[row], cfd[COP + row], cfd[2 * COP + row], cfd[3 * COP + row]);
            *(float4*)&ds[row * PITCH + 4 * lc4] = v;
        }
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = lrow + i * RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Cx) v = apply_pro(pro_x, xv[i], xv[i], cfx[row], cfx[CIP + row], cfx[2 * CIP + row]);
            *(float4*)&xs[row * PITCH + 4 * lc4] = v;
        }
    };

    if constexpr (!SPLIT) { issue_d(0); issue_x(0); }
    for (int ch = 0; ch < nchunks; ++ch) {
        if constexpr (SPLIT) {
            // all global loads of an operand tile are issued back-to-back (two latency exposures per chunk; the other resident
            // block's MFMA phase covers them), then transformed and staged
            issue_d(ch);
            stage_d();
            issue_x(ch);
            stage_x();
        } else {
            stage_d();
            stage_x();
            const int nx = ch + 1 < nchunks ? ch + 1 : ch;      // clamped: the last prefetch re-reads the last chunk
            issue_d(nx);
            issue_x(nx);
        }
        __syncthreads();
        if (g.rs_part && tid < COP) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KP; ++j) s += ds[tid * PITCH + j];
            rowsum += s;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 a4[MT], b4[NTL];
#pragma unroll
            for (int a = 0; a < MT; ++a)
                a4[a] = *(const float4*)&ds[((wco * MT + a) * 32 + (lane & 31)) * PITCH + 8 * s + 4 * (lane >> 5)];
#pragma unroll
            for (int b = 0; b < NTL; ++b)
                b4[b] = *(const float4*)&xs[((wci * NTL + b) * 32 + (lane & 31)) * PITCH + 8 * s + 4 * (lane >> 5)];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NTL; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&a4[a])[e],
                                                                          ((const float*)&b4[b])[e], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    const size_t blk = (size_t)n * gridDim.x + blockIdx.x;
    float* po = g.part + blk * COP * CIP;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTL; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int co = (wco * MT + a) * 32 + row;
                const int ci = (wci * NTL + b) * 32 + (lane & 31);
                po[co * CIP + ci] = acc[a][b][r];
            }
    if (g.rs_part && tid < COP) g.rs_part[blk * COP + tid] = rowsum;
}

// Reduce partials: out[f][co][ci] = sum over the blocks of frame-group f.  frames_per_out = N gives the
// total gradient, 1 gives per-frame products (needed for the SE gradient).  fp64, fixed order.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int nblk_per_out,
                                                            int COP, int CIP, int Cout, int Cin,
                                                            float* __restrict__ out) {
    // block = 256 elements x 4 slices of the partial range (latency-bound: more loads in flight), fixed combination order
    __shared__ double comb[4][256];
    const int el = threadIdx.x & 255, sl = threadIdx.x >> 8;
    const int idx = blockIdx.x * 256 + el;
    const size_t S = (size_t)COP * CIP;
    const bool live = idx < COP * CIP;
    double s = 0.0;
    if (live) {
        const float* src = part + (size_t)blockIdx.y * nblk_per_out * S + idx;
        const int b1 = (nblk_per_out * (sl + 1)) / 4;
        int b = (nblk_per_out * sl) / 4;
        for (; b + 8 <= b1; b += 8) {   // 8 independent loads in flight, summed in a fixed order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(b + j) * S];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; b < b1; ++b) s += (double)src[(size_t)b * S];
    }
    comb[sl][el] = s;
    __syncthreads();
    if (sl == 0 && live) {
        const int co = idx / CIP, ci = idx % CIP;
        if (co < Cout && ci < Cin)
            out[((size_t)blockIdx.y * Cout + co) * Cin + ci] = (float)(((comb[0][el] + comb[1][el]) + comb[2][el]) + comb[3][el]);
    }
}

// Wt[k][co] (zero padded [Kp][COUTP]) from W[co][k] (transpose=1) or W[k][co] (transpose=0), W row stride = ld
__global__ void pack_wt_kernel(const float* __restrict__ W, int rows_k, int cols_co, int ld, int transpose,
                               int Kp, int COUTP, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Kp * COUTP) return;
    const int k = idx / COUTP, co = idx % COUTP;
    float v = 0.f;
    if (k < rows_k && co < cols_co) v = transpose ? W[(size_t)co * ld + k] : W[(size_t)k * ld + co];
    out[idx] = v;
}

static int pw_coutp(int Cout) { return Cout > 128 ? 256 : (Cout > 64 ? 128 : (Cout > 32 ? 64 : 32)); }

// wide shapes (Cout > 64) run on the split kernels of pw_gemm_split.hip (bf16 matrix pipe, fp32 results); the library keeps no
// mutable state: which split a call uses follows from its arguments alone (magnitude bounds given -> two fp16 parts)
static bool use_split(int Cout) { return pw_coutp(Cout) >= 128; }

extern "C" int uncr_pw_coutp(int Cout) { return Cout <= 256 ? pw_coutp(Cout) : -1; }
extern "C" int uncr_pw_kpad(int Cin) { return ((Cin + 31) / 32) * 32; }
// statistics slots per (frame, channel) that uncr_pw_gemm(epi != 0) writes for this problem
extern "C" int uncr_pw_stat_slots(int N, int Cout, int P) {
    if (N <= 0 || Cout <= 0 || Cout > 256 || P <= 0) return -1;
    if (use_split(Cout)) return P % 128 ? -1 : pw_split_blocks_per_frame(N, P);
    const int tp = pw_coutp(Cout) >= 128 ? 128 : 256;
    return P % tp ? -1 : P / tp;
}
extern "C" int uncr_pw_tile_px(int Cout) {
    const int cp = pw_coutp(Cout);
    return (cp >= 128) ? 128 : 256;
}

extern "C" int uncr_pw_wt_floats(int rows_k, int cols_co) {
    if (cols_co > 256 || rows_k > 256 || cols_co <= 0 || rows_k <= 0) return -1;
    if (use_split(cols_co)) return (int)pw_split_wt_floats(rows_k, pw_coutp(cols_co));
    return uncr_pw_kpad(rows_k) * pw_coutp(cols_co);
}

extern "C" int uncr_pack_wt(const float* W, int rows_k, int cols_co, int ld, int transpose, float* out,
                            hipStream_t stream) {
    if (cols_co > 256 || rows_k > 256) return UNCR_ESHAPE;
    if (use_split(cols_co)) return pw_split_pack(W, rows_k, cols_co, ld, transpose, out, stream);
    const int Kp = uncr_pw_kpad(rows_k), CP = pw_coutp(cols_co);
    hipLaunchKernelGGL(pack_wt_kernel, dim3((Kp * CP + 255) / 256), dim3(256), 0, stream, W, rows_k, cols_co, ld,
                       transpose, Kp, CP, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_pack_wt_batch(const long long* desc, int n_items, hipStream_t stream) {
    if (!desc || n_items <= 0) return UNCR_EINVAL;
    return pw_pack_batch(desc, n_items, stream);
}

extern "C" int uncr_pw_gemm(const void* in, const void* in2, const float* Wt, void* out, const float* k0,
                            const float* k1, const float* k2, const float* kmu, const float* bias, int bias_stride_n,
                            const void* aux,
                            const float* e0, const float* e1, const float* e2, const float* e3, float* part, int N,
                            int Cin, int Cout, int P, int pro, int epi, int in_dt, int out_dt, float* amax_out,
                            const float* in_amax, int in_amax_n, const float* in2_amax, int in2_amax_n, int Pv,
                            hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin > 256 || Cout <= 0 || Cout > 256 || Pv <= 0 || Pv > P) return UNCR_ESHAPE;
    if (Pv < P && (in_dt != UNCR_F32 || out_dt != UNCR_F32)) return UNCR_EINVAL;      // padded planes: fp32 storage
    if ((in_amax && in_amax_n <= 0) || (in2_amax && in2_amax_n <= 0)) return UNCR_EINVAL;
    if (!in || !Wt || !out) return UNCR_EINVAL;
    if ((in_dt != UNCR_F32 && in_dt != UNCR_BF16) || (out_dt != UNCR_F32 && out_dt != UNCR_BF16)) return UNCR_EINVAL;
    if (pro == PRO_NORMBWD && !in2) return UNCR_EINVAL;
    // epi 9: out = relu(e0*(v + bias) + e1) with (sum, sum^2) statistics -- a ConvLayer's norm + ReLU applied to the accumulator
    // (wide kernels, no prologue; csrc/inconv.hip)
    // epi 10: out = aux + e0*(v + bias) + e1 with (sum, sum^2) statistics -- an eval-mode MBConv's closing BatchNorm and skip applied
    // to the accumulator of pw2 (wide kernels, 65..128 output channels, GELU prologue)
    if (epi < 0 || (epi > 4 && epi != 9 && epi != 10)) return UNCR_EINVAL;
    if (epi == 9 && (!use_split(Cout) || pro != PRO_NONE || !e0 || !e1)) return UNCR_EINVAL;
    if (epi == 10 && (!use_split(Cout) || pw_coutp(Cout) != 128 || pro != PRO_AFFINE_GELU || !e0 || !e1 || !aux)) return UNCR_EINVAL;
    if (epi && epi != 4 && !part) return UNCR_EINVAL;
    if ((epi == 2 || epi == 3) && !aux) return UNCR_EINVAL;
    if (epi == 3 && (!e0 || !e1 || !e2 || !e3)) return UNCR_EINVAL;
    if (epi == 4 && !use_split(Cout)) return UNCR_EINVAL;     // accumulate mode exists in the split kernels only
    const int tp = uncr_pw_tile_px(Cout);
    if (P % tp) return UNCR_ESHAPE;
    PwArgs g{in, in2, Wt, out, k0, k1, k2, bias, aux, e0, e1, e2, e3, (float2*)part, bias_stride_n, Cin, Cout, P, pro, epi};
    g.k3 = pro == PRO_NORMBWD ? kmu : nullptr;
    g.h2 = in_amax != nullptr;
    g.amax_out = amax_out;
    g.in_amax = in_amax; g.in_amax_n = in_amax_n;
    g.in2_amax = in2_amax; g.in2_amax_n = in2_amax_n;
    g.Pv = Pv < P ? Pv : 0;
    const int cp = pw_coutp(Cout);
    if (use_split(Cout)) {
        if (in_dt != out_dt) return UNCR_EINVAL;      // the wide kernels have one storage type for all activation operands
        switch (pro) {
            case PRO_NONE: return pw_split_launch_p0(g, N, cp, in_dt, stream);
            case PRO_AFFINE: return pw_split_launch_p1(g, N, cp, in_dt, stream);
            case PRO_AFFINE_GELU: return pw_split_launch_p2(g, N, cp, in_dt, stream);
            case PRO_NORMBWD:
                return pw_split_launch_p3(g, N, cp, in_dt, stream);
            case PRO_AFFINE_RELU: return pw_split_launch_p4(g, N, cp, in_dt, stream);
            default: return UNCR_EINVAL;
        }
    }
    // fp32-MFMA kernels (Cout <= 64): fp32 storage, bf16 inputs with fp32 outputs (head-like layers), or bf16 on both sides
    if (in_dt == UNCR_F32 && out_dt != UNCR_F32) return UNCR_EINVAL;
    dim3 grid(P / tp, N);
#define PW_GO(CT_, WN_, WM_, PRE2_, THREADS)                                                                                     \
    do {                                                                                                                         \
        if (in_dt == UNCR_F32) hipLaunchKernelGGL((pw_gemm_kernel<CT_, WN_, WM_, PRE2_>), grid, dim3(THREADS), 0, stream, g);     \
        else if (out_dt == UNCR_F32)                                                                                             \
            hipLaunchKernelGGL((pw_gemm_kernel<CT_, WN_, WM_, PRE2_, bf16_t, float>), grid, dim3(THREADS), 0, stream, g);         \
        else hipLaunchKernelGGL((pw_gemm_kernel<CT_, WN_, WM_, PRE2_, bf16_t, bf16_t>), grid, dim3(THREADS), 0, stream, g);       \
    } while (0)
    if (cp == 64) PW_GO(1, 2, 2, true, 256);
    else if (pro == PRO_NORMBWD) PW_GO(1, 1, 2, true, 128);
    else PW_GO(1, 1, 2, false, 128);
#undef PW_GO
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

#ifndef HEAD_KC
#define HEAD_KC 16      // contraction rows per staged chunk of the head's <= 32-channel GEMM (32: two blocks per CU, one wave per SIMD)
#endif
// out_conv + output nonlinearities in one kernel (narrow fp32-MFMA GEMM, Cout <= 64): uncrtaints.py:432-445
extern "C" int uncr_head_fwd(const void* y, const float* Wt, const float* bias, float* out, float* pre, int N, int Cin,
                             int Cout, int P, int n_mean, float scale, float eps, int var_mode, int in_dt,
                             hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin > 256 || Cout <= 0 || Cout > 64 || var_mode < 0 || var_mode > 2) return UNCR_ESHAPE;
    if (!y || !Wt || !out || (in_dt != UNCR_F32 && in_dt != UNCR_BF16)) return UNCR_EINVAL;
    const int tp = uncr_pw_tile_px(Cout);
    if (P % tp) return UNCR_ESHAPE;
    PwArgs g{y, nullptr, Wt, out, nullptr, nullptr, nullptr, bias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
             0, Cin, Cout, P, PRO_NONE, 7, nullptr, nullptr};
    g.head_nm = n_mean; g.head_var = var_mode; g.head_scale = scale; g.head_eps = eps; g.head_pre = pre;
    dim3 grid(P / tp, N);
    if (pw_coutp(Cout) == 64) {
        if (in_dt == UNCR_BF16) hipLaunchKernelGGL((pw_gemm_kernel<1, 2, 2, true, bf16_t, float>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((pw_gemm_kernel<1, 2, 2, true>), grid, dim3(256), 0, stream, g);
    } else {
        if (in_dt == UNCR_BF16) hipLaunchKernelGGL((pw_gemm_kernel<1, 1, 2, false, bf16_t, float, HEAD_KC>), grid, dim3(128), 0, stream, g);
        else hipLaunchKernelGGL((pw_gemm_kernel<1, 1, 2, false, float, float, HEAD_KC>), grid, dim3(128), 0, stream, g);
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// Backward of pw1 with the PreNorm backward and the skip connection in the epilogue (split kernels only):
//   out = dy + c1*(W^T . normbwd(in, in2)) + c2*x + c3,   statistics (sum out, sum out*xh3) if xh3 is given.
// c1..c3 are known before this GEMM because the sums they depend on follow from the weight-gradient products
// (uncr_prenorm_bwd_finish).
extern "C" int uncr_pw_gemm_dx_supported(int Cin, int Cout) { return (use_split(Cout) && pw_coutp(Cout) == 128 && Cin <= 256) ? 1 : 0; }
extern "C" int uncr_pw_gemm_dx(const void* in, const void* in2, const float* Wt, void* out, const float* k0,
                               const float* k1, const float* k2, const float* kmu, const void* dy, const void* x,
                               const void* xh3, const float* c1, const float* c2, const float* c3, const float* cmu,
                               const float* relu_a, const float* relu_b, const float* relu_mu, float* part, int N, int Cin, int Cout, int P,
                               int act, float* amax_out, const float* in_amax, int in_amax_n, const float* in2_amax,
                               int in2_amax_n, int Pv, hipStream_t stream) {
    if (N <= 0 || Cin <= 0 || Cin > 256 || Cout <= 0 || Cout > 256 || Pv <= 0 || Pv > P) return UNCR_ESHAPE;
    if (Pv < P && act != UNCR_F32) return UNCR_EINVAL;
    if (!in || !in2 || !Wt || !out || !dy || !x || !c1 || !c2 || !c3) return UNCR_EINVAL;
    if (act != UNCR_F32 && act != UNCR_BF16) return UNCR_EINVAL;
    if (xh3 && !part) return UNCR_EINVAL;
    // ReLU of the ConvLayer that produced x: relu_a AND relu_b with xh3 = that layer's pre-norm tensor (mask relu_a*xh3 + relu_b > 0),
    // or relu_a alone (any non-null pointer) WITHOUT xh3: the mask is [x > 0] itself and part (required) receives (sum out, sum out*x)
    const bool relu_x = relu_a && !relu_b && !xh3;
    if (relu_x && !part) return UNCR_EINVAL;
    if (!relu_x && (relu_a || relu_b) && !(relu_a && relu_b && xh3)) return UNCR_EINVAL;
    if (!uncr_pw_gemm_dx_supported(Cin, Cout)) return UNCR_EINVAL;
    if (P % uncr_pw_tile_px(Cout)) return UNCR_ESHAPE;
    PwArgs g{in, in2, Wt, out, k0, k1, k2, relu_x ? nullptr : relu_b, x, c1, c2, c3, (relu_a && !relu_x) ? relu_a : c3,
             (xh3 || relu_x) ? (float2*)part : nullptr, 0, Cin, Cout, P, PRO_NORMBWD, relu_x ? 8 : (relu_a ? 6 : 5), dy, xh3};
    g.k3 = kmu;
    g.emu = cmu;
    g.rmu = (relu_a && relu_b) ? relu_mu : nullptr;
    g.amax_out = amax_out;
    if ((in_amax && in_amax_n <= 0) || (in2_amax && in2_amax_n <= 0)) return UNCR_EINVAL;
    // both operand bounds given (fp32 storage): two scaled fp16 parts, as the dz GEMM
    g.h2 = in_amax != nullptr && in2_amax != nullptr;
    g.in_amax = in_amax; g.in_amax_n = in_amax_n;
    g.in2_amax = in2_amax; g.in2_amax_n = in2_amax_n;
    g.Pv = Pv < P ? Pv : 0;
    return pw_split_launch_p3(g, N, pw_coutp(Cout), act, stream);
}

// weight-gradient shapes: (COP, CIP) in {(128,256), (256,128), (128,32), (32,128), (64,256), (32,256), (64,128), (32,32), (64,32)}
static int wg_shape(int Cd, int Cx, int* cop, int* cip) {
    if (Cd > 128 && Cd <= 256 && Cx > 32 && Cx <= 128) { *cop = 256; *cip = 128; return 0; }
    if (Cd > 64 && Cd <= 128 && Cx > 32 && Cx <= 128) { *cop = 256; *cip = 128; return 0; }   // 128 x 128 (use_v): padded rows
    if (Cd > 64 && Cd <= 128 && Cx > 128 && Cx <= 256) { *cop = 128; *cip = 256; return 1; }
    if (Cd > 32 && Cd <= 128 && Cx <= 32) { *cop = 128; *cip = 32; return 2; }
    if (Cd <= 32 && Cx > 32 && Cx <= 128) { *cop = 32; *cip = 128; return 3; }
    if (Cd > 32 && Cd <= 64 && Cx > 128 && Cx <= 256) { *cop = 64; *cip = 256; return 4; }
    if (Cd <= 32 && Cx > 128 && Cx <= 256) { *cop = 32; *cip = 256; return 5; }              // keys of n_head <= 8
    if (Cd > 32 && Cd <= 64 && Cx > 32 && Cx <= 128) { *cop = 64; *cip = 128; return 6; }     // 64-wide blocks
    if (Cd <= 32 && Cx <= 32) { *cop = 32; *cip = 32; return 7; }
    if (Cd > 32 && Cd <= 64 && Cx <= 32) { *cop = 64; *cip = 32; return 8; }
    if (Cd > 128 && Cd <= 256 && Cx <= 32) { *cop = 256; *cip = 32; return 9; }               // L-TAE inconv of 32-wide encoders
    return -1;
}

extern "C" int uncr_wgrad_shape(int Cd, int Cx, int* cop, int* cip) { return wg_shape(Cd, Cx, cop, cip); }

// blocks (= partial products) per frame that uncr_pw_wgrad will use for this problem
extern "C" int uncr_wgrad_nbx(int N, int Cd, int Cx, int P, int pro_d, int pro_x, int rowsum, int act) {
    if (N <= 0 || P <= 0 || P % 32) return -1;
    if (act == UNCR_BF16 && pw_wgrad_a16_supported(Cd, Cx, pro_d, pro_x, rowsum != 0) && P % 64 == 0)
        return pw_wgrad_a16_nbx(N, P);
    if (act == UNCR_F32 && pw_wgrad_split_supported(Cd, Cx, pro_d, pro_x, rowsum != 0)) return pw_wgrad_split_nbx(N, P);
    // fp32-MFMA kernel: equal pixel ranges, aim for ~512-1024 blocks on the 256 CUs
    for (int cand = 4096; cand >= 256; cand >>= 1)
        if (P % cand == 0 && (long long)N * (P / cand) >= 512) return P / cand;
    return P % 256 == 0 ? P / 256 : P / 32;
}

extern "C" int uncr_pw_wgrad(const void* d, const void* d2, const void* x, const void* x2, const float* dk0,
                             const float* dk1, const float* dk2, const float* dkmu, const float* xk0, const float* xk1,
                             const float* xk2, float* part, float* rs_part, int N, int Cd, int Cx, int P, int NBX,
                             int pro_d, int pro_x, int act, const float* d_amax, int d_amax_n, const float* d2_amax,
                             int d2_amax_n, const float* x_ub, int Pv, hipStream_t stream) {
    int cop, cip;
    const int shp = wg_shape(Cd, Cx, &cop, &cip);
    if (shp < 0 || N <= 0 || NBX <= 0 || Pv <= 0 || Pv > P) return UNCR_ESHAPE;
    if (Pv < P && act != UNCR_F32) return UNCR_EINVAL;      // padded planes: fp32 storage
    const int pv = Pv < P ? Pv : 0;
    if (!d || !x || !part || (act != UNCR_F32 && act != UNCR_BF16)) return UNCR_EINVAL;
    if (pro_d == PRO_NORMBWD && !d2) return UNCR_EINVAL;
    if (pro_x == PRO_NORMBWD) return UNCR_EINVAL;   // norm-backward form is only built for the D operand
    if (act == UNCR_BF16 && pw_wgrad_a16_supported(Cd, Cx, pro_d, pro_x, rs_part != nullptr) && P % 64 == 0) {
        if (!dk0 || !dk1 || !dk2 || !xk0 || !xk1) return UNCR_EINVAL;
        return pw_wgrad_a16_launch(d, d2, x, dk0, dk1, dk2, dkmu, xk0, xk1, xk2, part, N, Cd, Cx, P, NBX, pro_x, stream);
    }
    if (act == UNCR_F32 && pw_wgrad_split_supported(Cd, Cx, pro_d, pro_x, rs_part != nullptr)) {
        if (!dk0 || !dk1 || !dk2 || !xk0 || !xk1) return UNCR_EINVAL;
        return pw_wgrad_split_launch((const float*)d, (const float*)d2, (const float*)x, dk0, dk1, dk2, dkmu, xk0, xk1, xk2, part, N,
                                     Cd, Cx, P, NBX, pro_x, d_amax, d_amax_n, d2_amax, d2_amax_n, x_ub, pv, stream);
    }
    if (P % NBX) return UNCR_ESHAPE;
    const int PXB = P / NBX;
    if (PXB % 32) return UNCR_ESHAPE;
    WgArgs g{d, d2, x, x2, dk0, dk1, dk2, xk0, xk1, xk2, part, rs_part, Cd, Cx, P, PXB, pro_d, pro_x};
    g.dk3 = pro_d == PRO_NORMBWD ? dkmu : nullptr;
    g.Pv = pv;
    dim3 grid(P / PXB, N);
    const size_t lds = (size_t)((cop + cip) * 36 + 4 * cop + 3 * cip) * sizeof(float);
    // bf16 operands on the fp32-MFMA kernels: every shape the two bf16-product kernels above do not take
    switch (shp) {
#define WG_LAUNCH_T(MT_, NT_, WCO_, WCI_, THREADS, TD, TX)                                                                 \
    if (pro_d == PRO_NORMBWD)                                                                                              \
        hipLaunchKernelGGL((pw_wgrad_kernel<MT_, NT_, WCO_, WCI_, true, TD, TX>), grid, dim3(THREADS), lds, stream, g);    \
    else                                                                                                                   \
        hipLaunchKernelGGL((pw_wgrad_kernel<MT_, NT_, WCO_, WCI_, false, TD, TX>), grid, dim3(THREADS), lds, stream, g);
#define WG_LAUNCH(MT_, NT_, WCO_, WCI_, THREADS) WG_LAUNCH_T(MT_, NT_, WCO_, WCI_, THREADS, float, float) break;
#define WG_LAUNCH_AB(MT_, NT_, WCO_, WCI_, THREADS)                                          \
    if (act == UNCR_BF16) { WG_LAUNCH_T(MT_, NT_, WCO_, WCI_, THREADS, bf16_t, bf16_t) }       \
    else { WG_LAUNCH_T(MT_, NT_, WCO_, WCI_, THREADS, float, float) }                          \
    break;
        case 0: WG_LAUNCH_AB(2, 4, 4, 1, 256)
        case 1: WG_LAUNCH_AB(2, 4, 2, 2, 256)
        case 2: WG_LAUNCH_AB(1, 1, 4, 1, 256)
        case 3: WG_LAUNCH_AB(1, 1, 1, 4, 256)
        case 4: WG_LAUNCH_AB(2, 4, 1, 2, 128)
        case 5: WG_LAUNCH_AB(1, 4, 1, 2, 128)
        case 6: WG_LAUNCH_AB(2, 4, 1, 1, 64)
        case 7: WG_LAUNCH_AB(1, 1, 1, 1, 64)
        case 8: WG_LAUNCH_AB(2, 1, 1, 1, 64)
        case 9: WG_LAUNCH_AB(2, 1, 4, 1, 256)
#undef WG_LAUNCH
#undef WG_LAUNCH_AB
#undef WG_LAUNCH_T
    }
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

// Padded planes of an any-size image (anysize.hip): the weight-gradient kernels sum the whole 32-pixel chunks below Pv; the last
// Pv % 32 pixels of every frame are added here, into the frame's FIRST partial (and its first row-sum partial): plain fp32 products of
// the same prologues, at most 31 terms per entry.  One thread per (co, ci); the operand rows come from L1 / L2.
__global__ __launch_bounds__(256) void wgrad_boundary_kernel(WgArgs g, int nbx, int COP, int CIP) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.Cd * g.Cx) return;
    const int co = i / g.Cx, ci = i - co * g.Cx;
    const int p0 = g.Pv / 32 * 32, np = g.Pv - p0;
    const float* dr = (const float*)g.d + ((size_t)n * g.Cd + co) * g.P + p0;
    const float* d2r = g.d2 ? (const float*)g.d2 + ((size_t)n * g.Cd + co) * g.P + p0 : dr;
    const float* xr = (const float*)g.x + ((size_t)n * g.Cx + ci) * g.P + p0;
    const int di = n * g.Cd + co, xi = n * g.Cx + ci;
    const float d0 = g.dk0 ? g.dk0[di] : 1.f, d1 = g.dk1 ? g.dk1[di] : 0.f, dmu = g.dk3 ? g.dk3[di] : 0.f;
    const float dc2 = g.dk2 ? g.dk2[di] : (g.pro_d == PRO_AFFINE_GELU ? 1.f : 0.f);
    const float x0 = g.xk0 ? g.xk0[xi] : 1.f, x1 = g.xk1 ? g.xk1[xi] : 0.f;
    const float xc2 = g.xk2 ? g.xk2[xi] : (g.pro_x == PRO_AFFINE_GELU ? 1.f : 0.f);
    auto pro = [](int kind, float v, float v2, float c0, float c1, float c2, float c3) -> float {
        if (kind == PRO_AFFINE) return fmaf(c0, v, c1);
        if (kind == PRO_AFFINE_GELU) return c2 * gelu_f(fmaf(c0, v, c1));
        if (kind == PRO_NORMBWD) return fmaf(c0, v, fmaf(c1, v2 - c3, c2));
        if (kind == PRO_AFFINE_RELU) return fmaxf(fmaf(c0, v, c1), 0.f);
        return v;
    };
    float acc = 0.f, rs = 0.f;
    for (int p = 0; p < np; ++p) {
        const float a = pro(g.pro_d, dr[p], d2r[p], d0, d1, dc2, dmu);
        acc = fmaf(a, pro(g.pro_x, xr[p], 0.f, x0, x1, xc2, 0.f), acc);
        rs += a;
    }
    g.part[((size_t)n * nbx * COP + co) * CIP + ci] += acc;
    if (g.rs_part && ci == 0) g.rs_part[(size_t)n * nbx * COP + co] += rs;
}
extern "C" int uncr_wgrad_boundary(const float* d, const float* d2, const float* x, const float* dk0, const float* dk1,
                                   const float* dk2, const float* dkmu, const float* xk0, const float* xk1, const float* xk2,
                                   float* part, float* rs_part, int N, int Cd, int Cx, int P, int Pv, int NBX, int pro_d, int pro_x,
                                   hipStream_t stream) {
    int cop, cip;
    if (wg_shape(Cd, Cx, &cop, &cip) < 0 || N <= 0 || NBX <= 0 || Pv <= 0 || Pv > P) return UNCR_ESHAPE;
    if (!d || !x || !part || (pro_d == PRO_NORMBWD && !d2) || pro_x == PRO_NORMBWD) return UNCR_EINVAL;
    if (Pv % 32 == 0) return UNCR_OK;
    WgArgs g{d, d2, x, nullptr, dk0, dk1, dk2, xk0, xk1, xk2, part, rs_part, Cd, Cx, P, 0, pro_d, pro_x};
    g.dk3 = pro_d == PRO_NORMBWD ? dkmu : nullptr;
    g.Pv = Pv;
    hipLaunchKernelGGL(wgrad_boundary_kernel, dim3((Cd * Cx + 255) / 256, N), dim3(256), 0, stream, g, NBX, cop, cip);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}

extern "C" int uncr_wgrad_reduce(const float* part, int n_out, int nblk_per_out, int COP, int CIP, int Cout, int Cin,
                                 float* out, hipStream_t stream) {
    if (n_out <= 0 || nblk_per_out <= 0) return UNCR_ESHAPE;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((COP * CIP + 255) / 256, n_out), dim3(1024), 0, stream, part,
                       nblk_per_out, COP, CIP, Cout, Cin, out);
    UNCR_LAUNCH_CHECK();
    return UNCR_OK;
}
