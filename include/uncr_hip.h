/* libuncr_hip -- C ABI of the MI355X (gfx950) kernels behind the UnCRtainTS `--model uncrtaints`
 * forward/backward hot path.
 *
 * The reference (PatrickTUM/UnCRtainTS) has no FFI: below `model/src/backbones/uncrtaints.py` everything is
 * stock torch.nn / ATen.  This header therefore defines the seam a maintainer would bind (ctypes stub shown in
 * INTEGRATION.md): one entry point per fused stage, forward and backward.  Each comment names the reference
 * code (file:line under /root/reference) whose work the entry point replaces.
 *
 * Conventions
 *   - all tensors are contiguous NCHW planes: [frames N][channels C][pixels P = H*W].  ACTIVATION tensors (declared
 *     `void*`: every [N][C][P] activation and activation gradient of in_conv, the MBConv blocks and the temporal
 *     aggregation) are stored as fp32 (UNCR_F32, the reference's arithmetic; BASELINE configs 1, 2, 4, 5) or as bf16
 *     (UNCR_BF16: "bf16 activations, fp32 accumulate", BASELINE config 3), selected per call by the `act` / `*_dt`
 *     argument.  Everything declared `float*` is fp32 in both modes: statistics partials, coefficients, weights and
 *     their gradients, the 32x32 L-TAE branch, the model outputs and the loss.  Kernels always compute in fp32; with
 *     bf16 storage a producer rounds once (to nearest even) at its store and takes its statistics from the ROUNDED values;
 *   - the CALLER owns every buffer (incl. partial-statistics workspaces); nothing is allocated, freed or
 *     retained here; kernels are enqueued on the passed stream and never synchronise;
 *   - return 0 on success, negative = argument/shape error, positive = hipError_t of the launch;
 *   - "part" buffers hold per-block partial sums float2[N*C][slots]; the *_slots()/tile helpers give the
 *     slot count of each producer for a given size;
 *   - a normalisation layer is never run on its own: producers emit partial (sum, sum^2), a finalize call
 *     turns them into per-(frame,channel) coefficients A,B, and the CONSUMER applies u = A*h + B in its
 *     prologue.  Backward likewise: dh = C1*du + C2*(h - mu) + C3 (four numbers per plane; a null `kmu` means mu = 0, the
 *     raw form).
 */
#ifndef UNCR_HIP_H
#define UNCR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* activation storage codes */
#define UNCR_F32 0
#define UNCR_BF16 1
/* prologue kinds (pw_gemm / pw_wgrad) */
#define UNCR_PRO_NONE 0
#define UNCR_PRO_AFFINE 1        /* A*v + B                              */
#define UNCR_PRO_AFFINE_GELU 2   /* S * gelu(A*v + B)                    */
#define UNCR_PRO_NORMBWD 3       /* C1*v + C2*(v2 - mu) + C3  (two operands; mu = the kmu array, 0 if null) */
#define UNCR_PRO_AFFINE_RELU 4
/* normalisation kinds */
#define UNCR_NORM_GROUP 0
#define UNCR_NORM_BATCH_TRAIN 1
#define UNCR_NORM_BATCH_EVAL 2
/* element-wise ops (uncr_ew) */
#define UNCR_EW_STATS_SQ 0
#define UNCR_EW_STATS_AUX 1
#define UNCR_EW_AFFINE_RELU 2
#define UNCR_EW_RESIDUAL 3
#define UNCR_EW_PASSB 4
#define UNCR_EW_PASSE 5
#define UNCR_EW_RELU_BWD 6
#define UNCR_EW_SE_POOL 7
#define UNCR_EW_HEAD_FWD 8
#define UNCR_EW_HEAD_BWD 9
#define UNCR_EW_RESIDUAL_RELU 10   /* out = a + relu(A*b + B) (ResidualConvBlock skip) */
/* head with the other variance nonlinearities of get_nonlinearity (uncrtaints.py:223-228): elu(a)+1+eps / identity */
#define UNCR_EW_HEAD_FWD_ELU 11
#define UNCR_EW_HEAD_BWD_ELU 12
#define UNCR_EW_HEAD_FWD_ID 13
#define UNCR_EW_HEAD_BWD_ID 14
/* stand-alone norm layers / SE (PreNorm uncrtaints.py:72-79, SE uncrtaints.py:82-97 called outside MBConv) */
#define UNCR_EW_AFFINE 15          /* out = A*a + B, stats (sum out, sum out^2) */
#define UNCR_EW_NORMBWD 16         /* out = C1*a + C2*(b - M) + C3 (M = k3 or 0) */
#define UNCR_EW_SE_POOL4 17        /* SE_POOL, four chunks of a plane per block: the same partial slots and values (P % 4096 == 0) */

int uncr_version(void);

/* ---- normalisation coefficients: nn.GroupNorm / nn.BatchNorm2d statistics
 *      (uncrtaints.py:16-22 get_norm_layer, utae.py:470-473, uncrtaints.py:72-79 PreNorm) ---- */
/* ub (nullable) [N*C]: upper bound on |coefA*h + coefB| per plane, taken from the partials' sums of squares (a block's
 * elements are bounded by the root of its sum of squares); needs `part` = (sum h, sum h^2) partials of the normalised tensor --
 * in BatchNorm eval mode too, where they serve nothing else.  Consumed by uncr_pw_gemm (in_amax) to scale its fp16 operand split. */
int uncr_norm_finalize_fwd(const float* part, int NP, int N, int C, int groups, int P, int kind,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, float* coefA, float* coefB, float* save_mean,
                           float* save_rstd, float* ub,
                           float* hb /* nullable (needs ub) [N*C]: the bound on |h| itself, for consumers that apply another map to h */,
                           const void* src /* nullable: the normalised tensor itself ([N*C] planes of P valid elements, src_stride elements
                                              apart, storage src_act).  Train-mode statistics sets with var <= 2^-6 mean^2 (|mean| >= 8
                                              sigma, where raw moments of fp32 slot sums lose digits) are then re-read once and
                                              take their second moment about the mean (csrc/bn_inline.h) */,
                           long long src_stride, int src_act, hipStream_t stream);
int uncr_norm_finalize_bwd(const float* part, int NP, int N, int C, int groups, int P, int kind,
                           const float* gamma, const float* save_mean, const float* save_rstd, float* c1,
                           float* c2, float* c3,
                           float* cmu /* [N*C] or null.  Given: CENTRED coefficients, dh = C1*du + C2*(h - cmu) + C3 (cmu = the
                                         norm's mean per plane): no per-plane rounding offset; consumers take cmu as their
                                         `kmu` argument.  Null: the raw form dh = C1*du + C2*h + C3 */,
                           float* dgamma, float* dbeta, float* scratch /* [2*N*C], GroupNorm */,
                           int centered /* 1: part.y = sum du*(h - mean) (uncr_dw_bwd with a mean array) */,
                           hipStream_t stream);
/* Synchronised BatchNorm for data parallelism (train mode): per-channel fp64 sums of the local partials -> the host
 * layer all-reduces them (RCCL) -> coefficients from the GLOBAL sums and element count; d gamma / d beta from the LOCAL
 * sums (they are averaged with the other gradients).  sums: [C][2] doubles = (S1, S2) in the partials' meaning. */
int uncr_bn_channel_sums(const float* part, int NP, int N, int C, double* sums, hipStream_t stream);
int uncr_bn_finalize_fwd_sums(const double* sums, double count, int N, int C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* coefA,
                              float* coefB, float* save_mean, float* save_rstd,
                              const float* part, int NP, float* ub /* nullable: as in uncr_norm_finalize_fwd, from the LOCAL partials */,
                              float* hb /* nullable, needs ub */, const double* csums /* nullable: the all-reduced output of uncr_bn_centred_sums */,
                              hipStream_t stream);
/* LOCAL centred sums [C][2] doubles = (sum (h - m0), sum (h - m0)^2) of every channel the GLOBAL raw sums put 8 sigma or more from zero
 * (m0 = the global raw mean in fp32; zeros for every other channel): the host all-reduces them too, and uncr_bn_finalize_fwd_sums takes
 * those channels' statistics from them (csrc/bn_inline.h: raw moments of fp32 slot sums lose such a set).  src: the normalised tensor */
int uncr_bn_centred_sums(const double* sums, double count, const void* src, int N, int C, int P, long long stride, int act,
                         double* out, hipStream_t stream);
int uncr_bn_finalize_bwd_sums(const double* sums_local, const double* sums_global, double count, int N, int C,
                              const float* gamma, const float* save_mean, const float* save_rstd, float* c1, float* c2,
                              float* c3, float* cmu, float* dgamma, float* dbeta, int centered, hipStream_t stream);

/* ---- element-wise family with fused coefficients + partial statistics
 *      (norm-apply/ReLU utae.py:470-494; residual add uncrtaints.py:142-146; SE avg-pool uncrtaints.py:85,95;
 *       output nonlinearities uncrtaints.py:384-388,441-445 and their autograd twins) ---- */
int uncr_ew_slots(int P);
/* act: storage of a, b, c, aux and out (the HEAD_* and RESIDUAL_RELU ops exist for fp32 only) */
int uncr_ew(int op, const void* a, const void* b, const void* c, const void* aux, void* out,
            const float* k0, const float* k1, const float* k2, const float* k3, float* part, int planes, int P,
            int C, int n_mean, float scale, float eps, int act,
            int Pv /* pixels of a plane that carry data: P, or fewer on the padded planes of an any-size image (fp32): the rest is a
                      zero tail on input, written as zeros, and left out of the statistics */,
            hipStream_t stream);
/* per-plane totals (fp64 accumulation, fixed order) of a [planes][slots] (sum0, sum1) partial array; either output may be null */
int uncr_part_sums(const float* part, int slots, int planes, float* out0, float* out1, hipStream_t stream);
/* out[n*C + c] = scale * mean of the statistics set of plane (n, c): groups > 0: mean [N*groups] (GroupNorm / InstanceNorm), 0: [C]
 * (BatchNorm) -- per-plane pivots for the centred backward statistics and the centred weight-gradient products */
int uncr_plane_means(const float* mean, int N, int C, int groups, float scale, float* out, hipStream_t stream);
/* InstanceNorm2d (groups == C) behind uncr_norm_finalize_fwd: every plane whose raw-moment variance is below 2^-6 mean^2 gets mean,
 * rstd, A, B (and the bounds ub / hb, nullable) recomputed from the tensor itself about its mean -- (sum h, sum h^2) of fp32 slot sums
 * resolve a variance to ~1e-7 mean^2 only (uncrtaints.py:16-22 InstanceNorm2d behind an un-normalised tensor, e.g. the decoder's
 * first PreNorm behind an eval-mode BatchNorm encoder).  x: [N*C] planes of P valid elements, `stride` elements apart; act: storage */
int uncr_instance_repair(const void* x, int N, int C, int P, long long stride, const float* gamma, const float* beta, float eps,
                         float* coefA, float* coefB, float* save_mean, float* save_rstd, float* ub, float* hb, int act,
                         hipStream_t stream);
/* dst = src converted between the storage types (model input -> bf16 activations; bf16 input gradient -> fp32) */
int uncr_cast(const void* src, void* dst, long long n, int src_dt, int dst_dt, hipStream_t stream);

/* ---- 1x1 convolutions as MFMA GEMMs with fp32 results (nn.Conv2d k=1: utae.py:476-484 in_conv/out_conv,
 *      uncrtaints.py:126 pw, :136 pw-linear; nn.Conv1d k=1 ltae.py:176,214; nn.Linear ltae.py:327,349).
 *      Cout <= 64: v_mfma_f32_32x32x2_f32.  Cout > 64: the operands are split into 16-bit parts for the bf16 / fp16 matrix pipe
 *      with fp32 accumulation and fp32-grade results: the exact 3-way bf16 split of both operands (six partial products on
 *      v_mfma_f32_32x32x16_bf16), or -- where a call carries magnitude bounds for its activation operand (in_amax below) -- two
 *      fp16 parts per operand (three products on v_mfma_f32_32x32x16_f16, 2^-22 relative accuracy): the activations are scaled
 *      per frame by a power of two derived from the bound, the weights per output channel at pack time, so no value can leave
 *      the fp16 range whatever the checkpoint or the data hold.  The library keeps no mutable state: the variant follows from
 *      the arguments of the call. ---- */
int uncr_pw_wt_floats(int rows_k, int cols_co);   /* floats to allocate for uncr_pack_wt's output */
int uncr_pw_coutp(int Cout);      /* padded output-channel count of the kernel variant */
int uncr_pw_kpad(int Cin);        /* padded reduction length */
int uncr_pw_tile_px(int Cout);    /* pixels per tile of the kernel variant (P must be a multiple) */
int uncr_pw_stat_slots(int N, int Cout, int P);   /* statistics slots per (frame, channel) written when epi != 0 */
int uncr_pack_wt(const float* W, int rows_k, int cols_co, int ld, int transpose, float* out, hipStream_t stream);
/* the same for many weights in one launch: desc = n_items x 8 int64 in DEVICE memory {W, out, rows_k, cols_co, ld,
 * transpose, 0, 0} */
int uncr_pack_wt_batch(const long long* desc, int n_items, hipStream_t stream);
/* in_dt: storage of in / in2; out_dt: storage of out and aux.  Cout > 64 (bf16 MFMA kernels): in_dt == out_dt; with bf16
 * the prologue's fp32 result is rounded once to bf16 and multiplied with the two leading weight parts (16 significant bits):
 * two products per MAC instead of six.  Cout <= 64 (fp32 MFMA kernels): fp32 outputs, fp32 or bf16 inputs. */
int uncr_pw_gemm(const void* in, const void* in2, const float* Wt, void* out, const float* k0,
                 const float* k1, const float* k2, const float* kmu /* PRO_NORMBWD: mean array or null */,
                 const float* bias, int bias_stride_n, const void* aux,
                 const float* e0, const float* e1, const float* e2, const float* e3 /* epi 3 coefficients; epi 9 (Cout > 64, pro NONE):
                 out = relu(e0*(v + bias) + e1) per (frame, output channel) with (sum, sum^2) statistics of the result; epi 10
                 (64 < Cout <= 128, pro AFFINE_GELU): out = aux + e0*(v + bias) + e1 with the same statistics -- the closing
                 BatchNorm of an eval-mode MBConv (running statistics) and its skip, uncrtaints.py:121-146, on the accumulator */,
                 float* part, int N, int Cin, int Cout, int P, int pro, int epi, int in_dt, int out_dt,
                 /* magnitude bookkeeping for the fp16 two-part split (all nullable / 0; Cout > 64, fp32 storage):
                  * amax_out [N][uncr_pw_stat_slots]: per-block max |stored output| (Cout <= 128 with epi 1 / 2);
                  * pro NORMBWD + epi 3 (Cout 256): in_amax [N][in_amax_n], in2_amax [N][in2_amax_n] = such arrays (or any
                  *   per-frame upper bounds) of the two prologue operands;
                  * pro AFFINE / AFFINE_GELU + epi 0 / 1 / 10: in_amax [N][in_amax_n = Cin] = upper bounds on |k0*in + k1| per plane
                  *   (uncr_norm_finalize_fwd's `ub` output), in2_amax unused.
                  * With the bounds given the GEMM multiplies in two fp16 parts scaled by a per-frame power of two derived from
                  * them; without, in the exact bf16 split */
                 float* amax_out, const float* in_amax, int in_amax_n, const float* in2_amax, int in2_amax_n,
                 int Pv /* pixels of a plane that carry data: P, or fewer on the padded planes of an any-size image (fp32 storage) --
                           the statistics then leave out every pixel tile (uncr_pw_tile_px) that reaches beyond Pv; uncr_fix_tail
                           behind the launch adds the boundary tile's valid pixels and zeroes the tail */,
                 hipStream_t stream);
/* out_conv (Conv2d k=1 + bias, uncrtaints.py:432-440) with the output nonlinearities (uncrtaints.py:441-445) in the GEMM
 * epilogue, Cout <= 64: channel < |n_mean| -> n_mean > 0 ? scale*sigmoid : identity; the others -> var_mode 0 softplus(beta 1,
 * threshold 20) + eps, 1 elu + 1 + eps, 2 identity.  pre (nullable) also receives the pre-activation for the backward
 * (uncr_ew HEAD_BWD*); without it, uncr_ew(HEAD_BWD*, C = -Cout) recovers the derivatives from the output (less accurate
 * where the variance is within rounding of eps). */
int uncr_head_fwd(const void* y, const float* Wt, const float* bias, float* out, float* pre, int N, int Cin, int Cout,
                  int P, int n_mean, float scale, float eps, int var_mode, int in_dt /* storage of y */, hipStream_t stream);
/* Backward of MBConv's pw1 (uncrtaints.py:100-146: x + block(PreNorm(x))) with the PreNorm backward and the skip
 * connection in the GEMM epilogue: out = dy + c1*(W^T . normbwd(in, in2; k0..k2)) + c2*x + c3; if xh3 (the h3 of the
 * block that produced x) is given, part receives (sum out, sum out*xh3) for that block's last norm backward.
 * c1..c3 come from uncr_norm_finalize_bwd on the sums uncr_prenorm_bwd_finish derives without a pass over da.
 * relu_a / relu_b [N*Cout] (both or neither; need xh3): x came out of a ConvLayer's norm + ReLU (in_conv, utae.py:453-520)
 * whose pre-norm output is xh3: the ReLU backward is applied here, out *= [relu_a*xh3 + relu_b > 0], and part gets the
 * statistics of the masked output for that norm's backward.  relu_a alone (any non-null pointer, relu_b and xh3 null; part
 * required): the mask is [x > 0] itself -- x IS the ReLU's output -- no third operand stream, part = (sum out, sum out*x)
 * (in_conv without its pre-norm tensor, uncr_inconv_*). */
int uncr_pw_gemm_dx_supported(int Cin, int Cout);
int uncr_pw_gemm_dx(const void* in, const void* in2, const float* Wt, void* out, const float* k0, const float* k1,
                    const float* k2, const float* kmu, const void* dy, const void* x, const void* xh3, const float* c1,
                    const float* c2, const float* c3, const float* cmu /* out = dy + c1*da + c2*(x - cmu) + c3 */,
                    const float* relu_a, const float* relu_b,
                    const float* relu_mu /* nullable [N*Cout], with relu_a AND relu_b: part = (sum out, sum out*(xh3 - relu_mu)), the
                    centred statistics of that norm's backward (uncr_norm_finalize_bwd centered = 1) */,
                    float* part, int N, int Cin, int Cout, int P, int act,
                    float* amax_out /* [N][uncr_pw_stat_slots] per-block max |out| or null (relu_a == null only) */,
                    const float* in_amax, int in_amax_n, const float* in2_amax, int in2_amax_n /* both or neither, fp32 storage:
                    bounds on |in| and |in2| per frame ([N][n]: per-block maxima or per-plane bounds, e.g. uncr_dw_bwd's amax_out and
                    uncr_norm_finalize_fwd's hb) -- the operand is then staged as two scaled fp16 parts, as in uncr_pw_gemm */,
                    int Pv /* as in uncr_pw_gemm */, hipStream_t stream);
/* wpart [N*nbx][COP][CIP] = the per-block partials of the per-frame products R[n] = sum_p du1n*x (uncr_pw_wgrad with the raw x,
 * its reduction is done here); part_b / part_f = the (sum du1, .) and (sum h1, .) partials [N*Ch][NPB|NPF][2] (part_f may be
 * null: no c2 term); c1..c3 [N*Ch] = norm-1 backward coefficients; A0, B0 [N*C] = PreNorm forward coefficients.
 * -> part0 [N*C][Ch/2][2] = partials of (sum da, sum da*x) for uncr_norm_finalize_bwd (every fp64 tile sum as a value slot and a
 * remainder slot), dW1 [Ch][C]. */
int uncr_prenorm_bwd_finish(const float* wpart, int nbx, int COP, int CIP, const float* W1, const float* part_b, int NPB,
                            const float* part_f, int NPF, const float* c1, const float* c2, const float* c3,
                            const float* cmu /* null: raw form; else part_f is required */, const float* A0,
                            const float* B0, float* part0, float* dW1, int N, int Ch /* % 4 == 0 */, int C /* % 32 == 0 */,
                            int P, const float* xmu /* nullable [N*C]: wpart holds products with x - xmu (a per-plane pivot, e.g. the
                            norm's mean); part0's second component is then the CENTRED sum da*(x - xmu) */, hipStream_t stream);
int uncr_wgrad_shape(int Cd, int Cx, int* cop, int* cip);
/* act: storage of d, d2, x.  bf16: the two wide shapes (256 x 128, 128 x 256) run one bf16 x bf16 product per MAC with fp32
 * accumulation (P % 64 == 0), every other shape the fp32 MFMA kernels.
 * fp32 storage, 128 x 256 with a norm-backward d and an affine + GELU x (MBConv's dW2): when the per-block maxima of |d| and
 * |d2| (d_amax [N][d_amax_n], d2_amax [N][d2_amax_n]: what the producers' amax_out left) and the per-plane bounds on x's affine
 * input (x_ub [N*Cx], uncr_norm_finalize_fwd's ub) are all given, the operands are staged as two row-scaled fp16 parts (three
 * products); with any of them null, as the exact three-part bf16 split (six products). */
int uncr_pw_wgrad(const void* d, const void* d2, const void* x, const void* x2, const float* dk0,
                  const float* dk1, const float* dk2, const float* dkmu /* PRO_NORMBWD on d: mean array or null */,
                  const float* xk0, const float* xk1, const float* xk2,
                  float* part /* [N*NBX][COP][CIP] */, float* rs_part, int N, int Cd, int Cx, int P,
                  int NBX /* blocks (partials) per frame, from uncr_wgrad_nbx */, int pro_d, int pro_x, int act,
                  const float* d_amax, int d_amax_n, const float* d2_amax, int d2_amax_n, const float* x_ub,
                  int Pv /* pixels of a plane that carry data: P, or fewer on the padded planes of an any-size image (fp32 storage) --
                            the kernel then sums the whole 32-pixel chunks below Pv and uncr_wgrad_boundary adds the last Pv % 32 */,
                  hipStream_t stream);
/* the last Pv % 32 pixels of every frame, added to the frame's first partial (and first row-sum partial) of a uncr_pw_wgrad launch with
 * the same operands, prologues and NBX; a no-op when Pv % 32 == 0 */
int uncr_wgrad_boundary(const float* d, const float* d2, const float* x, const float* dk0, const float* dk1, const float* dk2,
                        const float* dkmu, const float* xk0, const float* xk1, const float* xk2, float* part, float* rs_part, int N,
                        int Cd, int Cx, int P, int Pv, int NBX, int pro_d, int pro_x, hipStream_t stream);
int uncr_wgrad_nbx(int N, int Cd, int Cx, int P, int pro_d, int pro_x, int rowsum, int act);
int uncr_wgrad_reduce(const float* part, int n_out, int nblk_per_out, int COP, int CIP, int Cout, int Cin,
                      float* out, hipStream_t stream);

/* ---- depthwise 3x3 reflect (nn.Conv2d groups=C, padding_mode='reflect', uncrtaints.py:130-131) ---- */
/* variant: 0 = automatic (W == 256: the row-streaming kernels; other widths: the LDS-tiled kernels), 1 = LDS-tiled kernels for
 * every width (tests exercise both implementations on the same input) */
int uncr_dw_slots_fwd(int H);
int uncr_dw_slots_bwd(int H);
int uncr_dw_fwd(const void* in, const float* cA, const float* cB, const float* w, void* out, float* part,
                int N, int C, int H, int W, int act, int variant, hipStream_t stream);
/* The same depthwise forward with the train-mode BatchNorm of its INPUT finalised inside the kernel (every wave reduces the
 * N * fin_NP partial pairs of its own channel; csrc/bn_inline.h): replaces uncr_norm_finalize_fwd(kind = BATCH_TRAIN) + uncr_dw_fwd
 * -- one dependent ~5 us launch less per MBConv block (nn.BatchNorm2d in train mode, utae.py:470-473 via uncrtaints.py:16-22,128).
 * fin_part [N*C][fin_NP][2] = (sum, sum^2) partials of `in`; cA / cB [N*C], save_mean / save_rstd [C], ub / hb [N*C] (nullable, hb
 * needs ub) are OUTPUTS with the meaning of uncr_norm_finalize_fwd; running_mean / running_var (both or neither) are updated in
 * place.  Only where uncr_dw_fwd_bn_supported(H, W) (the row-streaming kernel). */
int uncr_dw_fwd_bn_supported(int H, int W);
int uncr_dw_fwd_bn(const void* in, const float* fin_part, int fin_NP, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* cA, float* cB,
                   float* save_mean, float* save_rstd, float* ub, float* hb, const float* w, void* out, float* part,
                   int N, int C, int H, int W, int act, hipStream_t stream);
int uncr_dw_bwd_emits_amax(int H, int W, int act, int variant);
int uncr_dw_bwd(const void* du2, const void* h2, const void* h1, const float* k1, const float* k2,
                const float* k3, const float* kmu /* dh2 = k1*du2 + k2*(h2 - kmu) + k3; null: kmu = 0 */,
                const float* cA1, const float* cB1, const float* w, void* du1, float* part,
                float* dw_part, const float* mean1 /* null: part.y = sum du1*h1; else sum du1*(h1 - mean), the
                well-conditioned form for uncr_norm_finalize_bwd(centered = 1) */,
                int mean_groups /* 0: mean1[c] (BatchNorm); G > 0: mean1[n*G + c/(C/G)] (GroupNorm) */,
                int N, int C, int H, int W, int act, int variant, float* amax_out /* nullable; [N*C][uncr_dw_slots_bwd(H)] max |du1| per statistics slot, only where
                                   uncr_dw_bwd_emits_amax(H, W, act, variant) */,
                hipStream_t stream);
int uncr_dw_wgrad_reduce(const float* dw_part, int N, int C, int NPT, float* dw, hipStream_t stream);

/* ---- any H x W (the reference takes any spatial size, uncrtaints.py:391-447).  For sizes outside the tuned tilings (H*W % 1024, W % 4)
 *      the host layer keeps full-resolution tensors as dense planes of H*W pixels + a ZERO tail up to the stride Pc =
 *      uncr_any_plane_stride(H, W) (0: the size needs none): flat kernels take the valid pixel count (Pv) next to the stride and keep the tail out
 *      of every reduction (uncr_ew, uncr_pw_gemm + uncr_fix_tail, uncr_pw_wgrad + uncr_wgrad_boundary), the 2-D kernels below read and write valid pixels only (csrc/anysize.hip).  fp32 storage. ---- */
int uncr_any_plane_stride(int H, int W);
int uncr_dw_any_slots(int H, int W, int bwd); /* statistics slots (row bands) per plane of uncr_dw_fwd_any (bwd 0) / uncr_dw_bwd_any (1); -1: W too wide */
int uncr_agg_any_slots(int Pc, int C, int NH);      /* ... of uncr_aggregate_any_fwd: Pc / 1024 on the float4 kernels (2, 4, 6, 8, 16, 32 channels per head), 8 on the scalar ones */
int uncr_embed_tail(const float* src /* [planes][P] */, float* dst /* [planes][Pc] */, int planes, int P, int Pc, hipStream_t stream);
int uncr_extract_tail(const float* src /* [planes][Pc] */, float* dst /* [planes][P] */, int planes, int P, int Pc, hipStream_t stream);
/* t [planes][Pc] is the output of a pointwise GEMM launched with Pv = P < Pc: its statistics left out every `unit`-pixel tile that
 * reaches into the tail.  Adds the valid pixels [P / unit * unit, P) of the boundary tile to the slot of the block that owned it --
 * mode 0: (sum t, sum t^2); mode 1: (sum t, sum t*aux), aux [planes][Pc]; mode 2 / part null: nothing -- and zeroes the tail. */
int uncr_fix_tail(float* t, const float* aux, float* part, int slots, int planes, int P, int Pc, int mode,
                  int unit /* uncr_pw_tile_px(Cout) of the producing GEMM */,
                  const float* pivot /* nullable [planes], mode 1: the second component is sum t*(aux - pivot) */, hipStream_t stream);
/* depthwise 3x3 reflect (uncrtaints.py:130-131) with the meaning of uncr_dw_fwd / uncr_dw_bwd on dense H x W planes of stride Pc: row
 * bands staged through LDS on the padded grid, any width up to 3998 (forward) / 2281 (backward), the zero tail of the result written too;
 * part [N*C][uncr_dw_any_slots(H, W, bwd)][2], dw_part [N*C][uncr_dw_any_slots(H, W, 1)][9] */
int uncr_dw_fwd_any(const float* in, const float* cA, const float* cB, const float* w, float* out, float* part, int N, int C, int H,
                    int W, int Pc, hipStream_t stream);
int uncr_dw_bwd_any(const float* du2, const float* h2, const float* h1, const float* k1, const float* k2, const float* k3,
                    const float* kmu, const float* cA1, const float* cB1, const float* w, float* du1, float* part, float* dw_part,
                    const float* mean1, int mean_groups, int N, int C, int H, int W, int Pc, hipStream_t stream);
/* adaptive max-pool (uncrtaints.py:403-404) on planes of stride pstride; idx = flat index inside the H x W image */
int uncr_maxpool_fwd_strided(const float* in, float* out, int* idx, int planes, int H, int W, int pstride, int OH, int OW,
                             hipStream_t stream);
int uncr_maxpool_bwd_strided(const float* dout, const int* idx, float* din, int planes, int H, int W, int pstride, int OH, int OW,
                             hipStream_t stream);
/* temporal aggregation (uncrtaints.py:156-221) with the meaning of uncr_aggregate_fwd / _bwd on planes of stride Pc, any up-sampling
 * ratio; part [B*C][uncr_agg_any_slots(Pc, C, NH)][2]; datt_up: [NH*B*T][Pc] scratch */
int uncr_aggregate_any_fwd(const float* e, const float* att, const int* pad, const float* dmask, unsigned long long seed,
                           const long long* seed_dev, float p_drop, int shared_mask, float* out, float* part, int B, int T, int C,
                           int NH, int H, int W, int Pc, int AH, int AW, hipStream_t stream);
int uncr_aggregate_any_bwd(const float* dg, const float* e, const float* att, const int* pad, const float* dmask,
                           unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask, float* de,
                           float* datt_up, float* datt, int B, int T, int C, int NH, int H, int W, int Pc, int AH, int AW,
                           hipStream_t stream);

/* ---- in_conv = Conv2d(Cin -> Cout, k = 1, bias) + GroupNorm + ReLU (utae.py:453-520 as built at uncrtaints.py:310-314) without
 *      its pre-norm tensor c0 = W x + b (csrc/inconv.hip): with Cin + 1 <= 16 the GroupNorm statistics of c0 are a quadratic form in
 *      the frame's augmented second-moment matrix M~ = sum_p [x;1][x;1]^T, and so are the sums the backward needs.
 *      forward : uncr_inconv_moments -> uncr_inconv_norm_from_moments -> uncr_pw_gemm(epi = 9, e0 = A, e1 = B) writes relu(norm(c0)).
 *      backward: the consumer's uncr_pw_gemm_dx masks with [x > 0] (relu_a alone, no xh3) and leaves (sum du, .) partials;
 *                R = uncr_pw_wgrad(du, x_in) per frame; uncr_inconv_bwd_finish -> dW, db, d gamma, d beta (fp64 algebra). ---- */
int uncr_inconv_moment_blocks(int P);
int uncr_inconv_moments(const void* x /* [N][Cin][pstride], storage `act` */, int N, int Cin, int P /* pixels that carry data */,
                        double* part /* [N][uncr_inconv_moment_blocks(P)][256] */, int act,
                        int pstride /* plane stride; 0 = P; > P: padded planes of an any-size image (zero tail, % 4 == 0) */,
                        hipStream_t stream);
int uncr_inconv_norm_from_moments(const double* part, int nblk, int N, int Cin, int Cout, int groups, const float* W /* [Cout][Cin] */,
                                  const float* bias /* [Cout] or null */, const float* gamma, const float* beta, float eps, float* coefA,
                                  float* coefB /* [N*Cout] */, float* save_mean, float* save_rstd /* [N*groups] */,
                                  double* mom /* [N][256] out: the reduced M~ per frame, row-major 16 x 16 */, hipStream_t stream);
int uncr_inconv_bwd_finish(const float* R /* [N][Cout][Cin] = sum_p du x^T */, const float* part /* [N*Cout][NP][2]: .x = sum_p du */,
                           int NP, const double* mom, const float* W, const float* bias, const float* gamma, const float* save_mean,
                           const float* save_rstd, int N, int Cin, int Cout, int groups, float* dW /* [Cout][Cin] */,
                           float* db /* [Cout] or null */, float* dgamma, float* dbeta, hipStream_t stream);
/* the same behind a train-mode BatchNorm (encoder_norm = 'batch'): statistics per channel over ALL frames from the sum of the frames'
 * moment matrices, fp64 (+ the running statistics' update); mom [N][256] scratch, momtot [256] out (kept for the backward).
 * groups = Cout in uncr_inconv_norm_from_moments / uncr_inconv_bwd_finish serves InstanceNorm (a constant plane -- a padded date --
 * gets A = 0, B = beta: the exact result). */
int uncr_inconv_bn_from_moments(const double* part, int nblk, int N, int Cin, int Cout, const float* W, const float* bias,
                                const float* gamma, const float* beta, float* running_mean /* nullable, with running_var */,
                                float* running_var, float momentum, float eps, float* coefA /* [N*Cout] */, float* coefB,
                                float* save_mean /* [Cout] */, float* save_rstd, double* mom, double* momtot, hipStream_t stream);
int uncr_inconv_bwd_finish_bn(const float* R /* [N][Cout][Cin] */, const float* part /* [N*Cout][NP][2]: .x = sum_p du */, int NP,
                              const double* momtot, const float* W, const float* bias, const float* gamma, const float* save_mean,
                              const float* save_rstd, int N, int Cin, int Cout, float* dW, float* db /* nullable */, float* dgamma,
                              float* dbeta, hipStream_t stream);

/* ---- squeeze-excite MLP (uncrtaints.py:82-97) ---- */
int uncr_se_mlp_fwd(const float* pool_part, int NP, int N, int C, int R, int P, const float* W1,
                    const float* W2, float* pooled, float* hid_pre, float* s, hipStream_t stream);
int uncr_se_mlp_bwd(const float* G, const float* Wpw, int N, int Co, int C, int R, int P, const float* W1,
                    const float* W2, const float* s, const float* pooled, const float* hid_pre, float* ds_pre,
                    float* dhid_pre, float* dpool_px, float* dWpw, float* dW1, float* dW2, hipStream_t stream);
/* dWpw / dW1 / dW2 all null: only the per-frame phase runs (ds_pre, dhid_pre, dpool_px); the weight gradients are then taken by
 * uncr_mbconv_param_grads together with the depthwise weight gradient -- ONE launch for the parameter-gradient reductions of an MBConv
 * backward that sit on no critical path (dw_part: uncr_dw_bwd's per-tile partials [N*Cdw][NPT][9] -> dwdw [Cdw][9], as
 * uncr_dw_wgrad_reduce). */
int uncr_mbconv_param_grads(const float* G, int N, int Co, int C, int R, const float* s, const float* pooled,
                            const float* hid_pre, const float* ds_pre, const float* dhid_pre, float* dWpw, float* dW1,
                            float* dW2, const float* dw_part, int Cdw, int NPT, float* dwdw, hipStream_t stream);

/* ---- L-TAE low-resolution branch (uncrtaints.py:403-404 max-pool; ltae.py:197-239 LTAE2dtiny;
 *      positional_encoding.py:5-31; ltae.py:341-385,431-458 attention) ---- */
int uncr_pad_mask(const float* x, int NF, long long frame_elems, float pad_value, int* mask,
                  hipStream_t stream);   /* pad-frame detection, uncrtaints.py:392-394 */
int uncr_maxpool_fwd(const void* in, float* out, int* idx, int planes, int H, int W, int OH, int OW, int act,
                     hipStream_t stream);
/* MBConv's closing residual y = x + A*h3 + B (uncrtaints.py:146) of the LAST encoder block with the L-TAE stage's
 * AdaptiveMaxPool2d((32,32)) (uncrtaints.py:403-404) taken on the fly: out [planes][H][W], down / idx [planes][OH][OW]
 * (same scan-order / NaN semantics as uncr_maxpool_fwd), part [planes][uncr_residual_pool_slots(H)][2] = (sum, sum^2)
 * or null.  Built for W == 256 with 8x8 windows (uncr_residual_pool_supported); other shapes use uncr_ew + uncr_maxpool_fwd. */
int uncr_residual_pool_supported(int H, int W, int OH, int OW);
int uncr_residual_pool_slots(int H);
int uncr_residual_pool(const void* x, const void* h3, const float* cA, const float* cB, void* out, float* part,
                       float* down, int* idx, int planes, int H, int W, int OH, int OW, int act, hipStream_t stream);
int uncr_maxpool_bwd(const float* dout, const int* idx, void* din, int planes, int H, int W, int OH, int OW, int act,
                     hipStream_t stream);
/* uncr_maxpool_bwd fused with the (sum de, sum de*h3) statistics pass of the last encoder block's backward: one pass over de and
 * h3 (the h3 of the block that produced the pooled tensor); part [planes][P/1024][2].  Disjoint windows, (W/OW) % 4 == 0. */
int uncr_pool_scatter_stats_supported(int H, int W, int OH, int OW);
int uncr_pool_scatter_stats(const float* dpool, const int* idx, void* de, const void* h3, float* part, int planes, int H, int W,
                            int OH, int OW, int act,
                            float* amax_out /* null, or [planes][P/1024]: per-block max |de| */, hipStream_t stream);
int uncr_ltae_gn_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean,
                     float* rstd, int B, int T, int C, int G, int S, hipStream_t stream);
int uncr_ltae_gn_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                     float* dx, float* gb_part, int B, int T, int C, int G, int S, hipStream_t stream);
int uncr_colsum(const float* part, int R, int K, float* out, hipStream_t stream);
/* the same for `batches` consecutive [R][K] arrays in one launch: out [batches][K] */
int uncr_colsum_batched(const float* part, int batches, int R, int K, float* out, hipStream_t stream);
/* agg_mode 'att_mean' (head-averaged attention, uncrtaints.py:179-188,211-219) and 'mean' (:189-192,220-221) helpers */
int uncr_bcast_scale(const float* src, int R, long long n, float scale, float* dst, hipStream_t stream);
int uncr_mean_weights(const int* pad, int NH, int B, int T, int S, float* out, hipStream_t stream);
int uncr_ltae_posbias(const float* dates, const float* denom, int d, const float* bin, float* out, int NF, int D,
                      int use_pe, hipStream_t stream);
int uncr_ltae_softmax_fwd(const float* k, const float* Q, const int* pad, float* att, int B, int T, int NH,
                          int DK, int S, hipStream_t stream);
int uncr_ltae_softmax_bwd(const float* datt, const float* att, const float* k, const float* Q, const int* pad,
                          float* dk, float* dq_part, int B, int T, int NH, int DK, int S, hipStream_t stream);

/* ---- LTAE2dtiny (ltae.py:145-239) fused per low-resolution pixel: GroupNorm over (C/NH channels x T dates), inconv, positional
 *      encoding, fc1_k, query product and the masked temporal softmax in ONE kernel per direction.  Everything between the
 *      GroupNorm and the softmax is linear, so the score is one linear functional of the normalised input per head:
 *      score = A' . xhat + B' with A' [NH][C], B' [NH][B*T] composed from the parameters by uncr_ltae_compose (fp64, one launch;
 *      also returns bias1 [B*T][D] = b_i + PE (dates / denom as in uncr_ltae_posbias; use_pe = 0: b_i alone), M = Wk Wi
 *      [NH*DK][C] and U = Wk bias1 + b_k [NH*DK][B*T] for the backward).  uncr_ltae_fused_bwd returns d(pooled features) and
 *      per-block partials of d A' ([B*nblk][NH][C], nblk = S*NH/256 blocks per sample) and d B' ([B*nblk][NH][T]);
 *      uncr_ltae_compose_bwd reduces them and applies the chain rule -> d Q, d fc1_k, d inconv, d gamma / d beta (gb [2][C]);
 *      scratch: NH*C + NH + B*T*NH + NH*2*C floats. ---- */
int uncr_ltae_fused_supported(int T, int C, int NH, int S);
int uncr_ltae_compose(const float* Q, const float* Wk, const float* bk, const float* Wi, const float* bin /* b_i or null */,
                      const float* dates /* [NF] */, const float* denom /* [dpe] */, int dpe, int use_pe, const float* gamma,
                      const float* beta, int NH, int DK, int D, int C, int NF, float* bias1, float* Ap, float* Bp, float* M,
                      float* U, hipStream_t stream);
int uncr_ltae_fused_fwd(const float* x, const float* Ap, const float* Bp, const int* pad, float eps, float* att, float* mean,
                        float* rstd, int B, int T, int C, int NH, int S, hipStream_t stream);
int uncr_ltae_fused_bwd(const float* datt, const float* att, const float* x, const float* Ap, const int* pad,
                        const float* mean, const float* rstd, float* dx, float* partA, float* partB, int B, int T, int C,
                        int NH, int S, hipStream_t stream);
int uncr_ltae_compose_bwd(const float* Q, const float* Wk, const float* Wi, const float* bias1, const float* gamma,
                          const float* beta, const float* M, const float* U, const float* partA, const float* partB, int nblk,
                          int NH, int DK, int D, int C /* <= 256 */, int NF, int T, float* scratch, float* dQ, float* dWk,
                          float* dbk, float* dWi, float* dbi, float* gb, hipStream_t stream);

/* ---- full-resolution temporal aggregation (Compact_Temporal_Aggregator 'att_group',
 *      uncrtaints.py:156-221: bilinear up-sample + dropout + pad mask + V-aggregate) ---- */
int uncr_agg_slots(int P);
int uncr_aggregate_fwd(const void* e, const float* att, const int* pad, const float* dmask,
                       unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask,
                       void* out, float* part, int B, int T, int C, int NH, int H, int W, int AH, int AW,
                       int act /* storage of e and out */, hipStream_t stream);
int uncr_aggregate_bwd(const void* dg, const void* e, const float* att, const int* pad, const float* dmask,
                       unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask,
                       void* de, float* datt_up, float* datt, int B, int T, int C, int NH, int H, int W, int AH,
                       int AW, int act /* storage of dg, e and de */, hipStream_t stream);
/* The same backward in two passes, for the case where the statistics pass of the block that produced e follows anyway (autograd of
 * uncrtaints.py:403-412: the gradient of e is the aggregator's de PLUS the max-pool's scatter of d(down)):
 *   uncr_aggregate_bwd_datt: the attention's gradient alone (reads dg and e, writes nothing at full resolution);
 *   uncr_aggregate_bwd_de:   de = a * dg + [pooled gradient dpool at the arg-max pidx of the OH x OW max-pool of e], written once,
 *                            with the (sum de, sum de*h3) partials `part` ([B*T*C][P / 1024] float2) and per-block maxima `amax` of
 *                            the producing block's norm-3 backward.  dpool / pidx, h3 / part and amax may be null (no scatter / no
 *                            statistics).  Windows disjoint with a width that is a multiple of 4 (uncr_aggregate_bwd_de_supported). */
int uncr_aggregate_bwd_datt(const void* dg, const void* e, const float* att, const int* pad, const float* dmask,
                            unsigned long long seed, const long long* seed_dev, float p_drop, int shared_mask,
                            float* datt_up, float* datt, int B, int T, int C, int NH, int H, int W, int AH, int AW,
                            int act, hipStream_t stream);
int uncr_aggregate_bwd_de_supported(int H, int W, int OH, int OW);
int uncr_aggregate_bwd_de(const void* dg, const float* att, const int* pad, const float* dmask, unsigned long long seed,
                          const long long* seed_dev, float p_drop, int shared_mask, void* de, const float* dpool,
                          const int* pidx, const void* h3, float* part, float* amax, int B, int T, int C, int NH, int H, int W,
                          int AH, int AW, int OH, int OW, int act, hipStream_t stream);
/* the aggregator's AvgPool branch (uncrtaints.py:197-204: feature map not larger than the attention map): the attention is
 * average-pooled with kernel = stride = k (= AW / H in the reference) to the feature map's size, no dropout; fp32.
 * Requires AH / k == H and AW / k == W.  datt [NH][B][T][AH][AW] (cells the pooling never reads get 0). */
int uncr_aggregate_pool_fwd(const float* e, const float* att, const int* pad, float* out, int B, int T, int C, int NH,
                            int H, int W, int AH, int AW, int k, hipStream_t stream);
int uncr_aggregate_pool_bwd(const float* dg, const float* e, const float* att, const int* pad, float* de, float* datt,
                            int B, int T, int C, int NH, int H, int W, int AH, int AW, int k, hipStream_t stream);

/* ---- dense 3x3 reflect convolution of ResidualConvBlock (uncrtaints.py:24-69, utae.py:478-487) as nine accumulating
 *      pointwise GEMMs on the padded grid (uncr_pw_gemm epi 4 with the input pointer shifted by dy*(W+2)+dx).  Glue:
 *      padding with the producing norm+ReLU (pro 4) / norm-backward (pro 3) fused in, un-padding with the next norm's
 *      statistics, adjoint of the reflect padding.  Padded tensors: [planes][uncr_conv3_plane_stride] floats with
 *      uncr_conv3_margin floats of (zero) slack before and after the whole tensor. ---- */
int uncr_conv3_plane_stride(int H, int W);
int uncr_conv3_margin(int W);
int uncr_pad2d(const float* src, const float* src2, float* dst, const float* k0, const float* k1, const float* k2,
               const float* kmu, int pro, int mode /* 0 reflect, 1 zero */, int planes, int H, int W, hipStream_t stream);
int uncr_unpad2d(const float* src, float* dst, float* part /* [planes][H*W/1024][2] or null */, int planes, int H,
                 int W, hipStream_t stream);
int uncr_unpad2d_reflect_adjoint(const float* src, float* dst, int planes, int H, int W, hipStream_t stream);
/* the same three on the dense planes of an any-size image: plane stride Ps (a multiple of 1024 >= H*W) on the un-padded side, whose
 * tail is never read and is written as zeros; part [planes][Ps/1024][2] */
int uncr_pad2d_strided(const float* src, const float* src2, float* dst, const float* k0, const float* k1, const float* k2,
                       const float* kmu, int pro, int mode, int planes, int H, int W, int Ps, hipStream_t stream);
int uncr_unpad2d_strided(const float* src, float* dst, float* part, int planes, int H, int W, int Ps, hipStream_t stream);
int uncr_unpad2d_reflect_adjoint_strided(const float* src, float* dst, int planes, int H, int W, int Ps, hipStream_t stream);

/* ---- use_v variant (uncrtaints.py:324-338,414-417; LTAE2d ltae.py:10-141): pieces that are not already covered by the
 *      entry points above.  include_v(cat(g, up(v))) = Wa*g + up(Wv*v + b), so only uncr_add_upsampled touches full
 *      resolution; the attention-weighted values reuse uncr_aggregate_fwd/bwd at H == AH. ---- */
int uncr_add_upsampled(const float* a, const float* z, float* out, float* part /* [planes][P/1024][2] or null */,
                       int planes, int H, int W, int AH, int AW, hipStream_t stream);
int uncr_bilinear_adjoint(const float* src, float* dst, int planes, int H, int W, int AH, int AW, hipStream_t stream);
/* the same pair on any-size planes (plane stride Pc >= H*W, tail written as zeros; part [planes][uncr_agg_any_slots()][2]) */
int uncr_add_upsampled_any(const float* a, const float* z, float* out, float* part, int planes, int H, int W, int Pc, int AH, int AW,
                           hipStream_t stream);
int uncr_bilinear_adjoint_any(const float* src, float* dst, int planes, int H, int W, int Pc, int AH, int AW, hipStream_t stream);
int uncr_add(const float* a, const float* b, float* out, long long n, hipStream_t stream);
int uncr_dropout(const float* a, float* out, long long n, unsigned long long seed, const long long* seed_dev, float p,
                 hipStream_t stream);

/* ---- the attention classes called on their own (ltae.py:388-458 ScaledDotProductAttention[Small], :244-307 / :312-385
 *      MultiHeadAttention[Small]): pixel-major rows q [q_rows][dk] (q_rows == m, or one query shared by each group of m / q_rows
 *      consecutive rows), k [m][T][dk], v [m][T][dv], pad [m][T] (non-zero = masked_fill(-1e3)); T <= 64.
 *      attn_sm = softmax_t(q.k / temperature) (saved for the backward), attn_out = the same after dropout p_drop (counter-based
 *      stream of (seed, seed_dev), nullable), out = attn_out @ v (nullable), comp = the masked, scaled scores (nullable).
 *      Backward: any of dattn / dout / dcomp may be null; dq_rows [m][dk] is per row (shared queries: sum the groups). ---- */
int uncr_sdpa_rows_fwd(const float* q, int q_rows, const float* k, const float* v, const int* pad, float temperature,
                       float* attn_sm, float* attn_out, float* out, float* comp, int m, int T, int dk, int dv, float p_drop,
                       unsigned long long seed, const long long* seed_dev, hipStream_t stream);
int uncr_sdpa_rows_bwd(const float* dattn, const float* dout, const float* dcomp, const float* q, int q_rows, const float* k,
                       const float* v, const int* pad, const float* attn_sm, float temperature, float* dq_rows, float* dk_out,
                       float* dv_out, int m, int T, int dk, int dv, float p_drop, unsigned long long seed,
                       const long long* seed_dev, hipStream_t stream);
/* dst [cols][dst_ld] = src [rows][cols] transposed, columns >= rows zero-filled: pixel-major rows <-> the channel-major planes of
 * uncr_pw_gemm (nn.Linear on rows = a 1x1 convolution on the transposed tensor, dst_ld = rows padded to the GEMM's pixel tile) */
int uncr_transpose2d(const float* src, float* dst, int rows, int cols, int dst_ld, hipStream_t stream);

/* ---- input assembly in front of the path: prepare_data_multi (model/train_reconstruct.py:161-179) stacks the
 *      per-date S1 [B,2,H,W] / S2 [B,13,H,W] tensors into x [B,T,C,H,W] (S1 channels first); with kind != 0 the
 *      loader's process_MS / process_SAR (data/dataLoader.py:38-61: clip, rescale, nan_to_num) is applied on the way.
 *      desc (DEVICE memory) = T*ngroups x 4 int64 {src pointer, channels, channel offset in x, kind}; kind: 0 copy,
 *      1 MS 'default', 2 MS 'resnet', 3 SAR 'default', 4 SAR 'resnet'. ---- */
int uncr_assemble_input(const long long* desc, float* x, int B, int T, int C, int P, int ngroups, hipStream_t stream);

/* ---- evaluation metrics right behind the path: img_metrics (model/src/learning/metrics.py:20-63) with the SSIM of
 *      util/pytorch_ssim/__init__.py:17-73.  target / pred / var [B][C][H][W]; win = the 11x11 window (121 floats, built
 *      by the caller exactly like create_window); out[0..8] = RMSE, MAE, PSNR, SAM, SSIM, nanmean error / ae / se / var,
 *      out[16 + b] = per-item SSIM; pixelwise (optional) [4][H*W] = x.nanmean(0).nanmean(0) of error, ae, se, var. ---- */
int uncr_img_metrics_work(int B, int C, int H, int W);
int uncr_img_metrics(const float* target, const float* pred, const float* var, const float* win, float* out,
                     float* pixelwise, float* work, int B, int C, int H, int W, hipStream_t stream);

/* ---- element-wise criteria of get_loss (losses.py:14-32): kind 0 GaussianNLLLoss (losses.py:46-128: var clamped to
 *      eps with identity gradient, optional 0.5*log(2 pi)), 1 nn.L1Loss, 2 nn.MSELoss.  The backward needs var of the
 *      full shape (the host expands a broadcast one); `inner` > 1 in the forward = var broadcast over the innermost
 *      `inner` elements. ---- */
int uncr_eltloss_blocks(long long n);
int uncr_eltloss_fwd(int kind, const float* pred, const float* targ, const float* var, float* loss_none,
                     float* vclamp, float* part /* [uncr_eltloss_blocks(n)] */, float* loss_out, int* neg_flag,
                     long long n, int inner, float eps, int full, int reduction, hipStream_t stream);
int uncr_eltloss_bwd(int kind, const float* pred, const float* targ, const float* var, const float* gscalar,
                     const float* gnone, float* dpred, float* dvar, long long n, float eps, int reduction,
                     hipStream_t stream);

/* ---- MGNLL loss (losses.py:131-218) and ensemble combine (ensemble_reconstruct.py:116-133) ---- */
int uncr_mgnll_blocks(int P);
/* vclamp (nullable) [B][K][P]: the clamped per-band variance max(var, eps) (iso: the one channel broadcast to K) -- what the
 * reference returns as the diagonal of its second result (losses.py:145,203-211). */
/* *_bstride: elements between consecutive samples of pred / var / dpred / dvar (0 = dense, K*H*W resp. Kv*H*W): the mean and the
 * variance may be channel slices of the head's [B, 13 + cov, H, W] output, read in place, and both gradients may be written
 * straight into the channel slices of one buffer of that shape (no slicing copies either way). */
int uncr_mgnll_fwd(const float* pred, const float* targ, const float* var, float* loss_none, float* vclamp, float* part,
                   float* loss_out, int* neg_flag, int B, int K, int Kv, int H, int W, float eps, int reduction,
                   long long pred_bstride, long long var_bstride, hipStream_t stream);
int uncr_mgnll_bwd(const float* pred, const float* targ, const float* var, const float* gscalar,
                   const float* gnone, float* dpred, float* dvar, int B, int K, int Kv, int H, int W, float eps,
                   int reduction, long long pred_bstride, long long var_bstride, long long dpred_bstride,
                   long long dvar_bstride, hipStream_t stream);
int uncr_ensemble_combine(const float* mu, const float* var, int M, long long n, int mode, float* mu_out,
                          float* var_out, hipStream_t stream);

/* ---- the optimizer step of the reference's train step (base_model.py:48,60-131: torch.optim.Adam(params, lr), default betas and eps)
 *      for up to uncr_adam_max_tensors() tensors in one launch.  desc (device) [n_tensors][5] int64 = {param, exp_avg, exp_avg_sq
 *      addresses, element count, address of the tensor's own step counter}; grads_host: HOST array [n_tensors] of the gradients' device addresses (they change every step; they
 *      travel by value in the kernel arguments, so a captured graph holds them without a copy node); chunks (device) [n_chunks][2]
 *      int32 = {tensor index, first element}, one block each, uncr_adam_chunk() elements per chunk.
 *      step counter: one device float PER TENSOR (torch.optim.Adam keeps a step per parameter; they diverge when layers are unfrozen
 *      later, train_reconstruct.py:657-660) holding the count t ALREADY incremented for this update; lr_dev (nullable): device float that
 *      overrides lr.  Arithmetic as torch.optim.Adam (L2 weight decay, no amsgrad): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
 *      p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). ---- */
int uncr_adam_chunk(void);
int uncr_adam_max_tensors(void);
int uncr_adam_step(const long long* desc, const long long* grads_host, int n_tensors, const int* chunks, int n_chunks, float lr,
                   const float* lr_dev, double beta1, double beta2, float eps, float weight_decay, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNCR_HIP_H */
