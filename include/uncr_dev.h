/* libuncr_dev -- development probes of the MI355X kernels (NOT the product ABI; see include/uncr_hip.h for that).
 * Same conventions: plain C types, caller-owned buffers, enqueue on the passed stream, 0 = success. */
#ifndef UNCR_DEV_H
#define UNCR_DEV_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* hipStream_t;

int uncr_debug_mfma_probe(float* out, int blocks, int iters, hipStream_t stream);   /* fp32-MFMA peak probe */
/* debug: y = erf_f(x) (what 0), gelu_f (1), gelu_grad_f (2), raw v_exp_f32 2^x (3) -- accuracy probes */
int uncr_debug_erf(const float* x, float* y, int n, int what, hipStream_t stream);
int uncr_debug_mfma_probe_bf16(float* out, int blocks, int iters, hipStream_t stream);   /* bf16-MFMA peak probe */
/* debug: out[32][32] = A[32][K] * B[K][32] through the 3-way bf16 split on v_mfma_f32_32x32x16_bf16
 * (terms = 1, 3, 6 or 9 partial products); numerics probe, K % 16 == 0 */
/* ds_read_b64_tr_b16 semantics probe: lds[i] = i (16-bit); lane l reads at element offset offs[l] (64 ints);
 * out[l*4 + j] = j-th value received (256 ints). */
int uncr_debug_tr_b16_probe(const int* offs, int* out, hipStream_t stream);
int uncr_debug_bf16split_probe(const float* A, const float* B, float* out, int K, int terms, hipStream_t stream);
/* HBM stream roofs: mode 0 read-only, 1 write-only, 2 copy, 3 two reads : one write, 4 one read : two writes, 6-8 the GEMMs' tiled pattern, 5 three reads :
 * one write; float4 lanes, contiguous slab per block, nt = non-temporal accesses.  a, b, c are read, x, y written, n_floats each. */
int uncr_debug_stream_probe(const float* a, const float* b, const float* c, float* x, float* y, long long n_floats,
                            int mode, int nt, int blocks, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNCR_DEV_H */
