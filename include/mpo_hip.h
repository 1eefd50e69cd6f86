/* libmpo_hip.so -- C ABI of the MI355X (gfx950) WSI-patch x omics fusion kernels.
 *
 * The reference (mattiagualtieri/multimodal-path-omic) is pure Python on stock PyTorch and has no
 * FFI of its own; its seam is the nn.Module attribute slots of the two models
 * (models/mcat/mcat.py:48-82, models/nacagat/nacagat.py:44-78).  Each entry point below replaces the
 * arithmetic behind one of those slots and is what a binding for that slot would call
 * (INTEGRATION.md shows the ctypes stubs).  Conventions, all entries:
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch allocates inputs, outputs, saved tensors and workspaces; the library allocates nothing
 *     persistent and keeps no pointer after returning);
 *   - asynchronous on the hipStream_t passed last; the compute entries are re-entrant and keep no state of their
 *     own.  Process-global state of the library: the thread-local last-error string, and the kernel SELECTORS
 *     mpo_set_* below (diagnostic switches that pick between two kernels computing the same values; they exist so
 *     that the tests can hold each special kernel against the general one).  A selector is read once per call, on
 *     the calling thread, before any launch: flipping one while another thread is inside a compute entry changes
 *     which kernel that call uses, never its result beyond the tolerance the two kernels are tested to.  Production
 *     code leaves them at their defaults;
 *   - returns 0 on success, non-zero on error (never throws); mpo_last_error() describes it;
 *   - row-major fp32 unless a dtype argument says otherwise; "bag" tensors may be fp32 or bf16
 *     (MPO_F32 / MPO_BF16): bf16 is a STORAGE format, accumulation is always fp32;
 *   - a window of n_slides slides is processed per call.  Bags are concatenated along rows
 *     ("ragged"): slide b owns rows cu_rows[b] .. cu_rows[b+1]-1 of the bag; cu_rows is a DEVICE
 *     int32 array of n_slides+1 entries; max_rows = longest bag and total_rows = cu_rows[n_slides]
 *     are passed from the host for grid sizing.  n_slides = 1 is the reference's per-slide call.
 *   - attention maps are ragged too: slide b's (n_q, M_b) block starts at float offset
 *     n_q * cu_rows[b], row stride M_b.
 */
#ifndef MPO_HIP_H
#define MPO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mpo_stream_t;   /* == hipStream_t */

enum { MPO_F32 = 0, MPO_BF16 = 1 };

/* Work plan of the bag passes over a ragged window (optional; NULL = every slide is cut into the same number of row
 * ranges, mpo_coattn_splits()).  With a plan every workgroup gets `rows_per_wg` rows (a multiple of 32) and slide b owns
 * workgroups wg_start[b] .. wg_start[b+1]-1 = ceil(M_b / rows_per_wg) of them: work proportional to the bag length, so a
 * window of 2k..30k-patch bags is balanced.  Choose rows_per_wg = round_up_32(ceil(total_rows / T)) with
 * T = mpo_coattn_target_workgroups() (one workgroup per CU); n_wg = wg_start[n_slides] <= T + n_slides.
 * wg_start is a DEVICE int32 array of n_slides + 1 entries; the struct itself is host memory. */
typedef struct mpo_bag_plan {
    const int32_t* wg_start;
    int32_t n_wg;
    int32_t rows_per_wg;
} mpo_bag_plan;
int mpo_coattn_target_workgroups(void);
enum { MPO_ACT_NONE_ = 0, MPO_ACT_RELU_ = 1, MPO_ACT_ELU_ = 2, MPO_ACT_TANH_ = 3, MPO_ACT_SIGMOID_ = 4 };

int mpo_abi_version(void);
const char* mpo_last_error(void);
/* Every entry launches on the caller's stream only: the library owns no stream and no event. */

/* ---- building block: y = act(alpha * (x W^T + b)) and its two backward products, on the fp32 MFMA.
 * Stands in for torch.nn.functional.linear on the 6 x 256-token tail. */
int mpo_linear_forward(const float* x, const float* weight, const float* bias, float* y,
                       int rows, int in_features, int out_features, float alpha, int act, mpo_stream_t stream);
int mpo_linear_backward_input(const float* dy, const float* weight, float* dx,
                              int rows, int in_features, int out_features, float alpha, int accumulate,
                              mpo_stream_t stream);
int mpo_linear_backward_weight(const float* dy, const float* x, float* dweight, float* dbias /* nullable */,
                               int rows, int in_features, int out_features, float alpha, mpo_stream_t stream);

/* ---- K1: MCAT genomic-guided co-attention = nn.MultiheadAttention(embed, num_heads=1)(query, bag, bag)
 * Replaces models/mcat/mcat.py:48 (constructor) / :97 (call); arithmetic of
 * torch/nn/functional.py:6206-6660 (packed in-projection, 1/sqrt(E) scaling, softmax, out_proj).
 *   query        [n_slides*n_q, embed] fp32        in_proj_weight [3*embed, embed], in_proj_bias [3*embed]
 *   out          [n_slides*n_q, embed] fp32        out_proj_weight [embed, embed],  out_proj_bias [embed]
 *   attn_map     NULL (need_weights=False) or ragged fp32 map (need_weights=True)
 *   saved        mpo_coattn_saved_floats() floats kept for the backward call
 */
size_t mpo_coattn_saved_floats(int n_slides, int n_q, int embed);
size_t mpo_coattn_workspace_bytes(int n_slides, int n_q, int embed, int max_rows);
int mpo_coattn_mcat_forward(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides,
                            int total_rows, int max_rows,
                            const float* query, int n_q, int embed,
                            const float* in_proj_weight, const float* in_proj_bias,
                            const float* out_proj_weight, const float* out_proj_bias,
                            float* out, float* attn_map, float* saved, const mpo_bag_plan* plan /* nullable */,
                            void* workspace, size_t workspace_bytes, mpo_stream_t stream);
/* ---- row f1: the patch layer fused with K1's forward, ONE pass over the raw patch matrix.
 * Replaces models/mcat/mcat.py:24-29,87 (self.H = Linear(1024, 256) + ReLU + Dropout) AND :97 (the co-attention call) for a
 * bf16-stored window:  h_bag = Dropout_p(ReLU(patches W_H^T + b_H))  is produced tile by tile on the MFMA, consumed for
 * the scores / online softmax / context while it is still in LDS, and written once (bf16, [total_rows, embed]) for
 * mpo_coattn_mcat_backward, which takes `saved` and h_bag exactly as after mpo_coattn_mcat_forward.
 *   patches [total_rows, patch_dim] bf16;  patch_weight [embed, patch_dim] fp32 (rounded to bf16 operands inside),
 *   patch_bias [embed] fp32.  Built for patch_dim 1024, embed 256 ('medium'), n_q <= 8.
 * Dropout: counter hash of (seed, offset [+ *rng_epoch << 40]), one draw per 16 elements, 8 random bits each: the realised drop
 * probability is round(256 p) / 256 (exact for the reference's 0.25) and the keep scale follows it; reserve
 * total_rows * embed / 16 + 1 counters. */
size_t mpo_patch_coattn_workspace_bytes(int n_slides, int n_q, int embed, int patch_dim);
int mpo_patch_coattn_mcat_forward(const void* patches, const int32_t* cu_rows, int n_slides, int total_rows, int max_rows,
                                  int patch_dim, const float* patch_weight, const float* patch_bias, float drop_p,
                                  uint64_t seed, uint64_t offset, const uint64_t* rng_epoch /* nullable */,
                                  const float* query, int n_q, int embed,
                                  const float* in_proj_weight, const float* in_proj_bias,
                                  const float* out_proj_weight, const float* out_proj_bias,
                                  void* h_bag, float* out, float* attn_map /* nullable */, float* saved,
                                  const mpo_bag_plan* plan /* nullable */, void* workspace, size_t workspace_bytes,
                                  mpo_stream_t stream);
/* The patch layer alone: h_bag [total_rows, embed] bf16 = dropout(relu(patches W^T + b)) (models/mcat/mcat.py:24-29,87), one
 * pass of the fused kernel with its co-attention slices off (NaCAGaT needs H_bag for more than one product; MCAT outside
 * the fused configuration).  Same dropout stream and realised rate as mpo_patch_coattn_mcat_forward.
 * patch_dim 1024; embed 128 / 256 / 512 = model_size small / medium / big (models/mcat/mcat.py:16-21) on the one kernel:
 * 128 as a 256-column block whose upper half is not stored, 512 as one pass per 256-column half.  Other widths: error. */
size_t mpo_patch_fc_workspace_bytes(int embed, int patch_dim);
int mpo_patch_fc_forward(const void* patches, const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int patch_dim,
                         const float* patch_weight, const float* patch_bias, int embed, float drop_p, uint64_t seed,
                         uint64_t offset, const uint64_t* rng_epoch, void* h_bag, const mpo_bag_plan* plan /* nullable */,
                         void* workspace, size_t workspace_bytes, mpo_stream_t stream);
/* The patch layer of an fp32-stored window (models/mcat/mcat.py:24-29,87 with fp32 patch features; ABI v11, x_scale: v13), 1024 -> 256:
 *   forward   H_bag = Dropout(ReLU(X W^T + b)) in fp32 storage; products as three fp16 MFMA terms of hi / lo operand splits with fp32
 *             accumulation (csrc/patch_fc_f32.hip).  Dropout: counter hash, 8 bits per element (realised p = round(256 p) / 256);
 *             the mask lives in H_bag as zeros.  x_scale: a power of two applied to X before its split and taken off the
 *             accumulator (exact); pass 2^(15 - e) for max |X| = m 2^e, m in [0.5, 1) (ops.patch_fc_f32 derives and caches it
 *             per tensor) so that the features use fp16's range; 1.0 = unscaled (|x| in ~[1e-3, 1.3e5] then).  W_H is scaled
 *             by its own maximum inside the call.  Non-finite features stay non-finite in their rows of H_bag.
 *   backward  d_weight = g^T X, d_bias = colsum(g) with g = d_h_bag (.) [H_bag > 0] * gate; gate = 1 / (1 - realised p)
 *             (1 without dropout); h_bag NULL: g = d_h_bag.  X needs no gradient (it is data).
 * workspace: caller-owned, mpo_patch_fc_f32_workspace_bytes(backward) bytes. */
size_t mpo_patch_fc_f32_workspace_bytes(int backward);
int mpo_patch_fc_f32_forward(const float* patches, int64_t total_rows, int patch_dim, const float* patch_weight,
                             const float* patch_bias, int embed, float drop_p, uint64_t seed, uint64_t offset,
                             const uint64_t* rng_epoch, float x_scale, float* h_bag, void* workspace, size_t workspace_bytes,
                             mpo_stream_t stream);
int mpo_patch_fc_f32_backward(const float* d_h_bag, const float* h_bag /* nullable */, const float* patches, int64_t total_rows,
                              int embed, int patch_dim, float gate, float* d_weight, float* d_bias, void* workspace,
                              size_t workspace_bytes, mpo_stream_t stream);

/* The fused bag pass alone (measurement): w_packed = embed * patch_dim bf16 values from mpo_pack_patch_weight (the weight
 * in the fragment order of the kernel's GEMM waves; one 512-KiB block per 256 rows of W_H, embed 128: one block whose upper
 * half is zero), qk2 [n_slides*n_q, embed].  The bag pass is the embed-256 form. */
int mpo_patch_coattn_fwd_bagpass(const void* patches, const void* w_packed, const float* bias, const int32_t* cu_rows, int n_slides,
                                 const float* qk2, void* h_bag, float* part_ml, float* part_ctx, int n_q, int max_rows,
                                 float drop_p, uint64_t seed, uint64_t offset, const mpo_bag_plan* plan /* nullable */,
                                 mpo_stream_t stream);
int mpo_pack_patch_weight(const float* weight /* [embed, patch_dim] */, void* packed, int embed, int patch_dim, mpo_stream_t stream);

/* d_attn_map (nullable): gradient arriving on the returned map; needs attn_map from the forward.
 * d_bag has the bag's dtype.  d_in_proj_bias[embed..2*embed) (the key bias) is exactly zero: a key
 * bias shifts every logit of a row equally and cancels in the softmax. */
int mpo_coattn_mcat_backward(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides,
                             int total_rows, int max_rows,
                             const float* query, int n_q, int embed,
                             const float* in_proj_weight, const float* out_proj_weight,
                             const float* saved, const float* attn_map,
                             const float* d_out, const float* d_attn_map,
                             float* d_query, int d_query_accumulate /* 1: d_query += (it arrives holding the gradient
                                of the query's other use, e.g. as the omic branch's tokens) */,
                             void* d_bag, float* d_bag_colsum /* nullable [embed] */,
                             float* d_in_proj_weight, float* d_in_proj_bias,
                             float* d_out_proj_weight, float* d_out_proj_bias,
                             float bag_relu_gate, const mpo_bag_plan* plan /* nullable */,
                             void* workspace, size_t workspace_bytes, mpo_stream_t stream);
/* d_bag_colsum (nullable): receives the column sums of d_bag, accumulated while the rows are written -- with the gate
 * below that is the bias gradient of the patch layer self.H, which otherwise costs a pass over the 480k x 256 d_bag.
 * bag_relu_gate: 0, or 1/(1-p) when the (bf16) bag is H = dropout_p(relu(.)) as in models/mcat/mcat.py:24-29,87 and
 * the caller wants d_bag already multiplied by that epilogue's derivative (H > 0 ? 1/(1-p) : 0): the H tile is still
 * in LDS when dH is formed, which saves two passes over the bag gradient. */

/* ---- epilogue of the patch layer self.H (models/mcat/mcat.py:24-29): h = dropout_p(relu(h + bias)) in place on the
 * bf16 product [rows, cols], and its derivative g = dy * (h > 0 ? 1/(1-p) : 0).  (The forward half dates from r01, when the
 * product X W^T was a library call; mpo_patch_fc_forward has covered every model width since ABI v13 and the Python host no
 * longer calls it.)  Backward: n = rows * cols elements; d_bias (nullable, [cols]) receives the column
 * sums of g -- the layer's bias gradient -- from the same pass (workspace of *_workspace_bytes then required). */
int mpo_patch_epilogue_forward(void* h_bf16, const float* bias, int64_t rows, int cols, float drop_p, uint64_t seed,
                               uint64_t offset, const uint64_t* rng_epoch, mpo_stream_t stream);
size_t mpo_patch_epilogue_backward_workspace_bytes(int64_t n, int cols);
int mpo_patch_epilogue_backward(const void* h_bf16, const void* dy_bf16, void* g_bf16, int64_t n, int cols, float drop_p,
                                float* d_bias /* nullable */, void* workspace, size_t workspace_bytes, mpo_stream_t stream);
/* out[c] = sum_r x[r][c] for a bf16 [rows, cols] tensor: the bias gradient of self.H (torch's reduce: 142 us) */
int mpo_colsum_bf16(const void* x_bf16, float* out, int64_t rows, int cols, mpo_stream_t stream);

/* ---- weight gradient of self.H (models/mcat/mcat.py:24-29: Linear(1024, 256)): d_weight [embed, patch_dim] fp32 =
 * g^T X over the whole window, g [rows, embed] bf16 = gradient w.r.t. the layer's pre-activation (what
 * mpo_coattn_mcat_backward with bag_relu_gate / mpo_nacagat_patch_grad / mpo_patch_epilogue_backward emit), X the raw bf16
 * patch matrix.  Hand-written split-row kernel (one workgroup per CU, fp32 partials in the workspace + a reduction
 * launch).  Built for embed 128 / 256 / 512 (512: one pass per 256 columns of g) and patch_dim 128, 256, 512, 1024 or 2048
 * (the narrow ones: NaCAGaT's key-projection weight gradient d_k^T H_bag); any number of rows -- the kernel's DMA offsets
 * are 32-bit, 4 GiB of patches and more go in row segments whose partial sums accumulate.  Other geometries: error.
 * workgroups (ABI v13): 0 = one per CU (256); fewer -- a multiple of 8 * patch_dim / 256, e.g. 224 at patch_dim 1024 -- leave
 * CUs free for a kernel of ANOTHER stream: the gradient all-reduce that a data-parallel step runs beside this product. */
size_t mpo_patch_weight_grad_workspace_bytes(int embed, int patch_dim);
int mpo_patch_weight_grad(const void* g_bf16, const void* patches_bf16, int64_t total_rows, int embed, int patch_dim,
                          float* d_weight, int workgroups, void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- optimiser step of the reference's default `adam` (models/mcat/main.py:284-300: torch.optim.Adam(lr, weight_decay))
 * over ONE flat parameter / gradient / moment buffer: g' = g + wd p; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
int mpo_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step,
                       const int32_t* step_dev /* nullable: device-resident step count, overrides `step` */,
                       mpo_stream_t stream);

/* Verification hook (tests): the 6 x d token-tail products run a branch-free GEMM body when the product is regular
 * (whole 16 x 16 tiles, K % 64 == 0, aligned operands, a gate known at compile time) and a general body otherwise; the two
 * are bit-identical.  enabled = 0 sends everything through the general body.  Returns the previous setting (default 1). */
int mpo_set_gemm_fast_path(int enabled);

/* Verification hook (tests): K1's backward bag pass has a two-waves-per-SIMD kernel for a bf16 bag at embed_dim 256 with at most 8
 * queries and no gradient on the map (csrc/coattn_bwd8.hip) and a general kernel for everything else (csrc/coattn_bwd.hip).
 * enabled = 0 sends every geometry through the general kernel.  Returns the previous setting (default 1).  ABI v11. */
int mpo_set_coattn_bwd_two_wave(int enabled);

/* K1 backward of an fp32-stored bag (embed 256, n_q <= 8, with or without a map gradient) runs on the vector ALUs in plain fp32
 * (csrc/coattn_bwd_f32.hip: with six queries every product is skinny; the matrix-pipe kernel needs split images, both
 * orientations and 1 KB of scratch per lane for the same result).  enabled = 0 sends fp32 bags through the general kernel as
 * the check of this one.  Returns the previous setting (default 1).  ABI v12. */
int mpo_set_coattn_bwd_f32_vector(int enabled);

/* K2 backward (models/blocks.py:184-187 differentiated): the query-side column accumulations  W1 K  and  W2 tanh(K)  and the
 * bag-side  dK  both need the fp32 key bag and the two gradient maps and nothing of each other; for n_q <= 6 at embed <= 256
 * ONE pass over K produces all three, on the vector ALUs in plain fp32 (bag_key_grad_kernel: with six queries the products
 * are too skinny for the matrix pipe).  enabled = 0 restores the two passes on the matrix pipe (bag_colacc_gated, then
 * bag_outer_gated; also what n_q > 6 and embed 512 run) as the check of the fused pass.  Returns the previous setting
 * (default 1).  ABI v12. */
int mpo_set_nacagat_one_pass_key_grad(int enabled);

/* rng_epoch += 1 and adam_step += 1 (either may be NULL) in one launch: the per-step device counters of a captured
 * training step (dropout epoch of every mpo_*_forward, step count of mpo_adam_step_flat). */
int mpo_step_counters_bump(uint64_t* rng_epoch, int32_t* adam_step, mpo_stream_t stream);

/* ---- K2: NaCAGaT narrow-gated co-attention core = models/blocks.py:114-206 (heads = 1):
 *   S = (q/sqrt(E)) k^T * (tanh(q) tanh(k)^T + 1)/2,  A = softmax(S),  A_drop = dropout(A, p) in training,
 *   out = (A_drop v) W_o^T + b_o,  returns (q, out, A_drop)  -- the map is post-dropout, as in the reference.
 * The caller supplies kbag = H W_k^T + b_k (a plain GEMM it owns, like self.H; dtype k_dtype) and hbag = H;
 * the value projection is folded out.  tanh(kbag) is recomputed on the fly in every pass, never stored.
 * k_dtype must be MPO_F32 even for a bf16 bag: the gate multiplies rounding errors of k (dS = (g+1) da),
 * so k is an intermediate that must not be stored in bf16 (hard part 4).
 *   score_maps  2 * n_q * total_rows floats (kept for backward);  attn_map  n_q * total_rows floats (output)
 *   seed/offset counter of the attention-weight dropout stream; pass the same pair to backward.
 * Backward takes gradients on all three returns (d_out, d_attn_map nullable, d_q_proj nullable) and emits
 * d_query, d_kbag (dk_dtype: a GRADIENT may be handed on in bf16), d_hbag (bag dtype) and the q / v / out-projection
 * gradients; the key slices of d_in_proj_* are zeroed (the caller back-propagates d_kbag through its GEMM).
 * d_kbag_colsum (nullable) receives the column sums of d_kbag (= the key bias gradient) from the pass that writes it. */
/* ---- ragged attention maps (slide b's [n_q][M_b] block at float offset n_q * cu_rows[b]), for the attention-regularised
 * loss CrossEntropySurvivalAttnRegLoss (models/loss.py:88-101: + lambda * ||A||_2 per slide):
 * out[b][q] = sum_m a[q][m] * b[q][m];   out_map block b = scale[b] * a block b. */
int mpo_map_block_dot(const float* a_map, const float* b_map, const int32_t* cu_rows, int n_slides, int n_q, float* out,
                      mpo_stream_t stream);
int mpo_map_block_scale(const float* a_map, const float* scale, const int32_t* cu_rows, int n_slides, int n_q, float* out,
                        mpo_stream_t stream);

/* K2's key projection for a bf16-stored bag: kbag[m][n] = sum_e hbag[m][e] w_k[n][e] + b_k[n] in fp32, with the fp32
 * weights split into three bf16 terms inside the kernel (all 24 mantissa bits; the bag is exact in bf16) -- replaces the k slice of
 * F.linear(key, in_proj_weight, in_proj_bias) at models/blocks.py:151-166 without an fp32 copy of the bag.
 * w_k = in_proj_weight + embed*embed, b_k = in_proj_bias + embed (nullable). */
int mpo_key_projection(const void* hbag_bf16, int64_t rows, int embed, const float* w_k, const float* b_k, float* kbag,
                       mpo_stream_t stream);
/* embed_dim 512 (model_size 'big', models/nacagat/nacagat.py:17-18; ABI v11): kbag, hbag, d_kbag and d_hbag of the two entries
 * below are in the SPLIT-HALVES layout [2][total_rows][256] -- columns 0..255 of every row, then columns 256..511 -- and each
 * bag pass runs once per half on the 256-wide kernels (maps are summed over the halves, column-indexed results sit side by
 * side).  Query-side tensors ([n_slides*n_q, 512]) and parameters keep their natural layout.  embed 128 / 256: row-major. */
size_t mpo_nacagat_saved_floats(int n_slides, int n_q, int embed);
size_t mpo_nacagat_workspace_bytes(int n_slides, int n_q, int embed, int max_rows, int total_rows);
int mpo_coattn_nacagat_forward(const void* kbag, int k_dtype, const void* hbag, int bag_dtype, const int32_t* cu_rows, int n_slides,
                               int total_rows, int max_rows, const float* query, int n_q, int embed,
                               const float* in_proj_weight, const float* in_proj_bias,
                               const float* out_proj_weight, const float* out_proj_bias,
                               float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                               float* q_proj, float* out, float* attn_map, float* score_maps,
                               float* saved, const mpo_bag_plan* plan /* nullable */,
                               void* workspace, size_t workspace_bytes, mpo_stream_t stream);
int mpo_coattn_nacagat_backward(const void* kbag, int k_dtype, const void* hbag, int bag_dtype,
                                const int32_t* cu_rows, int n_slides, int total_rows, int max_rows,
                                const float* query, int n_q, int embed,
                                const float* in_proj_weight, const float* in_proj_bias, const float* out_proj_weight,
                                float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                                const float* saved, const float* score_maps, const float* attn_map,
                                const float* d_out, const float* d_attn_map, const float* d_q_proj,
                                float* d_query, int d_query_accumulate /* ABI v14: d_query += (it already holds the gradient
                                                of the query's other consumers) */,
                                void* d_kbag, int dk_dtype, float* d_kbag_colsum /* nullable [embed] */,
                                void* d_hbag /* nullable when d_ctx is given */,
                                float* d_ctx /* nullable [n_slides*n_q][embed]: receives dL/d(A_drop V-side context); the
                                                dH outer product is then left to mpo_nacagat_patch_grad() */,
                                float* d_in_proj_weight, float* d_in_proj_bias,
                                float* d_out_proj_weight, float* d_out_proj_bias, const mpo_bag_plan* plan /* nullable */,
                                void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* Patch-side gradient of K2 for a bf16 bag, one pass (ABI v7):
 *   d_bag[m][e] = ( sum_q attn_map[q][m] d_ctx[q][e] + addend[m][e] ) * (hbag[m][e] > 0 ? relu_gate : 0)
 * addend = d_kbag W_k (the caller's GEMM back through K = H W_k^T + b_k, models/blocks.py:151-166); hbag is the bag
 * itself: for H = dropout(relu(.)) (models/nacagat/nacagat.py:20-25 via mcat.py:24-29) its sign is that layer's
 * ReLU/dropout derivative, relu_gate = 1/(1-p) (0: no gating).  d_bag may alias addend.  d_bias (nullable, [embed])
 * receives the column sums of d_bag = the bias gradient of the layer that produced the bag.
 * workspace: mpo_nacagat_workspace_bytes() of the same geometry is enough. */
int mpo_nacagat_patch_grad(const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int n_q, int embed,
                           const float* attn_map, const float* d_ctx, const void* addend_bf16, const void* hbag_bf16,
                           void* d_bag_bf16, float relu_gate, float* d_bias, const mpo_bag_plan* plan /* nullable */,
                           void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* The same gradient with the product back through the key projection INSIDE the pass (ABI v11; embed_dim 256):
 *   d_bag[m][e'] = ( sum_e d_kbag[m][e] W_k[e][e'] + sum_q attn_map[q][m] d_ctx[q][e'] ) * (hbag[m][e'] > 0 ? relu_gate : 0)
 * d_kbag bf16 [total_rows, embed] as mpo_coattn_nacagat_backward emits it, w_k = in_proj_weight rows [embed, 2 embed) (fp32,
 * [embed][embed]); reads d_kbag and hbag once, writes d_bag once -- no library GEMM, no second pass.  d_bag must not alias
 * d_kbag (the caller still needs it for dW_k). */
int mpo_nacagat_patch_grad_fused(const int32_t* cu_rows, int n_slides, int total_rows, int max_rows, int n_q, int embed,
                                 const float* attn_map, const float* d_ctx, const void* d_kbag_bf16, const float* w_k,
                                 const void* hbag_bf16, void* d_bag_bf16, float relu_gate, float* d_bias,
                                 const mpo_bag_plan* plan /* nullable */, void* workspace, size_t workspace_bytes,
                                 mpo_stream_t stream);

/* ==== the 6 x d token tail.  Parameter and gradient tensors are passed as arrays of device pointers in
 * the order listed per entry (the reference's state_dict order); dropout streams are counters of a counter-based generator (masks = a pure function of seed and counter)
 * (seed, offset): pass the same pair to forward and backward, reserve *_rng_span() counters per call.
 * rng_epoch (nullable, device uint64): added x 2^40 to every stream offset inside the kernels, so a HIP graph
 * that froze (seed, offset) at capture still draws fresh masks on every replay once the host bumps *rng_epoch
 * as part of the graph.  Eager callers pass NULL. ==== */

/* ---- K4: set-Transformer = nn.TransformerEncoder(post-norm layers, nhead, dim_feedforward, relu), no final
 * norm.  Replaces models/mcat/mcat.py:51-53,60-62 (call :101-102); torch/nn/modules/transformer.py:661.
 * x, y [n_slides*T, d]; per layer 12 pointers: self_attn.in_proj_weight, in_proj_bias, out_proj.weight,
 * out_proj.bias, linear1.weight, .bias, linear2.weight, .bias, norm1.weight, .bias, norm2.weight, .bias.
 * n_branches (1..4) batches that many independent encoders of IDENTICAL geometry -- the model's path_transformer
 * and omic_transformer -- into one launch sequence: x, y are [n_branches][n_slides*T][d], params / grads hold
 * n_branches * layers * 12 pointers (branch-major), and the *_floats / *_bytes / rng_span queries take
 * n_branches * n_slides as their n_slides.  The token tail is launch-latency-bound, so the second branch is free. */
size_t mpo_encoder_saved_floats(int n_slides, int T, int d, int ff, int heads, int layers);
size_t mpo_encoder_workspace_bytes(int n_slides, int T, int d, int ff);
uint64_t mpo_encoder_rng_span(int n_slides, int T, int d, int ff, int layers);
int mpo_encoder_forward(const float* x, int n_branches, int n_slides, int T, int d, int ff, int heads, int layers,
                        const float* const* params, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                        float* y, float* saved, mpo_stream_t stream);
int mpo_encoder_backward(const float* x, int n_branches, int n_slides, int T, int d, int ff, int heads, int layers,
                         const float* const* params, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                         const float* saved, const float* dy, float* dx, float* const* grads,
                         void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- f3: self-attention over the M rows of a bag -- the attention core of `nn.MultiheadAttention(embed, num_heads)` with
 * query = key = value = the bag (models/ge_nacagat/ge_nacagat.py:27,49: one head, the M x M map returned) and of the
 * `nn.TransformerEncoderLayer(nhead=8)` blocks over the same rows (:30-33,53; reached through mpo_encoder_* with T > 16).
 * qkv [n_bags][M][3 d] = the packed in_proj output (q | k | v; head h = columns h*d/heads .. of each part); out [n_bags][M][d]
 * is the per-head context BEFORE out_proj.  saved: mpo_bag_self_attention_saved_floats() floats (one log-sum-exp per head
 * and row -- no M x M state is kept).  attn_map (nullable, heads == 1 only) [n_bags][M][M] = softmax(q k^T / sqrt(d)).
 * drop_p: dropout on the probabilities (realised round(256 p) / 256; regenerated in the backward from seed / offset /
 * *rng_epoch); the map is the undropped softmax.  Head dimension d / heads in {16, 32, 64, 128, 256, 512}.
 * Several heads of width 32 (the encoder layers) run on bf16 MFMAs with every operand split into hi + lo (three products
 * per term: ~16 mantissa bits); the forward then also stores the operands' bf16 forms in `saved` for the backward.
 * mpo_set_bag_self_attention_bf16x3(0) keeps that geometry on the fp32 kernels as well (verification hook; returns the
 * previous setting, default 1).  The map carries no gradient (the reference only returns it). */
size_t mpo_bag_self_attention_saved_floats(int n_bags, int M, int d, int heads);
size_t mpo_bag_self_attention_workspace_bytes(int n_bags, int M, int d, int heads);
int mpo_set_bag_self_attention_bf16x3(int enabled);
int mpo_bag_self_attention_forward(const float* qkv, int n_bags, int M, int d, int heads, float drop_p, uint64_t seed, uint64_t offset,
                                   const uint64_t* rng_epoch, float* out, float* saved, float* attn_map, mpo_stream_t stream);
int mpo_bag_self_attention_backward(const float* qkv, const float* out, const float* saved, const float* d_out, int n_bags, int M, int d,
                                    int heads, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch, float* d_qkv,
                                    void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- K5: gated attention-MIL pooling = AttentionNetGated (models/blocks.py:13-48) + softmax pooling + rho
 * (models/mcat/mcat.py:105-109).  x [n_slides*L, d] -> scores [n_slides*L] (raw A), h [n_slides, d].
 * 8 pointers: attention_a.0.weight, .bias, attention_b.0.weight, .bias, attention_c.weight, .bias, rho.0.weight, .bias
 * n_branches (forward 1..4, backward 1..2) batches independent heads of identical geometry as for K4: tensors are
 * [n_branches][...], params / grads n_branches * 8 pointers, size queries take n_branches * n_slides. */
size_t mpo_gated_pool_saved_floats(int n_slides, int L, int d);
size_t mpo_gated_pool_workspace_bytes(int n_slides, int L, int d);
uint64_t mpo_gated_pool_rng_span(int n_slides, int L, int d);
int mpo_gated_pool_forward(const float* x, int n_branches, int n_slides, int L, int d, const float* const* params,
                           float head_drop_p, float rho_drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_epoch,
                           float* scores, float* h, int h_interleaved, float* saved, mpo_stream_t stream);
/* h_interleaved != 0: h (and dh in the backward) are [n_slides][n_branches][d] -- with the path and the omic branch batched
 * that IS the concatenated [h_path | h_omic] row ConcatFusion reads (models/fusion.py:17-19), no transposing copy. */
int mpo_gated_pool_backward(const float* x, int n_branches, int n_slides, int L, int d, const float* const* params,
                            float head_drop_p, float rho_drop_p, const float* saved, const float* h,
                            const float* dh, int h_interleaved, const float* d_scores_ext /* nullable */, float* dx,
                            float* const* grads,
                            void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- K6: ConcatFusion (models/fusion.py:7-19) + classifier + survival head (models/mcat/mcat.py:119-138).
 * hcat [n_slides, din] = [h_path | h_omic] -> hazards, survs, Y [n_slides, n_classes].
 * 6 pointers: fusion_layer.0.weight, .bias, fusion_layer.2.weight, .bias, classifier.weight, .bias */
size_t mpo_fusion_head_saved_floats(int n_slides, int hidden, int dout, int n_classes);
size_t mpo_fusion_head_workspace_bytes(int n_slides, int hidden, int dout, int n_classes);
int mpo_fusion_head_forward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                            const float* const* params, float* hazards, float* survs, float* y, float* saved,
                            mpo_stream_t stream);
int mpo_fusion_head_backward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                             const float* const* params, const float* saved, const float* hazards,
                             const float* survs, const float* y, const float* d_hazards, const float* d_survs,
                             const float* d_y, float* d_hcat, float* const* grads,
                             void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- survival head alone (for fusion layers other than `concat`, whose MLP K6 has built in): hazards = sigmoid(logits),
 * survs = cumprod(1 - hazards), Y = softmax(logits)  (models/mcat/mcat.py:130-138).  Gradients nullable. */
int mpo_survival_head_forward(const float* logits, int n_slides, int n_classes, float* hazards, float* survs, float* y,
                              mpo_stream_t stream);
int mpo_survival_head_backward(const float* hazards, const float* survs, const float* y, const float* d_hazards,
                               const float* d_survs, const float* d_y, int n_slides, int n_classes, float* d_logits,
                               mpo_stream_t stream);

/* ---- 'ces' survival loss = CrossEntropySurvivalLoss.forward, models/loss.py:5-28, for n_slides slides at once
 * (the reference calls it per slide, models/mcat/main.py:52); per-slide losses, no reduction.  risk (nullable)
 * receives -sum_j survs_j (models/mcat/main.py:56).  Backward takes the per-slide upstream gradient d_loss
 * (n_slides floats), or ONE device float broadcast to all slides when d_loss_is_scalar (loss.sum()/grad_acc_step). */
int mpo_ces_loss_forward(const float* hazards, const float* survs, const int64_t* label, const float* censorship,
                         int n_slides, int n_classes, float alpha, float eps, float* loss, float* risk, mpo_stream_t stream);
int mpo_ces_loss_backward(const float* hazards, const float* survs, const int64_t* label, const float* censorship,
                          int n_slides, int n_classes, float alpha, float eps, const float* d_loss, int d_loss_is_scalar,
                          float* d_hazards, float* d_survs, mpo_stream_t stream);

/* ---- training-step form of K6 + loss: ConcatFusion + classifier (models/fusion.py:17-19, mcat.py:126), then the survival
 * head (mcat.py:130-138), the `ces` loss (models/loss.py:5-28) and the backward of both in ONE launch.  slide_weight
 * (n_slides device floats) is the gradient the caller will send into the per-slide loss (1 / grad_acc_step in the
 * reference's loop, models/mcat/main.py:69-70); mpo_fusion_head_loss_backward continues from the stored d_logits.
 * Workspace of the backward: mpo_fusion_head_workspace_bytes. */
size_t mpo_fusion_head_loss_saved_floats(int n_slides, int hidden, int dout, int n_classes);
int mpo_fusion_head_loss_forward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                                 const float* const* params, const int64_t* label, const float* censorship,
                                 const float* slide_weight, float alpha, float eps, float* hazards, float* survs, float* y,
                                 float* loss, float* risk /* nullable */, float* saved, mpo_stream_t stream);
int mpo_fusion_head_loss_backward(const float* hcat, int n_slides, int din, int hidden, int dout, int n_classes,
                                  const float* const* params, const float* saved, float* d_hcat, float* const* grads,
                                  void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- K3: ContextualAttentionGate.forward (models/blocks.py:232-253) on rows of (Q, Q_hat).
 * 12 pointers: fc1.0.weight,.bias, fc2.0.weight,.bias, fc3.0.weight,.bias, G.1.weight,.bias, E.1.weight,.bias,
 * fc_c.0.weight,.bias */
size_t mpo_cag_saved_floats(int rows, int hidden);
size_t mpo_cag_workspace_bytes(int rows, int hidden);
/* ABI v14: residual / sum_out (both or neither): sum_out = residual + C in the forward's last launch -- the caller's
 * attn_output + CAG(query, q_proj) of models/blocks.py:110 without an element-wise pass; d_q_accumulate: d_q += . */
int mpo_cag_forward(const float* q, const float* q_hat, int rows, int dim, int hidden, const float* const* params,
                    float* c_out, float* saved, const float* residual /* nullable [rows, hidden] */,
                    float* sum_out /* nullable [rows, hidden] */, mpo_stream_t stream);
int mpo_cag_backward(const float* q, const float* q_hat, int rows, int dim, int hidden, const float* const* params,
                     const float* saved, const float* c_out, const float* d_c, float* d_q, int d_q_accumulate, float* d_q_hat,
                     float* const* grads, void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- omic SNNs self.G (models/mcat/mcat.py:32-45,90-92): per group Linear+ELU+AlphaDropout twice, all groups
 * per launch (grouped GEMM).  x[i] [n_slides, widths[i]] -> g_bag [n_slides, n_groups, d].  x, widths, params, grads are
 * HOST arrays (of device pointers / ints).  4 pointers per group: 0.0.weight, 0.0.bias, 1.0.weight, 1.0.bias. */
size_t mpo_omic_snn_saved_floats(int n_slides, int n_groups, int d);
size_t mpo_omic_snn_workspace_bytes(int n_slides, int n_groups, int d);
uint64_t mpo_omic_snn_rng_span(int n_slides, int n_groups, int d);
int mpo_omic_snn_forward(const float* const* x, const int* widths, int n_groups, int n_slides, int d,
                         const float* const* params, float drop_p, uint64_t seed, uint64_t offset,
                         const uint64_t* rng_epoch, float* g_bag, float* saved, mpo_stream_t stream);
int mpo_omic_snn_backward(const float* const* x, const int* widths, int n_groups, int n_slides, int d,
                          const float* const* params, float drop_p, uint64_t seed, uint64_t offset,
                          const uint64_t* rng_epoch, const float* g_bag, const float* saved, const float* d_g_bag,
                          float* const* grads, void* workspace, size_t workspace_bytes, mpo_stream_t stream);

/* ---- the two bag-pass kernels of K1 on their own (bench.py times them with HIP events for the
 * roofline line; tests use them for kernel-level checks).  qk2 = log2(e) * (q/sqrt(E)) W_k, [n_slides*n_q, embed].
 * part_ml [P*32], part_ctx / part_dqk [P*n_q*embed] with P >= mpo_coattn_target_workgroups() + n_slides partials. */
int mpo_coattn_splits(int n_slides, int max_rows);
int mpo_coattn_fwd_bagpass(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int embed,
                           const float* qk2, float* part_ml, float* part_ctx, float* raw_logits /* nullable */,
                           int n_q, int max_rows, const mpo_bag_plan* plan /* nullable */, mpo_stream_t stream);
/* K2's forward bag pass on its own (both score maps from one pass over the fp32 key bag): a_map, g_map [n_q*total_rows] */
int mpo_nacagat_fwd_bagpass(const float* kbag, const int32_t* cu_rows, int n_slides, int embed, const float* qs2,
                            const float* tq, float* a_map, float* g_map, int n_q, int max_rows,
                            const mpo_bag_plan* plan /* nullable */, mpo_stream_t stream);
int mpo_coattn_bwd_bagpass(const void* bag, int bag_dtype, const int32_t* cu_rows, int n_slides, int embed,
                           const float* qk2, const float* lse2, const float* dctx, const float* delta,
                           const float* d_attn_map /* nullable */, void* d_bag, float* part_dqk,
                           int n_q, int max_rows, const mpo_bag_plan* plan /* nullable */, mpo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPO_HIP_H */
