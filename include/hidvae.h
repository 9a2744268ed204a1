/* hidvae.h -- C ABI of the MI355X-native (gfx950) HiD-VAE tokenizer hot path.
 *
 * The reference (FDzhaozi/HiD-VAE) has NO native/FFI layer: its "operator interface" for this path
 * is the Python API of modules/h_rqvae.py, modules/encoder.py, modules/quantize.py, modules/loss.py
 * and init/kmeans.py.  Each entry point below names the reference function (file:line, relative to
 * the reference root) whose arithmetic it replaces; the Python mirror in hid-vae_amd/ binds them
 * with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; row-major, fp32, ids int64;
 *   - `ld*` arguments are row strides in ELEMENTS (so column slices / concatenated views need no copy);
 *   - the caller allocates every input, output and workspace; the library never allocates or
 *     retains device memory, never synchronises, and launches on `stream` (a hipStream_t);
 *   - return 0 on success, a negative HIDVAE_E* code otherwise; hidvae_last_error() gives the
 *     message of the calling thread's last failure.  Nothing throws across this boundary.
 *   - re-entrant: no mutable global state besides one-time kernel attribute setup.
 */
#ifndef HIDVAE_H
#define HIDVAE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIDVAE_OK 0
#define HIDVAE_EINVAL (-1)   /* bad shape / argument (mirrors the reference's asserts)            */
#define HIDVAE_EUNSUPPORTED (-2)
#define HIDVAE_ELAUNCH (-3)  /* hip runtime error at launch                                       */

#define HIDVAE_MAX_LEVELS 8
/* embed_dim is an ARGUMENT of every entry point that sees code vectors (rounds 1-2 fixed it at 32 as a constant of the ABI).  The fused
 * kernels are built for 32 (both h-configs); other widths -- a multiple of 4, at most 64: configs/rqvae_ml32m.gin uses 64 -- run the
 * same algorithm in a width-independent geometry (csrc/rq_generic.hip), exact fp32, not tuned.  hidvae_bottleneck_fwd and the
 * GUMBEL row kernels are 32 only. */

/* quantize.py:17-20 QuantizeForwardMode */
#define HIDVAE_MODE_GUMBEL 1
#define HIDVAE_MODE_STE 2
#define HIDVAE_MODE_ROTATION 3

/* quantize.py:22-24 QuantizeDistance: what the per-level search ranks the codes by.  L2 is |r - c|^2 (every shipped config; the fused
 * kernels); COSINE is -(r/|r| . c)/|c| (quantize.py:115-119; HRqVae never selects it -- Quantize on its own can): the width-independent
 * kernel at any embed_dim.  Outputs, losses and the backward are the same function of the chosen ids either way. */
#define HIDVAE_DIST_L2 0
#define HIDVAE_DIST_COSINE 1

/* GEMM operand layouts: C[M,N] = opA(A) . opB(B) */
#define HIDVAE_GEMM_NT 0 /* A[M,K] B[N,K]  : Linear forward   x W^T        (encoder.py:27) */
#define HIDVAE_GEMM_NN 1 /* A[M,K] B[K,N]  : input gradient   dY W                          */
#define HIDVAE_GEMM_TN 2 /* A[K,M] B[K,N]  : weight gradient  dY^T X                        */

/* GEMM epilogues */
#define HIDVAE_EPI_NONE 0
#define HIDVAE_EPI_SILU 1      /* C = silu(acc+bias); if aux != NULL also aux = acc+bias (pre-activation)   */
#define HIDVAE_EPI_RELU 2
#define HIDVAE_EPI_GELU 3      /* exact erf GELU (nn.GELU default, h_rqvae.py:136)                          */
#define HIDVAE_EPI_SIGMOID 4
#define HIDVAE_EPI_DSILU 16    /* C = acc * silu'(aux)      (backward through the activation)              */
#define HIDVAE_EPI_DRELU 17    /* C = acc * (aux > 0)        aux = forward OUTPUT or pre-activation         */
#define HIDVAE_EPI_DGELU 18    /* C = acc * gelu'(aux)       aux = pre-activation                           */
#define HIDVAE_EPI_DSIGMOID 19 /* C = acc * s(1-s), s = aux  aux = forward OUTPUT                           */

const char *hidvae_version(void);
const char *hidvae_last_error(void);

/* ---- workspace sizes.  The caller owns all device memory; this tells it how many BYTES the `workspace` / `scratch`
 * argument of an entry point must hold for a problem of dimensions dims[0..n_dims) (0 = the entry point needs none at that size
 * and NULL may be passed).  Host-only, no GPU call.  dims per op: */
#define HIDVAE_WS_GEMM 1                 /* M, N, K, split_k          -> hidvae_gemm_f32 workspace                      */
#define HIDVAE_WS_LINEAR_BWD 2           /* B, n_out, n_in, has_bias  -> hidvae_linear_bwd workspace                    */
#define HIDVAE_WS_COLSUM 3               /* M, N                      -> hidvae_colsum workspace                        */
#define HIDVAE_WS_CODEBOOK_GRAD 4        /* B, L, K                   -> hidvae_codebook_grad workspace                 */
#define HIDVAE_WS_LAYERNORM_PARAM_GRAD 5 /* M, N                                                                         */
#define HIDVAE_WS_LAYERNORM_BWD_ALL 6    /* M, N                                                                         */
#define HIDVAE_WS_BATCHNORM_FWD 7        /* M, N                                                                         */
#define HIDVAE_WS_BATCHNORM_BWD 8        /* M, N                                                                         */
#define HIDVAE_WS_ID_CENSUS 9            /* B  -> `scratch` of hidvae_id_stats / `census_scratch` of hidvae_bottleneck_fwd (zero-fill once) */
#define HIDVAE_WS_KMEANS 10              /* N, K                      -> shift_scratch of hidvae_kmeans_iter             */
#define HIDVAE_WS_TAG_LOSS 11            /* B, C                      -> row_loss + row_hit + zbuf of hidvae_tag_loss_fwd */
#define HIDVAE_WS_RQ_FORWARD 13           /* B, L, K                   -> hidvae_rq_forward workspace (0 below 65536 items)          */
#define HIDVAE_WS_LINEAR_BWD_ZEROED 12   /* B, n_out, n_in, has_bias  -> leading bytes of the hidvae_linear_bwd workspace that must be ZERO
                                            on entry (arrival counters of the balanced kernel; left zero on return).  Non-zero means: hand
                                            hidvae_linear_bwd a workspace that no launch running CONCURRENTLY shares (one per stream). */
int hidvae_query_workspace(int op, const int64_t *dims, int n_dims, int64_t *bytes);

/* ---- a2/a3/a9/a10: Linear layers (modules/encoder.py:23-36, h_rqvae.py:132-188,322-331) ------------
 * C[M,N] (ldc) = epilogue( opA(A) opB(B) + bias[N] ) [* mask * mask_scale].  fp32 in, fp32 MFMA.
 * split_k == 1: every output element is ONE fmaf chain over k in the fixed order ORDER-G16 (16-wide k-blocks
 * ascending, inside a block 0,4,8,12,1,5,9,13,...) -- reproducible bit for bit by oracle/exact.c; this is what the
 * forward (id-determining) GEMMs use.  split_k == 0: the library may split K over the waves of a workgroup and
 * reduce in fixed order (bit-reproducible run to run, not ORDER-G); split_k > 1 caps that split.
 * mask (optional, [M,N] at ldmask): dropout keep-mask multiplied in after the activation (nn.Dropout fused).
 * workspace: split_k * M * N floats, only read for batches large enough to take the LDS-tiled path (may be NULL).
 * accumulate != 0 adds into C instead of overwriting (gradient accumulation). */
int hidvae_gemm_f32(int layout, int64_t M, int64_t N, int64_t K,
                    const float *A, int64_t lda, const float *B, int64_t ldb,
                    const float *bias, float *C, int64_t ldc,
                    int epilogue, float *aux, int64_t ldaux,
                    const float *mask, int64_t ldmask, float mask_scale,
                    const unsigned long long *rng_state, unsigned rng_site, unsigned drop_threshold,
                    int split_k, float *workspace, int accumulate, void *stream);

/* Backward of y = x W^T (nn.Linear without bias; encoder.py:27-31 as autograd differentiates it) in ONE launch:
 *   dW [n_out, n_in] = g^T x          (g [B, n_out], x [B, n_in])
 *   dX [B, n_in]     = epi(g W)       (W [n_out, n_in]; dx_epilogue = HIDVAE_EPI_NONE or a backward code D* whose `aux`
 *                                      [B, n_in] is the previous layer's saved tensor, i.e. the activation derivative of the
 *                                      layer below is applied on the way out; dX == NULL: weight gradient only)
 * Both products are independent, so small problems share one grid (the dW tiles first, then the dX tiles) instead of paying
 * two launches; the arithmetic, split and summation order are those of hidvae_gemm_f32(split_k = 0) on each product, so
 * the results are bit-identical to the two separate calls.  Shapes outside the small-problem regime fall back to them.
 * Mid-size layers at batches 256 .. 4095 (hidvae_query_workspace(HIDVAE_WS_LINEAR_BWD_ZEROED, ...) > 0) run on the balanced
 * LDS-shared kernel instead when a workspace is given: 64x64 tiles whose k-steps are dealt evenly over the chip's workgroup slots,
 * partial tiles added in ascending k order by the last workgroup to arrive (deterministic, but a different summation order than
 * hidvae_gemm_f32).  It keeps arrival counters in the first ZEROED bytes of the workspace: zero on entry, zero again on return,
 * never shared with a launch that may run at the same time.  Without a workspace the paired kernels above are used.
 * accumulate_dw: dW += g^T x (gradient accumulation straight into a flat gradient buffer's slot).
 * db (optional): the bias gradient db[n_out] = column sums of g, from the same launch (fixed summation order; the fallback
 * uses hidvae_colsum and needs its workspace of ceil(B/64)*n_out floats when B > 16384).  workspace (optional) is also what lets the
 * unpaired dW product of a LARGE batch (B >= 4096) run as K-slabs on the LDS-tiled kernel: give it max(that, 16*n_out*n_in) floats.
 *   co_resident != 0: the caller runs launches on OTHER streams beside this one (the per-level streams of the tag heads).  The
 *             balanced kernel then takes eight-wave workgroups (two fit a CU) instead of sixteen-wave ones that hold every register
 *             of their CU for the whole launch: 0-6 % slower alone, but the streams' launches overlap each other's fill and drain
 *             (tagged step 1.385 -> 1.320 ms).  Results differ from the sixteen-wave form in the last bits (two k-groups per tile
 *             instead of four: another fixed summation order); each form is deterministic.
 *   dx_scale  multiplies the HIDVAE_EPI_DRELU result (1.0 elsewhere): the backward through ReLU -> Dropout(keep_scale) read off that
 *             layer's saved OUTPUT (aux = y = relu(.) * keep * keep_scale; y > 0 exactly where the unit was active and kept). */
int hidvae_linear_bwd(const float *g, int64_t ldg, const float *x, int64_t ldx, const float *W, int64_t ldw, int64_t B,
                      int64_t n_out, int64_t n_in, float *dW, int64_t lddw, int accumulate_dw, float *dX, int64_t lddx,
                      int dx_epilogue, float *aux, int64_t ldaux, float dx_scale, float *db, int accumulate_db, float *workspace,
                      int co_resident, void *stream);

/* ---- grouped launches: the SAME layer of several independent heads in one grid (the three tag-head levels of h_rqvae.py:526-549
 * run structurally identical Linear / LayerNorm chains of different widths).  problems_host: HOST array of n descriptors whose
 * fields mean exactly what the arguments of hidvae_gemm_f32 (split_k = 0) / hidvae_linear_bwd mean; up to 12 GEMM-class
 * sub-problems per launch (a backward descriptor counts dW, dX and db separately).  Every sub-problem keeps a fixed summation order
 * (K shared out over <= 8 waves, partials added in ascending wave order), so results are bit-reproducible run to run; small
 * problems' launch and drain latencies hide under the large ones' arithmetic.  Outside the direct kernels' regime (a problem with
 * >= 2048 output tiles, or more sub-problems than fit) the call falls back to one hidvae_gemm_f32 / hidvae_linear_bwd per problem
 * (`workspace`, optional, is only used there). */
typedef struct {
    const float *g; int64_t ldg;
    const float *x; int64_t ldx;
    const float *W; int64_t ldw;
    int64_t B, n_out, n_in;
    float *dW; int64_t lddw; int accumulate_dw;
    float *dX; int64_t lddx;
    int dx_epilogue;
    float *aux; int64_t ldaux;
    float *db; int accumulate_db;
    float *workspace;
} hidvae_linear_bwd_problem;
int hidvae_linear_bwd_group(const hidvae_linear_bwd_problem *problems_host, int n, void *stream);

/* out[n] (+)= sum_m X[m,n]   (bias gradients; fixed summation order, bit-reproducible).  One launch for M <= 16384 (no
 * workspace needed, may be NULL); above that two fixed-order passes through workspace >= ceil(M/64)*N floats. */
int hidvae_colsum(const float *X, int64_t M, int64_t N, int64_t ldx, float *out, int accumulate,
                  float *workspace, void *stream);

/* ---- a4: effective codebooks (modules/quantize.py:106; h_rqvae.py:295) -----------------------------
 * For each level i: cb_eff[i] = rows of E[i] L2-normalised iff normalize[i], cc[i][k] = |cb_eff[i][k]|^2.
 * E_host / normalize_host are HOST arrays of L entries (device pointers / flags).
 * cb_eff: [L][K][32], cc: [L][K] device workspaces. */
int hidvae_codebook_prepare(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K,
                            float *cb_eff, float *cc, int embed_dim, void *stream);

/* ---- a4-a7: fused L-level residual quantisation, forward ------------------------------------------
 * Replaces the level loop of HRqVae.get_semantic_ids (h_rqvae.py:515-552) and Quantize.forward
 * (quantize.py:100-153) incl. the rotation trick (quantize.py:34-45) and QuantizeLoss (loss.py:41-44),
 * plus the encoder's trailing L2 normalisation (encoder.py:32) when normalize_input != 0.
 *   y        [B,32] encoder output (before normalisation iff normalize_input)
 *   cb_eff   [L][K][32], cc [L][K] from hidvae_codebook_prepare
 *   mode     HIDVAE_MODE_STE | HIDVAE_MODE_ROTATION (GUMBEL is composed from GEMM + softmax entry points)
 *   training 0: o = e (eval branch, quantize.py:146-148)
 * outputs (any may be NULL except ids):
 *   z [B,32] level-0 input; ids [B,L] int64; emb_cat [B, ld_cat>=L*32] per-level o_i side by side
 *   (level-i concat_emb of h_rqvae.py:526 is the first (i+1)*32 columns); emb_sum [B,32] = sum_i o_i
 *   (decoder input, h_rqvae.py:607); res_cat [B, L*32] level inputs r_i; qloss [B] = sum_i loss_i.
 * With every output but ids NULL and training == 0 the launch is the corpus tokenisation of
 * HSemanticIdTokenizer.precompute_corpus_ids (modules/tokenizer/h_semids.py:109-195): only 8 L bytes per item leave it.
 *   workspace  hidvae_query_workspace(HIDVAE_WS_RQ_FORWARD, {B, L, K}) bytes, or NULL.  Batches from 65536 items on run a
 *   split-bf16 prefilter on bf16 MFMA first and then the exact fp32 search over the items the prefilter could not decide (their list
 *   lives in the workspace: 3 stream operations -- a 16-byte memset, the prefilter, the exact kernel -- all capturable); results are
 *   bit-identical to the exact kernel alone, which is what runs without a workspace (about 2x slower at corpus sizes). */
int hidvae_rq_forward(const float *y, int64_t B, int normalize_input,
                      const float *cb_eff, const float *cc, int L, int64_t K,
                      int mode, int training, float beta,
                      float *z, int64_t *ids, float *emb_cat, int64_t ld_cat,
                      float *emb_sum, float *res_cat, float *qloss, void *workspace, int embed_dim, int distance, void *stream);

/* ---- a2 (last two layers) + a5-a7 + a3 (first two layers) in ONE launch, for small batches (B <= 4096; codebooks + 34 KB of
 * activations must fit in LDS: L*(33*Kp+64)*4 bytes with Kp = K rounded up to 128, e.g. 3x256 or 2x512).  Training only.
 *   h2 = silu(h1 W2^T)   [B,N2]   (W2 [N2,K2]; h1 [B,K2] contiguous, activation already applied)      encoder.py:27-31
 *   y  = h2 W3^T         [B,32]   (W3 [32,N2])
 *   z, ids, emb_cat, emb_sum, qloss = hidvae_rq_forward(y, ...)  (same arguments and results)
 *   d0 = silu(emb_sum Wd0^T) [B,Nd0],  d1 = silu(d0 Wd1^T) [B,Nd1]
 * pre2 / pre_d0 / pre_d1 receive the pre-activations (the backward's saved tensors).  All widths multiples of 16; K2, N2, Nd0
 * <= 256.  Every output is bit-identical to the separate launches (hidvae_gemm_f32 split_k=1 + hidvae_rq_forward).
 * Optionally (all three pointers, or none) the launch also produces the debug statistics of h_rqvae.py:643-648 that
 * hidvae_id_stats computes: embs_norm [B,L] and *p_unique, same values bit for bit; census_scratch = 4*B+3 int64, zero-filled ONCE
 * by its owner and then only touched by this entry point (NOT shared with hidvae_id_stats: its slots carry the id tuple itself,
 * 10 bits per level, so L <= 4 and K <= 1024). */
int hidvae_bottleneck_fwd(const float *h1, int64_t B, int K2, int N2, const float *W2, const float *W3, float *pre2, float *h2,
                          float *y, int normalize_input, const float *cb_eff, const float *cc, int L, int64_t K, int mode,
                          float beta, float *z, int64_t *ids, float *emb_cat, int64_t ld_cat, float *emb_sum, float *qloss,
                          int Nd0, int Nd1, const float *Wd0, const float *Wd1, float *pre_d0, float *d0, float *pre_d1,
                          float *d1, float *embs_norm, float *p_unique, int64_t *census_scratch, void *stream);

/* ---- fused RQ backward (autograd of the above) --------------------------------
 *   g_cat [B, ld_gcat] grad wrt each o_i (NULL = 0); g_sum [B,32] grad wrt sum_i o_i (NULL = 0);
 *   g_z_in [g_z_rows,32] extra grad wrt the first g_z_rows rows of z (uniqueness loss; NULL = 0); gq: d(loss)/d(qloss[b]) (same for all b),
 *   times gq_items[b*gq_stride] when gq_items != NULL (gq_stride 0 broadcasts one device scalar).
 * outputs: g_y [B,32] grad wrt the encoder output; dE_rows [B, L*32]: per-item codebook-row gradient
 * contributions gq*2*(e_i - r_i), summed per code by hidvae_codebook_grad in ascending item order. */
int hidvae_rq_backward(const float *y, const float *z, int64_t B, int normalize_input,
                       const float *cb_eff, const float *cc, int L, int64_t K,
                       int mode, float beta, const int64_t *ids,
                       const float *g_cat, int64_t ld_gcat, const float *g_sum, const float *g_z_in,
                       int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows, int embed_dim, void *stream);
/* hidvae_rq_backward with the gradient of emb_cat handed over as the PREFIX SLICES the tag heads leave behind (slice s: a contiguous
 * [B, width_s] tensor = the gradient of the first width_s columns, width_s a multiple of 32) instead of their sum: added up on the
 * fly in slice order, from zero -- the bits of hidvae_sum_prefix_slices followed by hidvae_rq_backward, one launch less.  embed_dim 32. */
int hidvae_rq_backward_slices(const float *y, const float *z, int64_t B, int normalize_input, const float *cb_eff, const float *cc,
                              int L, int64_t K, int mode, float beta, const int64_t *ids, const float *const *g_slices_host,
                              const int32_t *slice_width_host, int n_slices, const float *g_sum, const float *g_z_in,
                              int64_t g_z_rows, float gq, const float *gq_items, int64_t gq_stride, float *g_y, float *dE_rows,
                              int embed_dim, void *stream);

/* gE[i][k][:] (+)= sum_{b: ids[b,i]==k} dE_rows[b, i*32:(i+1)*32], pushed through the row-normalise
 * Jacobian for levels with normalize[i] (E_host: raw tables, needed for |E_k|).  gE_host: L device ptrs.
 * Summation order: items ascending (two interleaved chains per code).  workspace (optional, L*K*ceil(B/2048)*32 floats): batches
 * above 2048 items are summed in slabs of 2048 items and the slabs added in ascending order (a second launch) -- the longest
 * code's chain no longer sets the time; identical results for B <= 2048, a different fixed order above. */
int hidvae_codebook_grad(const int64_t *ids, const float *dE_rows, int64_t B, int L, int64_t K,
                         const float *const *E_host, const float *cb_eff, const int32_t *normalize_host,
                         float *const *gE_host, int accumulate, float *workspace, int embed_dim, void *stream);

/* ---- a3/a14: decoder tail.  x_hat = y / max(|y|,1e-12) (encoder.py:32), recon[b] = sum (x_hat-x)^2
 * (loss.py:11-12) and, if g_y != NULL, g_y = d(sum_b gscale_b * recon[b]) / dy with gscale_b = gscale *
 * gscale_items[b*gs_stride] (just gscale when gscale_items == NULL; stride 0 broadcasts one device scalar).
 * n_cat > 0 (0 <= n_cat < N): the last n_cat columns are categorical (h_rqvae.py:610-613, rqvae.py:146-149, loss.py:15-33):
 * u = y / max(|y|,1e-12) over the whole row, the first N - n_cat columns of u are L2-normalised once more and enter as squared
 * error, the last n_cat enter as binary cross-entropy with logits against x; x_hat = cat(normalised head, tail of u). */
int hidvae_recon_fwd_bwd(const float *y, const float *x, int64_t B, int64_t N, int n_cat, float gscale,
                         const float *gscale_items, int64_t gs_stride, float *x_hat, float *recon, float *g_y,
                         void *stream);

/* rows L2-normalise forward / backward (modules/normalize.py:7-8; F.normalize in h_rqvae.py:212, loss.py:66).
 * out = x / max(|x|, eps); norms[m] = |x_m| saved for the backward.  The N == 32 form uses the RQ kernel's
 * summation order so that HRqVae.encode() reproduces the fused forward's z bit for bit. */
int hidvae_l2norm_fwd(const float *x, int64_t M, int64_t N, int64_t ldx, float eps, float *out, int64_t ldo,
                      float *norms, void *stream);
int hidvae_l2norm32_fwd(const float *x, int64_t M, int64_t ldx, float eps, float *out, int64_t ldo, float *norms,
                        void *stream);
int hidvae_l2norm_bwd(const float *g, int64_t ldg, const float *out, int64_t ldo, const float *norms,
                      int64_t M, int64_t N, float eps, float *gx, int64_t ldgx, int accumulate, void *stream);
/* two such normalisations of M rows each (widths N0, N1; contiguous outputs) in ONE launch, and their backward: TagAlignmentLoss
 * normalises the concatenated codebook embeddings and the projected tags side by side (loss.py:66-67).  Per row the arithmetic of
 * hidvae_l2norm_fwd / hidvae_l2norm_bwd (generic summation order at every width). */
int hidvae_l2norm_fwd_pair(const float *x0, int64_t ldx0, int64_t N0, float *out0, float *norms0, const float *x1, int64_t ldx1, int64_t N1,
                           float *out1, float *norms1, int64_t M, float eps, void *stream);
int hidvae_l2norm_bwd_pair(const float *g0, int64_t ldg0, const float *out0, const float *norms0, int64_t N0, float *gx0, const float *g1,
                           int64_t ldg1, const float *out1, const float *norms1, int64_t N1, float *gx1, int64_t M, float eps, void *stream);

/* ---- a12: SemanticIdUniquenessLoss exactly as HRqVae.forward calls it (h_rqvae.py:41-105 with the [L,B]
 * transposed ids of :630-631): level pairs (a<b) whose id vectors agree over the whole batch contribute
 * relu(cos(z[a], z[b]) - margin); loss = weight * mean over such pairs (0 if none).  g_rows [L,32] (optional) receives
 * d loss / d z[0:L].  a1: total loss = mean(recon)+mean(qloss)+w_a*align+w_p*pred+w_u*uniq (h_rqvae.py:634-640),
 * the uniqueness term evaluated in the same launch when ids != NULL.  align/pred/acc: HOST arrays of n_tag per-level
 * device scalars (n_tag = 0 for an untagged batch); tag_align = sum_i align_i / tag_div etc. (h_rqvae.py:561-563);
 * tagstats [3+3*n_tag] (optional) receives the three means followed by the three by-layer vectors; summary [6] (optional)
 * receives {loss, mean recon, mean qloss, tag align, tag pred, tag accuracy}: the row train_hidvae.py:770-790 logs. */
int hidvae_uniq_loss(const int64_t *ids, const float *z, int64_t B, int L, float weight, float margin, float *loss,
                     float *g_rows, int embed_dim, void *stream);
int hidvae_total_loss(const float *recon, const float *qloss, int64_t B,
                      const float *const *align_host, const float *const *pred_host, const float *const *acc_host,
                      int n_tag, float tag_div,
                      const int64_t *ids, const float *z, int L, float uniq_weight, float uniq_margin,
                      float w_a, float w_p, float w_u, float *loss, float *uniq, float *g_rows, float *tagstats,
                      float *summary, int embed_dim, void *stream);
/* backward of the above for a device scalar g = d/d loss: scal[0] = g/B (per-item grad of recon and qloss),
 * scal[1] = g*w_a, scal[2] = g*w_p (pass w_a/tag_div, w_p/tag_div to get the per-level gradients); g_z [B,32] (optional) = g*w_u*g_rows on rows < L, 0 elsewhere. */
int hidvae_total_loss_bwd(const float *g_loss, int64_t B, int L, float w_a, float w_p, float w_u, const float *g_rows,
                          float *scal, float *g_z, int embed_dim, void *stream);

/* The step's own pairing of the two above with the decoder tail (a3 + a14 + a1): one call forward, ONE launch backward.
 * loss_fwd: recon[b] = |normalize(y[b]) - x[b]|^2 (encoder.py:32, loss.py:11-12) for every row, then the total loss as
 *   hidvae_total_loss.
 * loss_bwd: g_y = (g_loss/B) d recon/d y;  scal / g_z exactly as hidvae_total_loss_bwd.
 * n_cat: categorical columns at the end of the row, as in hidvae_recon_fwd_bwd (0 on every shipped config).
 * expect_g (loss_bwd; 0 = no check): the loss gradient part of the backward was already seeded with (the tag heads' early backward,
 *   hidvae_amd/tagpath.py HeadsGradPort); if *g_loss differs, every output of the launch is NaN -- the step fails visibly instead of
 *   mixing two scalings. */
int hidvae_loss_fwd(const float *y, const float *x, int64_t B, int64_t N, int n_cat, const float *qloss,
                    const float *const *align_host, const float *const *pred_host, const float *const *acc_host, int n_tag,
                    float tag_div, const int64_t *ids, const float *z, int L, float uniq_weight, float uniq_margin, float w_a,
                    float w_p, float w_u, float *recon, float *loss, float *uniq, float *g_rows, float *tagstats,
                    float *summary, int embed_dim, void *stream);
int hidvae_loss_bwd(const float *g_loss, const float *y, const float *x, int64_t B, int64_t N, int n_cat, int L, float w_a, float w_p,
                    float w_u, const float *g_rows, float *g_y, float *scal, float *g_z, int embed_dim, float expect_g, void *stream);

/* ---- a14 as stand-alone modules: ReconstructionLoss.forward (loss.py:7-12) and the two halves of QuantizeLoss.forward
 * (loss.py:36-44) are sums s[m] = sum_j (a[m,j]-b[m,j])^2 over rows (lda/ldb row strides in elements):
 * out[m] = s + extra*s (extra = 0: the reconstruction loss; extra = commitment_weight: emb_loss + cw*query_loss, whose two terms
 * have the same value).  bwd: ga = g[m*g_stride] * scale_a * 2(a-b), gb = g[m*g_stride] * scale_b * 2(b-a); either may be NULL.
 * QuantizeLoss's stop-gradients make scale_a = commitment_weight (query) and scale_b = 1 (value). */
int hidvae_sqdiff_rows(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N, float extra, float *out,
                       void *stream);
int hidvae_sqdiff_rows_bwd(const float *g, int64_t g_stride, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M,
                           int64_t N, float scale_a, float scale_b, float *ga, float *gb, void *stream);

/* CategoricalReconstructionLoss.forward as a stand-alone module (loss.py:15-33), x_hat taken as given (row strides ldh / ldx):
 * out[m] = sum_{j < N-n_cat} (x_hat-x)^2 + sum_{j >= N-n_cat} BCE-with-logits(x_hat, x)   (out may be NULL);
 * g_xhat [M,N] (optional, needs g) = g[m*g_stride] * d out[m] / d x_hat. */
int hidvae_cat_recon_rows(const float *x_hat, int64_t ldh, const float *x, int64_t ldx, int64_t M, int64_t N, int n_cat, const float *g,
                          int64_t g_stride, float *out, float *g_xhat, void *stream);

/* ---- a13: debug statistics (h_rqvae.py:643-648) ----------------------------------------------------
 * embs_norm[b,i] = |emb_cat[b, i*32:(i+1)*32]|; *p_unique = (#distinct id tuples)/B computed by a
 * sort-free hash census (== the reference's O(B^2 L) triu expression), one launch.
 * scratch: 4*B + 3 int64 that the caller zero-fills ONCE when it allocates them and then leaves alone: the table is
 * generation-tagged and the kernel resets its own counters, so consecutive calls (same B, one at a time) share it. */
int hidvae_id_stats(const float *emb_cat, int64_t ld_cat, const int64_t *ids, int64_t B, int L,
                    float *embs_norm, float *p_unique, int64_t *scratch, int embed_dim, void *stream);

/* ---- a16: AdamW (torch.optim.AdamW defaults; train_hidvae.py:533-563,762-766) with the cosine schedule
 * evaluated ON DEVICE from a device step counter, so the whole step is graph-capturable.
 * hidvae_adamw_prepare: one tiny launch per optimizer step that depends on nothing (run it beside forward/backward):
 *   step_dev is int64[2] = {optimizer steps taken, scheduler steps taken before step 0 (non-zero only for a run resumed WITHOUT
 *   its optimizer state: the schedule continues while Adam's bias correction restarts, train_hidvae.py:621-640)};
 *   t = step_dev[0] + 1; s = t - 1 + step_dev[1]; lr_t = eta_min + (base_lr-eta_min)(1+cos(pi*s/T_max))/2 if T_max > 0
 *   (CosineAnnealingLR), else base_lr * gamma^floor(s/step_size) if step_size > 0 (StepLR, train_hidvae.py:641-642), else base_lr;
 *   hyper[i] = {1 - lr_t*wd_i, lr_t/(1-beta1^t), sqrt(1-beta2^t)} (double precision, as torch does on the host);
 *   step_dev[0] = t.
 * hidvae_adamw_step: one launch per <=128 tensors.  p/m/v/numel: DEVICE tables built once; g_host: HOST array of the
 *   n_tensors gradient device pointers (autograd hands out new buffers every step; they travel by value in the kernel
 *   arguments).  grad_scale multiplies g first (1/world_size after an all-reduce SUM). */
int hidvae_adamw_prepare(int64_t *step_dev, const float *base_lr_dev, const float *wd_dev, int n,
                         float beta1, float beta2, float eta_min, int64_t T_max, int64_t step_size, float gamma,
                         float *hyper_dev, void *stream);
/* hidvae_codebook_prepare and hidvae_adamw_prepare in ONE launch (one extra workgroup): both are start-of-step launches that
 * depend on nothing, each far shorter than a launch costs. */
int hidvae_codebook_prepare_adamw(const float *const *E_host, const int32_t *normalize_host, int L, int64_t K, float *cb_eff,
                                  float *cc, int64_t *step_dev, const float *base_lr_dev, const float *wd_dev, int n_tensors,
                                  float beta1, float beta2, float eta_min, int64_t T_max, int64_t step_size, float gamma,
                                  float *hyper_dev, int embed_dim, void *stream);
int hidvae_adamw_step(float *const *p_dev, const float *const *g_host, float *const *m_dev, float *const *v_dev,
                      const int64_t *numel_dev, const float *hyper_dev, int n_tensors, int64_t max_numel,
                      float beta1, float beta2, float eps, float grad_scale, void *stream);

/* ======== tag path (a8-a11): element / row kernels ================================================================ */
/* out = g * act'(ref) * mask*scale: backward through the activation (+dropout) fused into a Linear's epilogue.
 * act = HIDVAE_EPI_{NONE,RELU,GELU,SIGMOID,SILU}; ref = forward OUTPUT for RELU/SIGMOID, pre-activation for GELU/SILU. */
int hidvae_act_bwd(const float *g, const float *ref, int64_t numel, int act, const float *mask, float mask_scale,
                   float *out, void *stream);
/* out = a * b (op 0: TagPredictor gate, h_rqvae.py:208), a + b (op 1: residual add, :219) or a - b (op 2: residual update
 * of the Gumbel branch, h_rqvae.py:300), row strides in elements */
int hidvae_binary(int op, const float *a, int64_t lda, const float *b, int64_t ldb, int64_t M, int64_t N, float *out,
                  int64_t ldo, void *stream);
/* dst[M,N] = sum_s src_s[:, :width_s] zero-extended: gradient of the column-prefix views concat_emb = cat(embs[:i+1])
 * (h_rqvae.py:526) taken of one [B, L*32] buffer.  src_host / width_host: HOST arrays (n <= 24, NULL entries skipped). */
int hidvae_sum_prefix_slices(const float *const *src_host, const int32_t *width_host, int n, int64_t M, int64_t N,
                             float *dst, void *stream);
/* y = dropout(relu?(LayerNorm(x))) + residual   (nn.LayerNorm eps; h_rqvae.py:145,156,162,181,329) and its backward:
 * gx.  The residual's gradient is gy itself. */
/* Dropout inside the producing launch: instead of a keep-mask tensor (keep_mask, kept for injected draws) the launches that end in a
 * Dropout take (rng_state, rng_site, drop_threshold): the keep decision of element i is Philox4x32-10(key = rng_state[0], counter =
 * (i, site, rng_state[1])) >= drop_threshold, with drop_threshold = round(p_drop * 2^32).  rng_state: 2 x uint64 on the device,
 * {seed, step}; hidvae_rng_advance bumps `step` (once per forward pass), so replayed graphs draw fresh masks.  The backward never needs
 * the mask: it reads the gate off the forward OUTPUT (y > 0 exactly where the unit was active and kept).  hidvae_dropout_mask
 * materialises the 0/1 mask such a launch applies (tests; numel elements of site `site`). */
int hidvae_rng_advance(unsigned long long *rng_state, void *stream);
int hidvae_dropout_mask(float *out, int64_t numel, const unsigned long long *rng_state, unsigned site, unsigned threshold, void *stream);
int hidvae_layernorm_fwd(const float *x, int64_t M, int64_t N, const float *gamma, const float *beta, float eps, float *y,
                         float *mean, float *rstd, int relu, const float *keep_mask, float keep_scale,
                         const float *residual, const unsigned long long *rng_state, unsigned rng_site, unsigned drop_threshold,
                         void *stream);
int hidvae_layernorm_bwd(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                         const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask, float keep_scale,
                         float *gx, void *stream);
/* affine gradients (ggamma, gbeta) of the same op, fixed-order two-pass; workspace >= 2*ceil(M/128)*N floats.
 * Independent of hidvae_layernorm_bwd, so it can run on another stream. */
int hidvae_layernorm_param_grad(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask,
                                float keep_scale, float *ggamma, float *gbeta, int accumulate, float *workspace,
                                void *stream);
/* both of the above from one pass over (gy, x) + the fixed-order finish: gx (optional), ggamma, gbeta.
 * workspace: 2 * ceil(M/4) * N floats (2 * ceil(M/128) * N suffices when N > 1024, where the separate kernels run). */
int hidvae_layernorm_bwd_all(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                             const float *rstd, int64_t M, int64_t N, int relu, const float *keep_mask, float keep_scale,
                             float *gx, float *ggamma, float *gbeta, int accumulate, float *workspace, void *stream);
/* The same backward split at its natural seam, for chains of LayerNorms (the tag heads run 21 per step):
 *   hidvae_layernorm_bwd_partial       gx + the per-4-row partials of (ggamma, gbeta) into `partials` (2 * ceil(M/4) * N floats, N <= 1024);
 *                                      y_out (required when relu): the forward OUTPUT y = relu(h) * keep * keep_scale, off which the
 *                                      ReLU -> Dropout gate is read (y > 0 exactly where the unit was active and kept: no mask, no h);
 *                                      in_relu_scale != 0: the LayerNorm's INPUT was itself relu(.) * keep * in_relu_scale (Linear -> ReLU ->
 *                                      Dropout -> LayerNorm, h_rqvae.py:157-162) and gx is returned already taken through that gate;
 *                                      gy2 (optional, [M,N]): a second gradient of the same output, added to gy on the way in -- the residual
 *                                      path of TagPredictor's blocks (h_rqvae.py:165-186) hands the producer of f_n two gradients, which
 *                                      autograd would add in a launch of its own; gsum (optional, [M,N]): gy + gy2 written out (the next
 *                                      residual hop needs it)
 *   hidvae_layernorm_param_final_many  the fixed-order finish of MANY such partial sets in ONE launch (n <= 32 per launch; more are
 *                                      issued in slices): ggamma / gbeta (+)= column sums, same order as hidvae_layernorm_bwd_all */
typedef struct {
    const float *partials; int64_t M, N;
    float *ggamma, *gbeta;
    int accumulate;
} hidvae_ln_final;
int hidvae_layernorm_bwd_partial(const float *gy, const float *x, const float *gamma, const float *beta, const float *mean,
                                 const float *rstd, int64_t M, int64_t N, int relu, const float *y_out, float keep_scale,
                                 float in_relu_scale, const float *gy2, float *gsum, float *gx, float *partials, void *stream);
int hidvae_layernorm_param_final_many(const hidvae_ln_final *problems_host, int n, void *stream);
/* TagPredictor behind its attention gate (h_rqvae.py:132-188: feature_extractor, residual blocks, classifier) as ONE row-local
 * forward launch for levels whose layers are at most 256 wide.  A unit is  Linear(+bias) [-> ReLU -> Dropout(1)] [-> LayerNorm
 * [-> ReLU -> Dropout(2)] [+ carried residual]];  unit i reads unit i-1's output (unit 0 reads h [B, K] at row stride ldh).  `lin`
 * receives what the separate launches would save as the LayerNorm's input / the Linear's output (after ReLU -> Dropout(1) if act1),
 * `y`, `mean`, `rstd` the LayerNorm's outputs -- so hidvae_linear_bwd / hidvae_layernorm_bwd_partial run the backward unchanged.
 * carry: the unit's output becomes the carried residual; residual: the carried residual is added after the LayerNorm and the sum
 * becomes the new carried residual.  Dropout decisions: the counter-based generator on element index row * N + col of the given site
 * (threshold 0: no dropout), exactly as hidvae_gemm_f32 / hidvae_layernorm_fwd take them. */
typedef struct {
    const float *W, *bias, *gamma, *beta;
    int N, K;
    int act1, act2, residual, carry;
    unsigned drop_site1, drop_threshold1, drop_site2, drop_threshold2;
    float drop_scale1, drop_scale2, eps;
    float *lin, *y, *mean, *rstd;
    float *g_lin, *partials;   /* backward only: [B, N] gradient at the Linear's output; [ceil(B/4)][2][N] LayerNorm affine partials */
} hidvae_pred_unit;
int hidvae_predictor_fwd(const float *h, int64_t ldh, int64_t B, const hidvae_pred_unit *units_host, int n_units,
                         const unsigned long long *rng_state, void *stream);
/* the input-gradient chain of the same units in reverse, one launch: g_out [B, N_last] -> g_h [B, K_0]; per unit g_lin (from which
 * dW = g_lin^T in, db = colsum g_lin follow: hidvae_linear_bwd_group) and the LayerNorm partials (hidvae_layernorm_param_final_many
 * finishes them).  The gates are read off the tensors the forward saved (y > 0, lin > 0): no keep decision is re-drawn. */
int hidvae_predictor_bwd(const float *g_out, int64_t ldg, int64_t B, const hidvae_pred_unit *units_host, int n_units, float *g_h,
                         int64_t ldgh, void *stream);
/* The attention gate of TagPredictor (h_rqvae.py:128-139, :196-206) as ONE row-local launch each way (E = 32 (i+1) <= 128, a multiple of 4):
 *   a1 = relu(x W0^T + b0) [B,E/4], a2 = gelu(pre2 = a1 W2^T + b2) [B,E/2], a3 = sigmoid(a2 W4^T + b4) [B,E],
 *   h = x * a3, divided by max(|h|, eps) per row when normalize (nrm receives |h|)
 * backward: gx = d/dx through BOTH uses of x (the gate input and the attention input), and g3 / g2 / g1 = the gradients at the three
 * pre-activations, from which dW_k = g_k^T in_k and db_k = colsum g_k follow (hidvae_linear_bwd_group). */
int hidvae_gate_fwd(const float *x, int64_t ldx, int64_t B, int E, const float *W0, const float *b0, const float *W2,
                    const float *b2, const float *W4, const float *b4, int normalize, float eps, float *a1, float *pre2,
                    float *a2, float *a3, float *h, float *nrm, void *stream);
int hidvae_gate_bwd(const float *gh, int64_t ldgh, const float *x, int64_t ldx, int64_t B, int E, const float *W0,
                    const float *W2, const float *W4, int normalize, float eps, const float *a1, const float *pre2,
                    const float *a3, const float *nrm, float *gx, float *g3, float *g2, float *g1, void *stream);
/* BatchNorm1d (h_rqvae.py:325): y = dropout(relu?(BN(x))).  training != 0: batch statistics (biased variance for the
 * normalisation, unbiased for the running update with `momentum`), saved mean / rstd for the backward;
 * training == 0: running statistics.  num_batches_tracked (optional int64 device scalar) is incremented in training.
 * workspace (3*ceil(M/64)*N floats forward, 2*ceil(M/64)*N backward): selects the row-parallel form (per-chunk statistics merged
 * with Chan's update in ascending chunk order, then a row-parallel apply launch); NULL: one workgroup per 32 columns. */
int hidvae_batchnorm_fwd(const float *x, int64_t ldx, int64_t M, int64_t N, const float *gamma, const float *beta, float eps,
                         float momentum, int training, float *running_mean, float *running_var,
                         int64_t *num_batches_tracked, float *y, float *save_mean, float *save_rstd, int relu,
                         const float *keep_mask, float keep_scale, const unsigned long long *rng_state, unsigned rng_site,
                         unsigned drop_threshold, float *workspace, void *stream);
/* y_out (optional, instead of keep_mask): the forward OUTPUT, off which the ReLU -> Dropout gate is read */
int hidvae_batchnorm_bwd(const float *gy, const float *x, int64_t ldx, const float *gamma, const float *beta,
                         const float *save_mean, const float *save_rstd, int64_t M, int64_t N, int relu,
                         const float *keep_mask, float keep_scale, const float *y_out, float *gx, float *ggamma, float *gbeta,
                         int accumulate, float *workspace, void *stream);
/* InfoNCE (loss.py:54-85) on the similarity matrix S = normalize(c) normalize(t)^T [B,B] produced by hidvae_gemm_f32:
 * rows: loss = scale * mean_b( logsumexp_j(S_bj/tau) - S_bb/tau ), S overwritten by softmax(S/tau);
 * dlogits: P <- (g*scale/(B*tau)) (P - I) in place, g a device scalar -- feed it to two GEMMs for d c / d t. */
int hidvae_infonce_rows(float *S, int64_t B, float tau, float scale, float *row_loss, float *loss, void *stream);
/* The same loss WITHOUT the B x B matrix (B >= 4096; the 20,000-item warm-up forward of train_hidvae.py:692-696 would hold 1.6 GB of
 * softmax per level): the caller produces S in column chunks Sc [B, C] = n(c) n(t)[col0:col0+C]^T and hands each one over at once.
 *   lse_chunk   : online logsumexp per row (running maximum m [B] and sum l [B], chunks in ascending column order, first != 0 on the
 *                 first one), diag [B] captures S[b,b] when its column passes;
 *   lse_finish  : lse[b] = m + log l (kept for the backward), row_loss = lse - diag/tau, *loss = scale * mean;
 *   dlogits_chunk (backward, on a RECOMPUTED chunk): Sc <- (g*scale/(B*tau)) (exp(Sc/tau - lse) - I[:, col0:col0+C]) in place. */
int hidvae_infonce_lse_chunk(const float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0, float tau, float *m, float *l, float *diag,
                             int first, void *stream);
int hidvae_infonce_lse_finish(const float *m, const float *l, const float *diag, int64_t B, float tau, float scale, float *row_loss,
                              float *lse, float *loss, void *stream);
int hidvae_infonce_dlogits_chunk(float *Sc, int64_t B, int64_t C, int64_t ldc, int64_t col0, float tau, float scale, const float *lse,
                                 const float *g_dev, void *stream);
int hidvae_infonce_dlogits(float *P, int64_t B, float tau, float scale, const float *g_dev, void *stream);
/* The mixup plan of one training step for all L levels (loss.py:139-147: perm = randperm(n_valid), lam ~ Beta(alpha, alpha)):
 * targets [B, ld>=L] int64 (-1 = invalid row); uniforms [L, B+64] in [0,1) from the caller's generator (B sort keys + 64 spare
 * draws for the gamma sampler, per level), or NULL with rng_state: then the kernel draws them itself (counter-based, see hidvae_rng_advance).  partner [L,B]: partner[l][b] = the row mixed into b, a uniformly random permutation
 * of level l's valid rows among themselves, -1 on invalid rows; inverse [L,B]: inverse[l][partner[l][b]] = b; lam [L].  B <= 16384 (the sort keys and row numbers of a level live in LDS, 6 bytes per row). */
int hidvae_mixup_plan(const int64_t *targets, int64_t B, int L, int64_t ld_targets, const float *uniforms, float alpha,
                      int64_t *partner, int64_t *inverse, float *lam, const unsigned long long *rng_state, void *stream);
/* TagPredictionLoss (loss.py:107-265) with the model's layer_idx = 0 call.  target -1 = invalid row.
 * partner[b] (NULL = no mixup): original-index row mixed into row b with weight 1-lam (loss.py:144-154); lam is a
 * DEVICE scalar (it is drawn per step, also under graph replay).
 * focal != 0: focal loss with label smoothing `smooth` (0 under no_grad), gamma, alpha (loss.py:230-265);
 * focal == 0: cross entropy with label smoothing ce_label_smoothing + 0.05*KL(uniform || softmax(unmixed)+1e-8).
 * Writes loss/acc/n_valid (device scalars); row_loss,row_hit [B] and zbuf [B,C] are scratch; dmix/dkl [B,C] (optional)
 * hold what hidvae_tag_loss_bwd needs: g_logits = (g/n_valid)(lam*dmix[r] + (1-lam)*dmix[inverse[r]] + dkl[r]). */
int hidvae_tag_loss_fwd(const float *logits, int64_t B, int64_t C, const int64_t *target, const int64_t *partner,
                        const float *lam_dev,
                        int focal, float gamma, float alpha, float smooth, float ce_label_smoothing, float *loss,
                        float *acc, float *n_valid, float *row_loss, float *row_hit, float *zbuf, float *dmix, float *dkl,
                        void *stream);
int hidvae_tag_loss_bwd(const float *dmix, const float *dkl, const int64_t *target, const int64_t *inverse,
                        const float *lam_dev,
                        int64_t B, int64_t C, const float *g_dev, const float *n_valid, float *g_logits, void *stream);

/* ---- a15: one Lloyd iteration of the first-batch k-means codebook init (init/kmeans.py:34-77; quantize.py:91-95), D = 32:
 * assign[i] = argmin_k sum_d (x_id - c_kd)^2 (first minimum), new_centroids[k] = mean of its items (ascending item order;
 * an empty cluster takes x[reseed_idx[k]]), *shift = max_k |new_k - old_k|_2 (the reference stops below 1e-10).
 * shift_scratch: K floats. */
int hidvae_kmeans_iter(const float *x, int64_t N, const float *centroids, int64_t K, int32_t *assign,
                       const int64_t *reseed_idx, float *new_centroids, float *shift_scratch, float *shift, int embed_dim, void *stream);

/* ---- a6: GUMBEL_SOFTMAX training branch of one level (quantize.py:125-130, distributions/gumbel.py:8-18); row kernels
 * around the GEMMs (S = x cb^T, emb = P cb and their gradients run on hidvae_gemm_f32).  Any embed_dim <= 64.
 * rows_fwd : S [B,K] in -> P = softmax((-(|x|^2+cc-2S) + G)/T) in place, G from the uniform draws U [B,K]; ids = argmin dist.
 *            cosine != 0 (QuantizeDistance.COSINE, quantize.py:115-119): S arrives as x^ c^T of the NORMALISED operands, dist = -S
 *            (x and cc are not read then and may be NULL).
 * loss     : loss[b] = (1+beta)|x-emb|^2.        gemb: g_emb = g_out + g_l[b*stride]*2(emb-x).
 * rows_bwd : gP [B,K] (= g_emb cb^T) -> g_S in place (for the L2 distance: d dist / d S = -2; halve it for the cosine form),
 *            g_xx[b] = d/d|x_b|^2.
 * finish   : g_x += 2 x g_xx + g_l 2 beta (x-emb);  g_cb += 2 cb * (-colsum(g_S)/2). */
int hidvae_gumbel_rows_fwd(float *S, const float *x, const float *cc, const float *U, int64_t B, int64_t K,
                           float temperature, int64_t *ids, int embed_dim, int cosine, void *stream);
int hidvae_gumbel_loss(const float *x, const float *emb, int64_t B, float beta, float *loss, int embed_dim, void *stream);
int hidvae_gumbel_gemb(const float *g_out, const float *g_l, int64_t gl_stride, const float *x, const float *emb, int64_t B,
                       float *g_emb, int embed_dim, void *stream);
int hidvae_gumbel_rows_bwd(const float *P, float *gP, int64_t B, int64_t K, float temperature, float *g_xx, void *stream);
int hidvae_gumbel_finish(float *g_x, const float *x, const float *emb, const float *g_xx, const float *g_l,
                         int64_t gl_stride, float beta, int64_t B, float *g_cb, const float *cb, const float *gS_colsum,
                         int64_t K, int embed_dim, void *stream);

/* distributions/gumbel.py:8-18 as stand-alone functions: G = -log(-log(U+eps)+eps) elementwise; out = softmax((logits+G(U))/T)
 * per row.  U: uniform draws in [0,1) from the caller's generator (the library holds no RNG state). */
int hidvae_gumbel_noise(const float *U, int64_t n, float eps, float *out, void *stream);
int hidvae_gumbel_softmax_rows(const float *logits, const float *U, int64_t B, int64_t K, float temperature, float *out,
                               void *stream);
/* HRqVae.predict_tags (h_rqvae.py:716-722: torch.softmax(logits, -1).max(-1)): per row of logits [B, ld >= C] the first maximal
 * column -> pred[row * pred_stride] and its softmax probability 1 / sum_j exp(l_j - l_max) -> conf[row * conf_stride]; the strides
 * let a level write its column of the stacked [B, L] outputs in place. */
int hidvae_softmax_argmax_rows(const float *logits, int64_t B, int64_t C, int64_t ld_logits, int64_t *pred, int64_t pred_stride,
                               float *conf, int64_t conf_stride, void *stream);

/* ---- measurement aid: *slot = the device's constant-rate wall clock (wall_clock64: 100 MHz, 10 ns ticks), stored by a
 * one-wave kernel launched on `stream`.  Launched before and after another launch INSIDE a captured step it brackets that launch's
 * in-step duration (HIP events cannot be recorded inside a captured graph on ROCm); bench.py's per-kernel timeline uses it. */
int hidvae_timestamp(int64_t *slot, void *stream);

/* ---- the stage-2 transformer's padded -> jagged copy (reference ops/triton/jagged.py:9-124, a Triton kernel
 * there).  x [B, N, D] of any element type (strides in BYTES, D contiguous, row_bytes = D * element size); offsets [B+1] = the
 * exclusive scan of lengths (int64, device); values [offsets[B], D] receives the first lengths[b] rows of every entry back to
 * back.  jagged_to_padded is the adjoint (the reference's backward: padding rows become zero). */
int hidvae_padded_to_jagged(const void *x, int64_t stride_b_bytes, int64_t stride_n_bytes, const int64_t *offsets, void *values,
                            int64_t B, int64_t N, int64_t row_bytes, void *stream);
int hidvae_jagged_to_padded(const void *values, const int64_t *offsets, void *x, int64_t stride_b_bytes, int64_t stride_n_bytes,
                            int64_t B, int64_t N, int64_t row_bytes, void *stream);

/* ---- the training loop's batch formation (reference data/tags_processed.py:112-150: ItemData.__getitem__ indexes item_data,
 * tags_emb and tags_indices with the batch's ids; train_hidvae.py:698-701 hands the batch to the step).  ONE launch gathers the rows
 * idx[0 .. rows) of up to HIDVAE_GATHER_MAX resident tables (host arrays of n_tables device pointers: src[t] has src_rows[t] rows of
 * row_bytes[t] bytes, whole dwords, dword-aligned; dst[t] receives `rows` rows) -- e.g. straight into the step's input buffers.
 * idx: int64 on the device, every id in [0, src_rows[t]) (an id outside leaves that destination row untouched). */
#define HIDVAE_GATHER_MAX 4
int hidvae_gather_rows(const int64_t *idx, int64_t rows, int n_tables, const void *const *src, void *const *dst,
                       const int64_t *row_bytes, const int64_t *src_rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HIDVAE_H */
