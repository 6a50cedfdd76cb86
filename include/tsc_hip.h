/*
 * tsc_hip.h -- C-ABI of libtsc_hip.so, the MI355X (gfx950) backend of the feature-activation
 * hot path of charlieoneill11/transcoder-sae-circuits.
 *
 * The reference has no FFI: the path sits behind Python methods.  Each
 * entry point below names the reference code it replaces (paths under the reference's src/).
 * Conventions:
 *   - every pointer marked "device" is a HIP device pointer owned by the caller (in practice a
 *     torch tensor's data_ptr()); the library never returns memory the caller must free except
 *     the opaque dictionary handle (tsc_dict_destroy);
 *   - all tensors are dense row-major; fp32 unless said otherwise; token ids are int64;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     the call returns without synchronising;
 *   - return value 0 = ok, negative = error (tsc_last_error() gives the thread's message);
 *     nothing throws across the boundary.
 */
#ifndef TSC_HIP_H
#define TSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSC_OK 0
#define TSC_ERR_ARG (-1)         /* bad shape / null pointer / misaligned pointer            */
#define TSC_ERR_HIP (-2)         /* a HIP runtime call failed                                */
#define TSC_ERR_UNSUPPORTED (-3) /* shape outside what the kernels are built for             */

#define TSC_F32 0  /* fp32 inputs/weights, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)            */
#define TSC_BF16 1 /* bf16-rounded inputs/weights, fp32 accumulate (v_mfma_f32_32x32x16_bf16)  */
#define TSC_SCREEN_MAX_ROWS 65536 /* rows of the bf16 staging buffer of a TSC_BF16 / TSC_F32_SCREEN handle (100 MB at d_in 768) */
#define TSC_F32_SCREEN 2 /* TSC_F32 for every entry point; tsc_sparse_lens_pass screens with the bf16 MFMA first */

typedef struct tsc_dict tsc_dict_t; /* opaque: one transcoder or z-SAE resident on one GPU */
typedef void *tsc_stream_t;         /* hipStream_t */

/* Library version (major*100+minor) and the last error text of the calling thread. */
int tsc_version(void);
const char *tsc_last_error(void);

/*
 * Register one dictionary.  Replaces the parameter set of SparseTranscoder.__init__
 * (mlp_transcoder.py:70-99) / ZSAE.__init__ (z_sae.py:34-47).  Pointers are BORROWED and must
 * outlive the handle: W_enc [d_in,d_sae], b_enc [d_sae], b_dec [d_in], W_dec [d_sae,d_out],
 * out_bias [d_out] (= b_dec_out for a transcoder, = b_dec for a z-SAE; mlp_transcoder.py:131-147,
 * z_sae.py:62).  W_dec/out_bias may be NULL if only encode entry points are used.
 * compute_dtype TSC_BF16 (needs d_in % 128 == 0) makes one device-side transposed bf16 copy of W_enc and
 * a TSC_SCREEN_MAX_ROWS-row bf16 staging buffer for x (hipMalloc, freed by destroy); such a handle computes
 * relu(bf16(x - b_dec) @ bf16(W_enc) + b_enc) with fp32 accumulation, and its calls must not overlap
 * on different streams (they share the staging buffer).
 * compute_dtype TSC_F32_SCREEN (needs d_in % 128 == 0) is TSC_F32 plus the operands of the bf16 SCREEN of
 * tsc_sparse_lens_pass (the bf16 copy and staging buffer of TSC_BF16, and ||W_enc[:, n]||_2 per feature); results
 * of every entry point are fp32 results, see tsc_sparse_lens_pass.
 * Every handle also owns bias' = b_enc - b_dec @ W_enc ([d_sae] fp32, summed in fp64): the encode
 * kernels contract the raw x and add bias', which equals the reference's
 * (x - b_dec) @ W_enc + b_enc (mlp_transcoder.py:200-212) up to fp32 rounding -- the same folding
 * the reference itself uses in get_feature_scores (autointerpretability.py:228-233).
 * Requires d_in % 4 == 0, d_sae % 4 == 0 and d_out % 4 == 0, and 16-byte aligned pointers (all five).
 *
 * tsc_dict_refresh recomputes what the handle derived from the weights (bias', the transposed copy, the
 * bf16 copy); call it after the borrowed tensors were updated in place (an optimiser step on the
 * reference's nn.Parameters keeps the pointers but changes the values).
 *
 * Stream order: tsc_dict_create / tsc_dict_refresh ENQUEUE the derivation kernels on `stream` and
 * return; the derived operands are ordered on that stream only.  An encode call on another stream must
 * be ordered after it by the caller (an event, or a synchronize), like any other producer/consumer pair.
 */
int tsc_dict_create(tsc_dict_t **out, int d_in, int d_sae, int d_out, int compute_dtype,
                    const float *W_enc, const float *b_enc, const float *b_dec,
                    const float *W_dec, const float *out_bias, tsc_stream_t stream);
int tsc_dict_refresh(tsc_dict_t *dict, tsc_stream_t stream);
int tsc_dict_destroy(tsc_dict_t *dict);

/*
 * out[M,d_sae] = relu((x[M,d_in] - b_dec) @ W_enc + b_enc).
 * Replaces SparseTranscoder.encode (mlp_transcoder.py:194-213) and ZSAE.encode (z_sae.py:73-77).
 */
int tsc_encode_dense(const tsc_dict_t *dict, const float *x, int64_t M, float *out,
                     tsc_stream_t stream);
/*
 * out[M,d_sae] = (x - b_dec) @ W_enc + b_enc, the pre-activation: what the reference exposes as
 * hook_hidden_pre (mlp_transcoder.py:204-211) and what get_feature_scores' raw mode computes for selected
 * columns (autointerpretability.py:226-238).  A handle made from a few selected encoder columns turns that
 * column selection into a skinny MFMA GEMM (tsc_hip.FeatureProjector).
 */
int tsc_encode_dense_pre(const tsc_dict_t *dict, const float *x, int64_t M, float *out,
                         tsc_stream_t stream);

/*
 * Running per-feature top-k of max-activating tokens.  The state is caller-owned device memory:
 *   vals [n_slots, d_sae, k] fp32 and idx [n_slots, d_sae, k] int64.
 * A "slot" is an independent partial list set; the encode kernel splits each batch's token tiles
 * over n_slots groups of workgroups so that no two workgroups ever touch one list, and
 * tsc_topk_finalize merges the slots.  tsc_topk_recommended_slots(d_sae) returns the slot count
 * that fills the chip evenly and -- 8 where that is possible -- keeps the W_enc slabs of the column tiles
 * resident on one XCD within its L2 (any n_slots >= 1 is correct).
 * floor (optional, may be NULL): [d_sae] fp32 scratch.  When given and n_slots > 1, tsc_encode_topk first
 * writes each feature's current k-th best value over ALL slots into it and the encode kernel dismisses
 * candidates below it -- they cannot be in the merged result -- instead of only those below the k-th best
 * of their own slot (n_slots times fewer list updates).  A floor stays a valid (if weaker) bound for as long
 * as the lists only grow, so a caller may skip the recomputation on some calls: flags bit 0
 * (TSC_TOPK_FLOOR_KEEP) = "floor still holds what an earlier tsc_encode_topk on this state wrote and the lists
 * were not reset or replaced since: use it as it is".
 */
#define TSC_TOPK_FLOOR_KEEP 1
#define TSC_TOPK_MERGE_MAX 2048 /* n_lists * k of one merge; a state needs n_slots * k <= this */
typedef struct {
    float *vals;   /* device */
    int64_t *idx;  /* device */
    int32_t d_sae;
    int32_t k;       /* 1..64 */
    int32_t n_slots; /* 1..64, and n_slots * k <= TSC_TOPK_MERGE_MAX (checked on every call that takes a state) */
    int32_t flags;   /* TSC_TOPK_FLOOR_KEEP or 0 */
    float *floor;  /* device, [d_sae] scratch, or NULL */
} tsc_topk_state_t;

int tsc_topk_recommended_slots(int d_sae);

/*
 * Kernel-variant override for tests and measurements (no reference counterpart; results are identical for
 * every setting).  key/value: "f32_tile" or "bf16_tile" = 0 (auto) | 128 | 256; "staging" = 0 (auto: LDS-DMA) |
 * 1 (register staging); "lists" = 0 (auto: list values mirrored in LDS) | 1 (lists in HBM only).
 * Process-wide; not meant to be changed while calls are in flight.
 */
int tsc_set_tuning(const char *key, int value);
/* Fill every list with the padding entry (0.0, -1). */
int tsc_topk_state_reset(const tsc_topk_state_t *state, tsc_stream_t stream);

/*
 * Fused encode + top-k insert for tokens token_id_base .. token_id_base+M-1: the [M,d_sae]
 * activation matrix is never written.  Only activations > 0 are candidates; ranking is value
 * descending then token id ascending.  Replaces the loop body of MaxActAnalysis.active_examples
 * (max_act_analysis.py:215-249: encode, [:, feature], .cpu(), concat, argsort) for ALL features
 * at once, and the dense dump of max_activating.py:41-65,117-120.
 */
int tsc_encode_topk(const tsc_dict_t *dict, const float *x, int64_t M, int64_t token_id_base,
                    const tsc_topk_state_t *state, tsc_stream_t stream);

/*
 * tsc_encode_topk for callers that cannot allocate or carry a workspace: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (it has no native code and
 * no allocator to avoid).  Same arithmetic, same lists bit for bit; the only difference is where the
 * cross-slot floor scratch lives: not in state->floor (ignored, as is TSC_TOPK_FLOOR_KEEP: the floor is
 * recomputed on every call) but in a file-scope device array of the library, 512 KiB per GPU, held until the
 * process ends.  Limit: state->d_sae <= TSC_STATIC_FLOOR_MAX_D_SAE, larger states are refused with
 * TSC_ERR_UNSUPPORTED.  The call makes no hipMalloc / hipFree and calls no allocating routine.  Every call on
 * one GPU shares that GPU's array: calls that overlap on one GPU (different streams or threads) race on it --
 * serialise them, or use tsc_encode_topk with a floor per state.  (A TSC_BF16 handle still uses the staging
 * buffer tsc_dict_create allocated; this call allocates nothing for it either.)
 */
#define TSC_STATIC_FLOOR_MAX_D_SAE 131072
int tsc_encode_topk_static(const tsc_dict_t *dict, const float *x, int64_t M, int64_t token_id_base,
                           const tsc_topk_state_t *state, tsc_stream_t stream);

/*
 * tsc_encode_topk at the bf16 matrix rate with the lists of tsc_encode_topk, BIT FOR BIT (TSC_F32_SCREEN handles).
 * Per chunk of TSC_SCREEN_MAX_ROWS rows: (1) the bf16 MFMA kernel computes bf16(x - b_dec) @ bf16(W_enc) + b_enc + margin[n]
 * (margin: the proven bound of tsc_sparse_lens_pass) and queues every (token, feature) pair whose value reaches the
 * feature's floor -- the k-th best over all slots before the call, state->floor, required -- one queue per
 * workgroup in `workspace`; (2) a second kernel recomputes the queued pairs exactly in fp32, in the accumulation
 * order of the fp32 MFMA kernel (a scalar fma chain reproduces v_mfma_f32_32x32x2_f32 bit for bit), and folds them
 * into the lists; (3) the fp32 fused kernel is launched on the same chunk and returns at once unless a queue
 * overflowed in (1) -- then (2) skipped the chunk and (3) does it exactly, so nothing is ever lost and no host
 * round trip is needed.  Same ranking rule, same state layout, interchangeable with tsc_encode_topk call by call.
 * Worth it once the lists are warm (few candidates per launch); on empty lists every positive activation is a
 * candidate and step (3) usually takes over.  workspace: device memory of tsc_topk_screen_workspace_bytes(d_sae,
 * n_slots) bytes, scratch between calls (calls sharing one workspace must be ordered on one stream).
 * screen_row_bound as in tsc_sparse_lens_pass, over ALL rows of the call (NULL: computed per chunk).
 * No reference counterpart beyond tsc_encode_topk's (max_act_analysis.py:215-249).
 */
size_t tsc_topk_screen_workspace_bytes(int d_sae, int n_slots);
int tsc_encode_topk_screened(const tsc_dict_t *dict, const float *x, int64_t M, int64_t token_id_base,
                             const tsc_topk_state_t *state, void *workspace, size_t workspace_bytes,
                             const float *screen_row_bound, tsc_stream_t stream);

/*
 * Merge n_lists list sets laid out [n_lists, d_sae, k] into out_vals/out_idx [d_sae, k], sorted
 * (value desc, token id asc), short lists padded with (0.0, -1).  n_lists * k <= 2048.
 * tsc_topk_finalize = merge of the state's slots.  tsc_topk_merge is also the step after the
 * RCCL all-gather of per-rank lists (no reference counterpart: the reference is single-process,
 * SURVEY.md section 2.2).  out may not alias the inputs.
 */
int tsc_topk_merge(const float *vals, const int64_t *idx, int n_lists, int d_sae, int k,
                   float *out_vals, int64_t *out_idx, tsc_stream_t stream);
int tsc_topk_finalize(const tsc_topk_state_t *state, float *out_vals, int64_t *out_idx,
                      tsc_stream_t stream);

/*
 * Sparse emit: for every token row m, the active features (act > 0) sorted by value descending
 * (ties: feature id ascending): feat[m, 0..count[m]) int32, val[m, 0..count[m]) fp32, with row
 * pitch max_active (<= 8192).  count[m] is the TRUE number of active features; if it exceeds
 * max_active only max_active arbitrary ones were kept and the caller must retry with a larger
 * capacity.  Replaces `nonzero().numel()` + `topk(k=count)` of CircuitLens.get_active_features
 * (circuit_lens.py:612-614, 647-649) for a whole batch of positions.
 */
int tsc_encode_sparse(const tsc_dict_t *dict, const float *x, int64_t M, int max_active,
                      int32_t *feat, float *val, int32_t *count, tsc_stream_t stream);

/*
 * The sparse form of a dense activation matrix acts [M, d_sae] (e.g. tsc_encode_dense's output): per row the positive
 * entries as feat / val [M, max_active] in a FIXED order (not sorted by value; a function of the column only), count[m]
 * = their true number (entries beyond max_active are dropped: check count), or -1 - that number for a row that holds a
 * negative or NaN entry (not the output of a ReLU; a negative count makes every consumer skip the row's list).  With tsc_decode_sparse this replaces the
 * decoder half of SparseTranscoder.forward / ZSAE.forward on many rows -- `feature_acts @ W_dec` (mlp_transcoder.py:
 * 128-138, z_sae.py:62) multiplies a matrix that is 99.7 % zeros: 618 GFLOP per 16384 x 24576 batch -- by one streaming
 * read of the activations and a gather of the active features' W_dec rows.  The reconstruction then comes from exactly
 * the activations the caller gets back.  d_sae % 4 == 0, acts 16-byte aligned.
 */
int tsc_dense_to_sparse(const float *acts, int64_t M, int d_sae, int max_active, int32_t *feat, float *val, int32_t *count,
                        tsc_stream_t stream);

/*
 * rows[i, :] = val[i] * W_dec[feat[i], :] for i < n.  Replaces `W_dec[idx] * vals.unsqueeze(-1)`
 * (circuit_lens.py:616, 651-653, 1220-1222).  An entry with feat[i] outside 0..d_sae-1 (the padding value -1
 * of tsc_encode_sparse's rows) gives a zero row.
 */
int tsc_decode_rows(const tsc_dict_t *dict, const int32_t *feat, const float *val, int64_t n,
                    float *rows, tsc_stream_t stream);

/*
 * recon[m, :] = out_bias + sum_i val[m,i] * W_dec[feat[m,i], :], i < min(count[m], max_active).
 * Replaces the decoder half of SparseTranscoder.forward (mlp_transcoder.py:128-147) and
 * ZSAE.forward (z_sae.py:62) on the sparse form.
 */
int tsc_decode_sparse(const tsc_dict_t *dict, const int32_t *feat, const float *val,
                      const int32_t *count, int64_t M, int max_active, float *recon,
                      tsc_stream_t stream);

/*
 * Contribution scores, the north star's `feature_act x W_dec row . target direction`:
 *   scores[m, i, d] = val[m,i] * dot(W_dec[feat[m,i], :], dirs[d, :]),  i < min(count[m], max_active),
 * dirs [n_dir, d_out], n_dir <= 8.  Entries at i >= count[m] are written as 0.
 * Replaces the einsum of the lens functions on feature rows (circuit_lens.py:762-766, 955-957)
 * and get_transcoder_ixg (circuit_lens.py:1151-1164) restricted to the active features.  For an
 * attention dictionary pass dirs' = W_O[layer] (as [d_out, d_model]) @ dir, which folds the W_O
 * projection of circuit_lens.py:618-627 into the direction.
 */
int tsc_contrib_scores(const tsc_dict_t *dict, const int32_t *feat, const float *val,
                       const int32_t *count, int64_t M, int max_active, const float *dirs,
                       int n_dir, float *scores, tsc_stream_t stream);

/*
 * tsc_contrib_scores (n_dir = 1, a direction per row) for a window of rows of SEVERAL dictionaries in one launch:
 *   scores[i, m, e] = val[i, m, e] * dot(W_dec_i[feat[i, m, e], :], dirs[i][m * dir_row_stride ..]),  e < count[i, m],
 * where feat / val / count are windows of M rows of stacked pass outputs whose dictionaries lie dict_pitch_rows rows
 * apart (feat / val: [.., dict_pitch_rows, max_active], count: [.., dict_pitch_rows]; pass pointers to the window's
 * first row of dictionary 0), dirs is a HOST array of n_dicts device pointers (dir_row_stride 0 = one direction for
 * all M rows), and scores [n_dicts, M, out_max_active] is compact, zero beyond count.  Optional compact copies of the
 * window: feat_out [n_dicts, M, out_max_active] (-1 beyond count), count_out [n_dicts, M] -- what tsc_lens_assemble
 * takes.  out_max_active (0 = max_active) is the row pitch of these outputs: a caller that knows the window's counts
 * (it read them after the pass) asks for lists just long enough, so that the assemble step does not walk a capacity
 * that one crowded position of the pass dictated; counts above it are clamped to it.
 * This is the per-node lens call of the circuit search -- ONE position against one direction over the components of
 * all dictionaries (circuit_lens.py:753-778, 940-968, 974-1004, 1031-1054, 1085-1107: `vectors @ direction` on the
 * feature rows) -- without building the [components, d_model] matrix and without a launch per dictionary; attention
 * dictionaries take dirs' = W_O[layer] @ dir as in tsc_contrib_scores.  Bit-identical to tsc_contrib_scores.
 * n_dicts <= 32, same d_out for all.
 */
int tsc_contrib_scores_many(int n_dicts, const tsc_dict_t *const *dicts, int64_t M, int64_t dict_pitch_rows, int max_active,
                            const int32_t *feat, const float *val, const int32_t *count, const float *const *dirs,
                            int64_t dir_row_stride, float *scores, int32_t *feat_out, int32_t *count_out,
                            int out_max_active, tsc_stream_t stream);

/*
 * The CircuitLens decomposition of a BATCH of positions against one dictionary in two launches: the sparse emit
 * of tsc_encode_sparse, then one kernel that, per row, orders the emitted pairs (value descending) and reads each
 * active feature's W_dec row once for both
 *   recon[m, :]  = out_bias + sum_i val[m,i] * W_dec[feat[m,i], :]          (as tsc_decode_sparse; NULL = skip)
 *   scores[m, i] = val[m,i] * dot(W_dec[feat[m,i], :], dirs[m * dir_row_stride .. ])   (NULL = skip)
 * i.e. tsc_contrib_scores with n_dir = 1 and a direction PER ROW (dir_row_stride >= d_out: every prompt has its
 * own logit-difference direction, task_evaluation.py:141-151) or one shared direction (dir_row_stride = 0).
 * scores has row pitch max_active; entries at i >= count[m] are 0.  Outputs feat/val/count as tsc_encode_sparse.
 * flags: TSC_LENS_COUNT_IS_ZERO = the caller has already zeroed count (one memset for the counts of many
 * dictionaries); TSC_LENS_NO_SCREEN = plain fp32 emit even on a TSC_F32_SCREEN handle (see below).
 * Replaces, for all positions at once, the per-position z-SAE / transcoder forward, error terms, nonzero + topk,
 * W_dec gather (circuit_lens.py:585-653) and the lens dot product on the feature rows (:762-766).
 * Needs d_out <= 768 and max_active <= 2048 (else TSC_ERR_UNSUPPORTED: use the three separate calls).
 *
 * On a TSC_F32_SCREEN handle the emit is a two-stage EXACT computation: (1) the bf16 MFMA kernel (16x the fp32
 * matrix rate) computes bf16(x - b_dec) @ bf16(W_enc) + b_enc + margin[n] and emits every feature for which that
 * is positive -- margin[n] is a proven bound of the difference between the exact fp32 pre-activation and the bf16 one,
 * so no feature with a positive exact activation is missed; (2) the second kernel recomputes each candidate exactly
 * in fp32 (dot(x, W_enc^T[f]) + bias') and keeps the positive ones.  The margin uses the ACTUAL rounding errors of the
 * call's rows and of the dictionary's columns, not their worst case: with xc = x - b_dec, xt = bf16(xc), wt = bf16(w),
 *   |sum xt wt - sum xc w| <= ||xt - xc|| ||wt|| + ||xc|| ||wt - w||      (Cauchy-Schwarz on both rounding terms)
 *   margin[n] = bound[0] * max(||w_n||, ||wt_n||) + bound[1] * ||wt_n - w_n|| + g |b_enc[n]|,   g = 2 d_in 2^-24,
 *   bound[0] = max_m (||xt_m - xc_m|| + g ||xc_m|| + g (||x_m|| + ||b_dec||)),   bound[1] = max_m ||xc_m||
 * (the g terms: fp32 accumulation error of both sums).  On ordinary data this is 0.4x the worst-case bound
 * 2^-7 ||xc|| ||w|| (one bf16 rounding loses up to 2^-8 relative, 2^-9.3 in the root mean square).
 * Values are fp32 values (summed in a different order than tsc_encode_dense: they agree to ~1e-7 relative, and a
 * feature whose pre-activation is within that of zero may be active in one and not the other).  count[m] above
 * max_active then means "the CANDIDATES overflowed" (an upper bound of the active count): retry with that capacity.
 * screen_row_bound: device pointer to TWO floats >= (bound[0], bound[1]) above (tsc_hip.screen_row_bound computes them
 * with torch), or NULL = the library computes them (one more small kernel); ignored by other
 * handles.  At most TSC_SCREEN_MAX_ROWS rows per call on such a handle.
 */
#define TSC_LENS_COUNT_IS_ZERO 1
#define TSC_LENS_NO_SCREEN 2
#define TSC_LENS_ROW_BOUNDS 16 /* screen_row_bound(s) point at 2 + 2 M floats: the two maxima, then every row's own pair
                                 (bound[0], bound[1]) of the formula above taken without the max.  A candidate that passes
                                 with the maxima is re-tested with its row's pair: rows shorter than the longest row of the
                                 call get their own, tighter margin.  (Bounds the library computes always have this form.) */
#define TSC_LENS_EMIT_ONLY 4 /* stop after the emit: feat/val/count hold the unordered emitted pairs (on a screening
                                handle: the candidates); finish with tsc_sparse_lens_finish_many */
int tsc_sparse_lens_pass(const tsc_dict_t *dict, const float *x, int64_t M, int max_active, int32_t *feat,
                         float *val, int32_t *count, int flags, const float *dirs,
                         int64_t dir_row_stride, float *recon, float *scores, const float *screen_row_bound,
                         tsc_stream_t stream);

/*
 * The emits (tsc_sparse_lens_pass with TSC_LENS_EMIT_ONLY) of SEVERAL dictionaries on the same number of rows in one
 * call: dictionary i reads x[i] [M, d_in] and writes its slice of the stacked feat / val [n_dicts, M, max_active] and
 * count [n_dicts, M].  Screening handles of one input width (the CircuitLens case) run as ONE grouped launch on
 * streams[0]: a 512-row emit is 1.5 rounds of single-tile workgroups, and emits on different streams do not pack into
 * each other's idle half round -- one grid over all dictionaries does.  Otherwise dictionary i runs on
 * streams[i % n_streams].  Either way the caller orders ALL the streams before and after.  dicts, x,
 * screen_row_bounds (or NULL), streams: HOST arrays.  flags as tsc_sparse_lens_pass.  Results are those of the
 * separate calls (the order in which a row's pairs are emitted is unspecified either way).  Every handle may appear
 * ONCE (a handle's staging buffer serves one problem at a time): a repeated handle is refused with TSC_ERR_ARG.
 */
int tsc_sparse_lens_emit_many(int n_dicts, const tsc_dict_t *const *dicts, const float *const *x, int64_t M,
                              int max_active, int32_t *feat, float *val, int32_t *count, int flags,
                              const float *const *screen_row_bounds, const tsc_stream_t *streams, int n_streams);

/*
 * The second kernel of tsc_sparse_lens_pass for the rows of SEVERAL dictionaries in one launch (after their emits with
 * TSC_LENS_EMIT_ONLY, in stream order): at CircuitLens sizes one dictionary's rows are two workgroups per CU, all 24
 * dictionaries of a pass together fill the chip.  Outputs and emitted pairs are STACKED: feat / val [n_dicts, M,
 * max_active], count [n_dicts, M], recon [n_dicts, M, d_out] or NULL, scores [n_dicts, M, max_active] or NULL.
 * dicts, x, dirs: HOST arrays of n_dicts pointers (handles; the inputs each emit saw -- used by screening handles
 * only; per-dictionary directions [M, d_out] with row stride dir_row_stride, or one shared row with stride 0).
 * All handles screened or all plain (flags: TSC_LENS_NO_SCREEN as above), same d_in and d_out, n_dicts <= 32.
 * Results are those of tsc_sparse_lens_pass dictionary by dictionary, bit for bit.
 */
int tsc_sparse_lens_finish_many(int n_dicts, const tsc_dict_t *const *dicts, const float *const *x, int64_t M,
                                int max_active, int32_t *feat, float *val, int32_t *count, int flags,
                                const float *const *dirs, int64_t dir_row_stride, float *recon, float *scores,
                                tsc_stream_t stream);

/*
 * The tail of the CircuitLens pass for a batch of rows in ONE launch: per row, assemble the component scores in the
 * ActiveFeatures row order, divide by the row's sum -- normalize_activations, circuit_lens.py:551-554 -- and select the
 * k largest -- get_top_k_features, circuit_lens.py:161-164 (`activations.topk(k)`).
 * A row has W = n_head + sum_g (group_aux[g] + max_active) columns: n_head leading scalars (embed, pos_embed), then for
 * every group g its group_aux[g] auxiliary scalars (b_O, z error, z bias | error, bias) followed by the max_active
 * score slots of dictionary g (tsc_sparse_lens_pass's `scores`), of which the first count[g][row] are live; dead
 * slots read 0 in raw / acts, 0 in mask, and never rank.  group_aux: HOST array [n_groups], n_groups <= 64.
 * head [R, n_head], aux [R, sum_g group_aux[g]] (group after group), scores [n_groups, R, max_active],
 * count [n_groups, R]: device.  Optional device inputs: column_tags [n_head + sum_g group_aux[g] + n_groups] int32
 * -- the caller's non-negative label of every column class: the head scalars, the auxiliary scalars group after
 * group, then ONE per group for its score slots (CircuitLens packs component << 16 | layer) -- and feat
 * [n_groups, R, max_active], the feature ids behind the score slots (tsc_sparse_lens_pass's `feat`).
 * Outputs: raw / acts [R, W] fp32 and mask [R, W] bytes (each optional, NULL = skip; rows wider than 15360 columns
 * need all three), top_values [R, k] and top_info [R, k, 4] int32 = (column in the row, tag >> 16, tag & 0xffff,
 * feature id), all -1 (and -inf) where a row has fewer than k live columns, tag parts -1 without column_tags,
 * feature -1 on non-slot columns or without feat.  Order: value descending, column ascending; NaN never ranks.
 * k <= 256, n_head <= 256, group_aux[g] <= 256.
 */
int tsc_lens_assemble(int64_t R, int n_head, int n_groups, const int32_t *group_aux, int max_active, const float *head,
                      const float *aux, const float *scores, const int32_t *count, const int32_t *column_tags,
                      const int32_t *feat, int k, float *raw, float *acts, uint8_t *mask, float *top_values,
                      int32_t *top_info, tsc_stream_t stream);

/*
 * Many row-wise dot products in one launch:
 *   out[j][r * out_stride[j]] = sum_k (a[j][r * a_stride[j] + k] - c[j][r * c_stride[j] + k]) * b[j][r * b_stride[j] + k],  k < d,
 * for n_terms terms and R rows; an operand stride of 0 means one vector shared by all rows; c (the whole table or single
 * entries) may be NULL = nothing subtracted.  a, c, b, out and the stride tables are HOST arrays of n_terms entries; the
 * operands they point at are device memory, 16-byte aligned, strides 0 or >= d and multiples of 4; d % 4 == 0.
 * Replaces the auxiliary rows of CircuitLens.get_active_features dotted with a lens direction -- embed, pos_embed
 * (circuit_lens.py:570-575), b_O, the z-SAE error and bias through W_O (:591-604), the transcoder error and bias
 * (:640-645), each against the row's direction as in :762-766 -- for all layers and all positions of a batch at once
 * (62 terms for GPT-2 small); as separate tensor operations these were 17 launches over [layers, rows, 768] temporaries.
 * Sums run in a fixed order (lane-strided partial sums, then a 64-lane butterfly): results are reproducible.
 */
int tsc_rowdot_many(int n_terms, const float *const *a, const int64_t *a_stride, const float *const *c,
                    const int64_t *c_stride, const float *const *b, const int64_t *b_stride, float *const *out,
                    const int64_t *out_stride, int64_t R, int d, tsc_stream_t stream);

/*
 * Scores of a few selected features for every token, without touching the other d_sae - F columns:
 *   out[m, j] = act( dot(x[m,:] - b_dec, Pt[j,:]) + bias[j] ),  act = ReLU if apply_relu else identity,
 * Pt [F, d_in] holds the selected encoder columns as rows (W_enc[:, feats] transposed), bias [F],
 * b_dec [d_in] or NULL, F <= 64, d_in % 4 == 0 and d_in <= 4096.  HBM-bound on x (one pass).
 * Replaces `encode(acts)[:, feature]` of MaxActAnalysis.active_examples (max_act_analysis.py:231-232)
 * and both modes of get_feature_scores (autointerpretability.py:226-241), which compute the full
 * [tokens, d_sae] product to keep F columns.
 */
int tsc_project_features(const float *x, int64_t M, int d_in, const float *b_dec, const float *Pt,
                         const float *bias, int F, int apply_relu, float *out, tsc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TSC_HIP_H */
