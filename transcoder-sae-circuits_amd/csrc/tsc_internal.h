// Internal declarations shared by the HIP translation units of libtsc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsc_hip.h"

struct tsc_dict {
    int d_in, d_sae, d_out, compute_dtype;
    const float *W_enc, *b_enc, *b_dec, *W_dec, *out_bias;  // borrowed, device
    float *W_enc_t;                                         // owned (d_in % 64 == 0): W_enc transposed, [d_sae][d_in]
    void *W_enc_bf16;                                       // owned (TSC_BF16 only): bf16(W_enc) transposed, [d_sae][d_in]
    void *x_bf16;                                           // owned (TSC_BF16 only): bf16(x - b_dec) of one chunk of rows
    float *bias_fold;                                       // owned: b_enc - b_dec @ W_enc (fp64 sums), [d_sae]
    float *wnorm;                                           // owned (TSC_F32_SCREEN only): [2][d_sae] column norms of the screen's margin (encode_operands.hip)
    float *screen_bound;                                    // owned (TSC_F32_SCREEN only): 2 + 2 * TSC_BF16_WS_ROWS floats, the row bounds of the current call:
                                                            // the two maxima, then the pair of every row (row_bound_kernel)
};

// rows of a handle's bf16 staging buffer for x = rows per chunk of every bf16 / screened call (include/tsc_hip.h:
// TSC_SCREEN_MAX_ROWS).  65536: one chunk per sweep batch of 512 sequences x 128 positions; the screened top-k's
// queue entries (row << 8 | column) leave 24 bits for the row.
constexpr long long TSC_BF16_WS_ROWS = TSC_SCREEN_MAX_ROWS;

// error plumbing (capi.hip)
int tsc_fail(int code, const char *fmt, ...);
#define TSC_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return tsc_fail(TSC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                             \
    } while (0)

// encode_f32.hip
enum { TSC_MODE_DENSE = 0, TSC_MODE_TOPK = 1, TSC_MODE_SPARSE = 2 };
struct EncodeArgs {
    const float *x, *W, *b_enc, *b_dec;
    const void *Wt;      // W_enc transposed, [N][K] (fp32, or bf16 when .bf16): the LDS-DMA kernels' W operand
    int bf16;            // x and Wt are the bf16 operands prepared by encode_operands.hip (see tsc_launch_encode_bf16)
    int x_staged;        // tsc_launch_encode_bf16: the staging buffer already holds bf16(x - b_dec) of these rows (row_bound_kernel wrote it)
    const float *bias;   // b_enc - b_dec @ W_enc, or null: selects the LDS-DMA kernel when K % 64 == 0
    long long M;
    int K, N;
    int n_tiles_n, m_split;
    int no_relu;         // DENSE: store the pre-activation (hook_hidden_pre) instead of relu of it
    float *out;          // DENSE
    float *tk_vals;      // TOPK: [m_split][N][topk]
    long long *tk_idx;   //
    const float *tk_floor;  // [N] or null: k-th best over all slots at launch time
    int topk;
    long long token_base;
    int *sp_feat;        // SPARSE: [M][max_active]
    float *sp_val;
    int *sp_count;       // [M], zeroed by the caller
    int max_active;
    // bf16 SCREEN of the sparse lens pass (capi.hip: tsc_sparse_lens_pass on a TSC_F32_SCREEN handle): the bias of column n
    // is raised by screen_bound[0] * screen_wnorm[n] + screen_bound[1] * screen_wnorm[N + n], a proven bound of
    // |exact fp32 - bf16| for that column (row_bound_kernel, encode_operands.hip), so every
    // feature whose exact activation is positive is emitted as a candidate
    const float *screen_wnorm, *screen_bound;
    // optional: the bound pair of every row of this launch [M][2]; a candidate that passes the test with the maxima is
    // re-tested with its own row's pair before it is emitted (rows shorter than the longest one get a tighter margin)
    const float *screen_rows;
    // bf16 SCREEN of the fused top-k (screen_topk.hip): SPARSE epilogue, but a pair is emitted when its screened value
    // reaches the feature's floor (k-th best over all slots), and it goes to the emitting workgroup's own queue
    // sq_entries[w * sq_cap ..] as (row in this launch << 8 | column in tile); a full queue raises *sq_overflow
    const float *sq_floor;
    unsigned *sq_entries;
    int *sq_counters, *sq_overflow;
    int sq_cap;
    const int *run_flag;  // when set: the whole launch returns at once unless *run_flag != 0 (the fp32 fallback of the screen)
};
int tsc_launch_encode_f32(int mode, const EncodeArgs &a, hipStream_t stream);  // a.m_split <= 0: balanced
struct EmitGroupProblem {  // one dictionary of a grouped bf16 screen emit (encode_f32.hip: tsc_launch_emit_group)
    const void *x_bf16, *W_bf16;
    const float *b_enc, *wnorm, *bound, *rows;  // rows: per-row bound pairs [M][2] or null
    int *feat, *count;
    float *val;
    int N;
};
int tsc_launch_emit_group(int n, const EmitGroupProblem *p, long long M, int K, int max_active, hipStream_t stream);
int tsc_launch_convert_x_bf16(const float *x, const float *b_dec, long long rows, int K, void *x_bf16, hipStream_t stream);
int tsc_launch_convert_x_bf16_group(int n, const float *const *x, const float *const *b_dec, void *const *x_bf16,
                                    long long rows, int K, hipStream_t stream);
int tsc_encode_recommended_slots(int d_sae, int bf16);
int tsc_encode_tile_bn(int mode, const EncodeArgs &a);  // column-tile width the launcher will pick for these arguments
int tsc_encode_set_tuning(const char *key, int value);
int tsc_launch_fold_bias(const float *W, const float *b_enc, const float *b_dec, int K, int N, float *out,
                         hipStream_t stream);
int tsc_launch_encode_bf16(int mode, const EncodeArgs &a, const void *W_bf16, void *x_bf16, hipStream_t stream);
int tsc_launch_col_norms(const float *W, int K, int N, float *wnorm, hipStream_t stream);
// bound: 2 + 2 * M floats (maxima, then per-row pairs)
// xt_out (may be null): also store bf16(x - b_dec) [M][K] there, exactly what convert_x_kernel would (one pass over x instead of two)
int tsc_launch_row_bound(const float *x, long long M, int K, const float *b_dec, float *bound, hipStream_t stream,
                         void *xt_out = nullptr);
size_t tsc_bf16_workspace_bytes(int K);
int tsc_pack_w_bf16(const float *W, int K, int N, void *dst, hipStream_t stream);
int tsc_transpose_w_f32(const float *W, int K, int N, float *dst, hipStream_t stream);

// screen_topk.hip
size_t tsc_screen_topk_workspace_bytes(int d_sae, int n_slots);
int tsc_launch_screen_topk(const tsc_dict *d, const float *x, long long M, long long token_base, float *tk_vals,
                           long long *tk_idx, int n_slots, int k, const float *floor, void *workspace,
                           size_t workspace_bytes, const float *row_bound, hipStream_t stream);

// topk_merge.hip
int tsc_launch_topk_fill(float *vals, long long *idx, long long n, hipStream_t stream);
int tsc_launch_topk_floor(const float *vals, int n_slots, int d_sae, int k, float *floor, hipStream_t stream);
int tsc_launch_topk_merge(const float *vals, const long long *idx, int n_lists, int d_sae, int k,
                          float *out_vals, long long *out_idx, hipStream_t stream);

// sparse_ops.hip
int tsc_launch_sparse_sort(int *feat, float *val, const int *count, long long M, int max_active,
                           hipStream_t stream);
int tsc_launch_dense_to_sparse(const float *acts, long long M, int N, int max_active, int *feat, float *val, int *count,
                               hipStream_t stream);
int tsc_launch_decode_rows(const float *W_dec, int d_sae, int d_out, const int *feat, const float *val,
                           long long n, float *rows, hipStream_t stream);
int tsc_launch_decode_sparse(const float *W_dec, const float *bias, int d_out, const int *feat,
                             const float *val, const int *count, long long M, int max_active,
                             float *recon, hipStream_t stream);
int tsc_launch_contrib(const float *W_dec, int d_out, const int *feat, const float *val,
                       const int *count, long long M, int max_active, const float *dirs, int n_dir,
                       float *scores, hipStream_t stream);
int tsc_sparse_finish_supported(int d_out, int max_active);
int tsc_launch_sparse_finish_many(int n, const tsc_dict *const *dicts, const float *const *x, long long M,
                                  int max_active, int *feat, float *val, int *count, const float *const *dirs,
                                  long long dir_stride, float *recon, float *scores, int screened, hipStream_t stream);
int tsc_launch_lens_assemble(long long R, int n_head, int n_groups, const int *group_aux, int cap, const float *head,
                             const float *aux, const float *scores, const int *count, const int *tags, const int *feat,
                             int k, float *raw, float *acts, unsigned char *mask, float *top_v, int *top_info,
                             hipStream_t stream);
int tsc_launch_sparse_finish(const float *W_dec, const float *bias, int d_out, int *feat, float *val, int *count,
                             long long M, int max_active, const float *dirs, long long dir_stride, float *recon,
                             float *scores, const float *x, const float *W_enc_t, const float *bias_fold, int d_in,
                             hipStream_t stream);
int tsc_launch_contrib_many(int n, const tsc_dict *const *dicts, long long M, long long pitch_rows, int max_active,
                            const int *feat, const float *val, const int *count, const float *const *dirs,
                            long long dir_stride, float *scores, int *feat_out, int *count_out, int out_cap,
                            hipStream_t stream);
int tsc_launch_rowdot_many(int n, const float *const *a, const long long *sa, const float *const *c, const long long *sc,
                           const float *const *b, const long long *sb, float *const *out, const long long *so,
                           long long R, int d, hipStream_t stream);
int tsc_launch_project(const float *x, long long M, int d_in, const float *b_dec, const float *Pt,
                       const float *bias, int F, int relu, float *out, hipStream_t stream);
