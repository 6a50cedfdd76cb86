// extern "C" surface of libtsc_hip.so (declared in include/tsc_hip.h): argument checking, the
// dictionary handle, and the choice of launch geometry.  No torch types, no exceptions.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "tsc_internal.h"

static thread_local char g_err[512] = "";

int tsc_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_dict_x(const tsc_dict_t *d, const float *x, int64_t M) {
    if (!d) return tsc_fail(TSC_ERR_ARG, "null dictionary handle");
    if (M < 0) return tsc_fail(TSC_ERR_ARG, "negative row count %lld", (long long)M);
    if (M > 0 && !x) return tsc_fail(TSC_ERR_ARG, "null activation pointer");
    if (!aligned16(x)) return tsc_fail(TSC_ERR_ARG, "activation pointer not 16-byte aligned");
    return TSC_OK;
}

EncodeArgs base_args(const tsc_dict_t *d, const float *x, int64_t M) {
    EncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.W = d->W_enc; a.b_enc = d->b_enc; a.b_dec = d->b_dec; a.bias = d->bias_fold; a.Wt = d->W_enc_t;
    a.M = M; a.K = d->d_in; a.N = d->d_sae;
    a.m_split = 0;  // the launcher picks the tile geometry and the balanced split
    return a;
}

int run_encode(const tsc_dict_t *d, int mode, const EncodeArgs &a, hipStream_t s) {
    if (d->compute_dtype == TSC_BF16) return tsc_launch_encode_bf16(mode, a, d->W_enc_bf16, d->x_bf16, s);
    return tsc_launch_encode_f32(mode, a, s);
}

}  // namespace

// ---- the allocation-free variant (include/tsc_hip.h: tsc_encode_topk_static) --------------------------------
// The cross-slot floor of tsc_encode_topk normally lives in caller memory (state->floor).  Here it lives in
// this file-scope device array.  `scratch` is a host-side stand-in for the symbol, NOT a device address: the
// address on each GPU comes from hipGetSymbolAddress, once per device, and only that pointer is ever handed to
// a kernel.  Every GPU the library runs on holds its own 512-KiB copy until the process ends.
static __device__ __align__(256) float scratch[TSC_STATIC_FLOOR_MAX_D_SAE];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
float *g_scratch_ptr[MAX_DEVICES];  // per device: that device's address of `scratch`, or null until resolved

int tsc_static_scratch_init(float **out) {
    int dev = -1;
    TSC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return tsc_fail(TSC_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float *p = nullptr;
        TSC_HIP_CHECK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));  // of the CURRENT device
        if (!p) return tsc_fail(TSC_ERR_HIP, "hipGetSymbolAddress returned a null address for the static scratch");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return TSC_OK;
}
}  // namespace

extern "C" {

int tsc_version(void) { return 100; }
const char *tsc_last_error(void) { return g_err; }

int tsc_dict_create(tsc_dict_t **out, int d_in, int d_sae, int d_out, int compute_dtype,
                    const float *W_enc, const float *b_enc, const float *b_dec, const float *W_dec,
                    const float *out_bias, tsc_stream_t stream) {
    if (!out) return tsc_fail(TSC_ERR_ARG, "null output handle");
    *out = nullptr;
    if (d_in <= 0 || d_sae <= 0 || d_out <= 0) return tsc_fail(TSC_ERR_ARG, "non-positive dimension");
    if ((d_in & 3) || (d_sae & 3) || (d_out & 3))
        return tsc_fail(TSC_ERR_UNSUPPORTED, "d_in, d_sae and d_out must be multiples of 4 (got %d, %d, %d)",
                        d_in, d_sae, d_out);
    if (!W_enc || !b_enc || !b_dec) return tsc_fail(TSC_ERR_ARG, "W_enc, b_enc and b_dec are required");
    if (!aligned16(W_enc) || !aligned16(b_enc) || !aligned16(b_dec) || !aligned16(W_dec) || !aligned16(out_bias))
        return tsc_fail(TSC_ERR_ARG, "weight pointers must be 16-byte aligned");
    if (compute_dtype != TSC_F32 && compute_dtype != TSC_BF16 && compute_dtype != TSC_F32_SCREEN)
        return tsc_fail(TSC_ERR_ARG, "unknown compute dtype %d", compute_dtype);
    tsc_dict *d = new (std::nothrow) tsc_dict;
    if (!d) return tsc_fail(TSC_ERR_HIP, "out of host memory");
    memset(d, 0, sizeof(*d));
    d->d_in = d_in; d->d_sae = d_sae; d->d_out = d_out; d->compute_dtype = compute_dtype;
    d->W_enc = W_enc; d->b_enc = b_enc; d->b_dec = b_dec; d->W_dec = W_dec; d->out_bias = out_bias;
    if (compute_dtype == TSC_BF16 || compute_dtype == TSC_F32_SCREEN) {
        if (d_in & 127) {
            delete d;
            return tsc_fail(TSC_ERR_UNSUPPORTED, "bf16 path needs d_in %% 128 == 0 (got %d)", d_in);
        }
        hipError_t e = hipMalloc(&d->W_enc_bf16, (size_t)d_in * d_sae * 2);
        if (e == hipSuccess) e = hipMalloc(&d->x_bf16, tsc_bf16_workspace_bytes(d_in));
        if (e != hipSuccess) {
            tsc_dict_destroy(d);
            return tsc_fail(TSC_ERR_HIP, "hipMalloc of the bf16 operand buffers failed: %s", hipGetErrorString(e));
        }
    }
    {
        hipError_t e = hipSuccess;
        if (compute_dtype != TSC_BF16 && d_in % 64 == 0)  // operand of the LDS-DMA kernel; other d_in use register staging
            e = hipMalloc((void **)&d->W_enc_t, (size_t)d_in * d_sae * sizeof(float));
        if (e == hipSuccess && compute_dtype == TSC_F32_SCREEN) e = hipMalloc((void **)&d->wnorm, (size_t)d_sae * 2 * sizeof(float));
        if (e == hipSuccess && compute_dtype == TSC_F32_SCREEN)
            e = hipMalloc((void **)&d->screen_bound, (size_t)(2 + 2 * TSC_BF16_WS_ROWS) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&d->bias_fold, (size_t)d_sae * sizeof(float));
        int rc = e == hipSuccess ? tsc_dict_refresh(d, stream)
                                 : tsc_fail(TSC_ERR_HIP, "hipMalloc of the handle's derived operands failed: %s", hipGetErrorString(e));
        if (rc != TSC_OK) {
            tsc_dict_destroy(d);
            return rc;
        }
    }
    {
        float *unused = nullptr;  // set-up time: resolve this device's address of the static floor scratch
        int rc = tsc_static_scratch_init(&unused);
        if (rc != TSC_OK) {
            tsc_dict_destroy(d);
            return rc;
        }
    }
    *out = d;
    return TSC_OK;
}

int tsc_dict_refresh(tsc_dict_t *d, tsc_stream_t stream) {
    if (!d) return tsc_fail(TSC_ERR_ARG, "null dictionary handle");
    hipStream_t s = (hipStream_t)stream;
    if (d->W_enc_bf16) {
        int rc = tsc_pack_w_bf16(d->W_enc, d->d_in, d->d_sae, d->W_enc_bf16, s);
        if (rc != TSC_OK) return rc;
    }
    if (d->W_enc_t) {
        int rc = tsc_transpose_w_f32(d->W_enc, d->d_in, d->d_sae, d->W_enc_t, s);
        if (rc != TSC_OK) return rc;
    }
    if (d->wnorm) {
        int rc = tsc_launch_col_norms(d->W_enc, d->d_in, d->d_sae, d->wnorm, s);
        if (rc != TSC_OK) return rc;
    }
    return tsc_launch_fold_bias(d->W_enc, d->b_enc, d->b_dec, d->d_in, d->d_sae, d->bias_fold, s);
}

int tsc_dict_destroy(tsc_dict_t *d) {
    if (!d) return TSC_OK;
    if (d->W_enc_bf16) (void)hipFree(d->W_enc_bf16);
    if (d->x_bf16) (void)hipFree(d->x_bf16);
    if (d->W_enc_t) (void)hipFree(d->W_enc_t);
    if (d->bias_fold) (void)hipFree(d->bias_fold);
    if (d->wnorm) (void)hipFree(d->wnorm);
    if (d->screen_bound) (void)hipFree(d->screen_bound);
    delete d;
    return TSC_OK;
}

int tsc_encode_dense(const tsc_dict_t *d, const float *x, int64_t M, float *out, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    if (M == 0) return TSC_OK;
    if (!out) return tsc_fail(TSC_ERR_ARG, "null output pointer");
    EncodeArgs a = base_args(d, x, M);
    a.out = out;
    return run_encode(d, TSC_MODE_DENSE, a, (hipStream_t)stream);
}

int tsc_encode_dense_pre(const tsc_dict_t *d, const float *x, int64_t M, float *out, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    if (M == 0) return TSC_OK;
    if (!out) return tsc_fail(TSC_ERR_ARG, "null output pointer");
    EncodeArgs a = base_args(d, x, M);
    a.out = out;
    a.no_relu = 1;
    return run_encode(d, TSC_MODE_DENSE, a, (hipStream_t)stream);
}

int tsc_topk_recommended_slots(int d_sae) {
    if (d_sae <= 0) return 1;
    return tsc_encode_recommended_slots(d_sae, 0);
}

int tsc_set_tuning(const char *key, int value) { return tsc_encode_set_tuning(key, value); }

static int check_state(const tsc_topk_state_t *s) {
    if (!s || !s->vals || !s->idx) return tsc_fail(TSC_ERR_ARG, "null top-k state");
    if (s->d_sae <= 0 || s->k < 1 || s->k > 64 || s->n_slots < 1 || s->n_slots > 64)
        return tsc_fail(TSC_ERR_ARG, "top-k state out of range (d_sae %d, k %d, n_slots %d)", s->d_sae, s->k,
                        s->n_slots);
    // the slots of a state are merged in one pass (tsc_topk_finalize): a state that could be filled but not
    // finalized is refused on its first use, not after a whole sweep
    if (s->n_slots * s->k > TSC_TOPK_MERGE_MAX)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "top-k state: n_slots * k = %d exceeds the %d entries per feature that "
                        "tsc_topk_finalize merges (use fewer slots)", s->n_slots * s->k, TSC_TOPK_MERGE_MAX);
    return TSC_OK;
}

int tsc_topk_state_reset(const tsc_topk_state_t *s, tsc_stream_t stream) {
    int rc = check_state(s);
    if (rc) return rc;
    return tsc_launch_topk_fill(s->vals, (long long *)s->idx, (long long)s->n_slots * s->d_sae * s->k,
                                (hipStream_t)stream);
}

int tsc_encode_topk(const tsc_dict_t *d, const float *x, int64_t M, int64_t token_id_base,
                    const tsc_topk_state_t *s, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    rc = check_state(s);
    if (rc) return rc;
    if (s->d_sae != d->d_sae)
        return tsc_fail(TSC_ERR_ARG, "state d_sae %d does not match dictionary d_sae %d", s->d_sae, d->d_sae);
    if (token_id_base < 0) return tsc_fail(TSC_ERR_ARG, "negative token id base");
    if (M == 0) return TSC_OK;
    EncodeArgs a = base_args(d, x, M);
    a.tk_vals = s->vals;
    a.tk_idx = (long long *)s->idx;
    a.topk = s->k;
    a.token_base = token_id_base;
    a.m_split = s->n_slots;
    if (s->floor && s->n_slots > 1) {
        if (!(s->flags & TSC_TOPK_FLOOR_KEEP)) {
            rc = tsc_launch_topk_floor(s->vals, s->n_slots, s->d_sae, s->k, s->floor, (hipStream_t)stream);
            if (rc) return rc;
        }
        a.tk_floor = s->floor;
    }
    return run_encode(d, TSC_MODE_TOPK, a, (hipStream_t)stream);
}

int tsc_encode_topk_static(const tsc_dict_t *d, const float *x, int64_t M, int64_t token_id_base,
                           const tsc_topk_state_t *s, tsc_stream_t stream) {
    int rc = check_state(s);
    if (rc) return rc;
    if (s->d_sae > TSC_STATIC_FLOOR_MAX_D_SAE)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "tsc_encode_topk_static holds the floor of at most %d features in its "
                        "static scratch; this state has d_sae %d (use tsc_encode_topk with state->floor)",
                        TSC_STATIC_FLOOR_MAX_D_SAE, s->d_sae);
    float *floor = nullptr;
    rc = tsc_static_scratch_init(&floor);
    if (rc) return rc;
    // the shared array still holds the last call's floor (possibly of another state): never keep it.  The floor
    // kernel rewrites entries 0 .. d_sae-1, which are all this call reads, so nothing needs clearing.
    tsc_topk_state_t t = *s;
    t.floor = floor;
    t.flags = s->flags & ~TSC_TOPK_FLOOR_KEEP;
    return tsc_encode_topk(d, x, M, token_id_base, &t, stream);
}

size_t tsc_topk_screen_workspace_bytes(int d_sae, int n_slots) {
    if (d_sae <= 0 || n_slots < 1) return 0;
    return tsc_screen_topk_workspace_bytes(d_sae, n_slots);
}

int tsc_encode_topk_screened(const tsc_dict_t *d, const float *x, int64_t M, int64_t token_id_base,
                             const tsc_topk_state_t *s, void *workspace, size_t workspace_bytes,
                             const float *screen_row_bound, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    rc = check_state(s);
    if (rc) return rc;
    if (s->d_sae != d->d_sae)
        return tsc_fail(TSC_ERR_ARG, "state d_sae %d does not match dictionary d_sae %d", s->d_sae, d->d_sae);
    if (token_id_base < 0) return tsc_fail(TSC_ERR_ARG, "negative token id base");
    if (d->compute_dtype != TSC_F32_SCREEN) return tsc_fail(TSC_ERR_ARG, "tsc_encode_topk_screened needs a TSC_F32_SCREEN handle");
    if (!s->floor) return tsc_fail(TSC_ERR_ARG, "tsc_encode_topk_screened needs state->floor");
    if (M == 0) return TSC_OK;
    // the floor is what the screen compares against: the k-th best over ALL slots (also for a single slot)
    if (!(s->flags & TSC_TOPK_FLOOR_KEEP)) {
        rc = tsc_launch_topk_floor(s->vals, s->n_slots, s->d_sae, s->k, s->floor, (hipStream_t)stream);
        if (rc) return rc;
    }
    return tsc_launch_screen_topk(d, x, M, token_id_base, s->vals, (long long *)s->idx, s->n_slots, s->k, s->floor,
                                  workspace, workspace_bytes, screen_row_bound, (hipStream_t)stream);
}

int tsc_topk_merge(const float *vals, const int64_t *idx, int n_lists, int d_sae, int k, float *out_vals,
                   int64_t *out_idx, tsc_stream_t stream) {
    if (!vals || !idx || !out_vals || !out_idx) return tsc_fail(TSC_ERR_ARG, "null list pointer");
    if (n_lists < 1 || d_sae < 1 || k < 1) return tsc_fail(TSC_ERR_ARG, "non-positive merge dimension");
    if (out_vals == vals || (const int64_t *)out_idx == idx)
        return tsc_fail(TSC_ERR_ARG, "merge output may not alias its input");
    return tsc_launch_topk_merge(vals, (const long long *)idx, n_lists, d_sae, k, out_vals,
                                 (long long *)out_idx, (hipStream_t)stream);
}

int tsc_topk_finalize(const tsc_topk_state_t *s, float *out_vals, int64_t *out_idx, tsc_stream_t stream) {
    int rc = check_state(s);
    if (rc) return rc;
    return tsc_topk_merge(s->vals, s->idx, s->n_slots, s->d_sae, s->k, out_vals, out_idx, stream);
}

int tsc_encode_sparse(const tsc_dict_t *d, const float *x, int64_t M, int max_active, int32_t *feat,
                      float *val, int32_t *count, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    if (M == 0) return TSC_OK;
    if (!feat || !val || !count) return tsc_fail(TSC_ERR_ARG, "null sparse output pointer");
    if (max_active < 1 || max_active > 8192) return tsc_fail(TSC_ERR_ARG, "max_active %d outside 1..8192", max_active);
    hipStream_t s = (hipStream_t)stream;
    TSC_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)M * sizeof(int32_t), s));
    EncodeArgs a = base_args(d, x, M);
    a.sp_feat = feat; a.sp_val = val; a.sp_count = count; a.max_active = max_active;
    rc = run_encode(d, TSC_MODE_SPARSE, a, s);
    if (rc) return rc;
    return tsc_launch_sparse_sort(feat, val, count, M, max_active, s);
}

static int check_decoder(const tsc_dict_t *d) {
    if (!d) return tsc_fail(TSC_ERR_ARG, "null dictionary handle");
    if (!d->W_dec) return tsc_fail(TSC_ERR_ARG, "dictionary was created without W_dec");
    return TSC_OK;
}

int tsc_sparse_lens_pass(const tsc_dict_t *d, const float *x, int64_t M, int max_active, int32_t *feat, float *val,
                         int32_t *count, int flags, const float *dirs, int64_t dir_row_stride, float *recon,
                         float *scores, const float *screen_row_bound, tsc_stream_t stream) {
    int rc = check_dict_x(d, x, M);
    if (rc) return rc;
    rc = check_decoder(d);
    if (rc) return rc;
    if (M == 0) return TSC_OK;
    if (!feat || !val || !count) return tsc_fail(TSC_ERR_ARG, "null sparse output pointer");
    if (scores && (!dirs || !aligned16(dirs) || (dir_row_stride != 0 && (dir_row_stride < d->d_out || (dir_row_stride & 3)))))
        return tsc_fail(TSC_ERR_ARG, "scores need 16-byte aligned directions with row stride 0 or >= d_out (multiple of 4)");
    if (recon && !aligned16(recon)) return tsc_fail(TSC_ERR_ARG, "recon must be 16-byte aligned");
    if (max_active < 1 || !tsc_sparse_finish_supported(d->d_out, max_active))
        return tsc_fail(TSC_ERR_UNSUPPORTED, "tsc_sparse_lens_pass needs d_out <= 768 and 1 <= max_active <= 2048 (got %d, %d); "
                        "use tsc_encode_sparse + tsc_decode_sparse + tsc_contrib_scores", d->d_out, max_active);
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & TSC_LENS_COUNT_IS_ZERO)) TSC_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)M * sizeof(int32_t), s));
    EncodeArgs a = base_args(d, x, M);
    a.sp_feat = feat; a.sp_val = val; a.sp_count = count; a.max_active = max_active;
    if (d->compute_dtype == TSC_F32_SCREEN && d->W_enc_t && !(flags & TSC_LENS_NO_SCREEN)) {
        // bf16 MFMA screen (16x the fp32 matrix rate) -> candidates; the finish kernel recomputes them exactly in fp32
        if (M > TSC_SCREEN_MAX_ROWS)
            return tsc_fail(TSC_ERR_UNSUPPORTED, "screened lens pass takes at most %d rows per call (got %lld)", TSC_SCREEN_MAX_ROWS, (long long)M);
        bool per_row = (flags & TSC_LENS_ROW_BOUNDS) != 0;
        if (!screen_row_bound) {
            rc = tsc_launch_row_bound(x, M, d->d_in, d->b_dec, d->screen_bound, s);
            if (rc) return rc;
            screen_row_bound = d->screen_bound;
            per_row = true;
        }
        a.screen_wnorm = d->wnorm;
        a.screen_bound = screen_row_bound;
        a.screen_rows = per_row ? screen_row_bound + 2 : nullptr;
        rc = tsc_launch_encode_bf16(TSC_MODE_SPARSE, a, d->W_enc_bf16, d->x_bf16, s);
        if (rc || (flags & TSC_LENS_EMIT_ONLY)) return rc;
        return tsc_launch_sparse_finish(d->W_dec, d->out_bias, d->d_out, feat, val, count, M, max_active, dirs,
                                        dir_row_stride, recon, scores, x, d->W_enc_t, d->bias_fold, d->d_in, s);
    }
    rc = run_encode(d, TSC_MODE_SPARSE, a, s);
    if (rc || (flags & TSC_LENS_EMIT_ONLY)) return rc;
    return tsc_launch_sparse_finish(d->W_dec, d->out_bias, d->d_out, feat, val, count, M, max_active, dirs,
                                    dir_row_stride, recon, scores, nullptr, nullptr, nullptr, 0, s);
}

int tsc_sparse_lens_emit_many(int n_dicts, const tsc_dict_t *const *dicts, const float *const *x, int64_t M,
                              int max_active, int32_t *feat, float *val, int32_t *count, int flags,
                              const float *const *screen_row_bounds, const tsc_stream_t *streams, int n_streams) {
    if (n_dicts < 0 || M < 0) return tsc_fail(TSC_ERR_ARG, "negative count");
    if (n_dicts == 0 || M == 0) return TSC_OK;
    if (!dicts || !x || !feat || !val || !count) return tsc_fail(TSC_ERR_ARG, "null lens_emit_many pointer");
    if (n_streams < 1 || !streams) return tsc_fail(TSC_ERR_ARG, "lens_emit_many needs at least one stream");
    // a handle listed twice would share its bf16 staging buffer and bound scalar between two problems of one launch
    // (or race across streams on the fallback below): refused
    for (int i = 0; i < n_dicts && i < 64; ++i)
        for (int j = 0; j < i; ++j)
            if (dicts[i] && dicts[i] == dicts[j])
                return tsc_fail(TSC_ERR_ARG, "lens_emit_many: dictionary handle %d is listed twice (entries %d and %d)", i, j, i);
    // Screening handles of one input width: ONE grouped launch on streams[0] (a 512-row emit is 1.5 rounds of
    // single-tile workgroups; emits on different streams do not pack into each other's idle half round, one grid does)
    bool group = n_dicts > 1 && n_dicts <= 32 && M <= TSC_SCREEN_MAX_ROWS && !(flags & TSC_LENS_NO_SCREEN);
    for (int i = 0; i < n_dicts && group; ++i)
        group = dicts[i] && dicts[i]->compute_dtype == TSC_F32_SCREEN && dicts[i]->W_enc_t && dicts[i]->W_enc_bf16 &&
                dicts[i]->d_in == dicts[0]->d_in && x[i] && aligned16(x[i]);
    if (group) {
        if (max_active < 1 || !tsc_sparse_finish_supported(dicts[0]->d_out, max_active))
            return tsc_fail(TSC_ERR_UNSUPPORTED, "tsc_sparse_lens_emit_many needs d_out <= 768 and 1 <= max_active <= 2048");
        hipStream_t s = (hipStream_t)streams[0];
        if (!(flags & TSC_LENS_COUNT_IS_ZERO)) TSC_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)n_dicts * M * sizeof(int32_t), s));
        EmitGroupProblem p[32];
        const float *b_decs[32];
        void *stagings[32];
        for (int i = 0; i < n_dicts; ++i) { b_decs[i] = dicts[i]->b_dec; stagings[i] = dicts[i]->x_bf16; }
        int rc = tsc_launch_convert_x_bf16_group(n_dicts, x, b_decs, stagings, M, dicts[0]->d_in, s);
        if (rc) return rc;
        for (int i = 0; i < n_dicts; ++i) {
            const tsc_dict_t *d = dicts[i];
            const float *bound = screen_row_bounds ? screen_row_bounds[i] : nullptr;
            bool per_row = (flags & TSC_LENS_ROW_BOUNDS) != 0;
            if (!bound) {
                rc = tsc_launch_row_bound(x[i], M, d->d_in, d->b_dec, d->screen_bound, s);
                if (rc) return rc;
                bound = d->screen_bound;
                per_row = true;
            }
            p[i].rows = per_row ? bound + 2 : nullptr;
            p[i].x_bf16 = d->x_bf16; p[i].W_bf16 = d->W_enc_bf16; p[i].b_enc = d->b_enc; p[i].wnorm = d->wnorm;
            p[i].bound = bound; p[i].feat = feat + (size_t)i * M * max_active; p[i].val = val + (size_t)i * M * max_active;
            p[i].count = count + (size_t)i * M; p[i].N = d->d_sae;
        }
        return tsc_launch_emit_group(n_dicts, p, M, dicts[0]->d_in, max_active, s);
    }
    for (int i = 0; i < n_dicts; ++i) {
        const int rc = tsc_sparse_lens_pass(dicts[i], x[i], M, max_active, feat + (size_t)i * M * max_active,
                                            val + (size_t)i * M * max_active, count + (size_t)i * M,
                                            flags | TSC_LENS_EMIT_ONLY, nullptr, 0, nullptr, nullptr,
                                            screen_row_bounds ? screen_row_bounds[i] : nullptr, streams[i % n_streams]);
        if (rc) return rc;
    }
    return TSC_OK;
}

int tsc_sparse_lens_finish_many(int n_dicts, const tsc_dict_t *const *dicts, const float *const *x, int64_t M,
                                int max_active, int32_t *feat, float *val, int32_t *count, int flags,
                                const float *const *dirs, int64_t dir_row_stride, float *recon, float *scores,
                                tsc_stream_t stream) {
    if (n_dicts < 0 || M < 0) return tsc_fail(TSC_ERR_ARG, "negative count");
    if (n_dicts == 0 || M == 0) return TSC_OK;
    if (!dicts || !feat || !val || !count) return tsc_fail(TSC_ERR_ARG, "null lens_finish_many pointer");
    if (max_active < 1) return tsc_fail(TSC_ERR_ARG, "max_active %d < 1", max_active);
    if (recon && !aligned16(recon)) return tsc_fail(TSC_ERR_ARG, "recon must be 16-byte aligned");
    int screened = -1;
    for (int i = 0; i < n_dicts; ++i) {
        if (!dicts[i]) return tsc_fail(TSC_ERR_ARG, "null dictionary handle %d", i);
        const int sc = dicts[i]->compute_dtype == TSC_F32_SCREEN && dicts[i]->W_enc_t && !(flags & TSC_LENS_NO_SCREEN);
        if (screened >= 0 && sc != screened)
            return tsc_fail(TSC_ERR_ARG, "grouped finish: the dictionaries must all be screened or all plain");
        screened = sc;
        if (sc && (!x || !x[i] || !aligned16(x[i]))) return tsc_fail(TSC_ERR_ARG, "screened finish needs the 16-byte aligned inputs x[%d]", i);
        if (scores && (!dirs || !dirs[i] || !aligned16(dirs[i])))
            return tsc_fail(TSC_ERR_ARG, "scores need 16-byte aligned directions (dictionary %d)", i);
    }
    if (scores && dir_row_stride != 0 && (dir_row_stride < dicts[0]->d_out || (dir_row_stride & 3)))
        return tsc_fail(TSC_ERR_ARG, "direction row stride must be 0 or >= d_out and a multiple of 4");
    return tsc_launch_sparse_finish_many(n_dicts, dicts, x, M, max_active, feat, val, count, dirs, dir_row_stride,
                                         recon, scores, screened, (hipStream_t)stream);
}

int tsc_lens_assemble(int64_t R, int n_head, int n_groups, const int32_t *group_aux, int max_active, const float *head,
                      const float *aux, const float *scores, const int32_t *count, const int32_t *column_tags,
                      const int32_t *feat, int k, float *raw, float *acts, uint8_t *mask, float *top_values,
                      int32_t *top_info, tsc_stream_t stream) {
    if (R < 0) return tsc_fail(TSC_ERR_ARG, "negative row count");
    if (R == 0) return TSC_OK;
    if (n_head < 0 || n_groups < 1 || !group_aux || max_active < 1 || k < 1 || k > 256)
        return tsc_fail(TSC_ERR_ARG, "bad lens_assemble shape (n_head %d, n_groups %d, max_active %d, k %d)", n_head,
                        n_groups, max_active, k);
    if (!scores || !count || !top_values || !top_info || (n_head > 0 && !head))
        return tsc_fail(TSC_ERR_ARG, "null lens_assemble pointer");
    int n_aux = 0;
    for (int g = 0; g < n_groups && g < 64; ++g) n_aux += group_aux[g] > 0 ? group_aux[g] : 0;
    if (n_aux > 0 && !aux) return tsc_fail(TSC_ERR_ARG, "auxiliary scalars declared but aux is null");
    return tsc_launch_lens_assemble(R, n_head, n_groups, group_aux, max_active, head, aux, scores, count, column_tags,
                                    feat, k, raw, acts, mask, top_values, top_info, (hipStream_t)stream);
}

int tsc_contrib_scores_many(int n_dicts, const tsc_dict_t *const *dicts, int64_t M, int64_t dict_pitch_rows, int max_active,
                            const int32_t *feat, const float *val, const int32_t *count, const float *const *dirs,
                            int64_t dir_row_stride, float *scores, int32_t *feat_out, int32_t *count_out,
                            int out_max_active, tsc_stream_t stream) {
    if (n_dicts < 0 || M < 0) return tsc_fail(TSC_ERR_ARG, "negative count");
    if (n_dicts == 0 || M == 0) return TSC_OK;
    if (!dicts || !feat || !val || !count || !dirs || !scores) return tsc_fail(TSC_ERR_ARG, "null contrib_scores_many pointer");
    if (max_active < 1 || dict_pitch_rows < M || out_max_active < 0 || out_max_active > max_active)
        return tsc_fail(TSC_ERR_ARG, "bad contrib_scores_many shape");
    for (int i = 0; i < n_dicts; ++i) {
        if (!dicts[i]) return tsc_fail(TSC_ERR_ARG, "null dictionary handle %d", i);
        if (!dirs[i] || !aligned16(dirs[i])) return tsc_fail(TSC_ERR_ARG, "direction %d must be a 16-byte aligned device pointer", i);
    }
    if (dir_row_stride != 0 && (dir_row_stride < dicts[0]->d_out || (dir_row_stride & 3)))
        return tsc_fail(TSC_ERR_ARG, "direction row stride must be 0 or >= d_out and a multiple of 4");
    return tsc_launch_contrib_many(n_dicts, dicts, M, dict_pitch_rows, max_active, feat, val, count, dirs, dir_row_stride,
                                   scores, feat_out, count_out, out_max_active, (hipStream_t)stream);
}

int tsc_rowdot_many(int n_terms, const float *const *a, const int64_t *a_stride, const float *const *c,
                    const int64_t *c_stride, const float *const *b, const int64_t *b_stride, float *const *out,
                    const int64_t *out_stride, int64_t R, int d, tsc_stream_t stream) {
    if (n_terms < 0 || R < 0) return tsc_fail(TSC_ERR_ARG, "negative count");
    if (n_terms == 0 || R == 0) return TSC_OK;
    if (d <= 0 || (d & 3)) return tsc_fail(TSC_ERR_UNSUPPORTED, "rowdot: d %d must be a positive multiple of 4", d);
    if (!a || !a_stride || !b || !b_stride || !out || !out_stride || (c && !c_stride))
        return tsc_fail(TSC_ERR_ARG, "null rowdot table");
    for (int j = 0; j < n_terms; ++j) {
        if (!a[j] || !b[j] || !out[j] || !aligned16(a[j]) || !aligned16(b[j]) || (c && !aligned16(c[j])))
            return tsc_fail(TSC_ERR_ARG, "rowdot term %d: null or misaligned operand", j);
        const int64_t s[4] = {a_stride[j], b_stride[j], c && c[j] ? c_stride[j] : 0, out_stride[j]};
        for (int q = 0; q < 3; ++q)
            if (s[q] != 0 && (s[q] < d || (s[q] & 3) || s[q] > 0x7fffffff))
                return tsc_fail(TSC_ERR_ARG, "rowdot term %d: operand row stride must be 0 or >= d, a multiple of 4", j);
        if (s[3] < 1 || s[3] > 0x7fffffff) return tsc_fail(TSC_ERR_ARG, "rowdot term %d: output stride must be >= 1", j);
    }
    return tsc_launch_rowdot_many(n_terms, a, (const long long *)a_stride, c, (const long long *)c_stride, b,
                                  (const long long *)b_stride, out, (const long long *)out_stride, R, d, (hipStream_t)stream);
}

int tsc_dense_to_sparse(const float *acts, int64_t M, int d_sae, int max_active, int32_t *feat, float *val, int32_t *count,
                        tsc_stream_t stream) {
    if (M < 0) return tsc_fail(TSC_ERR_ARG, "negative row count");
    if (M == 0) return TSC_OK;
    if (d_sae <= 0 || (d_sae & 3)) return tsc_fail(TSC_ERR_UNSUPPORTED, "d_sae %d must be a positive multiple of 4", d_sae);
    if (max_active < 1) return tsc_fail(TSC_ERR_ARG, "max_active %d < 1", max_active);
    if (!acts || !feat || !val || !count || !aligned16(acts)) return tsc_fail(TSC_ERR_ARG, "bad dense_to_sparse pointer");
    return tsc_launch_dense_to_sparse(acts, M, d_sae, max_active, feat, val, count, (hipStream_t)stream);
}

int tsc_decode_rows(const tsc_dict_t *d, const int32_t *feat, const float *val, int64_t n, float *rows,
                    tsc_stream_t stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    if (n < 0) return tsc_fail(TSC_ERR_ARG, "negative row count");
    if (n == 0) return TSC_OK;
    if (!feat || !val || !rows || !aligned16(rows)) return tsc_fail(TSC_ERR_ARG, "bad decode_rows pointer");
    return tsc_launch_decode_rows(d->W_dec, d->d_sae, d->d_out, feat, val, n, rows, (hipStream_t)stream);
}

int tsc_decode_sparse(const tsc_dict_t *d, const int32_t *feat, const float *val, const int32_t *count,
                      int64_t M, int max_active, float *recon, tsc_stream_t stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    if (M < 0 || max_active < 1) return tsc_fail(TSC_ERR_ARG, "bad decode_sparse shape");
    if (M == 0) return TSC_OK;
    if (!feat || !val || !count || !recon || !aligned16(recon))
        return tsc_fail(TSC_ERR_ARG, "bad decode_sparse pointer");
    return tsc_launch_decode_sparse(d->W_dec, d->out_bias, d->d_out, feat, val, count, M, max_active, recon,
                                    (hipStream_t)stream);
}

int tsc_contrib_scores(const tsc_dict_t *d, const int32_t *feat, const float *val, const int32_t *count,
                       int64_t M, int max_active, const float *dirs, int n_dir, float *scores,
                       tsc_stream_t stream) {
    int rc = check_decoder(d);
    if (rc) return rc;
    if (M < 0 || max_active < 1) return tsc_fail(TSC_ERR_ARG, "bad contrib_scores shape");
    if (M == 0) return TSC_OK;
    if (!feat || !val || !count || !dirs || !scores || !aligned16(dirs))
        return tsc_fail(TSC_ERR_ARG, "bad contrib_scores pointer");
    return tsc_launch_contrib(d->W_dec, d->d_out, feat, val, count, M, max_active, dirs, n_dir, scores,
                              (hipStream_t)stream);
}

int tsc_project_features(const float *x, int64_t M, int d_in, const float *b_dec, const float *Pt,
                         const float *bias, int F, int apply_relu, float *out, tsc_stream_t stream) {
    if (M < 0 || F < 0) return tsc_fail(TSC_ERR_ARG, "negative shape");
    if (M == 0 || F == 0) return TSC_OK;
    if (d_in <= 0 || (d_in & 3) || d_in > 4096)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "d_in %d must be a multiple of 4 and at most 4096", d_in);
    if (F > 64) return tsc_fail(TSC_ERR_UNSUPPORTED, "at most 64 features per call (got %d)", F);
    if (!x || !Pt || !bias || !out) return tsc_fail(TSC_ERR_ARG, "null pointer");
    if (!aligned16(x) || !aligned16(Pt) || !aligned16(b_dec))
        return tsc_fail(TSC_ERR_ARG, "x, Pt and b_dec must be 16-byte aligned");
    return tsc_launch_project(x, M, d_in, b_dec, Pt, bias, F, apply_relu, out, (hipStream_t)stream);
}

}  // extern "C"
