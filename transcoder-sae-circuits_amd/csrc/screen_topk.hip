// The fused encode + per-feature top-k at the bf16 matrix rate with fp32 results (tsc_encode_topk_screened).
//
// No reference counterpart beyond what tsc_encode_topk replaces (reference src/max_act_analysis.py:215-249 for all
// features at once); the point is the MI355X: its bf16 MFMA is 16x its fp32 MFMA, and the top-k only needs exact values
// for the few (token, feature) pairs that can enter a list.  Per chunk of rows, three launches on one stream:
//
//   K1  encode_f32_kernel<SPARSE, BF16> (encode_f32.hip) computes bf16(x - b_dec) @ bf16(W_enc) + b_enc + margin[n] and
//       queues every pair whose value reaches the feature's floor (k-th best over all slots before the launch).
//       margin[n] bounds |exact fp32 pre-activation - bf16 one| (encode_operands.hip: row_bound_kernel), so every pair
//       whose EXACT activation could enter the merged top-k is queued.  One queue per workgroup, no global queue.
//   K2  screen_insert_kernel (here): the same workgroup grid; each workgroup recomputes its queued pairs exactly -- a
//       scalar fmaf chain over k in the order the fp32 MFMA kernel accumulates (k = 8g+s then 8g+4+s, s = 0..3,
//       starting from bias'), which reproduces v_mfma_f32_32x32x2_f32 bit for bit -- and the thread that owns a feature
//       folds its pairs into that feature's list of the workgroup's slot, exactly as the fp32 kernel's owner does.
//   K3  the fp32 fused kernel on the same chunk, which returns at once unless a queue overflowed in K1 (then K2 has
//       skipped the chunk and K3 does all of it exactly).  Overflow needs no host round trip and loses nothing.
//
// The finished lists are therefore the lists of tsc_encode_topk, bit for bit (tests/test_screened_topk_gpu.py).
#include <cstring>

#include "tsc_internal.h"

namespace {

constexpr int SQ_CAP_256 = 8192;   // queue entries per workgroup with 256-wide column tiles (half with 128-wide ones)
constexpr int SQ_BATCH = 1024;
constexpr int SQ_KL = 20;  // list length up to which an owner mirrors its list's values in LDS
constexpr long long SQ_CHUNK_ROWS = TSC_BF16_WS_ROWS;  // = the bf16 staging buffer of the handle; a queue entry keeps 24 bits of row

__device__ __forceinline__ bool below(float v, long long id, float w, long long wid) {
    return v < w || (v == w && id > wid);  // (v, id) ranks below (w, wid): value descending, token id ascending
}

struct InsertArgs {
    const float *x, *Wt, *bias;
    int K, N, k, n_tiles_n, m_split, cap;
    long long token_base;
    float *tk_vals;
    long long *tk_idx;
    const unsigned *entries;
    const int *counters, *overflow;
};

template <int BN>
__global__ __launch_bounds__(256) void screen_insert_kernel(const InsertArgs p) {
    __shared__ unsigned s_ent[SQ_BATCH];
    __shared__ float s_val[SQ_BATCH];
    __shared__ float s_lv[BN * SQ_KL];  // k <= SQ_KL: the values of the owner's list (HBM is written through)
    if (*p.overflow) return;  // K3 redoes the whole chunk exactly
    const int w = blockIdx.x, tid = threadIdx.x;
    int n_cand = p.counters[w];
    if (n_cand <= 0) return;
    if (n_cand > p.cap) n_cand = p.cap;
    int nt, slot;  // the blockIdx -> (column tile, slot) map of encode_f32_kernel
    if ((p.n_tiles_n & 7) == 0) {
        const int q = w >> 3;
        nt = (w & 7) + 8 * (q / p.m_split);
        slot = q % p.m_split;
    } else {
        nt = w / p.m_split;
        slot = w % p.m_split;
    }
    const int n0 = nt * BN;
    // owner state of feature n0 + tid, loaded from the list on the first pair that concerns it
    bool loaded = false;
    int cnt = 0, wpos = 0;
    float wv = 0.f;
    long long wid = 0;
    const int k = p.k;
    const long long off = ((long long)slot * p.N + n0 + (tid < BN ? tid : 0)) * k;
    float *lv = p.tk_vals + off;
    long long *lid = p.tk_idx + off;
    for (int b0 = 0; b0 < n_cand; b0 += SQ_BATCH) {
        const int nb = n_cand - b0 < SQ_BATCH ? n_cand - b0 : SQ_BATCH;
        for (int e = tid; e < nb; e += 256) {
            const unsigned ent = p.entries[(size_t)w * p.cap + b0 + e];
            const int n = n0 + (int)(ent & 255u);
            const float4 *xp = (const float4 *)(p.x + (size_t)(ent >> 8) * p.K);
            const float4 *wp = (const float4 *)(p.Wt + (size_t)n * p.K);
            float acc = p.bias[n];
            // the accumulation order of the fp32 MFMA kernel, one rounding per product.  A lane walks its own two rows:
            // the 8 loads of one 128-byte line are issued back to back (first x's line, then W's), so that the line is
            // still in the 32-KB L1 when its other pieces are asked for -- interleaving x and W loads and fetching two
            // lines ahead was measured 8x amplified in L2 traffic.  (Staging the rows through LDS in 64-byte segments
            // was tried: coalesced, but two barriers per 16 k make it latency-bound, 540 us against 230.)
            for (int g0 = 0; g0 < (p.K >> 3); g0 += 4) {
                float4 xv[8], wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = xp[2 * g0 + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wp[2 * g0 + j];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    acc = fmaf(xv[j].x, wv[j].x, acc); acc = fmaf(xv[j + 1].x, wv[j + 1].x, acc);
                    acc = fmaf(xv[j].y, wv[j].y, acc); acc = fmaf(xv[j + 1].y, wv[j + 1].y, acc);
                    acc = fmaf(xv[j].z, wv[j].z, acc); acc = fmaf(xv[j + 1].z, wv[j + 1].z, acc);
                    acc = fmaf(xv[j].w, wv[j].w, acc); acc = fmaf(xv[j + 1].w, wv[j + 1].w, acc);
                }
            }
            s_ent[e] = ent;
            s_val[e] = acc;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            for (int e = 0; e < nb; ++e) {
                const unsigned ent = s_ent[e];
                const float v = s_val[e];
                if ((int)(ent & 255u) != tid || !(v > 0.f)) continue;
                const long long id = p.token_base + (long long)(ent >> 8);
                if (k <= SQ_KL) {
                    // values mirrored in LDS: finding the new lowest entry after a replacement reads no HBM; ids are
                    // fetched only to break exact ties (the same scheme as the fp32 kernel's owner, encode_f32.hip)
                    float *mv = &s_lv[tid * SQ_KL];
                    if (!loaded) {
                        loaded = true;
                        float ev[SQ_KL];
#pragma unroll
                        for (int i = 0; i < SQ_KL; ++i) ev[i] = lv[i < k ? i : k - 1];  // independent loads
                        wid = -1;
#pragma unroll
                        for (int i = 0; i < SQ_KL; ++i)
                            if (i < k) {
                                mv[i] = ev[i];
                                if (ev[i] > 0.f) {  // lists are compact and live values are positive
                                    if (cnt == 0 || ev[i] <= wv) { wv = ev[i]; wpos = i; }  // among equal values: resolved below
                                    cnt = i + 1;
                                }
                            }
                        if (cnt > 0) {  // exact lowest-ranked entry: among equal lowest values the largest id
                            int ties = 0;
                            for (int i = 0; i < cnt; ++i) ties += mv[i] == wv;
                            if (ties > 1) {
                                long long best = -1;
                                for (int i = 0; i < cnt; ++i)
                                    if (mv[i] == wv) {
                                        const long long t = lid[i];
                                        if (t > best) { best = t; wpos = i; }
                                    }
                                wid = best;
                            }
                        }
                    }
                    if (cnt < k) {
                        lv[cnt] = v;
                        lid[cnt] = id;
                        mv[cnt] = v;
                        if (cnt == 0 || v < wv) { wv = v; wid = id; wpos = cnt; }
                        else if (v == wv) {
                            if (wid < 0) wid = lid[wpos];
                            if (id > wid) { wid = id; wpos = cnt; }
                        }
                        ++cnt;
                        continue;
                    }
                    if (v < wv) continue;
                    if (v == wv) {
                        if (wid < 0) wid = lid[wpos];
                        if (id > wid) continue;
                    }
                    const int pos = wpos;
                    lv[pos] = v;
                    lid[pos] = id;
                    mv[pos] = v;
                    float lowest = v;
                    int mp = pos, ties = 1;
                    for (int i = 0; i < k; ++i) {
                        if (i == pos) continue;
                        const float e2 = mv[i];
                        if (e2 < lowest) { lowest = e2; mp = i; ties = 1; }
                        else if (e2 == lowest) ++ties;
                    }
                    wv = lowest; wpos = mp;
                    wid = mp == pos ? id : -1;
                    if (ties > 1) {
                        long long best = -1;
                        for (int i = 0; i < k; ++i)
                            if (mv[i] == lowest) {
                                const long long t = i == pos ? id : lid[i];
                                if (t > best) { best = t; wpos = i; }
                            }
                        wid = best;
                    }
                    continue;
                }
                if (!loaded) {  // lists are compact: live entries first
                    loaded = true;
                    for (int i = 0; i < k; ++i) {
                        const long long eid = lid[i];
                        if (eid < 0) break;
                        const float ev = lv[i];
                        if (cnt == 0 || below(ev, eid, wv, wid)) { wv = ev; wid = eid; wpos = i; }
                        cnt = i + 1;
                    }
                }
                if (cnt < k) {
                    lv[cnt] = v;
                    lid[cnt] = id;
                    if (cnt == 0 || below(v, id, wv, wid)) { wv = v; wid = id; wpos = cnt; }
                    ++cnt;
                } else if (below(wv, wid, v, id)) {
                    lv[wpos] = v;
                    lid[wpos] = id;
                    wv = v; wid = id;
                    const int self = wpos;
                    for (int i = 0; i < k; ++i) {
                        if (i == self) continue;
                        const float ev = lv[i];
                        const long long eid = lid[i];
                        if (below(ev, eid, wv, wid)) { wv = ev; wid = eid; wpos = i; }
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

size_t tsc_screen_topk_workspace_bytes(int d_sae, int n_slots) {
    const size_t wg128 = (size_t)((d_sae + 127) / 128) * n_slots, wg256 = (size_t)((d_sae + 255) / 256) * n_slots;
    return 256 + ((wg128 * sizeof(int) + 255) & ~(size_t)255) + wg256 * SQ_CAP_256 * sizeof(unsigned) * 2;
}

int tsc_launch_screen_topk(const tsc_dict *d, const float *x, long long M, long long token_base, float *tk_vals,
                           long long *tk_idx, int n_slots, int k, const float *floor, void *workspace,
                           size_t workspace_bytes, const float *row_bound, hipStream_t stream) {
    if (!d->W_enc_bf16 || !d->x_bf16 || !d->W_enc_t || !d->wnorm)
        return tsc_fail(TSC_ERR_ARG, "tsc_encode_topk_screened needs a TSC_F32_SCREEN dictionary handle");
    if (!floor) return tsc_fail(TSC_ERR_ARG, "tsc_encode_topk_screened needs the state's floor scratch");
    if (!workspace || workspace_bytes < tsc_screen_topk_workspace_bytes(d->d_sae, n_slots))
        return tsc_fail(TSC_ERR_ARG, "screen workspace too small: %zu bytes given, %zu needed", workspace_bytes,
                        tsc_screen_topk_workspace_bytes(d->d_sae, n_slots));
    int *overflow = (int *)workspace;
    int *counters = (int *)((char *)workspace + 256);
    const size_t wg128 = (size_t)((d->d_sae + 127) / 128) * n_slots;
    const size_t counters_bytes = (wg128 * sizeof(int) + 255) & ~(size_t)255;
    unsigned *entries = (unsigned *)((char *)workspace + 256 + counters_bytes);
    for (long long r0 = 0; r0 < M; r0 += SQ_CHUNK_ROWS) {
        const long long rows = M - r0 < SQ_CHUNK_ROWS ? M - r0 : SQ_CHUNK_ROWS;
        const float *xc = x + r0 * d->d_in;
        float *bound = d->screen_bound;
        if (row_bound) bound = const_cast<float *>(row_bound);
        else {
            // (one pass over x: the row bounds and the screen's bf16 operand together)
            int rc = tsc_launch_row_bound(xc, rows, d->d_in, d->b_dec, d->screen_bound, stream, d->x_bf16);
            if (rc) return rc;
        }
        TSC_HIP_CHECK(hipMemsetAsync(workspace, 0, 256 + counters_bytes, stream));
        EncodeArgs a;
        memset(&a, 0, sizeof(a));
        a.x = xc; a.W = d->W_enc; a.b_enc = d->b_enc; a.b_dec = d->b_dec; a.bias = d->bias_fold; a.Wt = d->W_enc_t;
        a.M = rows; a.K = d->d_in; a.N = d->d_sae;
        a.m_split = n_slots;
        EncodeArgs s = a;  // K1: the screen
        s.x_staged = row_bound ? 0 : 1;
        s.screen_wnorm = d->wnorm; s.screen_bound = bound;
        s.screen_rows = row_bound ? nullptr : bound + 2;  // the library's own bounds carry every row's pair
        s.sq_floor = floor; s.sq_entries = entries; s.sq_counters = counters; s.sq_overflow = overflow;
        EncodeArgs probe = s;
        probe.bf16 = 1;
        const int bn = tsc_encode_tile_bn(TSC_MODE_SPARSE, probe);
        s.sq_cap = bn == 256 ? SQ_CAP_256 : SQ_CAP_256 / 2;
        int rc = tsc_launch_encode_bf16(TSC_MODE_SPARSE, s, d->W_enc_bf16, d->x_bf16, stream);
        if (rc) return rc;
        InsertArgs p;  // K2: exact recompute + insert
        p.x = xc; p.Wt = d->W_enc_t; p.bias = d->bias_fold; p.K = d->d_in; p.N = d->d_sae; p.k = k;
        p.n_tiles_n = (d->d_sae + bn - 1) / bn; p.m_split = n_slots; p.cap = s.sq_cap;
        p.token_base = token_base + r0; p.tk_vals = tk_vals; p.tk_idx = tk_idx;
        p.entries = entries; p.counters = counters; p.overflow = overflow;
        const dim3 grid((unsigned)(p.n_tiles_n * n_slots)), block(256);
        if (bn == 256) hipLaunchKernelGGL(screen_insert_kernel<256>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(screen_insert_kernel<128>, grid, block, 0, stream, p);
        TSC_HIP_CHECK(hipGetLastError());
        EncodeArgs f = a;  // K3: exact fallback, a no-op launch unless a queue overflowed
        f.tk_vals = tk_vals; f.tk_idx = tk_idx; f.topk = k; f.token_base = token_base + r0; f.tk_floor = floor;
        f.run_flag = overflow;
        rc = tsc_launch_encode_f32(TSC_MODE_TOPK, f, stream);
        if (rc) return rc;
    }
    return TSC_OK;
}
