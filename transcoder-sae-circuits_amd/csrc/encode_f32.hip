// Dictionary encode on the gfx950 fp32 matrix pipe, with three epilogues.
//
//   acts[m, n] = relu( sum_k (x[m,k] - b_dec[k]) * W_enc[k,n] + b_enc[n] )
//
// replaces SparseTranscoder.encode (reference src/mlp_transcoder.py:194-213) and ZSAE.encode
// (src/z_sae.py:73-77).  One workgroup (4 waves, one per SIMD; two workgroups per CU) owns one
// 128-feature column tile of the dictionary and walks a strided set of 128-token row tiles, so the
// per-feature state of the fused top-k epilogue stays with that workgroup for the whole launch.
//
//   main loop : K in steps of 32; x and W_enc tiles prefetched global->VGPR during the MFMAs of the
//               previous step, written to a double-buffered LDS image after them (one barrier per
//               step); `x - b_dec` is applied while staging (the reference subtracts before the
//               contraction, mlp_transcoder.py:200-202).  Each wave computes 64x64 outputs as 2x2
//               v_mfma_f32_32x32x2_f32 accumulators: exact fp32, one rounding per product.
//   DENSE     : + b_enc, ReLU, streaming store of the [M, d_sae] matrix.
//   TOPK      : + b_enc, compare against the feature's current k-th best; survivors go through an
//               LDS queue to the thread that owns the feature, which updates the feature's list
//               in HBM.  The activation matrix is never written.  (max_act_analysis.py:232-249 for
//               all features at once.)
//   SPARSE    : + b_enc, every activation > 0 is appended to its token's (feature, value) list
//               (circuit_lens.py:612-614 / 647-649 before the sort).
#include "tsc_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LDA = BK + 4;  // 144-byte rows: ds_read_b128 of 16 consecutive rows hits 16 distinct slots
constexpr int LDB = BN;      // [k][n] image, read one dword per lane (32 consecutive n: conflict-free)
constexpr int QCAP = 512;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) Smem {
    float A[2][BM * LDA];
    float B[2][BK * LDB];
    float thr[BN];
    float q_val[QCAP];
    int q_loc[QCAP];  // (row in tile << 8) | column in tile
    // per-feature list state, owned by thread t < BN: entries held, and the lowest-ranked entry
    long long f_worst_id[BN];
    float f_worst_v[BN];
    int f_worst_pos[BN];
    int f_cnt[BN];
    int q_count;
    int pad[3];
};

__device__ __forceinline__ bool ranks_below(float v, long long id, float w, long long wid) {
    // true if (v, id) ranks below (w, wid): value descending, token id ascending
    return v < w || (v == w && id > wid);
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void encode_f32_kernel(const EncodeArgs a) {
    __shared__ Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

    // Blocks b and b+8 share an XCD (round-robin dispatch): give every XCD its own column tiles
    // so the 393-KB W_enc slabs of the tiles resident on one XCD stay in that XCD's 4-MiB L2, and
    // the token tiles they stream are shared.  A speed heuristic only.
    int nt, slot;
    {
        const int w = blockIdx.x;
        if ((a.n_tiles_n & 7) == 0) {
            const int q = w >> 3;
            nt = (w & 7) + 8 * (q / a.m_split);
            slot = q % a.m_split;
        } else {
            nt = w / a.m_split;
            slot = w % a.m_split;
        }
    }
    const int n0 = nt * BN;
    const long long n_mtiles = (a.M + BM - 1) / BM;
    const int nk = (a.K + BK - 1) / BK;

    // epilogue constants of this lane: two features (column li of the two 32-wide blocks)
    float be[2];
    bool nok[2];
#pragma unroll
    for (int bn = 0; bn < 2; ++bn) {
        const int n = n0 + wn0 + bn * 32 + li;
        nok[bn] = n < a.N;
        be[bn] = nok[bn] ? a.b_enc[n] : 0.f;
    }

    // TOPK: thread t < BN owns feature n0 + t of list set `slot`; its view of the list (kept in
    // LDS, not registers: the main loop is register-bound) is rebuilt from HBM at start-up
    if (MODE == TSC_MODE_TOPK) {
        if (tid < BN) {
            int cnt = 0, wpos = 0;
            float wv = 0.f;
            long long wid = 0;
            if (n0 + tid < a.N) {
                const long long off = ((long long)slot * a.N + n0 + tid) * a.topk;
                for (int i0 = 0; i0 < a.topk; i0 += 8) {  // lists are compact: live entries first
                    float ev[8];
                    long long eid[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = i0 + j < a.topk ? i0 + j : a.topk - 1;
                        ev[j] = a.tk_vals[off + i];
                        eid[j] = a.tk_idx[off + i];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (i0 + j < a.topk && eid[j] >= 0) {
                            if (cnt == 0 || ranks_below(ev[j], eid[j], wv, wid)) {
                                wv = ev[j]; wid = eid[j]; wpos = i0 + j;
                            }
                            cnt = i0 + j + 1;
                        }
                }
            }
            sm.f_cnt[tid] = cnt;
            sm.f_worst_pos[tid] = wpos;
            sm.f_worst_v[tid] = wv;
            sm.f_worst_id[tid] = wid;
            sm.thr[tid] = cnt == a.topk ? wv : 0.f;
        }
        if (tid == 0) sm.q_count = 0;
        __syncthreads();
    }

    f32x4 ra[4], rb[4], rbd;
    const int a_row = tid >> 3, a_col = (tid & 7) * 4;
    const int b_row = tid >> 5, b_col = (tid & 31) * 4;

    // Loads are unconditional from clamped (always valid) addresses: a bounds-checked
    // "load or zero" makes hipcc branch around every load and drain the queue per element.
    // Rows past M and columns past N then hold finite copies of real data and are masked in the
    // epilogue; k past K is zeroed on the W side while staging (so x garbage there multiplies 0).
    // Addresses are a wave-uniform 64-bit base plus a 32-bit lane offset (saddr form): 64-bit
    // per-lane pointers would cost 2 VGPRs per load in a loop that has none to spare.
    const int nc = n0 + b_col < a.N - 4 ? n0 + b_col : a.N - 4;
    auto gload = [&](long long m0, int k0) {
        const int kc = k0 + a_col < a.K - 4 ? k0 + a_col : a.K - 4;
        rbd = *(const f32x4 *)(a.b_dec + kc);
        const char *xt = (const char *)(a.x + m0 * a.K);
        const long long rows_left = a.M - m0;  // >= 1
        const int rmax = rows_left > BM ? BM - 1 : (int)rows_left - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = a_row + 32 * i < rmax ? a_row + 32 * i : rmax;
            ra[i] = *(const f32x4 *)(xt + (unsigned)(r * a.K + kc) * 4u);
        }
        const char *wt = (const char *)a.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int kr = k0 + b_row + 8 * i;
            kr = kr < a.K ? kr : a.K - 1;
            rb[i] = *(const f32x4 *)(wt + ((size_t)kr * a.N + nc) * 4u);
        }
    };
    auto sstore = [&](int buf, int k0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(f32x4 *)&sm.A[buf][(a_row + 32 * i) * LDA + a_col] = ra[i] - rbd;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(f32x4 *)&sm.B[buf][(b_row + 8 * i) * LDB + b_col] = (k0 + b_row + 8 * i < a.K) ? rb[i] : z;
    };

    // Fragments of one group of 8 k (4 MFMA k-steps): MFMA step s contracts k = 8*kk + 4*lh + s
    // (lane half lh).  Any bijection works as long as A and B agree, and this one makes the A
    // fragment a single 16-byte LDS read.
    struct Frag {
        f32x4 af[2], bf[2];
    };
    auto ldfrag = [&](int buf, int kk, Frag &f) {
        const float *As = sm.A[buf];
        const float *Bs = sm.B[buf];
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
            f.af[bm] = *(const f32x4 *)&As[(wm0 + bm * 32 + li) * LDA + kk * 8 + lh * 4];
#pragma unroll
        for (int bn = 0; bn < 2; ++bn)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                f.bf[bn][s] = Bs[(kk * 8 + lh * 4 + s) * LDB + wn0 + bn * 32 + li];
    };

    // The workgroup's work is one continuous stream of k-steps over all its token tiles: the
    // prefetch of a tile's first k-step is issued during the previous tile's last one, and the
    // epilogue runs with the next tile's first k-step already staged.
    int parity = 0;
    long long mt = slot;
    if (mt < n_mtiles) {
        gload(mt * BM, 0);
        sstore(0, 0);
    }
    __syncthreads();
    Frag F, G;
    if (mt < n_mtiles) ldfrag(0, 0, F);

    for (; mt < n_mtiles; mt += a.m_split) {
        const long long m0 = mt * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
#pragma unroll
            for (int bn = 0; bn < 2; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bm][bn][r] = 0.f;

#define TSC_MFMA(FR, S0, S1)                                                                   \
    _Pragma("unroll") for (int s = S0; s < S1; ++s) _Pragma("unroll") for (int bm = 0; bm < 2; ++bm) \
        _Pragma("unroll") for (int bn = 0; bn < 2; ++bn) acc[bm][bn] =                         \
            __builtin_amdgcn_mfma_f32_32x32x2f32(FR.af[bm][s], FR.bf[bn][s], acc[bm][bn], 0, 0, 0);

        for (int kt = 0; kt < nk; ++kt) {
            // what to prefetch: the next k-step of this tile, or the first of the next tile
            const bool last_k = kt + 1 == nk;
            const long long next_m0 = last_k ? (mt + a.m_split) * BM : m0;
            const int next_k0 = last_k ? 0 : (kt + 1) * BK;
            const bool more = !last_k || mt + a.m_split < n_mtiles;
#ifndef TSC_EXP_NOLOAD
            if (more) gload(next_m0, next_k0);
#endif
            ldfrag(parity, 1, G);
            TSC_MFMA(F, 0, 4)
            ldfrag(parity, 2, F);
            TSC_MFMA(G, 0, 4)
            ldfrag(parity, 3, G);
            TSC_MFMA(F, 0, 4)
            // stage the prefetched k-step into the other buffer (free since the last barrier),
            // cover the LDS write latency and the barrier with the first half of the last group,
            // and the first fragment read of the new buffer with the second half
            if (more) sstore(parity ^ 1, next_k0);
            TSC_MFMA(G, 0, 2)
            __syncthreads();
            if (more) ldfrag(parity ^ 1, 0, F);
            TSC_MFMA(G, 2, 4)
            parity ^= 1;
        }
#undef TSC_MFMA

        // ---------------------------------------------------------------- epilogues
        // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
        // The lane's row/column bases are made opaque once per tile: otherwise hipcc hoists the 64
        // per-element row ids / addresses out of the tile loop and spills them around the k-loop.
        int rbase = wm0 + 4 * lh, cbase = wn0 + li;
        asm volatile("" : "+v"(rbase), "+v"(cbase));
        const long long rows_left = a.M - m0;
        const int lim = (rows_left > BM ? BM : (int)rows_left) - rbase;  // element row offset c is valid iff c < lim
#define TSC_FOR_ACC(BODY)                                                                     \
    _Pragma("unroll") for (int bm = 0; bm < 2; ++bm) _Pragma("unroll") for (int bn = 0; bn < 2; ++bn) \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
        const int c = bm * 32 + (r & 3) + 8 * (r >> 2);                                       \
        const float v = acc[bm][bn][r];                                                       \
        BODY                                                                                  \
    }

#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
#pragma unroll
            for (int bn = 0; bn < 2; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bm][bn][r] += be[bn];

        if (MODE == TSC_MODE_DENSE) {
            float *obase = a.out + (m0 + rbase) * a.N + (n0 + cbase);
            TSC_FOR_ACC({
                if (nok[bn] && c < lim)
                    __builtin_nontemporal_store(v > 0.f ? v : 0.f, obase + (size_t)c * a.N + bn * 32);
            })
        } else if (MODE == TSC_MODE_SPARSE) {
            TSC_FOR_ACC({
                if (v > 0.f && nok[bn] && c < lim) {
                    const long long row = m0 + rbase + c;
                    const int pos = atomicAdd(a.sp_count + row, 1);
                    if (pos < a.max_active) {
                        a.sp_feat[row * a.max_active + pos] = n0 + cbase + bn * 32;
                        a.sp_val[row * a.max_active + pos] = v;
                    }
                }
            })
        } else {
            // Lane-local pre-check: the lane's largest value per feature against that feature's
            // k-th best (rows past M are copies of the last real row, so they cannot raise an alarm
            // that the exact check below would not dismiss).  Lanes with a candidate push it on a
            // bounded LDS queue; the thread that owns the feature folds it into the feature's list
            // in HBM.  A lane whose push finds the queue full keeps its place and the round is
            // repeated, so nothing is dropped whatever the data.  Cost: one barrier per tile when
            // nobody has a candidate, two when somebody has.
            float thr[2];
            int any = 0;
#pragma unroll
            for (int bn = 0; bn < 2; ++bn) {
                thr[bn] = nok[bn] ? sm.thr[cbase + bn * 32] : __builtin_inff();
                float mx = acc[0][bn][0];
#pragma unroll
                for (int bm = 0; bm < 2; ++bm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[bm][bn][r]);
                any |= (mx > 0.f && mx >= thr[bn]) ? 1 : 0;
            }
            int consumed = 0;
            for (;;) {  // sm.q_count == 0 on entry
                if (any) {
                    int ord = 0, blocked = 0;
                    TSC_FOR_ACC({
                        if (v > 0.f && v >= thr[bn] && c < lim) {
                            if (ord >= consumed && !blocked) {
                                const int pos = atomicAdd(&sm.q_count, 1);
                                if (pos < QCAP) {
                                    sm.q_val[pos] = v;
                                    sm.q_loc[pos] = ((rbase + c) << 8) | (cbase + bn * 32);
                                    ++consumed;
                                } else {
                                    blocked = 1;
                                }
                            }
                            ++ord;
                        }
                    })
                }
                __syncthreads();
                const int pushed = sm.q_count;  // > QCAP: some lane is blocked and needs another round
                if (pushed == 0) break;
                if (tid < BN && n0 + tid < a.N) {
                    const int nq = pushed < QCAP ? pushed : QCAP;
                    const int k = a.topk;
                    const long long off = ((long long)slot * a.N + n0 + tid) * k;
                    float *lv = a.tk_vals + off;
                    long long *lid = a.tk_idx + off;
                    int cnt = sm.f_cnt[tid], wpos = sm.f_worst_pos[tid];
                    float wv = sm.f_worst_v[tid];
                    long long wid = sm.f_worst_id[tid];
                    for (int e = 0; e < nq; ++e) {
                        const int loc = sm.q_loc[e];
                        if ((loc & 255) != tid) continue;
                        const float v = sm.q_val[e];
                        const long long id = a.token_base + m0 + (loc >> 8);
                        if (cnt < k) {
                            lv[cnt] = v;
                            lid[cnt] = id;
                            if (cnt == 0 || ranks_below(v, id, wv, wid)) {
                                wv = v; wid = id; wpos = cnt;
                            }
                            ++cnt;
                        } else if (ranks_below(wv, wid, v, id)) {
                            const int p = wpos;
                            lv[p] = v;
                            lid[p] = id;
                            wv = v; wid = id;
                            // find the new lowest-ranked entry; position p is taken from registers,
                            // the others are read eight at a time so the loads overlap
                            for (int i0 = 0; i0 < k; i0 += 8) {
                                float ev[8];
                                long long eid[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const int i = i0 + j < k ? i0 + j : k - 1;
                                    ev[j] = lv[i];
                                    eid[j] = lid[i];
                                }
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    if (i0 + j < k && i0 + j != p && ranks_below(ev[j], eid[j], wv, wid)) {
                                        wv = ev[j]; wid = eid[j]; wpos = i0 + j;
                                    }
                            }
                        }
                    }
                    sm.f_cnt[tid] = cnt;
                    sm.f_worst_pos[tid] = wpos;
                    sm.f_worst_v[tid] = wv;
                    sm.f_worst_id[tid] = wid;
                    sm.thr[tid] = cnt == k ? wv : 0.f;
                }
                __syncthreads();  // lists and thresholds updated; everyone has read q_count
                if (tid == 0) sm.q_count = 0;
                if (pushed <= QCAP) break;  // reset is ordered before the next tile's pushes by the k-loop barriers
                __syncthreads();
            }
        }
#undef TSC_FOR_ACC
    }
}

template <int MODE>
int launch(const EncodeArgs &a, hipStream_t stream) {
    const long long grid = (long long)a.n_tiles_n * a.m_split;
    if (grid <= 0 || grid > 0x7fffffffLL) return tsc_fail(TSC_ERR_ARG, "encode grid %lld out of range", grid);
    hipLaunchKernelGGL(encode_f32_kernel<MODE>, dim3((unsigned)grid), dim3(NT), 0, stream, a);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

}  // namespace

int tsc_launch_encode_f32(int mode, const EncodeArgs &a, hipStream_t stream) {
    if (a.M <= 0) return TSC_OK;
    switch (mode) {
        case TSC_MODE_DENSE: return launch<TSC_MODE_DENSE>(a, stream);
        case TSC_MODE_TOPK: return launch<TSC_MODE_TOPK>(a, stream);
        case TSC_MODE_SPARSE: return launch<TSC_MODE_SPARSE>(a, stream);
    }
    return tsc_fail(TSC_ERR_ARG, "unknown encode mode %d", mode);
}
