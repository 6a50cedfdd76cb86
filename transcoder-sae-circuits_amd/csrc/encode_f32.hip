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
    int q_count;
    int pad[3];
};

struct FeatState {  // the owning thread's view of one feature's list
    int cnt, worst_pos;
    float worst_v;
    long long worst_id;
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

    // TOPK: thread t < BN owns feature n0 + t of list set `slot`
    FeatState st = {0, 0, 0.f, 0};
    float *lv = nullptr;
    long long *lid = nullptr;
    if (MODE == TSC_MODE_TOPK) {
        if (tid < BN) {
            float t = 0.f;
            if (n0 + tid < a.N) {
                const long long off = ((long long)slot * a.N + n0 + tid) * a.topk;
                lv = a.tk_vals + off;
                lid = a.tk_idx + off;
                for (int i = 0; i < a.topk; ++i) {
                    const long long id = lid[i];
                    if (id < 0) break;
                    const float v = lv[i];
                    if (i == 0 || ranks_below(v, id, st.worst_v, st.worst_id)) {
                        st.worst_v = v; st.worst_id = id; st.worst_pos = i;
                    }
                    st.cnt = i + 1;
                }
                t = st.cnt == a.topk ? st.worst_v : 0.f;
            }
            sm.thr[tid] = t;
        }
        __syncthreads();
    }

    f32x4 ra[4], rb[4], rbd;
    const int a_row = tid >> 3, a_col = (tid & 7) * 4;
    const int b_row = tid >> 5, b_col = (tid & 31) * 4;

    // Loads are unconditional from clamped (always valid) addresses: a bounds-checked
    // "load or zero" makes hipcc branch around every load and drain the queue per element.
    // Rows past M and columns past N then hold finite copies of real data and are masked in the
    // epilogue; k past K is zeroed on the W side while staging (so x garbage there multiplies 0).
    const int nc = n0 + b_col < a.N - 4 ? n0 + b_col : a.N - 4;
    auto gload = [&](long long m0, int k0) {
        const int kc = k0 + a_col < a.K - 4 ? k0 + a_col : a.K - 4;
        rbd = *(const f32x4 *)(a.b_dec + kc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long r = m0 + a_row + 32 * i;
            r = r < a.M ? r : a.M - 1;
            ra[i] = *(const f32x4 *)(a.x + r * a.K + kc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int kr = k0 + b_row + 8 * i;
            kr = kr < a.K ? kr : a.K - 1;
            rb[i] = *(const f32x4 *)(a.W + (long long)kr * a.N + nc);
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

    for (long long mt = slot; mt < n_mtiles; mt += a.m_split) {
        const long long m0 = mt * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
#pragma unroll
            for (int bn = 0; bn < 2; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bm][bn][r] = 0.f;

        gload(m0, 0);
        sstore(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) gload(m0, (kt + 1) * BK);
            const float *As = sm.A[kt & 1];
            const float *Bs = sm.B[kt & 1];
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                // MFMA step s of this group contracts k = 8*kk + 4*lh + s (lane half lh): any
                // bijection works as long as A and B agree, and this one makes the A fragment one
                // 16-byte LDS read.
                f32x4 af[2], bf[2];
#pragma unroll
                for (int bm = 0; bm < 2; ++bm)
                    af[bm] = *(const f32x4 *)&As[(wm0 + bm * 32 + li) * LDA + kk * 8 + lh * 4];
#pragma unroll
                for (int bn = 0; bn < 2; ++bn)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        bf[bn][s] = Bs[(kk * 8 + lh * 4 + s) * LDB + wn0 + bn * 32 + li];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int bm = 0; bm < 2; ++bm)
#pragma unroll
                        for (int bn = 0; bn < 2; ++bn)
                            acc[bm][bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[bm][s], bf[bn][s],
                                                                               acc[bm][bn], 0, 0, 0);
            }
            if (more) sstore((kt + 1) & 1, (kt + 1) * BK);
            __syncthreads();
        }

        // ---------------------------------------------------------------- epilogues
        // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#define TSC_FOR_ACC(BODY)                                                                     \
    _Pragma("unroll") for (int bm = 0; bm < 2; ++bm) _Pragma("unroll") for (int bn = 0; bn < 2; ++bn) \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
        const int ml = wm0 + bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;                       \
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
            TSC_FOR_ACC({
                const long long row = m0 + ml;
                if (nok[bn] && row < a.M)
                    __builtin_nontemporal_store(v > 0.f ? v : 0.f,
                                                a.out + row * a.N + (n0 + wn0 + bn * 32 + li));
            })
        } else if (MODE == TSC_MODE_SPARSE) {
            TSC_FOR_ACC({
                const long long row = m0 + ml;
                if (v > 0.f && nok[bn] && row < a.M) {
                    const int pos = atomicAdd(a.sp_count + row, 1);
                    if (pos < a.max_active) {
                        a.sp_feat[row * a.max_active + pos] = n0 + wn0 + bn * 32 + li;
                        a.sp_val[row * a.max_active + pos] = v;
                    }
                }
            })
        } else {
            float thr[2];
            thr[0] = sm.thr[wn0 + li];
            thr[1] = sm.thr[wn0 + 32 + li];
            int any = 0;
            TSC_FOR_ACC({ any |= (v > 0.f && v >= thr[bn] && nok[bn] && m0 + ml < a.M) ? 1 : 0; })
            if (__syncthreads_or(any)) {
                // Rare path.  Candidates are handed to the owning threads through a bounded LDS
                // queue; a lane whose push finds the queue full keeps its place and the round is
                // repeated, so nothing is ever dropped whatever the data.
                int consumed = 0;
                for (;;) {
                    if (tid == 0) sm.q_count = 0;
                    __syncthreads();
                    int ord = 0, blocked = 0;
                    TSC_FOR_ACC({
                        if (v > 0.f && v >= thr[bn] && nok[bn] && m0 + ml < a.M) {
                            if (ord >= consumed && !blocked) {
                                const int pos = atomicAdd(&sm.q_count, 1);
                                if (pos < QCAP) {
                                    sm.q_val[pos] = v;
                                    sm.q_loc[pos] = (ml << 8) | (wn0 + bn * 32 + li);
                                    ++consumed;
                                } else {
                                    blocked = 1;
                                }
                            }
                            ++ord;
                        }
                    })
                    __syncthreads();
                    if (tid < BN) {
                        const int nq = sm.q_count < QCAP ? sm.q_count : QCAP;
                        if (lv != nullptr) {
                            for (int e = 0; e < nq; ++e) {
                                const int loc = sm.q_loc[e];
                                if ((loc & 255) != tid) continue;
                                const float v = sm.q_val[e];
                                const long long id = a.token_base + m0 + (loc >> 8);
                                if (st.cnt < a.topk) {
                                    lv[st.cnt] = v;
                                    lid[st.cnt] = id;
                                    if (st.cnt == 0 || ranks_below(v, id, st.worst_v, st.worst_id)) {
                                        st.worst_v = v; st.worst_id = id; st.worst_pos = st.cnt;
                                    }
                                    ++st.cnt;
                                } else if (ranks_below(st.worst_v, st.worst_id, v, id)) {
                                    const int p = st.worst_pos;
                                    lv[p] = v;
                                    lid[p] = id;
                                    st.worst_v = v; st.worst_id = id;  // rescan; slot p from registers
                                    for (int i = 0; i < a.topk; ++i) {
                                        if (i == p) continue;
                                        const float ev = lv[i];
                                        const long long eid = lid[i];
                                        if (ranks_below(ev, eid, st.worst_v, st.worst_id)) {
                                            st.worst_v = ev; st.worst_id = eid; st.worst_pos = i;
                                        }
                                    }
                                }
                            }
                            sm.thr[tid] = st.cnt == a.topk ? st.worst_v : 0.f;
                        }
                    }
                    if (!__syncthreads_or(blocked)) break;
                }
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
