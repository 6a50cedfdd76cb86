// Dictionary encode on the gfx950 matrix pipe (fp32, or bf16 operands), with three epilogues.
//
//   acts[m, n] = relu( sum_k (x[m,k] - b_dec[k]) * W_enc[k,n] + b_enc[n] )
//
// replaces SparseTranscoder.encode (reference src/mlp_transcoder.py:194-213) and ZSAE.encode
// (src/z_sae.py:73-77).  One workgroup owns one column tile of the dictionary and walks a strided set of
// token row tiles, so the per-feature state of the fused top-k epilogue stays with that workgroup for the
// whole launch.  DESIGN.md section 4.1 has the measurements behind every choice below.
//
//   main loop : LDS-DMA variant (d_in % 64 == 0): the raw x tile and the tile of the handle's transposed
//               W_enc copy go global -> LDS by global_load_lds_dwordx4 (8 rows x 128 bytes per piece,
//               16-byte chunks XOR-swizzled through the source address), two LDS buffers, one barrier per
//               k-step, k-loop unrolled by two so that every LDS address is an immediate; b_dec is folded
//               into the bias (bias' = b_enc - b_dec @ W_enc).  Register-staged variant (any d_in % 4 == 0):
//               tiles prefetched global -> VGPR, `x - b_dec` applied while they are written to LDS, as the
//               reference subtracts before the contraction (mlp_transcoder.py:200-202).
//               fp32: v_mfma_f32_32x32x2_f32, exact fp32, one rounding per product.
//               bf16: v_mfma_f32_32x32x16_bf16 on operands prepared by encode_operands.hip.
//   DENSE     : + bias, ReLU, streaming store of the [M, d_sae] matrix.
//   TOPK      : + bias, compare against the feature's current k-th best (its own slot's, and the cross-slot
//               floor); survivors go through per-feature LDS slots to the thread that owns the feature,
//               which updates the feature's list (values mirrored in LDS, HBM written through).  The
//               activation matrix is never written.  (max_act_analysis.py:232-249 for all features at once.)
//   SPARSE    : + bias, every activation > 0 is appended to its token's (feature, value) list
//               (circuit_lens.py:612-614 / 647-649 before the sort).
//
// Two tile geometries are compiled (Cfg): 128x128 (4 waves, two workgroups per CU) and 256x256 (8 waves,
// one workgroup per CU); pick_large() chooses per call.  On this pipe every instruction a wave issues costs
// the matrix pipe about as much as it would cost the VALU, so the design rule throughout is: as few
// non-MFMA instructions per MFMA as possible.
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "tsc_internal.h"

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;  // 144-byte rows: ds_read_b128 of 16 consecutive rows hits 16 distinct slots
constexpr int QD = 4;   // candidate slots per feature and round
constexpr int KL = 20;  // list length up to which the lists' values are mirrored in LDS

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WMB_, int WNB_, int WVM_, int WVN_, int OCC_>
struct Cfg {
    static constexpr int WMB = WMB_, WNB = WNB_, OCC = OCC_;  // 32x32 blocks per wave; workgroups per CU
    static constexpr int WVM = WVM_, WVN = WVN_, NW = WVM * WVN, NT = 64 * NW;  // waves as WVM x WVN
    static constexpr int BM = WVM * WMB * 32, BN = WVN * WNB * 32;
    static constexpr int WQ = WMB * WNB * 32;  // SPARSE: entries a wave stages per round (1/32 of its elements)
    static constexpr int A_LD = BM * 8 / NT;                  // float4 loads of x per thread and k-step
    static constexpr int B_TPR = BN / 4;                      // threads per W row
    static constexpr int B_RPP = NT / B_TPR;                  // W rows per pass
    static constexpr int B_LD = BK / B_RPP;                   // float4 loads of W per thread and k-step
    static_assert(BN <= 256, "one owner thread per feature, column index packed in 8 bits");
    static_assert(BM <= 256, "row index packed in 8 bits");
};
typedef Cfg<2, 2, 2, 2, 2> CfgSmall;  // 128x128, 4 waves of 64x64, two workgroups per CU
typedef Cfg<4, 2, 2, 4, 1> CfgLarge;  // 256x256, 8 waves of 128x64, one workgroup per CU
// (Cfg<4, 4, 2, 2, 1> -- four waves of 128x128, accumulators in AGPRs, a third less LDS fragment traffic per MFMA --
// compiles to 640-700 spilled registers per lane with this body: hipcc keeps copies of the accumulators in VGPRs)

template <class C, int MODE, bool DMA, bool LLDS>
struct __attribute__((aligned(16))) Smem {
    static constexpr int TN = MODE == TSC_MODE_TOPK ? C::BN : 1;  // the per-feature state exists in the TOPK kernels only
    static constexpr bool SP = MODE == TSC_MODE_SPARSE;
    float A[2][C::BM * (DMA ? BK : LDA)];  // register staging: padded rows; LDS-DMA: 32-float rows, 16-byte chunks XOR-swizzled
    float B[2][BK * C::BN];  // [k][n] image, read one dword per lane (32 consecutive n: conflict-free)
    float lv[LLDS ? C::BN * KL : 4];  // TOPK, k <= KL: the values of the workgroup's lists, [feature][KL]
    float thr[TN];
    float q_val[TN * QD];         // candidates of one round, [feature][QD]
    unsigned char q_row[TN * QD];  // their rows in the tile
    int f_nq[TN];                  // candidates pushed for the feature this round (may exceed QD)
    // per-feature list state, owned by thread t < BN: entries held, and the lowest-ranked entry
    long long f_worst_id[TN];
    float f_worst_v[TN];
    unsigned char f_worst_pos[TN];  // k <= 64
    unsigned char f_cnt[TN];
    float f_floor[TN];  // k-th best of the feature over all slots when the launch started (0: unknown)
    int pushed, again;  // round flags: somebody pushed / somebody found its feature's slots full
    // SPARSE: every wave stages the entries of its part of the tile in a queue of its own (no atomics, no barrier: the
    // wave that wrote a queue is the one that drains it), then issues their global appends together
    // (two arrays, two ds_write_b32: one 8-byte write wants value and place in a register PAIR, the value is an
    // accumulator, and the pairing constraint spills 200 registers -- measured: emit 670 -> 1010 us)
    float wq_val[SP ? C::NW : 1][SP ? C::WQ : 1];
    int wq_loc[SP ? C::NW : 1][SP ? C::WQ : 1];  // (row in tile << 8) | column in tile
    // bf16 screen with per-row bounds (EncodeArgs::screen_rows): the bound pairs of the tile's rows (filled by LDS-DMA with
    // the tile's first k-step; two buffers, by tile parity: a wave may still drain the previous tile), the two margin norms
    // and the emit threshold of the workgroup's columns, and the call's maxima.  Everything the re-test of a candidate
    // needs sits in LDS: the k-loop runs at the register limit, and four more live registers spill (measured: the
    // 24-dictionary emit 725 -> 917 us).
    float2 rowb[2][SP && DMA ? C::BM : 1];
    float coln[3][SP && DMA ? C::BN : 1];
    float2 rbmax;
};

__device__ __forceinline__ bool ranks_below(float v, long long id, float w, long long wid) {
    // true if (v, id) ranks below (w, wid): value descending, token id ascending
    return v < w || (v == w && id > wid);
}

#ifdef TSC_STAMPS  // diagnosis builds only: cycles per phase, summed over the workgroups' waves 0 and 3
__device__ unsigned long long g_stamps[16];
#define TSC_STAMP(i) { const unsigned long long t_ = clock64(); st_acc[i] += t_ - st_last; st_last = t_; }
#else
#define TSC_STAMP(i)
#endif

// cache-policy bits of the LDS-DMA loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef TSC_X_AUX
#define TSC_X_AUX 0
#endif
#ifndef TSC_W_AUX
#define TSC_W_AUX 0
#endif

#ifdef TSC_NO_PIN
#define TSC_PIN
#else
#define TSC_PIN __builtin_amdgcn_sched_barrier(0);
#endif

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// KALIGNED: K % 32 == 0 (no k tail).  DMA (needs K % 64 == 0 and a.bias = b_enc - b_dec @ W_enc):
// tiles go global -> LDS by global_load_lds_dwordx4, no register staging.
// LLDS (TOPK with DMA, k <= KL): the values of the lists live in LDS for the whole launch (HBM is written
// through), so folding a candidate into its list needs no load from HBM.
// BF16 (needs DMA; K % 128 == 0): a.x is bf16(x - b_dec) [M][K] and a.W is bf16(W_enc) transposed [N][K],
// both prepared by encode_bf16.hip; a k-step is 64 k (the same 128-byte rows), the W tile is staged exactly
// like the x tile, and one v_mfma_f32_32x32x16_bf16 replaces four fp32 MFMAs.
// (the body is a device function so that the grouped launch below can run it for several dictionaries; `wg` is the
// workgroup's index within ITS problem's grid)
template <class C, int MODE, bool KALIGNED, bool DMA, bool LLDS, bool BF16>
__device__ __forceinline__ void encode_body(const EncodeArgs &a, const int wg) {
    constexpr int WMB = C::WMB, WNB = C::WNB, BM = C::BM, BN = C::BN;
    constexpr int LDA_ = DMA ? BK : LDA;
    constexpr int ES = BF16 ? 2 : 4;  // bytes per staged element; a k-step is 128 bytes of every row
    static_assert(!BF16 || DMA, "the bf16 operands are staged by LDS-DMA only");
    constexpr int NT = C::NT;
    constexpr int APW = BM / 8 / C::NW, BPW = BN / 8 / C::NW;  // 1-KiB DMA pieces per wave and k-step (x tile, W tile)
    static_assert(!DMA || KALIGNED, "DMA staging has no k tail handling");
    static_assert(!LLDS || (DMA && MODE == TSC_MODE_TOPK), "LDS lists belong to the fused top-k DMA kernel");
    __shared__ Smem<C, MODE, DMA, LLDS> sm;
    if (MODE == TSC_MODE_TOPK && a.run_flag && *a.run_flag == 0) return;  // fallback launch of the screen: nothing overflowed
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / C::WVN) * (WMB * 32), wn0 = (wave % C::WVN) * (WNB * 32);

    // Blocks b and b+8 share an XCD (round-robin dispatch): give every XCD its own column tiles
    // so the W_enc slabs of the tiles resident on one XCD stay in that XCD's 4-MiB L2, and the
    // token tiles they stream are shared.  A speed heuristic only.
    int nt, slot;
    {
        const int w = wg;
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
    const int nk = DMA ? a.K / (128 / ES) : (a.K + BK - 1) / BK;
#ifdef TSC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif

    // epilogue constants of this lane: WNB features (column li of each 32-wide block)
    float be[WNB];
    bool nok[WNB];
#pragma unroll
    for (int bn = 0; bn < WNB; ++bn) {
        const int n = n0 + wn0 + bn * 32 + li;
        nok[bn] = n < a.N;
        be[bn] = nok[bn] ? (DMA ? a.bias[n] : a.b_enc[n]) : 0.f;
        if (BF16 && MODE == TSC_MODE_SPARSE && a.screen_wnorm && nok[bn])  // bf16 screen: see row_bound_kernel (encode_operands.hip)
            be[bn] += a.screen_bound[0] * a.screen_wnorm[n] + a.screen_bound[1] * a.screen_wnorm[a.N + n] +
                      2.f * (float)a.K * 5.9604645e-8f * fabsf(be[bn]);
    }
    // per-row bounds of the bf16 screen (wave-uniform).  Not with a single row (the DMA below moves two rows per lane) and
    // not when a row of the call is not finite: the maxima are then infinite, every pair is a candidate and stays one
    // (inf - inf would dismiss them all).
    bool has_rb = false;
    if (BF16 && MODE == TSC_MODE_SPARSE && DMA && a.screen_rows && a.M > 1)
        has_rb = a.screen_bound[0] < __builtin_inff() && a.screen_bound[1] < __builtin_inff();
    // SPARSE: a value is emitted when it exceeds this (0 = every positive activation; the top-k screen: just below the
    // feature's floor, so that a value equal to the k-th best -- which a smaller token id may still outrank -- passes;
    // columns past N: nothing exceeds +inf, so the scan needs no column predicate)
    auto emit_threshold = [&](int n) -> float {
        if (n >= a.N) return __builtin_inff();
        const float fl = a.sq_floor ? a.sq_floor[n] : 0.f;
        return fl > 0.f ? __uint_as_float(__float_as_uint(fl) - 1u) : 0.f;
    };
    float sp_thr[WNB];
#pragma unroll
    for (int bn = 0; bn < WNB; ++bn) sp_thr[bn] = MODE == TSC_MODE_SPARSE ? emit_threshold(n0 + wn0 + bn * 32 + li) : 0.f;

    // TOPK: thread t < BN owns feature n0 + t of list set `slot`; its view of the list (kept in
    // LDS, not registers) is rebuilt from HBM at start-up
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
                            if constexpr (LLDS) sm.lv[tid * KL + i0 + j] = ev[j];
                        }
                }
            }
            sm.f_cnt[tid] = cnt;
            sm.f_worst_pos[tid] = wpos;
            sm.f_worst_v[tid] = wv;
            sm.f_worst_id[tid] = wid;
            const float fl = a.tk_floor && n0 + tid < a.N ? a.tk_floor[n0 + tid] : 0.f;
            sm.f_floor[tid] = fl;
            sm.thr[tid] = fmaxf(cnt == a.topk ? wv : 0.f, fl);
        }
        if (tid < BN) sm.f_nq[tid] = 0;
        if (tid == 0) sm.pushed = sm.again = 0;
        __syncthreads();
    }

    f32x4 ra[DMA ? 1 : C::A_LD], rb[DMA ? 1 : C::B_LD], rbd;
    const int a_row = tid >> 3, a_col = (tid & 7) * 4;
    const int b_row = tid / C::B_TPR, b_col = (tid % C::B_TPR) * 4;

    // Loads are unconditional from clamped (always valid) addresses: a bounds-checked
    // "load or zero" makes hipcc branch around every load and drain the queue per element.
    // Rows past M and columns past N then hold finite copies of real data and are masked in the
    // epilogue; k past K is zeroed on the W side while staging (so x garbage there multiplies 0).
    // Addresses are a wave-uniform 64-bit base (advanced by scalar adds per k-step) plus a 32-bit
    // lane offset that is constant per token tile: vector-side address arithmetic per k-step costs
    // matrix throughput directly.
    //
    // DMA staging: one global_load_lds_dwordx4 writes 64 lanes x 16 B = 1 KiB of LDS linearly, from
    // per-lane source addresses.  x piece p = rows 8p..8p+7 of the tile, 128 B (32 k) per row; the
    // 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7), which makes the A-fragment
    // ds_read_b128 of 16 consecutive rows conflict-free without padding; the swizzle is applied to
    // the source address.  W piece q = floats 256q..256q+255 of the [32][BN] image, unswizzled.
    unsigned a_voff[DMA ? APW : C::A_LD], b_voff[DMA ? BPW : C::B_LD];
    const unsigned bd_voff = (unsigned)a_col * 4u;
    if constexpr (DMA) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {  // rows of the W tile are features: same piece shape and swizzle as the x tile
            const int row = (wave * BPW + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int n = n0 + row < a.N ? n0 + row : a.N - 1;
            b_voff[j] = (unsigned)n * (unsigned)a.K * (unsigned)ES + (unsigned)c * 16u + (4096u - (j & 3) * 1024u);
        }
    } else {
        const int nc = n0 + b_col < a.N - 4 ? n0 + b_col : a.N - 4;
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i)
            b_voff[i] = ((unsigned)(b_row + C::B_RPP * i) * (unsigned)a.N + (unsigned)nc) * 4u;
    }
    auto set_tile = [&](long long m0) {  // lane offsets of the x rows of the tile being prefetched
        const long long rows_left = a.M - m0;  // >= 1
        const int rmax = rows_left > BM ? BM - 1 : (int)rows_left - 1;
        if constexpr (DMA) {
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int row = (wave * APW + j) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                a_voff[j] = (unsigned)(row < rmax ? row : rmax) * (unsigned)a.K * (unsigned)ES + (unsigned)c * 16u +
                            (4096u - (j & 3) * 1024u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::A_LD; ++i) {
                const int r = a_row + (NT / 8) * i < rmax ? a_row + (NT / 8) * i : rmax;
                a_voff[i] = (unsigned)(r * a.K + a_col) * 4u;
            }
        }
    };
    // DMA piece j of this wave (j < APW: x, else W) of the k-step at (xt, wt) into LDS buffer `buf`.
    // lds_a / lds_b: LDS byte addresses of this wave's first x / W piece in buffer 0.
    const unsigned lds_a0 = (unsigned)(size_t)(lptr_t)&sm.A[0][wave * APW * 256];
    const unsigned lds_b0 = (unsigned)(size_t)(lptr_t)&sm.B[0][wave * BPW * 256];
    // The instruction's immediate offset applies to the global AND the LDS address: pieces j = 4g .. 4g+3 of an
    // operand share one LDS base (one M0 write instead of four) and take their 1-KiB stride from the
    // immediate; the lane offsets carry the opposite bias (4096 - 1024 * (j & 3), against a tile base lowered by
    // 4096) so that the global address is unchanged.
#define TSC_GLDS(G, L, Q, AUX)                                                                         \
    switch (Q) {                                                                                       \
        case 0: __builtin_amdgcn_global_load_lds(G, L, 16, 0, AUX); break;                             \
        case 1: __builtin_amdgcn_global_load_lds(G, L, 16, 1024, AUX); break;                          \
        case 2: __builtin_amdgcn_global_load_lds(G, L, 16, 2048, AUX); break;                          \
        default: __builtin_amdgcn_global_load_lds(G, L, 16, 3072, AUX); break;                         \
    }
    auto dma_piece = [&](int j, const char *xt, const char *wt, int buf, unsigned lds_a, unsigned lds_b) {
#ifdef TSC_ABL_NODMA  // diagnosis builds only (wrong results): what the k-loop costs without its global -> LDS traffic
        return;
#endif
#ifdef TSC_ABL_NODMA_B
        if (j >= APW) return;
#endif
        if (j < APW) {
            TSC_GLDS((gptr_t)(xt - 4096 + a_voff[j]),
                     (lptr_t)(size_t)(lds_a + buf * (unsigned)sizeof(sm.A[0]) + (j >> 2) * 4096), j & 3, TSC_X_AUX)
        } else {
            const int jb = j - APW;
            TSC_GLDS((gptr_t)(wt - 4096 + b_voff[jb]),
                     (lptr_t)(size_t)(lds_b + buf * (unsigned)sizeof(sm.B[0]) + (jb >> 2) * 4096), jb & 3, TSC_W_AUX)
        }
    };
    auto gload = [&](long long m0, int k0) {
        if (KALIGNED) {
            const char *xt = (const char *)(a.x + m0 * a.K + k0);
            const char *wt = (const char *)(a.W + (long long)k0 * a.N);
            rbd = *(const f32x4 *)((const char *)(a.b_dec + k0) + bd_voff);
#pragma unroll
            for (int i = 0; i < C::A_LD; ++i) ra[i] = *(const f32x4 *)(xt + a_voff[i]);
#pragma unroll
            for (int i = 0; i < C::B_LD; ++i) rb[i] = *(const f32x4 *)(wt + b_voff[i]);
        } else {
            const int nc = n0 + b_col < a.N - 4 ? n0 + b_col : a.N - 4;
            const int kc = k0 + a_col < a.K - 4 ? k0 + a_col : a.K - 4;
            rbd = *(const f32x4 *)(a.b_dec + kc);
            const char *xt = (const char *)(a.x + m0 * a.K);
#pragma unroll
            for (int i = 0; i < C::A_LD; ++i)
                ra[i] = *(const f32x4 *)(xt + a_voff[i] + (long long)(kc - a_col) * 4);  // kc < a_col when K < 32: signed
            const char *wt = (const char *)a.W;
#pragma unroll
            for (int i = 0; i < C::B_LD; ++i) {
                int kr = k0 + b_row + C::B_RPP * i;
                kr = kr < a.K ? kr : a.K - 1;
                rb[i] = *(const f32x4 *)(wt + ((size_t)kr * a.N + nc) * 4u);
            }
        }
    };
    auto sstore = [&](int buf, int k0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i)
            *(f32x4 *)&sm.A[buf][(a_row + (NT / 8) * i) * LDA + a_col] = ra[i] - rbd;
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i)
            *(f32x4 *)&sm.B[buf][(b_row + C::B_RPP * i) * BN + b_col] =
                (KALIGNED || k0 + b_row + C::B_RPP * i < a.K) ? rb[i] : z;
    };

    // Fragments of one group of 8 k (4 MFMA k-steps): MFMA step s contracts k = 8*kk + 4*lh + s
    // (lane half lh).  Any bijection works as long as A and B agree, and this one makes the A
    // fragment a single 16-byte LDS read.
    struct Frag {
        f32x4 af[WMB], bf[WNB];
    };
    auto ldfrag = [&](int buf, int kk, Frag &f) {
#ifdef TSC_ABL_NOFRAG  // diagnosis builds only (wrong results): what the k-loop costs without its fragment reads
        asm volatile("" : "+v"(f.af[0]), "+v"(f.bf[0]));
        return;
#endif
        const float *As = sm.A[buf];
        const float *Bs = sm.B[buf];
#pragma unroll
        for (int bm = 0; bm < WMB; ++bm)
            f.af[bm] = *(const f32x4 *)&As[(wm0 + bm * 32 + li) * LDA_ +
                                           (DMA ? (((kk * 2 + lh) ^ ((li >> 1) & 7)) * 4) : kk * 8 + lh * 4)];
#ifdef TSC_ABL_NOFRAG_B
        return;
#endif
#pragma unroll
        for (int bn = 0; bn < WNB; ++bn) {
            if constexpr (DMA) {
                f.bf[bn] = *(const f32x4 *)&Bs[(wn0 + bn * 32 + li) * BK + (((kk * 2 + lh) ^ ((li >> 1) & 7)) * 4)];
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) f.bf[bn][s] = Bs[(kk * 8 + lh * 4 + s) * BN + wn0 + bn * 32 + li];
            }
        }
    };

    int ep_tiles = 0;
    auto epilogue = [&](f32x16 (&acc)[WMB][WNB], const long long m0) {
        // ---------------------------------------------------------------- epilogues
        // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
        // The lane's row/column bases are made opaque once per tile: otherwise hipcc hoists the
        // per-element row ids / addresses out of the tile loop and spills them around the k-loop.
        int rbase = wm0 + 4 * lh, cbase = wn0 + li;
        asm volatile("" : "+v"(rbase), "+v"(cbase));
        const long long rows_left = a.M - m0;
        const int lim = (rows_left > BM ? BM : (int)rows_left) - rbase;  // element row offset c is valid iff c < lim
#define TSC_FOR_ACC(BODY)                                                                       \
    _Pragma("unroll") for (int bm = 0; bm < WMB; ++bm) _Pragma("unroll") for (int bn = 0; bn < WNB; ++bn) \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                        \
        const int c = bm * 32 + (r & 3) + 8 * (r >> 2);                                         \
        const float v = acc[bm][bn][r];                                                         \
        BODY                                                                                    \
    }

        // the bias is already in the accumulators: they start from it (see the tile loop)

        if (MODE == TSC_MODE_DENSE) {
            if (rows_left >= BM && n0 + BN <= a.N) {
                // Interior tile: no predicates.  The address is a scalar row base (tile origin + element row * N)
                // plus one 32-bit lane offset per column block, so an element costs a v_max and a store.
                const char *tile = (const char *)(a.out + m0 * a.N + n0);
                unsigned vo[WNB];
#pragma unroll
                for (int bn = 0; bn < WNB; ++bn) vo[bn] = ((unsigned)rbase * (unsigned)a.N + (unsigned)(cbase + bn * 32)) * 4u;
                const size_t row_bytes = (size_t)a.N * 4;
                if (a.no_relu) {
                    TSC_FOR_ACC({ __builtin_nontemporal_store(v, (float *)(tile + c * row_bytes + vo[bn])); })
                } else {
                    TSC_FOR_ACC({ __builtin_nontemporal_store(fmaxf(v, 0.f), (float *)(tile + c * row_bytes + vo[bn])); })
                }
            } else {
                float *obase = a.out + (m0 + rbase) * a.N + (n0 + cbase);
                TSC_FOR_ACC({
                    if (nok[bn] && c < lim)
                        __builtin_nontemporal_store((v > 0.f || a.no_relu) ? v : 0.f, obase + (size_t)c * a.N + bn * 32);
                })
            }
        } else if (MODE == TSC_MODE_SPARSE) {
            // Append (feature, value) to the token's list.  A wave-level round trip per emitted element (an atomic on a
            // shared counter, or the global append itself) costs far more than the scan: at the 0.3 % of a CircuitLens
            // screen a fifth of the 128 per-lane elements have a candidate in SOME lane of the wave.  So the scan only
            // stages: the position of an entry in the wave's own LDS queue is the wave's running count (a scalar) plus the
            // lane's rank in the ballot, which needs no atomic and no wait.  The wave then drains its queue, one entry per
            // lane, so the global appends of a tile go out together.  A queue holds 1/32 of the wave's elements; when more
            // pass, the scan is repeated for the next window of the count -- whatever the data, nothing is dropped.
            const int par = ep_tiles & 1;
            ++ep_tiles;
            constexpr int WQ = C::WQ;
            float *qv = sm.wq_val[wave];
            int *ql = sm.wq_loc[wave];
            for (int win = 0;; win += WQ) {
                int cnt = 0;  // wave-uniform: candidates of the elements scanned so far
                // (opaque per window: otherwise hipcc hoists all 128 ballots out of the window loop and spills them)
                float thr_w[WNB];
#pragma unroll
                for (int bn = 0; bn < WNB; ++bn) {
                    thr_w[bn] = sp_thr[bn];
                    asm volatile("" : "+v"(thr_w[bn]));
                }
                int loc_w = (rbase << 8) | cbase;  // an entry's place in the tile: this plus a constant (cbase + bn * 32 < 256)
                asm volatile("" : "+v"(loc_w));
#pragma unroll
                for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
                    for (int bn = 0; bn < WNB; ++bn) {
                        if (a.sq_entries) {
                            // top-k screen: candidates are a few per million elements, so the lane's largest value of the
                            // 32x32 block is tested first (15 v_max) and 16 compare-and-branch pairs are skipped; at the
                            // lens pass's densities nearly every block has a candidate in some lane and the test is left out
                            float mx = acc[bm][bn][0];
#pragma unroll
                            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[bm][bn][r]);
                            if (!__builtin_amdgcn_ballot_w64(mx > thr_w[bn])) continue;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[bm][bn][r];
                            const bool p = v > thr_w[bn];
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
                            // (marked unlikely: hipcc then lays the no-candidate case out as the fall-through -- a taken
                            // branch per element costs -- and moves the staging out of line; emit 680-690 -> 668 us)
                            if (__builtin_expect(m != 0, 0)) {
                                if (p) {
                                    const unsigned q = (unsigned)(cnt - win) +
                                                       __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                    if (q < (unsigned)WQ) {
                                        qv[q] = v;
                                        ql[q] = loc_w + (((bm * 32 + (r & 3) + 8 * (r >> 2)) << 8) | (bn * 32));
                                    }
                                }
                                cnt += __builtin_popcountll(m);
                            }
                        }
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int staged = cnt - win < WQ ? cnt - win : WQ;
                for (int e0 = 0; e0 < staged; e0 += 64) {
                    const int e = e0 + lane;
                    bool live = e < staged;
                    const int loc = live ? ql[e] : 0;
                    float v = live ? qv[e] : 0.f;
                    const int row = loc >> 8, col = loc & 255;
                    live = live && m0 + row < a.M;  // rows past M are copies of the last row
                    if (BF16 && has_rb && live && (m0 + row + 1 < a.M || !(a.M & 1))) {
                        // v carries the margin of the call's LONGEST row; this row's own margin is smaller by
                        // (A_max - A_row) wnorm[n] + (B_max - B_row) wdiff[n] (row_bound_kernel): re-test with it
                        const float2 rb = sm.rowb[par][row], mx = sm.rbmax;
                        v -= (mx.x - rb.x) * sm.coln[0][col] + (mx.y - rb.y) * sm.coln[1][col];
                        live = v > sm.coln[2][col];
                    }
                    if (a.sq_entries) {  // top-k screen: the workgroup's own candidate queue, one reservation per wave
                        const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
                        if (lm) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(a.sq_counters + wg, __builtin_popcountll(lm));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (live) {
                                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
                                if (pos < a.sq_cap) a.sq_entries[(size_t)wg * a.sq_cap + pos] = ((unsigned)(m0 + row) << 8) | (unsigned)col;
                                else *a.sq_overflow = 1;
                            }
                        }
                    } else if (live) {
                        const long long grow = m0 + row;
                        const int pos = atomicAdd(a.sp_count + grow, 1);
                        if (pos < a.max_active) {
                            a.sp_feat[grow * a.max_active + pos] = n0 + col;
                            a.sp_val[grow * a.max_active + pos] = v;
                        }
                    }
                }
                if (cnt <= win + WQ) break;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the drain's reads before the next window's writes
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // Lane-local pre-check per 32x32 block: the lane's largest value of the block against
            // its feature's k-th best (rows past M are copies of the last real row, so they cannot
            // raise an alarm that the exact check below would not dismiss).  Lanes with a candidate
            // push it into one of the QD slots of its feature; the thread that owns the feature
            // folds its slots into the feature's list.  A lane that finds the slots full keeps its
            // place and the round is repeated, so nothing is dropped whatever the data.  Cost: one
            // barrier per tile when nobody has a candidate, two when somebody has.
            TSC_STAMP(0)
            float thr[WNB];
#pragma unroll
            // columns past N: NaN fails every comparison (an infinite threshold would admit an infinite activation of
            // the clamped copy column, and its owner would then write past the slot's lists)
            for (int bn = 0; bn < WNB; ++bn) thr[bn] = nok[bn] ? sm.thr[cbase + bn * 32] : __builtin_nanf("");
            unsigned hot = 0;  // bit bm * WNB + bn: the block holds a candidate of this lane
#pragma unroll
            for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
                for (int bn = 0; bn < WNB; ++bn) {
                    float mx = acc[bm][bn][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[bm][bn][r]);
                    hot |= (mx > 0.f && mx >= thr[bn]) ? 1u << (bm * WNB + bn) : 0u;
                }
            int consumed = 0;
            TSC_STAMP(1)
            for (;;) {  // sm.pushed == sm.again == 0 and sm.f_nq[] == 0 on entry
                if (hot) {
                    int ord = 0, blocked = 0;
                    // This branch is cold.  Its compares are pure, so hipcc would evaluate them ahead of the
                    // branch on every tile (and spill the resulting masks): tie their operands to the branch.
                    float thr_c[WNB], zero = 0.f;
                    int lim_c = lim, rbase_c = rbase, cbase_c = cbase;
#pragma unroll
                    for (int bn = 0; bn < WNB; ++bn) {
                        thr_c[bn] = thr[bn];
                        asm volatile("" : "+v"(thr_c[bn]));
                    }
                    asm volatile("" : "+v"(zero), "+v"(lim_c), "+v"(rbase_c), "+v"(cbase_c));
#pragma unroll
                    for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
                        for (int bn = 0; bn < WNB; ++bn)
                            if (hot & (1u << (bm * WNB + bn))) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int c = bm * 32 + (r & 3) + 8 * (r >> 2);
                                    const float v = acc[bm][bn][r];
                                    if (v > zero && v >= thr_c[bn] && c < lim_c) {
                                        if (ord >= consumed && !blocked) {
                                            const int col = cbase_c + bn * 32;
                                            const int pos = atomicAdd(&sm.f_nq[col], 1);
                                            if (pos < QD) {
                                                sm.q_val[col * QD + pos] = v;
                                                sm.q_row[col * QD + pos] = (unsigned char)(rbase_c + c);
                                                sm.pushed = 1;
                                                ++consumed;
                                            } else {
                                                sm.again = 1;
                                                blocked = 1;
                                            }
                                        }
                                        ++ord;
                                    }
                                }
                            }
                }
                TSC_STAMP(2)
                __syncthreads();
                TSC_STAMP(3)
                if (!sm.pushed) break;
                const int again = sm.again;
                const int nq = tid < BN ? (sm.f_nq[tid] < QD ? sm.f_nq[tid] : QD) : 0;
                if (nq > 0) {  // features past N never receive candidates (thr = NaN)
                    const int k = a.topk;
                    const long long off = ((long long)slot * a.N + n0 + tid) * k;
                    float *lv = a.tk_vals + off;
                    long long *lid = a.tk_idx + off;
                    int cnt = sm.f_cnt[tid], wpos = sm.f_worst_pos[tid];
                    float wv = sm.f_worst_v[tid];
                    long long wid = sm.f_worst_id[tid];
                    for (int e = 0; e < nq; ++e) {
                        const float v = sm.q_val[tid * QD + e];
                        const long long id = a.token_base + m0 + sm.q_row[tid * QD + e];
                        if (cnt < k) {
                            lv[cnt] = v;
                            lid[cnt] = id;
                            if constexpr (LLDS) sm.lv[tid * KL + cnt] = v;
                            if (cnt == 0 || ranks_below(v, id, wv, wid)) {
                                wv = v; wid = id; wpos = cnt;
                            }
                            ++cnt;
                            continue;
                        }
                        if constexpr (LLDS) {
                            // wid < 0: the id of the lowest-ranked entry has not been needed yet
                            if (v < wv) continue;
                            if (v == wv) {
                                if (wid < 0) wid = lid[wpos];
                                if (id > wid) continue;
                            }
                            const int p = wpos;
                            lv[p] = v;
                            lid[p] = id;
                            sm.lv[tid * KL + p] = v;
                            // the new lowest value, from LDS; among equal values the largest id ranks
                            // lowest, which takes the ids from HBM -- only when there is such a tie
                            float mv = v;
                            int mp = p, ties = 1;
                            for (int i = 0; i < k; ++i) {
                                const float x = sm.lv[tid * KL + i];
                                if (i == p) continue;
                                if (x < mv) { mv = x; mp = i; ties = 1; }
                                else if (x == mv) ++ties;
                            }
                            wv = mv; wpos = mp;
                            wid = mp == p ? id : -1;
                            if (ties > 1) {
                                long long best = -1;
                                for (int i = 0; i < k; ++i)
                                    if (sm.lv[tid * KL + i] == mv) {
                                        const long long t = i == p ? id : lid[i];
                                        if (t > best) { best = t; wpos = i; }
                                    }
                                wid = best;
                            }
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
                    sm.thr[tid] = fmaxf(cnt == k ? wv : 0.f, sm.f_floor[tid]);
                }
                if (tid < BN) sm.f_nq[tid] = 0;
                TSC_STAMP(4)
                __syncthreads();  // lists and thresholds updated; everyone has read the round flags
                TSC_STAMP(5)
                if (tid == 0) sm.pushed = sm.again = 0;
                if (!again) break;  // the resets are ordered before the next tile's pushes by the k-loop barriers
                // thr[] is deliberately not refreshed: `consumed` counts this lane's candidates in
                // the order of one fixed threshold; the owner's exact check dismisses stale ones
                __syncthreads();
            }
        }
#undef TSC_FOR_ACC
    };

#define TSC_MFMA_S(FR, s)                                                                        \
    _Pragma("unroll") for (int bm = 0; bm < WMB; ++bm) _Pragma("unroll") for (int bn = 0; bn < WNB; ++bn) \
        acc[bm][bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(FR.af[bm][s], FR.bf[bn][s], acc[bm][bn], 0, 0, 0);
#define TSC_MFMA(FR, S0, S1) _Pragma("unroll") for (int s = S0; s < S1; ++s) { TSC_MFMA_S(FR, s) }
#define TSC_MFMA_B(FR, bm, bn)                                                                               \
    acc[bm][bn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FR.af[bm]),             \
                                                          __builtin_bit_cast(bf16x8, FR.bf[bn]), acc[bm][bn], 0, 0, 0);
#define TSC_MFMA_BALL(FR)                                                                                    \
    _Pragma("unroll") for (int bm = 0; bm < WMB; ++bm) _Pragma("unroll") for (int bn = 0; bn < WNB; ++bn) { TSC_MFMA_B(FR, bm, bn) }


    // The workgroup's work is one continuous stream of k-steps over all its token tiles: the
    // prefetch of a tile's first k-step is issued during the previous tile's last one, and the
    // epilogue runs with the next tile's first k-step already staged.
    int parity = 0;
    long long mt = slot;
    if (mt < n_mtiles) {
        set_tile(mt * BM);
        if constexpr (DMA) {
            const char *xt = (const char *)a.x + (size_t)mt * BM * a.K * ES;
#pragma unroll
            for (int j = 0; j < APW + BPW; ++j) dma_piece(j, xt, (const char *)a.Wt, 0, lds_a0, lds_b0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            gload(mt * BM, 0);
            sstore(0, 0);
        }
    }
    if (BF16 && has_rb) {
        if (tid < BN) {
            const int n = n0 + tid < a.N ? n0 + tid : a.N - 1;
            sm.coln[0][tid] = a.screen_wnorm[n];
            sm.coln[1][tid] = a.screen_wnorm[a.N + n];
            sm.coln[2][tid] = emit_threshold(n0 + tid);
        }
        if (tid == 0) sm.rbmax = make_float2(a.screen_bound[0], a.screen_bound[1]);
    }
    __syncthreads();
    Frag F, G;
    if (mt < n_mtiles) ldfrag(0, 0, F);

    for (; mt < n_mtiles; mt += a.m_split) {
        const long long m0 = mt * BM;
        f32x16 acc[WMB][WNB];
#pragma unroll
        for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
            for (int bn = 0; bn < WNB; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bm][bn][r] = be[bn];  // start from the bias: no add in the epilogue

        // The order inside a k-step is pinned with sched_barrier(0): left alone, hipcc sinks every
        // fragment read to just before its first use and hoists the barrier above the MFMAs meant
        // to cover it.
        if constexpr (DMA) {
            // Every instruction a wave issues costs the matrix pipe about as much as it would cost
            // the VALU, so the k-loop is unrolled by two (nk is even: K % 64 == 0): the LDS buffer
            // is then a compile-time constant, every LDS address an immediate offset, and there is
            // no per-step branch.  A k-step computes from buffer PAR and fills buffer PAR ^ 1 with
            // the next k-step -- or, in the tile's last step, with the first k-step of the
            // workgroup's next tile (of this tile again if there is none: nobody reads that).  The
            // pieces are issued one per MFMA k-step over the first two groups (an issue burst
            // would leave the matrix pipe idle) and have the last two groups to land.
            constexpr int PPS = (APW + BPW) / 8;
            const bool has_next = mt + a.m_split < n_mtiles;
            const long long next_m0 = has_next ? (mt + a.m_split) * BM : m0;
            const char *xt = (const char *)a.x + (size_t)m0 * a.K * ES + 128;  // k-step 1 of this tile
            const char *wt = (const char *)a.Wt + 128;
            if (BF16 && has_rb && wave < BM / 128) {
                // the bound pairs of this tile's rows, two rows (16 bytes) per lane, straight into LDS; the first k-step's
                // vmcnt(0) waits for them, the later k-steps' barriers publish them.  The buffer alternates with the tile
                // parity: a slower wave may still drain the previous tile's queue (two tiles back it cannot: the k-loop's
                // barriers lie in between).  Lanes past M - 2 read the last two rows: with an odd M the slot of row M - 1
                // then holds row M - 2's pair, so the drain does not refine that row.
                long long r0 = m0 + wave * 128 + lane * 2;
                if (r0 > a.M - 2) r0 = a.M - 2;
                __builtin_amdgcn_global_load_lds((gptr_t)(a.screen_rows + 2 * r0), (lptr_t)&sm.rowb[ep_tiles & 1][wave * 128], 16, 0, 0);
            }
            auto kstep = [&](auto par_) {
                constexpr int PAR = decltype(par_)::value;
                // keep the tile bases in scalar registers (saddr + 32-bit lane offset addressing);
                // otherwise loop strength reduction turns every piece's address into a 64-bit
                // vector induction variable
                const char *xp = xt, *wp = wt;
                // likewise the piece destinations: one base plus an immediate each, not 16 hoisted constants
                unsigned la = lds_a0, lb = lds_b0;
                asm volatile("" : "+s"(xp), "+s"(wp), "+s"(la), "+s"(lb));
#pragma unroll
                for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(b_voff[j]));  // same, for the hoisted zero-extensions
#pragma unroll
                for (int j = 0; j < APW; ++j) asm volatile("" : "+v"(a_voff[j]));
                if constexpr (BF16) {
                    // 4 MFMAs (32 cycles each) per 16-k group: the pieces go out two per MFMA in the
                    // first group and have the other three to land
                    constexpr int PPM = (APW + BPW) / (WMB * WNB);
                    // A fragment read is issued right AFTER the first MFMA of the group that runs meanwhile, not
                    // before it: hipcc waits lgkmcnt(0) for that group's own (older) fragment, which would
                    // otherwise also wait for the read just issued.
#define TSC_BGROUP(FR, NEXT_READ)                                                                        \
    _Pragma("unroll") for (int bm = 0; bm < WMB; ++bm) _Pragma("unroll") for (int bn = 0; bn < WNB; ++bn) { \
        TSC_PIN TSC_MFMA_B(FR, bm, bn)                                                                   \
        if (bm == 0 && bn == 0) { TSC_PIN NEXT_READ; }                                                   \
    }
#pragma unroll
                    for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
                        for (int bn = 0; bn < WNB; ++bn) {
                            TSC_PIN
#pragma unroll
                            for (int jj = 0; jj < PPM; ++jj) dma_piece((bm * WNB + bn) * PPM + jj, xp, wp, PAR ^ 1, la, lb);
                            TSC_PIN TSC_MFMA_B(F, bm, bn)
                            if (bm == 0 && bn == 0) { TSC_PIN ldfrag(PAR, 1, G); }
                        }
                    TSC_BGROUP(G, ldfrag(PAR, 2, F))
                    TSC_BGROUP(F, ldfrag(PAR, 3, G))
#pragma unroll
                    for (int bn = 0; bn < WNB; ++bn) { TSC_PIN TSC_MFMA_B(G, 0, bn) }
                    TSC_PIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    TSC_PIN ldfrag(PAR ^ 1, 0, F);
#pragma unroll
                    for (int bm = 1; bm < WMB; ++bm)
#pragma unroll
                        for (int bn = 0; bn < WNB; ++bn) { TSC_PIN TSC_MFMA_B(G, bm, bn) }
                    TSC_PIN
#undef TSC_BGROUP
                } else {
                    // (fragment reads one MFMA step late: see the bf16 branch)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        TSC_PIN
#pragma unroll
                        for (int jj = 0; jj < PPS; ++jj) dma_piece(s * PPS + jj, xp, wp, PAR ^ 1, la, lb);
                        TSC_PIN TSC_MFMA_S(F, s)
                        if (s == 0) { TSC_PIN ldfrag(PAR, 1, G); }
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        TSC_PIN
#pragma unroll
                        for (int jj = 0; jj < PPS; ++jj) dma_piece((4 + s) * PPS + jj, xp, wp, PAR ^ 1, la, lb);
                        TSC_PIN TSC_MFMA_S(G, s)
                        if (s == 0) { TSC_PIN ldfrag(PAR, 2, F); }
                    }
                    TSC_PIN TSC_MFMA_S(F, 0)
                    TSC_PIN ldfrag(PAR, 3, G);
                    TSC_PIN TSC_MFMA(F, 1, 4)
                    TSC_PIN TSC_MFMA(G, 0, 2)
                    TSC_PIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed
                    __syncthreads();
                    TSC_PIN ldfrag(PAR ^ 1, 0, F);
                    TSC_PIN TSC_MFMA(G, 2, 4)
                    TSC_PIN
                }
            };
            for (int kt = 0; kt < nk; kt += 2) {
                kstep(std::integral_constant<int, 0>());
                if (kt + 2 == nk) {
                    set_tile(next_m0);
                    xt = (const char *)a.x + (size_t)next_m0 * a.K * ES;
                    wt = (const char *)a.Wt;
                } else {
                    xt += 128;
                    wt += 128;
                }
                kstep(std::integral_constant<int, 1>());
                xt += 128;
                wt += 128;
            }
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                // what to prefetch: the next k-step of this tile, or the first of the next tile
                const bool last_k = kt + 1 == nk;
                const long long next_m0 = last_k ? (mt + a.m_split) * BM : m0;
                const int next_k0 = last_k ? 0 : (kt + 1) * BK;
                const bool more = !last_k || mt + a.m_split < n_mtiles;
                if (last_k && more) set_tile(next_m0);
                if (more) gload(next_m0, next_k0);
                TSC_PIN ldfrag(parity, 1, G);
                TSC_PIN TSC_MFMA(F, 0, 4)
                TSC_PIN ldfrag(parity, 2, F);
                TSC_PIN TSC_MFMA(G, 0, 4)
                TSC_PIN ldfrag(parity, 3, G);
                TSC_PIN TSC_MFMA(F, 0, 4)
                // stage the prefetched k-step into the other buffer (free since the last barrier),
                // cover the LDS write latency and the barrier with the first half of the last group,
                // and the first fragment read of the new buffer with the second half
                TSC_PIN if (more) sstore(parity ^ 1, next_k0);
                TSC_PIN TSC_MFMA(G, 0, 2)
                TSC_PIN __syncthreads();
                TSC_PIN if (more) ldfrag(parity ^ 1, 0, F);
                TSC_PIN TSC_MFMA(G, 2, 4)
                TSC_PIN
                parity ^= 1;
            }
        }
#undef TSC_MFMA_S
#undef TSC_MFMA_B
#undef TSC_MFMA_BALL
#undef TSC_MFMA

#ifndef TSC_ABL_NOEPI  // diagnosis builds only (no output): what the launch costs without its epilogues
        epilogue(acc, m0);
#elif defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int bm = 0; bm < WMB; ++bm)
#pragma unroll
            for (int bn = 0; bn < WNB; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[bm][bn][r]));  // every accumulator stays live
#endif
    }
#ifdef TSC_STAMPS
    if (MODE == TSC_MODE_TOPK && (tid == 0 || tid == 192)) {
        const int o = tid == 0 ? 0 : 8;
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[o + i], st_acc[i]);
    }
#endif
}

template <class C, int MODE, bool KALIGNED, bool DMA, bool LLDS, bool BF16>
__global__ __launch_bounds__(C::NT, C::OCC) void encode_f32_kernel(const EncodeArgs a) {
    encode_body<C, MODE, KALIGNED, DMA, LLDS, BF16>(a, blockIdx.x);
}

// The bf16 SPARSE emit (the screen of the lens pass) of SEVERAL dictionaries on the same number of rows in one launch:
// at CircuitLens sizes one dictionary's emit is 768 single-tile workgroups on 512 resident slots -- 1.5 rounds, each
// workgroup a chain of exposed latencies -- and emits on different streams do not pack into each other's idle half
// round; one grid over all of them does.  What differs per dictionary travels in the kernel arguments.
constexpr int GROUP_MAX = 32;
struct EmitGroupArgs {
    struct One {
        const void *x, *Wt;                  // bf16(x - b_dec) [M][K] (the handle's staging), bf16(W_enc) transposed [N][K]
        const float *bias, *wnorm, *bound, *rows;  // b_enc, column norms, the row bounds of this call (maxima; per row or null)
        int *sp_feat, *sp_count;
        float *sp_val;
        int N, n_tiles_n, wg_end;            // wg_end: first workgroup of the NEXT problem
    } d[GROUP_MAX];
    long long M;
    int K, max_active, m_split, n;
};
template <class C>
__global__ __launch_bounds__(C::NT, C::OCC) void encode_group_kernel(const EmitGroupArgs g) {
    int p = 0;  // uniform: the problem's fields are scalar loads
    while (p + 1 < g.n && (int)blockIdx.x >= g.d[p].wg_end) ++p;
    const int wg0 = p ? g.d[p - 1].wg_end : 0;
    EncodeArgs a = {};
    a.x = (const float *)g.d[p].x; a.Wt = (const float *)g.d[p].Wt; a.bias = g.d[p].bias;
    a.screen_wnorm = g.d[p].wnorm; a.screen_bound = g.d[p].bound; a.screen_rows = g.d[p].rows;
    a.sp_feat = g.d[p].sp_feat; a.sp_val = g.d[p].sp_val; a.sp_count = g.d[p].sp_count;
    a.N = g.d[p].N; a.n_tiles_n = g.d[p].n_tiles_n;
    a.M = g.M; a.K = g.K; a.max_active = g.max_active; a.m_split = g.m_split;
    encode_body<C, TSC_MODE_SPARSE, true, true, false, true>(a, (int)blockIdx.x - wg0);
}

// bias[n] = b_enc[n] - sum_k b_dec[k] * W[k,n], summed in fp64 and rounded once: lets the DMA kernel
// contract the raw x (relu((x - b_dec) @ W + b_enc) == relu(x @ W + bias) up to fp32 rounding).
__global__ __launch_bounds__(256) void fold_bias_kernel(const float *__restrict__ W, const float *__restrict__ b_enc,
                                                        const float *__restrict__ b_dec, int K, int N,
                                                        float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)b_dec[k] * (double)W[(size_t)k * N + n];
    out[n] = (float)((double)b_enc[n] - s);
}

// Variant selection.  Each knob is "auto" (0) unless forced, either by an environment variable read once
// (TSC_F32_TILE / TSC_BF16_TILE = 128|256, TSC_F32_STAGING = reg, TSC_F32_LISTS = hbm) or at run time through
// tsc_set_tuning (the tests use it to cover every variant in one process).
struct Tuning {
    int f32_tile = 0, bf16_tile = 0;  // 0 auto, 128, 256
    int staging = 0;                  // 0 auto (LDS-DMA when possible), 1 register staging
    int lists = 0;                    // 0 auto (LDS mirror when possible), 1 lists in HBM only
    Tuning() {
        const char *e;
        if ((e = getenv("TSC_F32_TILE"))) f32_tile = atoi(e);
        if ((e = getenv("TSC_BF16_TILE"))) bf16_tile = atoi(e);
        if ((e = getenv("TSC_F32_STAGING")) && e[0] == 'r') staging = 1;
        if ((e = getenv("TSC_F32_LISTS")) && e[0] == 'h') lists = 1;
    }
};
Tuning &tuning() {
    static Tuning t;
    return t;
}
bool use_dma() { return tuning().staging == 0; }
bool use_lds_lists() { return tuning().lists == 0; }

// Auto geometry.  256x256 (8 waves, one workgroup per CU) reads each operand byte from L2 half as often and
// issues fewer non-MFMA instructions per MFMA: fp32 dense 148 vs 146 TFLOP/s, bf16 top-k 965 vs 915 (1000 vs
// 760 at d_sae = 131072); for the fp32 fused top-k the two are equally fast, and the wide tile cuts the L2 miss
// traffic from 9.7 to 1.9 GB per launch.  It is used whenever its grid still covers the chip; short inputs
// and skinny dictionaries take 128x128.
bool pick_large(int mode, const EncodeArgs &a) {
    const int forced = a.bf16 ? tuning().bf16_tile : tuning().f32_tile;
    if (forced == 128) return false;
    if (forced == 256) return true;
    if (a.N <= CfgSmall::BN) return false;  // a skinny dictionary (tsc_hip.FeatureProjector) would waste the wide tile
    const long long tiles = (long long)((a.N + CfgLarge::BN - 1) / CfgLarge::BN) * ((a.M + CfgLarge::BM - 1) / CfgLarge::BM);
    if (mode == TSC_MODE_TOPK || (mode == TSC_MODE_SPARSE && a.sq_entries))  // the slots are fixed by the state: every one of them must get a wide tile
        return (long long)((a.N + CfgLarge::BN - 1) / CfgLarge::BN) * a.m_split >= 256 &&
               (a.M + CfgLarge::BM - 1) / CfgLarge::BM >= a.m_split;
    return tiles >= 256;
}

// number of token-tile groups per column tile such that the grid is a whole number of waves of
// resident workgroups (every workgroup does the same amount of work, so a ragged last wave of
// workgroups is pure idle time).  Eight or more groups are tried first: the workgroups resident on an XCD are
// consecutive in (column tile, group), so with s groups its 32 CUs hold 32/s column tiles -- at s = 8 their W_enc
// slabs (4 x 393 KB bf16) stay in the XCD's 4-MiB L2 and every slab read after the first is an L2 hit, where one group
// per column tile has 32 slabs (12.6 MB) thrashing it and re-reads every slab from the Infinity Cache per token tile
// (measured, bf16 fused top-k at d_sae 131072 x 16384 tokens: 1.07 PFLOP/s with 1 group, 1.12 / 1.17 / 1.18 with
// 2 / 4 / 8, 1.12 with 16; fp32 145.6 -> 147.2 TFLOP/s; tools/exp_slots.py).
int balanced_split(int n_tiles_n, long long n_tiles_m, int resident, int cap) {
    if (cap > n_tiles_m) cap = (int)(n_tiles_m < 1 ? 1 : n_tiles_m);
    for (int s = 8; s <= cap && s <= 16; ++s)
        if (((long long)n_tiles_n * s) % resident == 0) return s;
    for (int s = 1; s <= cap; ++s)
        if (((long long)n_tiles_n * s) % resident == 0) return s;
    int s = (2 * resident + n_tiles_n - 1) / n_tiles_n;
    if (s < 1) s = 1;
    return s > cap ? cap : s;
}

template <class C, int MODE>
int launch(EncodeArgs a, hipStream_t stream) {
    a.n_tiles_n = (a.N + C::BN - 1) / C::BN;
    if (a.m_split <= 0)
        a.m_split = balanced_split(a.n_tiles_n, (a.M + C::BM - 1) / C::BM, 256 * C::OCC, 4096);
    const long long grid = (long long)a.n_tiles_n * a.m_split;
    if (grid <= 0 || grid > 0x7fffffffLL) return tsc_fail(TSC_ERR_ARG, "encode grid %lld out of range", grid);
    // the LDS-DMA kernels address the transposed operand with 32-bit lane offsets
    const bool fits32 = (unsigned long long)a.N * a.K * (a.bf16 ? 2 : 4) + 8192ull < (1ull << 32);
    const bool dma = a.K % (2 * BK) == 0 && a.bias && a.Wt && fits32 && use_dma();
    const dim3 g((unsigned)grid), b(C::NT);
    // the mirror of the lists' values must fit in LDS next to the staging buffers of every resident workgroup
    constexpr bool TK = MODE == TSC_MODE_TOPK && sizeof(Smem<C, TSC_MODE_TOPK, true, true>) * C::OCC <= 160 * 1024;
    const bool lists = TK && a.topk <= KL && use_lds_lists();
    if (a.bf16) {
        if (a.K % 128 != 0 || !a.bias || !a.Wt) return tsc_fail(TSC_ERR_UNSUPPORTED, "bf16 encode needs d_in %% 128 == 0");
        if (!fits32) return tsc_fail(TSC_ERR_UNSUPPORTED, "bf16 encode needs d_in * d_sae * 2 < 4 GiB");
        if (lists)
            hipLaunchKernelGGL((encode_f32_kernel<C, MODE, true, true, TK, true>), g, b, 0, stream, a);
        else
            hipLaunchKernelGGL((encode_f32_kernel<C, MODE, true, true, false, true>), g, b, 0, stream, a);
    } else if (dma && lists)
        hipLaunchKernelGGL((encode_f32_kernel<C, MODE, true, true, TK, false>), g, b, 0, stream, a);
    else if (dma)
        hipLaunchKernelGGL((encode_f32_kernel<C, MODE, true, true, false, false>), g, b, 0, stream, a);
    else if (a.K % BK == 0)
        hipLaunchKernelGGL((encode_f32_kernel<C, MODE, true, false, false, false>), g, b, 0, stream, a);
    else
        hipLaunchKernelGGL((encode_f32_kernel<C, MODE, false, false, false, false>), g, b, 0, stream, a);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

template <class C>
int dispatch(int mode, const EncodeArgs &a, hipStream_t stream) {
    switch (mode) {
        case TSC_MODE_DENSE: return launch<C, TSC_MODE_DENSE>(a, stream);
        case TSC_MODE_TOPK: return launch<C, TSC_MODE_TOPK>(a, stream);
        case TSC_MODE_SPARSE: return launch<C, TSC_MODE_SPARSE>(a, stream);
    }
    return tsc_fail(TSC_ERR_ARG, "unknown encode mode %d", mode);
}

}  // namespace

#ifdef TSC_STAMPS
extern "C" int tsc_debug_read_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int tsc_launch_fold_bias(const float *W, const float *b_enc, const float *b_dec, int K, int N, float *out,
                         hipStream_t stream) {
    hipLaunchKernelGGL(fold_bias_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, W, b_enc, b_dec, K, N, out);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

// a.m_split <= 0 asks for the balanced split; a.n_tiles_n is filled in here
int tsc_launch_encode_f32(int mode, const EncodeArgs &a, hipStream_t stream) {
    if (a.M <= 0) return TSC_OK;
    return pick_large(mode, a) ? dispatch<CfgLarge>(mode, a, stream) : dispatch<CfgSmall>(mode, a, stream);
}

// grouped bf16 screen emit: `n` problems that share M, K and max_active; p[i] carries the per-dictionary pointers
int tsc_launch_emit_group(int n, const EmitGroupProblem *p, long long M, int K, int max_active, hipStream_t stream) {
    if (n <= 0 || M <= 0) return TSC_OK;
    if (n > GROUP_MAX) return tsc_fail(TSC_ERR_UNSUPPORTED, "at most %d dictionaries per grouped emit (got %d)", GROUP_MAX, n);
    if (K % 128 != 0) return tsc_fail(TSC_ERR_UNSUPPORTED, "grouped bf16 emit needs d_in %% 128 == 0");
    // the wide tile: a grouped grid has no round quantisation (ungrouped, 192 wide tiles on 256 CUs lose to 768 small
    // ones on 512 slots), and a wide-tile k-step does four times the work of a small one for about the same latency
    typedef CfgLarge C;
    EmitGroupArgs g;
    memset(&g, 0, sizeof(g));
    const long long n_mtiles = (M + C::BM - 1) / C::BM;
    g.M = M; g.K = K; g.max_active = max_active; g.n = n;
    // one workgroup per tile up to 8 token tiles: a workgroup that walks the token tiles of its column tile re-reads
    // the W tile from the Infinity Cache each time (measured at 512 / 1024 rows: 745 / 1368 us against 683 / 1165)
    g.m_split = (int)(n_mtiles < 8 ? n_mtiles : 8);
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        if ((unsigned long long)p[i].N * K * 2 + 8192ull >= (1ull << 32))
            return tsc_fail(TSC_ERR_UNSUPPORTED, "bf16 encode needs d_in * d_sae * 2 < 4 GiB");
        g.d[i].x = p[i].x_bf16; g.d[i].Wt = p[i].W_bf16; g.d[i].bias = p[i].b_enc; g.d[i].wnorm = p[i].wnorm;
        g.d[i].bound = p[i].bound; g.d[i].rows = p[i].rows; g.d[i].sp_feat = p[i].feat; g.d[i].sp_val = p[i].val; g.d[i].sp_count = p[i].count;
        g.d[i].N = p[i].N; g.d[i].n_tiles_n = (p[i].N + C::BN - 1) / C::BN;
        total += (long long)g.d[i].n_tiles_n * g.m_split;
        g.d[i].wg_end = (int)total;
    }
    if (total > 0x7fffffffLL) return tsc_fail(TSC_ERR_UNSUPPORTED, "grouped emit grid out of range");
    hipLaunchKernelGGL((encode_group_kernel<C>), dim3((unsigned)total), dim3(C::NT), 0, stream, g);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_encode_tile_bn(int mode, const EncodeArgs &a) { return pick_large(mode, a) ? CfgLarge::BN : CfgSmall::BN; }

// slot count that makes the fused top-k grid a whole number of waves of resident workgroups
int tsc_encode_recommended_slots(int d_sae, int bf16) {
    const int forced = bf16 ? tuning().bf16_tile : tuning().f32_tile;
    const bool large = forced != 128;
    if (large) return balanced_split((d_sae + CfgLarge::BN - 1) / CfgLarge::BN, 1 << 30, 256 * CfgLarge::OCC, 16);
    return balanced_split((d_sae + CfgSmall::BN - 1) / CfgSmall::BN, 1 << 30, 256 * CfgSmall::OCC, 16);
}

int tsc_encode_set_tuning(const char *key, int value) {
    Tuning &t = tuning();
    const std::string k(key ? key : "");
    if (k == "f32_tile" && (value == 0 || value == 128 || value == 256)) t.f32_tile = value;
    else if (k == "bf16_tile" && (value == 0 || value == 128 || value == 256)) t.bf16_tile = value;
    else if (k == "staging" && (value == 0 || value == 1)) t.staging = value;
    else if (k == "lists" && (value == 0 || value == 1)) t.lists = value;
    else return tsc_fail(TSC_ERR_ARG, "unknown tuning knob or value: %s = %d", k.c_str(), value);
    return TSC_OK;
}
