// Sparse-feature kernels of the CircuitLens contribution pass.  All are HBM-bound gathers of
// coalesced W_dec rows (d_out * 4 = 3072 bytes each for GPT-2 small); none is reshaped into a GEMM.
//
//   sparse_sort    : per token, order the (feature, value) pairs emitted by the SPARSE encode
//                    epilogue by value descending -- `topk(k=count)` of reference
//                    src/circuit_lens.py:612-614 / 647-649.
//   decode_rows    : rows[i] = val[i] * W_dec[feat[i]]          (circuit_lens.py:616, 651-653)
//   decode_sparse  : recon[m] = bias + sum_i val * W_dec[feat]  (mlp_transcoder.py:128-147, z_sae.py:62)
//   contrib        : scores[m,i,d] = val * dot(W_dec[feat], dirs[d])  (circuit_lens.py:762-766,
//                    955-957, 1151-1164 restricted to active features)
#include <cstring>

#include "tsc_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SORT_MAX = 8192;
constexpr int DIR_MAX = 8;

// grid = M, block = 256.  Short lists (the normal case, c ~ 64): rank sort in LDS, rank(e) = #entries that precede
// e -- c^2 / 256 compares per thread and one pass.  Long lists (c > RANK_SORT_MAX, up to the 8192 the emit can hold):
// bitonic sort of 64-bit keys (~value bits : feature id), ascending = value descending then feature ascending
// (activations are positive floats, whose bit patterns order like the values) -- c log^2 c instead of c^2.
constexpr int RANK_SORT_MAX = 512;
__global__ __launch_bounds__(256) void sparse_sort_kernel(int *feat, float *val, const int *count,
                                                         int max_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long long m = blockIdx.x;
    int c = count[m];
    if (c > max_active) c = max_active;
    int *frow = feat + m * max_active;
    float *vrow = val + m * max_active;
    if (c <= RANK_SORT_MAX) {
        float *sv = (float *)smem_raw;
        int *sf = (int *)(smem_raw + sizeof(float) * RANK_SORT_MAX);
        for (int e = threadIdx.x; e < c; e += blockDim.x) {
            sv[e] = vrow[e];
            sf[e] = frow[e];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < c; e += blockDim.x) {
            const float v = sv[e];
            const int f = sf[e];
            int rank = 0;
            for (int j = 0; j < c; ++j) {
                const float w = sv[j];
                rank += (w > v || (w == v && sf[j] < f)) ? 1 : 0;
            }
            vrow[rank] = v;
            frow[rank] = f;
        }
    } else {
        unsigned long long *key = (unsigned long long *)smem_raw;
        int p2 = RANK_SORT_MAX * 2;
        while (p2 < c) p2 <<= 1;
        for (int e = threadIdx.x; e < p2; e += blockDim.x)
            key[e] = e < c ? ((unsigned long long)(~__float_as_uint(vrow[e])) << 32) | (unsigned)frow[e] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= p2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < p2; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = key[i], b = key[l];
                        if ((a > b) == ((i & k) == 0)) { key[i] = b; key[l] = a; }
                    }
                }
                __syncthreads();
            }
        for (int e = threadIdx.x; e < c; e += blockDim.x) {
            vrow[e] = __uint_as_float(~(unsigned)(key[e] >> 32));
            frow[e] = (int)(unsigned)key[e];
        }
    }
    // zero the unused tail so the buffers are fully defined
    for (int e = c + threadIdx.x; e < max_active; e += blockDim.x) {
        vrow[e] = 0.f;
        frow[e] = -1;
    }
}

// Dense activations -> per-row (feature, value) lists: one wavefront per row walks the row in coalesced 1-KiB steps
// (eight in flight), positives are compacted with ballots -- list position = positives before the element in the order
// (step, component, lane), no atomics -- so the order, and with it the summation order of a later decode, is fixed.
// count[m] is the true number of positives (entries beyond max_active are dropped: the caller checks); a row that holds a
// negative or NaN entry (not the output of a ReLU) reports -1 - that number: NEGATIVE, so that a decode queued before the
// caller has looked reads none of its list (every consumer clamps its loop at count) instead of unwritten entries.
__global__ __launch_bounds__(256) void dense_to_sparse_kernel(const float *__restrict__ acts, long long M, int N, int max_active,
                                                             int *__restrict__ feat, float *__restrict__ val, int *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const long long m = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (m >= M) return;
    const f32x4 *row = (const f32x4 *)(acts + m * N);
    int *frow = feat + m * max_active;
    float *vrow = val + m * max_active;
    const int chunks = N >> 2;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
    bool odd = false;
    for (int c0 = 0; c0 < chunks; c0 += 64 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ch = c0 + 64 * u + lane;
            v[u] = ch < chunks ? __builtin_nontemporal_load(row + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = v[u][j] > 0.f;
                odd |= !(v[u][j] >= 0.f);
                const unsigned long long mask = __ballot(hit);
                if (mask) {  // wave-uniform
                    if (hit) {
                        const int pos = base + __popcll(mask & below);
                        if (pos < max_active) {
                            frow[pos] = (c0 + 64 * u + lane) * 4 + j;
                            vrow[pos] = v[u][j];
                        }
                    }
                    base += __popcll(mask);
                }
            }
    }
    if (__ballot(odd)) base = -1 - base;
    if (lane == 0) count[m] = base;
}

// one wavefront per row; each lane moves float4 chunks
__global__ __launch_bounds__(256) void decode_rows_kernel(const float *W_dec, int d_sae, int d_out, const int *feat,
                                                         const float *val, long long n, float *rows) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = d_out >> 2;
    for (long long i = wave; i < n; i += n_waves) {
        const int f = feat[i];
        const float v = val[i];
        f32x4 *dst = (f32x4 *)(rows + i * d_out);
        if (f < 0 || f >= d_sae) {  // the sparse emit pads its rows with feature -1: such an entry decodes to zero
            for (int c = lane; c < chunks; c += 64) dst[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const f32x4 *src = (const f32x4 *)(W_dec + (long long)f * d_out);
        for (int c = lane; c < chunks; c += 64) dst[c] = src[c] * v;
    }
}

// grid = M; each thread owns float4 column chunks and walks the token's active list, U rows in flight
// (U = 8 for short inputs, where one workgroup per token leaves the pass latency-bound; U = 1 for long ones,
// where occupancy hides the latency and the registers are better spent on more resident workgroups)
template <int U>
__global__ __launch_bounds__(256) void decode_sparse_kernel(const float *W_dec, const float *bias, int d_out,
                                                           const int *feat, const float *val,
                                                           const int *count, int max_active, float *recon) {
    __shared__ int sf[256 + 8];
    __shared__ float sv[256 + 8];
    const long long m = blockIdx.x;
    int c = count[m];
    if (c > max_active) c = max_active;
    const int chunks = d_out >> 2;
    // up to 4 chunks per thread (d_out <= 4096)
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < c; base += 256) {
        const int nb = c - base < 256 ? c - base : 256;
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            sf[threadIdx.x] = feat[m * max_active + base + threadIdx.x];
            sv[threadIdx.x] = val[m * max_active + base + threadIdx.x];
        }
        if (threadIdx.x < 8) {  // padding of the last group of eight
            sf[nb + threadIdx.x] = 0;
            sv[nb + threadIdx.x] = 0.f;
        }
        __syncthreads();
        // (entries past nb are staged as feature 0 with value 0)
        for (int e0 = 0; e0 < nb; e0 += U) {
            f32x4 w[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 *src = (const f32x4 *)(W_dec + (long long)sf[e0 + u] * d_out);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = threadIdx.x + 256 * j;
                    if (ch < chunks) w[u][j] = src[ch];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float v = sv[e0 + u];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = threadIdx.x + 256 * j;
                    if (ch < chunks) acc[j] += w[u][j] * v;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = threadIdx.x + 256 * j;
        if (ch < chunks) {
            f32x4 b = bias ? ((const f32x4 *)bias)[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
            ((f32x4 *)(recon + m * d_out))[ch] = acc[j] + b;
        }
    }
}

// Workgroups walk tokens; the directions are staged in LDS once per workgroup.  A quarter wave (16 lanes)
// takes one active feature: its W_dec row is read as 16-lane x 16-byte runs with all of a lane's loads in
// flight at once (12 for d_out = 768), dotted against every direction, and reduced over 16 lanes.  Only the
// token's `count` live entries are visited; the tail of its score row is zero-filled.
template <int NDIR>
__global__ __launch_bounds__(256) void contrib_kernel(const float *W_dec, int d_out, const int *feat,
                                                     const float *val, const int *count, long long M,
                                                     int max_active, const float *dirs, float *scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *sd = (f32x4 *)smem_raw;  // [NDIR][d_out/4]
    const int chunks = d_out >> 2;
    for (int i = threadIdx.x; i < NDIR * chunks; i += blockDim.x) sd[i] = ((const f32x4 *)dirs)[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;  // 16 groups of 16 lanes
    for (long long m = blockIdx.x; m < M; m += gridDim.x) {
        int c = count[m];
        if (c > max_active) c = max_active;
        const long long base = m * max_active;
        for (int e = grp; e < c; e += 16) {
            const float v = val[base + e];
            const f32x4 *src = (const f32x4 *)(W_dec + (long long)feat[base + e] * d_out);
            float part[NDIR];
#pragma unroll
            for (int d = 0; d < NDIR; ++d) part[d] = 0.f;
            for (int ch0 = 0; ch0 < chunks; ch0 += 16 * 12) {
                f32x4 w[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int ch = ch0 + l16 + 16 * j;
                    w[j] = ch < chunks ? src[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int ch = ch0 + l16 + 16 * j;
                    if (ch < chunks) {
#pragma unroll
                        for (int d = 0; d < NDIR; ++d) {
                            const f32x4 q = sd[d * chunks + ch];
                            part[d] += w[j].x * q.x + w[j].y * q.y + w[j].z * q.z + w[j].w * q.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < NDIR; ++d)
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) part[d] += __shfl_xor(part[d], off);
            if (l16 < NDIR) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < NDIR; ++d) s = l16 == d ? part[d] : s;
                scores[(base + e) * NDIR + l16] = s * v;
            }
        }
        for (int i = c * NDIR + threadIdx.x; i < max_active * NDIR; i += blockDim.x) scores[base * NDIR + i] = 0.f;
    }
}

// contrib_kernel<1> for the rows of SEVERAL dictionaries in one launch, each with its own direction(s): the per-node lens
// calls of the circuit search score ONE position against one direction over all 24 dictionaries -- 24 launches of a
// few microseconds of work each otherwise.  grid = n_dicts * M, one workgroup per (dictionary, row); the (feature,
// value) lists are read from stacked buffers whose dictionaries lie `pitch_rows` rows apart (a window of M rows of a
// larger pass), scores -- and optionally the lists themselves -- are written compactly as [n_dicts, M, max_active].
// Same per-lane arithmetic and reduction order as contrib_kernel<1>: the scores are bit-identical to tsc_contrib_scores.
constexpr int CM_MAX = 32;
struct ContribManyArgs {
    struct One {
        const float *W_dec, *dirs;
    } d[CM_MAX];
    long long M, pitch_rows, dir_stride;
    int d_out, max_active, out_cap, n;  // out_cap: row pitch of the compact outputs (<= max_active)
    const int *feat, *count;
    const float *val;
    float *scores;
    int *feat_out, *count_out;  // optional compact copies [n_dicts, M, max_active] / [n_dicts, M]
};
__global__ __launch_bounds__(256) void contrib_many_kernel(const ContribManyArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *sd = (f32x4 *)smem_raw;  // the row's direction
    const int di = (int)(blockIdx.x / p.M);  // uniform
    const long long m = blockIdx.x - di * p.M;
    const ContribManyArgs::One &o = p.d[di];
    const int chunks = p.d_out >> 2;
    for (int i = threadIdx.x; i < chunks; i += 256) sd[i] = ((const f32x4 *)(o.dirs + m * p.dir_stride))[i];
    __syncthreads();
    const long long src = (di * p.pitch_rows + m) * p.max_active, dst = ((long long)di * p.M + m) * p.out_cap;
    int c = p.count[di * p.pitch_rows + m];
    if (c > p.out_cap) c = p.out_cap;  // (out_cap <= max_active)
    if (p.count_out && threadIdx.x == 0) p.count_out[(long long)di * p.M + m] = c;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    for (int e = grp; e < c; e += 16) {
        const float v = p.val[src + e];
        const f32x4 *w_row = (const f32x4 *)(o.W_dec + (long long)p.feat[src + e] * p.d_out);
        float part = 0.f;
        for (int ch0 = 0; ch0 < chunks; ch0 += 16 * 12) {
            f32x4 w[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ch = ch0 + l16 + 16 * j;
                w[j] = ch < chunks ? w_row[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ch = ch0 + l16 + 16 * j;
                if (ch < chunks) {
                    const f32x4 q = sd[ch];
                    part += w[j].x * q.x + w[j].y * q.y + w[j].z * q.z + w[j].w * q.w;
                }
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (l16 == 0) p.scores[dst + e] = part * v;
    }
    for (int e = c + threadIdx.x; e < p.out_cap; e += 256) p.scores[dst + e] = 0.f;
    if (p.feat_out)
        for (int e = threadIdx.x; e < p.out_cap; e += 256) p.feat_out[dst + e] = e < c ? p.feat[src + e] : -1;
}

// The CircuitLens pass for a batch of positions after the sparse emit, ONE launch per dictionary: per token,
// order the emitted (feature, value) pairs (as sparse_sort_kernel), then read each active feature's W_dec row
// ONCE and use it twice: for the reconstruction (as decode_sparse_kernel) and for the score against the token's
// own direction (as contrib_kernel<1>, same per-lane arithmetic and reduction order, so the scores are bit-identical
// to tsc_contrib_scores).  grid = M, block = 256: 16 quarter waves, each takes the entries e = q, q + 16, ...
constexpr int FIN_MAX_ACTIVE = 2048;  // sorted entries staged in LDS
constexpr int FIN_CPL = 12;           // float4 chunks of a W_dec row per lane of a quarter wave: d_out <= 768
// SCREENED: the emitted pairs are CANDIDATES from the bf16 screen (every feature whose exact activation is positive is
// among them, plus the near misses).  Phase A recomputes each candidate's pre-activation exactly in fp32 --
// dot(x[m], W_enc^T[f]) + bias'[f], a quarter wave per candidate over the contiguous row of the transposed copy --
// and keeps the positive ones; count[m] becomes the true number of active features (unless the candidates
// overflowed the capacity: then it is left above max_active so that the caller retries).
// One row (= one workgroup of 256 threads).  frow / vrow / count_m: the row's emitted pairs and their number; dir_row,
// recon_row, score_row: the row's direction and outputs (score_row == nullptr: no scores; recon_row == nullptr: no
// reconstruction); x_row, W_enc_t, bias_fold, d_in: the exact recompute of a screened emit.
template <bool SCREENED>
__device__ __forceinline__ void finish_row(const float *W_dec, const float *bias, int d_out, int *frow, float *vrow,
                                           int *count_m, int max_active, const float *dir_row, float *recon_row,
                                           float *score_row, const float *x_row, const float *W_enc_t,
                                           const float *bias_fold, int d_in) {
    __shared__ float sv[FIN_MAX_ACTIVE];
    __shared__ int sf[FIN_MAX_ACTIVE];
    __shared__ f32x4 sd[FIN_CPL * 16];        // the token's direction
    __shared__ f32x4 red[4][FIN_CPL * 16];    // per-wave partial reconstructions
    __shared__ int n_live;
    const int tid = threadIdx.x;
    const int c_emitted = *count_m;
    int c = c_emitted;
    if (c > max_active) c = max_active;
    const int chunks = d_out >> 2;
    for (int e = tid; e < c; e += 256) {
        sv[e] = vrow[e];
        sf[e] = frow[e];
    }
    if (score_row)
        for (int i = tid; i < chunks; i += 256) sd[i] = ((const f32x4 *)dir_row)[i];
    if (SCREENED && tid == 0) n_live = 0;
    __syncthreads();
    if constexpr (SCREENED) {
        const int l16 = tid & 15, grp = tid >> 4;
        const int kchunks = d_in >> 2;  // <= FIN_CPL * 16
        f32x4 xr[FIN_CPL];
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) {
            const int ch = l16 + 16 * j;
            xr[j] = ch < kchunks ? ((const f32x4 *)x_row)[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int e = grp; e < c; e += 16) {
            const int f = sf[e];
            const f32x4 *src = (const f32x4 *)(W_enc_t + (long long)f * d_in);
            f32x4 w[FIN_CPL];
#pragma unroll
            for (int j = 0; j < FIN_CPL; ++j) {
                const int ch = l16 + 16 * j;
                w[j] = ch < kchunks ? src[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < FIN_CPL; ++j)
                part += w[j].x * xr[j].x + w[j].y * xr[j].y + w[j].z * xr[j].z + w[j].w * xr[j].w;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (l16 == 0) {
                const float pre = part + bias_fold[f];
                sv[e] = pre > 0.f ? pre : -1.f;  // -1 marks a near miss of the screen
                if (pre > 0.f) atomicAdd(&n_live, 1);
            }
        }
        __syncthreads();
    }
    // rank sort (value descending, feature id ascending); a thread holds its entries' ranks across the barrier
    constexpr int EPT = FIN_MAX_ACTIVE / 256;
    float ev[EPT];
    int ef[EPT], er[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + 256 * j;
        ev[j] = 0.f; ef[j] = 0; er[j] = -1;
        if (e < c && (!SCREENED || sv[e] > 0.f)) {
            const float v = sv[e];
            const int f = sf[e];
            int rank = 0;
            for (int i = 0; i < c; ++i) {  // near misses hold -1 and precede nothing
                const float w = sv[i];
                rank += (w > v || (w == v && sf[i] < f)) ? 1 : 0;
            }
            ev[j] = v; ef[j] = f; er[j] = rank;
        }
    }
    __syncthreads();
    if constexpr (SCREENED) {
        c = n_live;
        if (tid == 0 && c_emitted <= max_active) *count_m = c;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (er[j] >= 0) {
            sv[er[j]] = ev[j]; sf[er[j]] = ef[j];
            vrow[er[j]] = ev[j]; frow[er[j]] = ef[j];
        }
    for (int e = c + tid; e < max_active; e += 256) {  // defined tail, as sparse_sort_kernel
        vrow[e] = 0.f;
        frow[e] = -1;
    }
    __syncthreads();
    const int l16 = tid & 15, grp = tid >> 4;
    f32x4 acc[FIN_CPL];
#pragma unroll
    for (int j = 0; j < FIN_CPL; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int e = grp; e < c; e += 16) {
        const float v = sv[e];
        const f32x4 *src = (const f32x4 *)(W_dec + (long long)sf[e] * d_out);
        f32x4 w[FIN_CPL];
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) {
            const int ch = l16 + 16 * j;
            w[j] = ch < chunks ? src[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (score_row) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < FIN_CPL; ++j) {
                const int ch = l16 + 16 * j;
                if (ch < chunks) {
                    const f32x4 q = sd[ch];
                    part += w[j].x * q.x + w[j].y * q.y + w[j].z * q.z + w[j].w * q.w;
                }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (l16 == 0) score_row[e] = part * v;
        }
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) acc[j] += w[j] * v;
    }
    if (score_row)
        for (int e = c + tid; e < max_active; e += 256) score_row[e] = 0.f;
    if (recon_row) {
        // quarter waves of one wave -> lanes 0..15; waves -> LDS; fixed order, so the result is reproducible
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) {
#pragma unroll
            for (int comp = 0; comp < 4; ++comp) {
                float x = acc[j][comp];
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                acc[j][comp] = x;
            }
        }
        if ((tid & 63) < 16) {
#pragma unroll
            for (int j = 0; j < FIN_CPL; ++j) red[tid >> 6][l16 + 16 * j] = acc[j];
        }
        __syncthreads();
        for (int ch = tid; ch < chunks; ch += 256) {
            f32x4 b = bias ? ((const f32x4 *)bias)[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
            ((f32x4 *)recon_row)[ch] = ((red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch])) + b;
        }
    }
}

template <bool SCREENED>
__global__ __launch_bounds__(256) void sparse_finish_kernel(const float *W_dec, const float *bias, int d_out, int *feat,
                                                           float *val, int *count, int max_active,
                                                           const float *dirs, long long dir_stride, float *recon,
                                                           float *scores, const float *x, const float *W_enc_t,
                                                           const float *bias_fold, int d_in) {
    const long long m = blockIdx.x;
    finish_row<SCREENED>(W_dec, bias, d_out, feat + m * max_active, val + m * max_active, count + m, max_active,
                         scores ? dirs + m * dir_stride : nullptr, recon ? recon + m * d_out : nullptr,
                         scores ? scores + m * max_active : nullptr, SCREENED ? x + m * d_in : nullptr, W_enc_t, bias_fold,
                         d_in);
}

// The same for the rows of SEVERAL dictionaries in one launch (grid = n_dicts * M): at CircuitLens sizes one
// dictionary's 512 rows are two workgroups per CU; all dictionaries of a pass together keep every CU full.  Outputs
// are stacked [n_dicts, M, ...]; what differs per dictionary travels in the kernel arguments.
constexpr int FIN_MANY_MAX = 32;
struct FinishManyArgs {
    struct One {
        const float *W_dec, *bias, *dirs, *x, *W_enc_t, *bias_fold;
    } d[FIN_MANY_MAX];
    long long M, dir_stride;
    int d_out, d_in, max_active;
    int *feat, *count;
    float *val, *recon, *scores;
};
template <bool SCREENED>
__global__ __launch_bounds__(256) void sparse_finish_many_kernel(const FinishManyArgs p) {
    const int di = (int)(blockIdx.x / p.M);  // uniform: the dictionary's pointers are scalar loads
    const long long m = blockIdx.x - di * p.M, r = di * p.M + m;
    const FinishManyArgs::One &o = p.d[di];
    finish_row<SCREENED>(o.W_dec, o.bias, p.d_out, p.feat + r * p.max_active, p.val + r * p.max_active, p.count + r,
                         p.max_active, p.scores ? o.dirs + m * p.dir_stride : nullptr,
                         p.recon ? p.recon + r * p.d_out : nullptr, p.scores ? p.scores + r * p.max_active : nullptr,
                         SCREENED ? o.x + m * p.d_in : nullptr, o.W_enc_t, o.bias_fold, p.d_in);
}

// The tail of the CircuitLens pass for a batch of rows, ONE launch: assemble each row's component scores in the
// ActiveFeatures row order (head scalars, then per group its auxiliary scalars followed by the scores of its
// dictionary's live features), divide by the row's sum (normalize_activations, reference src/circuit_lens.py:551-554)
// and select the k largest (get_top_k_features, :161-164: `activations.topk(k)`).  grid = rows, block = 256.
// Column layout of a row (W columns): [n_head] then for g < n_groups: [aux_n[g]] [cap score slots of dictionary g].
// A score slot is live iff slot < count[g][row]; dead slots read 0 in raw / acts and never rank.
// Selection order: value descending, column ascending (deterministic; torch.topk leaves ties unspecified); NaN
// never ranks.  The row is kept in LDS when it fits (W <= lds_cols), else the rounds re-read `acts` (L2-hot).
constexpr int ASM_MAX_GROUPS = 64;
constexpr int ASM_CAND_MAX = 2048;  // selection candidates held in LDS
constexpr int ASM_VPT = 60;  // ranked values a thread keeps in registers: rows of up to 256 * 60 = 15360 columns
struct AssembleArgs {
    long long R;
    int n_head, n_groups, cap, W, k, n_aux_total, lds_cols;
    int aux_n[ASM_MAX_GROUPS];      // auxiliary scalars in front of each group's score slots
    int aux_off[ASM_MAX_GROUPS];    // offset of the group's auxiliary scalars in a row of `aux`
    int col_off[ASM_MAX_GROUPS];    // first column of the group
    const float *head, *aux, *scores;
    const int *count;
    const int *tags;   // optional [n_head + n_aux_total + n_groups]: the caller's label of every column class
    const int *feat;   // optional [n_groups, R, cap]: the feature id behind every score slot
    float *raw, *acts;
    unsigned char *mask;
    float *top_v;
    int *top_info;     // [R, k, 4]: column, tag >> 16, tag & 0xffff, feature id
};

__global__ __launch_bounds__(256) void lens_assemble_kernel(const AssembleArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *row = (float *)smem_raw;  // [lds_cols] raw values when the row fits
    __shared__ unsigned live_bits[(15 * 1024 + 31) / 32];  // ... and which of them are live (a VALUE cannot say: a live
                                                            // score may be -inf or any other bit pattern)
    __shared__ float red_v[4];
    __shared__ int red_c[4];
    __shared__ float s_total;
    __shared__ int s_cnt[ASM_MAX_GROUPS], s_col[ASM_MAX_GROUPS], s_aux[ASM_MAX_GROUPS], s_aoff[ASM_MAX_GROUPS];
    __shared__ float m_v[256], cand_v[ASM_CAND_MAX], t_v;
    __shared__ int m_c[256], cand_c[ASM_CAND_MAX], t_c, n_cand;
    const long long r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool in_lds = p.W <= p.lds_cols;
    const float NINF = -__builtin_inff();
    float *raw_row = p.raw ? p.raw + r * p.W : nullptr;
    float *acts_row = p.acts ? p.acts + r * p.W : nullptr;
    unsigned char *mask_row = p.mask ? p.mask + r * p.W : nullptr;
    // pass 1, group by group (the group index is uniform: its tables stay in scalar registers, the score slots are
    // read coalesced): raw values, mask, the row sum in a fixed order
    float part = 0.f;
    auto put = [&](int c, float v, bool live) {
        part += v;
        if (raw_row) raw_row[c] = v;
        if (mask_row) mask_row[c] = live ? 1 : 0;
        if (in_lds) {
            row[c] = v;  // the raw value for now; scaled in pass 2
            if (live) atomicOr(&live_bits[c >> 5], 1u << (c & 31));
        }
    };
    if (in_lds)
        for (int i = tid; i < (p.W + 31) / 32; i += 256) live_bits[i] = 0u;
    if (tid < p.n_groups) {
        const int cnt = p.count[(long long)tid * p.R + r];
        s_cnt[tid] = cnt > p.cap ? p.cap : cnt;
        s_col[tid] = p.col_off[tid];
        s_aux[tid] = p.aux_n[tid];
        s_aoff[tid] = p.aux_off[tid];
    }
    __syncthreads();
    // what the caller wants to know about a winning column: its class label and the feature behind a score slot
    auto emit = [&](int j, float v, int c) {
        int tag = -1, feature = -1;
        if (c >= 0) {
            int cls;  // index into tags: head columns, then the auxiliary scalars group after group, then one per group
            if (c < p.n_head) cls = c;
            else {
                int g = 0;
                while (g + 1 < p.n_groups && c >= s_col[g + 1]) ++g;
                const int o = c - s_col[g];
                if (o < s_aux[g]) cls = p.n_head + s_aoff[g] + o;
                else {
                    cls = p.n_head + p.n_aux_total + g;
                    if (p.feat) feature = p.feat[((long long)g * p.R + r) * p.cap + (o - s_aux[g])];
                }
            }
            if (p.tags) tag = p.tags[cls];
        }
        int *info = p.top_info + (r * p.k + j) * 4;
        p.top_v[r * p.k + j] = v;
        info[0] = c;
        info[1] = tag < 0 ? -1 : tag >> 16;
        info[2] = tag < 0 ? -1 : tag & 0xffff;
        info[3] = feature;
    };
    if (tid < p.n_head) put(tid, p.head[r * p.n_head + tid], true);
    for (int g = 0; g < p.n_groups; ++g)
        if (tid < p.aux_n[g]) put(p.col_off[g] + tid, p.aux[r * p.n_aux_total + p.aux_off[g] + tid], true);
    // score slots: eight groups' loads in flight before the first store (the stores would otherwise serialise one
    // load round trip per group: nothing tells the compiler that the outputs do not alias the scores)
    const float *__restrict__ sc = p.scores + r * p.cap;
    const long long gstride = p.R * p.cap;
    for (int slot = tid; slot < p.cap; slot += 256)
        for (int g0 = 0; g0 < p.n_groups; g0 += 8) {
            float v[8];
            bool live[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int g = g0 + u < p.n_groups ? g0 + u : p.n_groups - 1;
                live[u] = slot < s_cnt[g];
                v[u] = live[u] ? sc[g * gstride + slot] : 0.f;  // (dead slots are not read: 6 of 7 at CircuitLens densities)
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (g0 + u < p.n_groups) put(p.col_off[g0 + u] + p.aux_n[g0 + u] + slot, v[u], live[u]);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red_v[wave] = part;
    __syncthreads();
    if (tid == 0) s_total = (red_v[0] + red_v[1]) + (red_v[2] + red_v[3]);
    __syncthreads();
    const float total = s_total;
    // pass 2: acts = raw / total (a division, as the reference divides).  Columns are walked with stride 256 from
    // here on: column c belongs to thread c & 255, which keeps its ranked values (dead = -inf) in REGISTERS when the
    // row went through LDS (W <= 256 * ASM_VPT) -- the k selection rounds then touch no memory.
    float vals[ASM_VPT];
    float bv = NINF;
    int bc = 0x7fffffff;
    if (in_lds) {
#pragma unroll
        for (int i = 0; i < ASM_VPT; ++i) {
            const int c = tid + 256 * i;
            vals[i] = NINF;
            if (c < p.W) {
                const float v = row[c];
                const bool live = (live_bits[c >> 5] >> (c & 31)) & 1u;
                const float a = live ? v / total : 0.f;
                if (acts_row) acts_row[c] = a;
                if (live) vals[i] = a;
                if (live && a > bv) { bv = a; bc = c; }  // ascending c: the first of equal values stays
            }
        }
    } else {  // rows too wide for LDS: the launcher insists on raw, acts and mask, and the rounds re-read them
        for (int c = tid; c < p.W; c += 256) {
            const bool live = mask_row[c] != 0;
            const float a = live ? raw_row[c] / total : 0.f;
            acts_row[c] = a;
            if (live && a > bv) { bv = a; bc = c; }
        }
    }
    // Selection without k serial block-wide rounds.  An element of the global top-k can only belong to one of the k
    // threads with the largest per-thread maxima, and must rank at or above the k-th largest of those maxima (T):
    // those threads push their qualifying elements (typically a dozen or two in total) into an LDS list, which is
    // then rank-sorted in one step.  Order everywhere: value descending, column ascending.
    if (tid < p.k) emit(tid, NINF, -1);
    m_v[tid] = bv;
    m_c[tid] = bc;
    if (tid == 0) { n_cand = 0; t_v = NINF; t_c = 0x7fffffff; }
    __syncthreads();
    {
        int rank = 0;
        for (int i = 0; i < 256; ++i) {
            const float ov = m_v[i];
            const int oc = m_c[i];
            rank += (ov > bv || (ov == bv && oc < bc)) ? 1 : 0;
        }
        const bool has = bv > NINF;
        if (has && rank == p.k - 1) { t_v = bv; t_c = bc; }  // fewer than k live threads: T stays "everything qualifies"
        __syncthreads();
        const float tv = t_v;
        const int tc = t_c;
        if (has && rank < p.k) {
            auto push = [&](float v, int c) {
                if (v > NINF && (v > tv || (v == tv && c <= tc))) {
                    const int pos = atomicAdd(&n_cand, 1);
                    if (pos < ASM_CAND_MAX) { cand_v[pos] = v; cand_c[pos] = c; }
                }
            };
            if (in_lds) {
#pragma unroll
                for (int i = 0; i < ASM_VPT; ++i) push(vals[i], tid + 256 * i);
            } else {
                for (int c = tid; c < p.W; c += 256) push(mask_row[c] ? acts_row[c] : NINF, c);
            }
        }
        __syncthreads();
    }
    const int n = n_cand;
    if (n <= ASM_CAND_MAX) {
        for (int e = tid; e < n; e += 256) {
            const float v = cand_v[e];
            const int c = cand_c[e];
            int rank = 0;
            for (int i = 0; i < n; ++i) {
                const float ov = cand_v[i];
                const int oc = cand_c[i];
                rank += (ov > v || (ov == v && oc < c)) ? 1 : 0;
            }
            if (rank < p.k) emit(rank, v, c);
        }
        return;
    }
    // (more candidates than the list holds -- k * columns-per-thread can exceed it for wide rows and large k --:)
    // k rounds: block-wide best of the threads' candidates; only the winner's thread looks for its next one (the
    // others' candidates are still their best below the new winner)
    for (int j = 0; j < p.k; ++j) {
        float wv = bv;
        int wc = bc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(wv, off);
            const int oc = __shfl_xor(wc, off);
            if (ov > wv || (ov == wv && oc < wc)) { wv = ov; wc = oc; }
        }
        if (lane == 0) { red_v[wave] = wv; red_c[wave] = wc; }
        __syncthreads();
        wv = red_v[0]; wc = red_c[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > wv || (red_v[w] == wv && red_c[w] < wc)) { wv = red_v[w]; wc = red_c[w]; }
        __syncthreads();
        const bool found = wc != 0x7fffffff;
        if (!found) break;  // fewer than k live columns: the rest stay empty (pre-filled above)
        if (tid == 0) emit(j, wv, wc);
        if ((wc & 255) == tid) {  // my candidate won: my largest (value, -column) strictly below it
            bv = NINF; bc = 0x7fffffff;
            if (in_lds) {
#pragma unroll
                for (int i = 0; i < ASM_VPT; ++i) {
                    const int c = tid + 256 * i;
                    const float v = vals[i];
                    const bool below = v < wv || (v == wv && c > wc);
                    if (below && v > bv) { bv = v; bc = c; }
                }
            } else {
                for (int c = tid; c < p.W; c += 256) {
                    const float v = mask_row[c] ? acts_row[c] : NINF;
                    const bool below = v < wv || (v == wv && c > wc);
                    if (below && v > bv) { bv = v; bc = c; }
                }
            }
        }
    }
}

// Many row-wise dot products in one launch: out_j[r] = sum_k (a_j[r][k] - c_j[r][k]) * b_j[r][k] for every term j and
// row r, each operand either per row (stride >= d) or one shared vector (stride 0), c optional.  This is the whole
// auxiliary part of the CircuitLens direction lens -- embed, pos_embed, b_O, z error, z bias, transcoder error and
// transcoder bias of every layer against the row's direction (reference src/circuit_lens.py:570-575, 591-604, 640-645
// dotted as in :762-766) -- which as torch ops is 17 launches over [L, R, 768] temporaries.  grid = rows, one
// workgroup of four waves per row; wave w takes terms w, w + 4, ...; a lane holds three float4 per operand (d = 768).
constexpr int RD_MAX_TERMS = 64;
struct RowDotArgs {
    struct Term {
        const float *a, *c, *b;
        float *out;
        int sa, sc, sb, so;  // element strides between rows (0: shared)
    } t[RD_MAX_TERMS];
    long long R;
    int d, n;
};
__global__ __launch_bounds__(256) void rowdot_kernel(const RowDotArgs p) {
    const long long r = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = p.d >> 2;
    // (the terms are spread over gridDim.y workgroups: one row's 62 dot products on one workgroup are a chain of 16
    // dependent round trips)
    for (int j = blockIdx.y * 4 + wave; j < p.n; j += 4 * gridDim.y) {  // j is wave-uniform: the term's fields are scalar loads
        const RowDotArgs::Term &t = p.t[j];
        const f32x4 *a = (const f32x4 *)(t.a + r * t.sa), *b = (const f32x4 *)(t.b + r * t.sb);
        const f32x4 *c = t.c ? (const f32x4 *)(t.c + r * t.sc) : nullptr;
        float part = 0.f;
        for (int ch0 = 0; ch0 < chunks; ch0 += 64 * 3) {
            f32x4 av[3], bv[3], cv[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int ch = ch0 + lane + 64 * u;
                const bool ok = ch < chunks;
                av[u] = ok ? a[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
                bv[u] = ok ? b[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
                cv[u] = (ok && c) ? c[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const f32x4 e = av[u] - cv[u];
                part += e.x * bv[u].x + e.y * bv[u].y + e.z * bv[u].z + e.w * bv[u].w;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) t.out[r * t.so] = part;
    }
}

// one wavefront per token row: the row (minus b_dec) is held in registers, each selected encoder
// column (a contiguous row of Pt, L2-resident) is dotted against it
__global__ __launch_bounds__(256) void project_kernel(const float *x, long long M, int d_in, const float *b_dec,
                                                     const float *Pt, const float *bias, int F, int relu,
                                                     float *out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = d_in >> 2;  // <= 1024: up to 16 float4 per lane
    for (long long m = wave; m < M; m += n_waves) {
        f32x4 xr[16];
        const f32x4 *src = (const f32x4 *)(x + m * d_in);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ch = lane + 64 * j;
            if (ch < chunks) {
                xr[j] = src[ch];
                if (b_dec) xr[j] -= ((const f32x4 *)b_dec)[ch];
            } else {
                xr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        for (int f = 0; f < F; ++f) {
            const f32x4 *col = (const f32x4 *)(Pt + (long long)f * d_in);
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ch = lane + 64 * j;
                if (ch < chunks) {
                    const f32x4 w = col[ch];
                    part += xr[j].x * w.x + xr[j].y * w.y + xr[j].z * w.z + xr[j].w * w.w;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0) {
                float v = part + bias[f];
                out[m * F + f] = (relu && v < 0.f) ? 0.f : v;
            }
        }
    }
}

template <int NDIR>
int launch_contrib(const float *W_dec, int d_out, const int *feat, const float *val, const int *count,
                   long long M, int max_active, const float *dirs, float *scores, hipStream_t stream) {
    long long blocks = M < 4096 ? M : 4096;
    const size_t lds = (size_t)NDIR * d_out * sizeof(float);
    hipLaunchKernelGGL(contrib_kernel<NDIR>, dim3((unsigned)blocks), dim3(256), lds, stream, W_dec, d_out,
                       feat, val, count, M, max_active, dirs, scores);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

}  // namespace

int tsc_launch_sparse_sort(int *feat, float *val, const int *count, long long M, int max_active,
                           hipStream_t stream) {
    if (M <= 0) return TSC_OK;
    if (max_active > SORT_MAX)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "max_active %d exceeds %d", max_active, SORT_MAX);
    size_t lds = (size_t)RANK_SORT_MAX * 8;  // rank sort: two arrays of RANK_SORT_MAX; bitonic: next power of two of max_active keys
    if (max_active > RANK_SORT_MAX) {
        size_t p2 = (size_t)RANK_SORT_MAX * 2;
        while (p2 < (size_t)max_active) p2 <<= 1;
        lds = p2 * 8;
    }
    hipLaunchKernelGGL(sparse_sort_kernel, dim3((unsigned)M), dim3(256), lds, stream, feat, val, count,
                       max_active);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_dense_to_sparse(const float *acts, long long M, int N, int max_active, int *feat, float *val, int *count,
                               hipStream_t stream) {
    if (M <= 0) return TSC_OK;
    const long long blocks = (M + 3) / 4;
    if (blocks > 0x7fffffffLL) return tsc_fail(TSC_ERR_UNSUPPORTED, "dense_to_sparse grid out of range");
    hipLaunchKernelGGL(dense_to_sparse_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, acts, M, N, max_active, feat, val,
                       count);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_decode_rows(const float *W_dec, int d_sae, int d_out, const int *feat, const float *val, long long n,
                           float *rows, hipStream_t stream) {
    if (n <= 0) return TSC_OK;
    long long blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(decode_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, W_dec, d_sae, d_out, feat,
                       val, n, rows);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_decode_sparse(const float *W_dec, const float *bias, int d_out, const int *feat,
                             const float *val, const int *count, long long M, int max_active, float *recon,
                             hipStream_t stream) {
    if (M <= 0) return TSC_OK;
    if (d_out > 4096) return tsc_fail(TSC_ERR_UNSUPPORTED, "d_out %d exceeds 4096", d_out);
    if (M <= 2048)
        hipLaunchKernelGGL(decode_sparse_kernel<8>, dim3((unsigned)M), dim3(256), 0, stream, W_dec, bias, d_out,
                           feat, val, count, max_active, recon);
    else
        hipLaunchKernelGGL(decode_sparse_kernel<1>, dim3((unsigned)M), dim3(256), 0, stream, W_dec, bias, d_out,
                           feat, val, count, max_active, recon);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_sparse_finish_supported(int d_out, int max_active) {
    return d_out <= FIN_CPL * 16 * 4 && (d_out & 3) == 0 && max_active <= FIN_MAX_ACTIVE;
}

int tsc_launch_sparse_finish(const float *W_dec, const float *bias, int d_out, int *feat, float *val, int *count,
                             long long M, int max_active, const float *dirs, long long dir_stride, float *recon,
                             float *scores, const float *x, const float *W_enc_t, const float *bias_fold, int d_in,
                             hipStream_t stream) {
    if (M <= 0) return TSC_OK;
    if (!tsc_sparse_finish_supported(d_out, max_active))
        return tsc_fail(TSC_ERR_UNSUPPORTED, "fused lens pass needs d_out <= %d and max_active <= %d (got %d, %d)",
                        FIN_CPL * 64, FIN_MAX_ACTIVE, d_out, max_active);
    if (x) {
        if (!W_enc_t || !bias_fold || d_in > FIN_CPL * 64 || (d_in & 3))
            return tsc_fail(TSC_ERR_UNSUPPORTED, "screened lens pass needs the transposed W_enc copy and d_in <= %d", FIN_CPL * 64);
        hipLaunchKernelGGL(sparse_finish_kernel<true>, dim3((unsigned)M), dim3(256), 0, stream, W_dec, bias, d_out, feat,
                           val, count, max_active, dirs, dir_stride, recon, scores, x, W_enc_t, bias_fold, d_in);
    } else {
        hipLaunchKernelGGL(sparse_finish_kernel<false>, dim3((unsigned)M), dim3(256), 0, stream, W_dec, bias, d_out, feat,
                           val, count, max_active, dirs, dir_stride, recon, scores, x, W_enc_t, bias_fold, d_in);
    }
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_sparse_finish_many(int n, const tsc_dict *const *dicts, const float *const *x, long long M,
                                  int max_active, int *feat, float *val, int *count, const float *const *dirs,
                                  long long dir_stride, float *recon, float *scores, int screened, hipStream_t stream) {
    if (M <= 0 || n <= 0) return TSC_OK;
    if (n > FIN_MANY_MAX) return tsc_fail(TSC_ERR_UNSUPPORTED, "at most %d dictionaries per grouped finish (got %d)", FIN_MANY_MAX, n);
    if ((long long)n * M > 0x7fffffffLL) return tsc_fail(TSC_ERR_UNSUPPORTED, "grouped finish grid out of range");
    FinishManyArgs p;
    memset(&p, 0, sizeof(p));
    const int d_out = dicts[0]->d_out, d_in = dicts[0]->d_in;
    if (!tsc_sparse_finish_supported(d_out, max_active))
        return tsc_fail(TSC_ERR_UNSUPPORTED, "fused lens pass needs d_out <= %d and max_active <= %d (got %d, %d)",
                        FIN_CPL * 64, FIN_MAX_ACTIVE, d_out, max_active);
    for (int i = 0; i < n; ++i) {
        const tsc_dict *d = dicts[i];
        if (d->d_out != d_out || d->d_in != d_in) return tsc_fail(TSC_ERR_ARG, "grouped finish: dictionary %d has other d_in / d_out", i);
        if (!d->W_dec) return tsc_fail(TSC_ERR_ARG, "dictionary %d was created without W_dec", i);
        if (screened && (!d->W_enc_t || !d->bias_fold || !x || !x[i] || d_in > FIN_CPL * 64 || (d_in & 3)))
            return tsc_fail(TSC_ERR_UNSUPPORTED, "grouped screened finish needs TSC_F32_SCREEN handles, their inputs and d_in <= %d", FIN_CPL * 64);
        p.d[i].W_dec = d->W_dec; p.d[i].bias = d->out_bias; p.d[i].dirs = dirs ? dirs[i] : nullptr;
        p.d[i].x = screened ? x[i] : nullptr; p.d[i].W_enc_t = d->W_enc_t; p.d[i].bias_fold = d->bias_fold;
    }
    p.M = M; p.dir_stride = dir_stride; p.d_out = d_out; p.d_in = d_in; p.max_active = max_active;
    p.feat = feat; p.count = count; p.val = val; p.recon = recon; p.scores = scores;
    const dim3 grid((unsigned)(n * M)), block(256);
    if (screened) hipLaunchKernelGGL(sparse_finish_many_kernel<true>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(sparse_finish_many_kernel<false>, grid, block, 0, stream, p);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_lens_assemble(long long R, int n_head, int n_groups, const int *group_aux, int cap, const float *head,
                             const float *aux, const float *scores, const int *count, const int *tags, const int *feat,
                             int k, float *raw, float *acts, unsigned char *mask, float *top_v, int *top_info,
                             hipStream_t stream) {
    if (R <= 0) return TSC_OK;
    if (n_groups < 1 || n_groups > ASM_MAX_GROUPS)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "lens assemble takes 1..%d groups (got %d)", ASM_MAX_GROUPS, n_groups);
    if (n_head > 256) return tsc_fail(TSC_ERR_UNSUPPORTED, "lens assemble takes at most 256 head scalars (got %d)", n_head);
    AssembleArgs p;
    p.R = R; p.n_head = n_head; p.n_groups = n_groups; p.cap = cap; p.k = k;
    int col = n_head, aoff = 0;
    for (int g = 0; g < n_groups; ++g) {
        if (group_aux[g] < 0 || group_aux[g] > 256)
            return tsc_fail(TSC_ERR_ARG, "auxiliary count of group %d outside 0..256", g);
        p.aux_n[g] = group_aux[g]; p.aux_off[g] = aoff; p.col_off[g] = col;
        aoff += group_aux[g];
        col += group_aux[g] + cap;
    }
    for (int g = n_groups; g < ASM_MAX_GROUPS; ++g) p.aux_n[g] = p.aux_off[g] = p.col_off[g] = 0;
    p.W = col; p.n_aux_total = aoff;
    p.lds_cols = 15 * 1024;  // 60 KiB of dynamic LDS: two workgroups per CU
    p.head = head; p.aux = aux; p.scores = scores; p.count = count;
    p.tags = tags; p.feat = feat;
    p.raw = raw; p.acts = acts; p.mask = mask; p.top_v = top_v; p.top_info = top_info;
    if (p.W > p.lds_cols && (!acts || !raw || !mask))
        return tsc_fail(TSC_ERR_UNSUPPORTED, "lens assemble: rows of %d columns do not fit in LDS, pass the raw, acts and mask outputs", p.W);
    const size_t lds = p.W <= p.lds_cols ? (size_t)p.W * sizeof(float) : 16;
    hipLaunchKernelGGL(lens_assemble_kernel, dim3((unsigned)R), dim3(256), lds, stream, p);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_contrib_many(int n, const tsc_dict *const *dicts, long long M, long long pitch_rows, int max_active,
                            const int *feat, const float *val, const int *count, const float *const *dirs,
                            long long dir_stride, float *scores, int *feat_out, int *count_out, int out_cap,
                            hipStream_t stream) {
    if (n <= 0 || M <= 0) return TSC_OK;
    if (n > CM_MAX) return tsc_fail(TSC_ERR_UNSUPPORTED, "at most %d dictionaries per grouped contribution call (got %d)", CM_MAX, n);
    if ((long long)n * M > 0x7fffffffLL) return tsc_fail(TSC_ERR_UNSUPPORTED, "grouped contribution grid out of range");
    ContribManyArgs p;
    memset(&p, 0, sizeof(p));
    const int d_out = dicts[0]->d_out;
    if ((size_t)d_out * 4 > 60 * 1024) return tsc_fail(TSC_ERR_UNSUPPORTED, "d_out too large for the LDS staging");
    for (int i = 0; i < n; ++i) {
        if (!dicts[i]->W_dec) return tsc_fail(TSC_ERR_ARG, "dictionary %d was created without W_dec", i);
        if (dicts[i]->d_out != d_out) return tsc_fail(TSC_ERR_ARG, "grouped contribution: dictionary %d has another d_out", i);
        p.d[i].W_dec = dicts[i]->W_dec;
        p.d[i].dirs = dirs[i];
    }
    p.M = M; p.pitch_rows = pitch_rows; p.dir_stride = dir_stride; p.d_out = d_out; p.max_active = max_active; p.n = n;
    p.out_cap = out_cap > 0 && out_cap < max_active ? out_cap : max_active;
    p.feat = feat; p.val = val; p.count = count; p.scores = scores; p.feat_out = feat_out; p.count_out = count_out;
    hipLaunchKernelGGL(contrib_many_kernel, dim3((unsigned)(n * M)), dim3(256), (size_t)d_out * 4, stream, p);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_rowdot_many(int n, const float *const *a, const long long *sa, const float *const *c, const long long *sc,
                           const float *const *b, const long long *sb, float *const *out, const long long *so,
                           long long R, int d, hipStream_t stream) {
    if (R <= 0 || n <= 0) return TSC_OK;
    for (int j0 = 0; j0 < n; j0 += RD_MAX_TERMS) {
        RowDotArgs p;
        memset(&p, 0, sizeof(p));
        p.R = R; p.d = d; p.n = n - j0 < RD_MAX_TERMS ? n - j0 : RD_MAX_TERMS;
        for (int j = 0; j < p.n; ++j) {
            RowDotArgs::Term &t = p.t[j];
            t.a = a[j0 + j]; t.b = b[j0 + j]; t.c = c ? c[j0 + j] : nullptr; t.out = out[j0 + j];
            t.sa = (int)sa[j0 + j]; t.sb = (int)sb[j0 + j]; t.sc = c && sc ? (int)sc[j0 + j] : 0; t.so = (int)so[j0 + j];
        }
        // a workgroup walks its share of the terms one dependent round trip after the other: spread the terms over
        // gridDim.y until about 4096 workgroups (two rounds of resident ones) are in flight
        unsigned split = (unsigned)((4096 + R - 1) / R);
        if (split > (unsigned)((p.n + 3) / 4)) split = (unsigned)((p.n + 3) / 4);
        if (split < 1) split = 1;
        hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)R, split), dim3(256), 0, stream, p);
        TSC_HIP_CHECK(hipGetLastError());
    }
    return TSC_OK;
}

int tsc_launch_contrib(const float *W_dec, int d_out, const int *feat, const float *val, const int *count,
                       long long M, int max_active, const float *dirs, int n_dir, float *scores,
                       hipStream_t stream) {
    if (M <= 0) return TSC_OK;
    if (n_dir < 1 || n_dir > DIR_MAX) return tsc_fail(TSC_ERR_ARG, "n_dir %d outside 1..%d", n_dir, DIR_MAX);
    if ((size_t)n_dir * d_out * 4 > 60 * 1024)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "n_dir * d_out too large for the LDS staging");
#define TSC_CASE(N) \
    case N: return launch_contrib<N>(W_dec, d_out, feat, val, count, M, max_active, dirs, scores, stream);
    switch (n_dir) {
        TSC_CASE(1) TSC_CASE(2) TSC_CASE(3) TSC_CASE(4) TSC_CASE(5) TSC_CASE(6) TSC_CASE(7) TSC_CASE(8)
    }
#undef TSC_CASE
    return TSC_ERR_ARG;
}

int tsc_launch_project(const float *x, long long M, int d_in, const float *b_dec, const float *Pt,
                       const float *bias, int F, int relu, float *out, hipStream_t stream) {
    if (M <= 0 || F <= 0) return TSC_OK;
    long long blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, M, d_in, b_dec, Pt, bias, F,
                       relu, out);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}
