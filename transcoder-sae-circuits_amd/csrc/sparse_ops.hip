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
template <bool SCREENED>
__global__ __launch_bounds__(256) void sparse_finish_kernel(const float *W_dec, const float *bias, int d_out, int *feat,
                                                           float *val, int *count, int max_active,
                                                           const float *dirs, long long dir_stride, float *recon,
                                                           float *scores, const float *x, const float *W_enc_t,
                                                           const float *bias_fold, int d_in) {
    __shared__ float sv[FIN_MAX_ACTIVE];
    __shared__ int sf[FIN_MAX_ACTIVE];
    __shared__ f32x4 sd[FIN_CPL * 16];        // the token's direction
    __shared__ f32x4 red[4][FIN_CPL * 16];    // per-wave partial reconstructions
    __shared__ int n_live;
    const long long m = blockIdx.x;
    const int tid = threadIdx.x;
    const int c_emitted = count[m];
    int c = c_emitted;
    if (c > max_active) c = max_active;
    int *frow = feat + m * max_active;
    float *vrow = val + m * max_active;
    const int chunks = d_out >> 2;
    for (int e = tid; e < c; e += 256) {
        sv[e] = vrow[e];
        sf[e] = frow[e];
    }
    if (scores)
        for (int i = tid; i < chunks; i += 256) sd[i] = ((const f32x4 *)(dirs + m * dir_stride))[i];
    if (SCREENED && tid == 0) n_live = 0;
    __syncthreads();
    if constexpr (SCREENED) {
        const int l16 = tid & 15, grp = tid >> 4;
        const int kchunks = d_in >> 2;  // <= FIN_CPL * 16
        f32x4 xr[FIN_CPL];
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) {
            const int ch = l16 + 16 * j;
            xr[j] = ch < kchunks ? ((const f32x4 *)(x + m * d_in))[ch] : (f32x4){0.f, 0.f, 0.f, 0.f};
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
        if (tid == 0 && c_emitted <= max_active) count[m] = c;
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
        if (scores) {
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
            if (l16 == 0) scores[m * max_active + e] = part * v;
        }
#pragma unroll
        for (int j = 0; j < FIN_CPL; ++j) acc[j] += w[j] * v;
    }
    if (scores)
        for (int e = c + tid; e < max_active; e += 256) scores[m * max_active + e] = 0.f;
    if (recon) {
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
            ((f32x4 *)(recon + m * d_out))[ch] = ((red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch])) + b;
        }
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
