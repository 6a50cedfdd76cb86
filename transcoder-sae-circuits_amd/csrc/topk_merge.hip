// Per-feature top-k list maintenance outside the encode kernel: reset, and the k-way merge used
// both to fold the encode kernel's slots into the final lists and to combine the lists of several
// ranks after the RCCL all-gather.  There is no reference counterpart beyond
// `argsort(descending=True)[:num_active]` (reference src/max_act_analysis.py:245-249), whose
// result these kernels reproduce with the tie rule "equal values: smaller token id first".
// HBM-bound integer/compare work; one wavefront per feature.
#include "tsc_internal.h"

namespace {

constexpr int MERGE_MAX = TSC_TOPK_MERGE_MAX;  // n_lists * k entries staged in LDS per feature

__global__ void topk_fill_kernel(float *vals, long long *idx, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        vals[i] = 0.f;
        idx[i] = -1;
    }
}

// floor[f] = the k-th largest value over the feature's n_slots unsorted lists (0 while fewer than k entries
// are live: padding is 0 and live values are > 0).  One thread per feature keeps the KC largest values seen
// in registers with a branch-free insertion (max/min ripple); values only, no ids: a candidate below the
// floor cannot enter the merged top-k whatever the tie rule.
template <int KC>
__global__ __launch_bounds__(128) void topk_floor_kernel(const float *__restrict__ vals, int n_slots, int d_sae,
                                                         int k, float *__restrict__ floor) {
    const int f = blockIdx.x * 128 + threadIdx.x;
    if (f >= d_sae) return;
    float t[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) t[j] = 0.f;
    for (int s = 0; s < n_slots; ++s) {
        const float *l = vals + ((size_t)s * d_sae + f) * k;
        for (int i = 0; i < k; ++i) {
            float v = l[i];
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const float hi = fmaxf(t[j], v);
                v = fminf(t[j], v);
                t[j] = hi;
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < KC; ++j) r = j == k - 1 ? t[j] : r;
    floor[f] = r;
}

__device__ __forceinline__ bool ranks_above(float v, long long id, float w, long long wid) {
    return v > w || (v == w && id < wid);
}

// Register variant for n_lists * k <= 64 * EPL: one wavefront per feature (4 features per workgroup), every
// lane keeps EPL entries in registers; k rounds of (lane-local best, wave arg-max by shuffles, retire the
// winner).  No LDS, no barrier.
template <int EPL>
__global__ __launch_bounds__(256) void topk_merge_reg_kernel(const float *__restrict__ vals,
                                                            const long long *__restrict__ idx, int n_lists,
                                                            int d_sae, int k, float *__restrict__ out_vals,
                                                            long long *__restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= d_sae) return;
    const int total = n_lists * k;
    float ev[EPL];
    long long eid[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int e = lane + 64 * j;
        ev[j] = -1.f;
        eid[j] = -1;
        if (e < total) {
            const int l = e / k, i = e - l * k;
            const long long off = ((long long)l * d_sae + f) * k + i;
            const long long id = idx[off];
            const float v = vals[off];
            if (id >= 0 && v > 0.f) { ev[j] = v; eid[j] = id; }
        }
    }
    for (int round = 0; round < k; ++round) {
        float bv = -1.f;
        long long bid = -1;
        int be = -1;  // global entry number of the lane's best: breaks ties between identical (value, id) pairs
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (ev[j] > 0.f && (be < 0 || ranks_above(ev[j], eid[j], bv, bid))) {
                bv = ev[j]; bid = eid[j]; be = lane + 64 * j;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const long long oid = __shfl_xor(bid, off);
            const int oe = __shfl_xor(be, off);
            const bool take = oe >= 0 && (be < 0 || ranks_above(ov, oid, bv, bid) ||
                                          (ov == bv && oid == bid && oe < be));
            if (take) { bv = ov; bid = oid; be = oe; }
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (be == lane + 64 * j) ev[j] = -1.f;
        if (lane == 0) {
            out_vals[(long long)f * k + round] = be >= 0 ? bv : 0.f;
            out_idx[(long long)f * k + round] = be >= 0 ? bid : -1;
        }
    }
}

// grid = d_sae, block = 64.  Entries with id < 0 or value <= 0 are padding.
__global__ __launch_bounds__(64) void topk_merge_kernel(const float *vals, const long long *idx,
                                                        int n_lists, int d_sae, int k,
                                                        float *out_vals, long long *out_idx) {
    __shared__ float sv[MERGE_MAX];
    __shared__ long long sid[MERGE_MAX];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int total = n_lists * k;
    for (int e = lane; e < total; e += 64) {
        const int l = e / k, i = e - l * k;
        const long long off = ((long long)l * d_sae + f) * k + i;
        const long long id = idx[off];
        const float v = vals[off];
        const bool live = id >= 0 && v > 0.f;
        sv[e] = live ? v : -1.f;
        sid[e] = live ? id : -1;
    }
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        float bv = -1.f;
        long long bid = -1;
        int be = -1;
        for (int e = lane; e < total; e += 64) {
            const float v = sv[e];
            if (v > 0.f && (be < 0 || ranks_above(v, sid[e], bv, bid))) {
                bv = v; bid = sid[e]; be = e;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const long long oid = __shfl_xor(bid, off);
            const int oe = __shfl_xor(be, off);
            // identical (value, id) pairs can come from two lists (the same token seen twice):
            // break that tie by entry index so every lane agrees on one winner
            const bool take = oe >= 0 && (be < 0 || ranks_above(ov, oid, bv, bid) ||
                                          (ov == bv && oid == bid && oe < be));
            if (take) { bv = ov; bid = oid; be = oe; }
        }
        if (lane == 0) {
            out_vals[(long long)f * k + round] = be >= 0 ? bv : 0.f;
            out_idx[(long long)f * k + round] = be >= 0 ? bid : -1;
            if (be >= 0) sv[be] = -1.f;
        }
        __syncthreads();
    }
}

}  // namespace

int tsc_launch_topk_fill(float *vals, long long *idx, long long n, hipStream_t stream) {
    if (n <= 0) return TSC_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(topk_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, vals, idx, n);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_topk_floor(const float *vals, int n_slots, int d_sae, int k, float *floor, hipStream_t stream) {
    const dim3 g((d_sae + 127) / 128), b(128);
    if (k <= 8)
        hipLaunchKernelGGL(topk_floor_kernel<8>, g, b, 0, stream, vals, n_slots, d_sae, k, floor);
    else if (k <= 20)
        hipLaunchKernelGGL(topk_floor_kernel<20>, g, b, 0, stream, vals, n_slots, d_sae, k, floor);
    else if (k <= 32)
        hipLaunchKernelGGL(topk_floor_kernel<32>, g, b, 0, stream, vals, n_slots, d_sae, k, floor);
    else
        hipLaunchKernelGGL(topk_floor_kernel<64>, g, b, 0, stream, vals, n_slots, d_sae, k, floor);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_topk_merge(const float *vals, const long long *idx, int n_lists, int d_sae, int k,
                          float *out_vals, long long *out_idx, hipStream_t stream) {
    if ((long long)n_lists * k > MERGE_MAX)
        return tsc_fail(TSC_ERR_UNSUPPORTED, "merge of %d lists x k=%d exceeds %d entries per feature",
                        n_lists, k, MERGE_MAX);
    const int total = n_lists * k;
    const dim3 g4((d_sae + 3) / 4), b4(256);
    if (total <= 64)
        hipLaunchKernelGGL(topk_merge_reg_kernel<1>, g4, b4, 0, stream, vals, idx, n_lists, d_sae, k, out_vals, out_idx);
    else if (total <= 128)
        hipLaunchKernelGGL(topk_merge_reg_kernel<2>, g4, b4, 0, stream, vals, idx, n_lists, d_sae, k, out_vals, out_idx);
    else if (total <= 192)
        hipLaunchKernelGGL(topk_merge_reg_kernel<3>, g4, b4, 0, stream, vals, idx, n_lists, d_sae, k, out_vals, out_idx);
    else if (total <= 256)
        hipLaunchKernelGGL(topk_merge_reg_kernel<4>, g4, b4, 0, stream, vals, idx, n_lists, d_sae, k, out_vals, out_idx);
    else
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)d_sae), dim3(64), 0, stream, vals, idx,
                           n_lists, d_sae, k, out_vals, out_idx);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}
