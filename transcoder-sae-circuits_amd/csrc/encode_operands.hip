// Operand preparation for the LDS-DMA encode kernels (encode_f32.hip).
//
// fp32: the handle keeps W_enc transposed, [d_sae][d_in], so that a feature's k run is contiguous: the W tile
// is then staged by the same 8-rows-by-128-bytes DMA pieces as the x tile and a B fragment is one
// ds_read_b128 (transpose_w_kernel, at tsc_dict_create / tsc_dict_refresh).
//
// bf16 encode path (TSC_BF16 dictionaries; BASELINE config 5, the wide-dictionary stress case).
//
//   acts = relu( bf16(x - b_dec) @ bf16(W_enc) + b_enc ),  products exact, fp32 accumulation
//
// The contraction itself is encode_f32_kernel<..., BF16 = true> (encode_f32.hip): same tiling, LDS-DMA
// staging and epilogues as the fp32 path, with v_mfma_f32_32x32x16_bf16.  This file prepares its operands:
//   * W_enc once per handle (and on tsc_dict_refresh): rounded to bf16 (nearest even, like torch's
//     .bfloat16()) and transposed to [d_sae][d_in], so that a feature's k run is contiguous and the W tile is
//     staged by the same 8-rows-by-128-bytes DMA pieces as the x tile;
//   * x per call: (x - b_dec) rounded to bf16 into a handle-owned workspace of TSC_BF16_WS_ROWS rows; longer
//     inputs are processed in chunks of that many rows (stream order makes the reuse safe; calls on one bf16
//     handle must therefore not overlap on different streams).
#include <cstring>

#include "tsc_internal.h"

namespace {


__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// out[n][k0..k0+7] = bf16(W[k0..k0+7][n]); thread = (n, k0/8): reads coalesced over n, 16-byte writes
__global__ __launch_bounds__(256) void pack_w_kernel(const float *__restrict__ W, int K, int N,
                                                    unsigned short *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * 8;
    if (n >= N) return;
    union { unsigned short h[8]; uint4 v; } u;
#pragma unroll
    for (int j = 0; j < 8; ++j) u.h[j] = f32_to_bf16_rne(W[(size_t)(k0 + j) * N + n]);
    *(uint4 *)(out + (size_t)n * K + k0) = u.v;
}

// out[n][k0..k0+7] = W[k0..k0+7][n]
__global__ __launch_bounds__(256) void transpose_w_kernel(const float *__restrict__ W, int K, int N,
                                                         float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * 8;
    if (n >= N) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(size_t)(k0 + j) * N + n];
    *(float4 *)(out + (size_t)n * K + k0) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4 *)(out + (size_t)n * K + k0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// out[m][k] = bf16(x[m][k] - b_dec[k]); 8 elements per thread
__global__ __launch_bounds__(256) void convert_x_kernel(const float *__restrict__ x, const float *__restrict__ b_dec,
                                                       long long n8, int K8, unsigned short *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int k0 = (int)(i % K8) * 8;
    const float4 a0 = *(const float4 *)(x + i * 8), a1 = *(const float4 *)(x + i * 8 + 4);
    const float4 b0 = *(const float4 *)(b_dec + k0), b1 = *(const float4 *)(b_dec + k0 + 4);
    union { unsigned short h[8]; uint4 v; } u;
    u.h[0] = f32_to_bf16_rne(a0.x - b0.x); u.h[1] = f32_to_bf16_rne(a0.y - b0.y);
    u.h[2] = f32_to_bf16_rne(a0.z - b0.z); u.h[3] = f32_to_bf16_rne(a0.w - b0.w);
    u.h[4] = f32_to_bf16_rne(a1.x - b1.x); u.h[5] = f32_to_bf16_rne(a1.y - b1.y);
    u.h[6] = f32_to_bf16_rne(a1.z - b1.z); u.h[7] = f32_to_bf16_rne(a1.w - b1.w);
    *(uint4 *)(out + i * 8) = u.v;
}

// the same for several (x, b_dec, out) triples of one shape in one launch: grid.y = the triple
struct ConvertGroupArgs {
    const float *x[32], *b_dec[32];
    unsigned short *out[32];
    long long n8;
    int K8;
};
__global__ __launch_bounds__(256) void convert_x_group_kernel(const ConvertGroupArgs g) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n8) return;
    const int p = blockIdx.y;  // uniform
    const float *x = g.x[p], *b_dec = g.b_dec[p];
    const int k0 = (int)(i % g.K8) * 8;
    const float4 a0 = *(const float4 *)(x + i * 8), a1 = *(const float4 *)(x + i * 8 + 4);
    const float4 b0 = *(const float4 *)(b_dec + k0), b1 = *(const float4 *)(b_dec + k0 + 4);
    union { unsigned short h[8]; uint4 v; } u;
    u.h[0] = f32_to_bf16_rne(a0.x - b0.x); u.h[1] = f32_to_bf16_rne(a0.y - b0.y);
    u.h[2] = f32_to_bf16_rne(a0.z - b0.z); u.h[3] = f32_to_bf16_rne(a0.w - b0.w);
    u.h[4] = f32_to_bf16_rne(a1.x - b1.x); u.h[5] = f32_to_bf16_rne(a1.y - b1.y);
    u.h[6] = f32_to_bf16_rne(a1.z - b1.z); u.h[7] = f32_to_bf16_rne(a1.w - b1.w);
    *(uint4 *)(g.out[p] + i * 8) = u.v;
}

// Column side of the screen's margin, summed in fp64 and rounded UP (they scale an error bound):
//   wnorm[n]     = max(||W[:, n]||_2, ||bf16(W[:, n])||_2)
//   wnorm[N + n] = ||W[:, n] - bf16(W[:, n])||_2      -- the ACTUAL rounding error of the column, not its worst case
__global__ __launch_bounds__(256) void col_norm_kernel(const float *__restrict__ W, int K, int N, float *__restrict__ wnorm) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, st = 0.0, sd = 0.0;
    for (int k = 0; k < K; ++k) {
        const float wf = W[(size_t)k * N + n];
        const float wt = __uint_as_float((unsigned)f32_to_bf16_rne(wf) << 16);
        const double w = (double)wf, t = (double)wt, d = w - t;
        s += w * w;
        st += t * t;
        sd += d * d;
    }
    wnorm[n] = __double2float_ru(sqrt(s > st ? s : st) * (1.0 + 1e-6));
    wnorm[N + n] = __double2float_ru(sqrt(sd) * (1.0 + 1e-6));
}

// Row side of the bf16 screen's margin, one wave per row, maxima over the rows by an atomic on the floats' bits (the
// values are non-negative, so unsigned order is float order).  With xc = fl(x - b_dec), xt = bf16(xc) (what
// convert_x_kernel stores), wt = bf16(w):
//   sum_k xt_k wt_k - sum_k xc_k w_k = sum_k (xt_k - xc_k) wt_k + sum_k xc_k (wt_k - w_k)
//   |...| <= ||xt - xc|| ||wt|| + ||xc|| ||wt - w||                                   (Cauchy-Schwarz, twice)
// with the ACTUAL rounding errors of this row and of the column (a bf16 rounding loses up to 2^-8 relative, 2^-9.3 in
// the root mean square: the worst case over every operand, 2^-7 ||xc|| ||w||, is 2.4x this on ordinary data), plus the
// fp32 accumulation error of the exact kernel (which contracts the raw x against the folded bias) and of the screen
// itself, g/2 (||x|| ||w|| + ||b_dec|| ||w||) + g/2 ||xt|| ||wt||, g = 2 K 2^-24, and the 2^-24 of fl(x - b_dec):
//   bound[0] = max_m ( ||xt_m - xc_m|| + g ||xc_m|| + g (||x_m|| + ||b_dec||) )      multiplies wnorm[n]
//   bound[1] = max_m ||xc_m||                                                         multiplies wnorm[N + n]
// (both scaled by 1.0001 for the fp32 rounding of the norms themselves; the screen kernel adds g |b_enc[n]| per
// column for the bias both chains start from).  margin[n] = bound[0] wnorm[n] + bound[1] wnorm[N + n] + g |b_enc[n]|
// bounds |exact pre-activation - screened pre-activation| of column n for every row.
__global__ __launch_bounds__(256) void row_bound_kernel(const float *__restrict__ x, long long M, int K,
                                                       const float *__restrict__ b_dec, float *__restrict__ bound,
                                                       unsigned short *__restrict__ xt_out) {
    __shared__ float wave_max[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_waves = (long long)gridDim.x * 4;
    const float g = 2.f * (float)K * 5.9604645e-8f;
    const int chunks = K >> 2;
    float sb = 0.f;
    for (int c = lane; c < chunks; c += 64) {
        const float4 b = ((const float4 *)b_dec)[c];
        sb += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sb += __shfl_xor(sb, off);
    const float nb = sqrtf(sb);
    float best0 = 0.f, best1 = 0.f;
    auto rerr = [](float v) { return v - __uint_as_float((unsigned)f32_to_bf16_rne(v) << 16); };  // exact in fp32
    for (long long row = (long long)blockIdx.x * 4 + wave; row < M; row += n_waves) {
        float sc = 0.f, sr = 0.f, sd = 0.f;
        const float4 *xr = (const float4 *)(x + row * K);
        for (int c = lane; c < chunks; c += 64) {
            const float4 v = xr[c], b = ((const float4 *)b_dec)[c];
            const float dx = v.x - b.x, dy = v.y - b.y, dz = v.z - b.z, dw = v.w - b.w;
            const float ex = rerr(dx), ey = rerr(dy), ez = rerr(dz), ew = rerr(dw);
            if (xt_out) {  // the screen's x operand, rounded exactly as convert_x_kernel rounds it
                union { unsigned short h[4]; uint2 v; } u;
                u.h[0] = f32_to_bf16_rne(dx); u.h[1] = f32_to_bf16_rne(dy); u.h[2] = f32_to_bf16_rne(dz); u.h[3] = f32_to_bf16_rne(dw);
                *(uint2 *)(xt_out + row * K + 4 * c) = u.v;
            }
            sc += dx * dx + dy * dy + dz * dz + dw * dw;
            sd += ex * ex + ey * ey + ez * ez + ew * ew;
            sr += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sc += __shfl_xor(sc, off);
            sr += __shfl_xor(sr, off);
            sd += __shfl_xor(sd, off);
        }
        sd += (float)K * 1.2e-38f;  // squares that underflowed: each lost less than the smallest normal number
        const float nc = sqrtf(sc);
        const float r0 = (sqrtf(sd) + g * nc + g * (sqrtf(sr) + nb)) * 1.0001f, r1 = nc * 1.0001f;
        if (lane == 0) {  // the row's own pair: the epilogue re-tests a candidate with it (EncodeArgs::screen_rows)
            bound[2 + 2 * row] = r0;
            bound[3 + 2 * row] = r1;
        }
        best0 = fmaxf(best0, r0);
        best1 = fmaxf(best1, r1);
    }
    if (lane == 0) { wave_max[0][wave] = best0; wave_max[1][wave] = best1; }
    __syncthreads();
    if (threadIdx.x < 2) {  // non-negative floats order like their bit patterns
        const float *w = wave_max[threadIdx.x];
        atomicMax((unsigned *)bound + threadIdx.x, __float_as_uint(fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]))));
    }
}

}  // namespace

int tsc_launch_col_norms(const float *W, int K, int N, float *wnorm, hipStream_t stream) {
    hipLaunchKernelGGL(col_norm_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, W, K, N, wnorm);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_row_bound(const float *x, long long M, int K, const float *b_dec, float *bound, hipStream_t stream,
                         void *xt_out) {
    if (xt_out && M > TSC_BF16_WS_ROWS) return tsc_fail(TSC_ERR_UNSUPPORTED, "the bf16 staging buffer holds %d rows", (int)TSC_BF16_WS_ROWS);
    TSC_HIP_CHECK(hipMemsetAsync(bound, 0, 2 * sizeof(float), stream));  // the maxima; the per-row pairs are all written
    long long blocks = (M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(row_bound_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, M, K, b_dec, bound,
                       (unsigned short *)xt_out);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

size_t tsc_bf16_workspace_bytes(int K) { return (size_t)TSC_BF16_WS_ROWS * K * 2; }

int tsc_pack_w_bf16(const float *W, int K, int N, void *dst, hipStream_t stream) {
    if (K % 8) return tsc_fail(TSC_ERR_UNSUPPORTED, "bf16 pack needs d_in %% 8 == 0");
    hipLaunchKernelGGL(pack_w_kernel, dim3((N + 255) / 256, K / 8), dim3(256), 0, stream, W, K, N,
                       (unsigned short *)dst);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_transpose_w_f32(const float *W, int K, int N, float *dst, hipStream_t stream) {
    if (K % 8) return tsc_fail(TSC_ERR_UNSUPPORTED, "W_enc transpose needs d_in %% 8 == 0");
    hipLaunchKernelGGL(transpose_w_kernel, dim3((N + 255) / 256, K / 8), dim3(256), 0, stream, W, K, N, dst);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_convert_x_bf16(const float *x, const float *b_dec, long long rows, int K, void *x_bf16, hipStream_t stream) {
    if (rows <= 0) return TSC_OK;
    if (rows > TSC_BF16_WS_ROWS) return tsc_fail(TSC_ERR_UNSUPPORTED, "the bf16 staging buffer holds %d rows", (int)TSC_BF16_WS_ROWS);
    const long long n8 = rows * (K / 8);
    hipLaunchKernelGGL(convert_x_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, x, b_dec, n8, K / 8,
                       (unsigned short *)x_bf16);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_convert_x_bf16_group(int n, const float *const *x, const float *const *b_dec, void *const *x_bf16,
                                    long long rows, int K, hipStream_t stream) {
    if (rows <= 0 || n <= 0) return TSC_OK;
    if (n > 32) return tsc_fail(TSC_ERR_UNSUPPORTED, "at most 32 inputs per grouped conversion");
    if (rows > TSC_BF16_WS_ROWS) return tsc_fail(TSC_ERR_UNSUPPORTED, "the bf16 staging buffer holds %d rows", (int)TSC_BF16_WS_ROWS);
    ConvertGroupArgs g;
    memset(&g, 0, sizeof(g));
    for (int i = 0; i < n; ++i) { g.x[i] = x[i]; g.b_dec[i] = b_dec[i]; g.out[i] = (unsigned short *)x_bf16[i]; }
    g.n8 = rows * (K / 8);
    g.K8 = K / 8;
    hipLaunchKernelGGL(convert_x_group_kernel, dim3((unsigned)((g.n8 + 255) / 256), (unsigned)n), dim3(256), 0, stream, g);
    TSC_HIP_CHECK(hipGetLastError());
    return TSC_OK;
}

int tsc_launch_encode_bf16(int mode, const EncodeArgs &a, const void *W_bf16, void *x_bf16, hipStream_t stream) {
    // (a.screen_wnorm / a.screen_bound travel with the copy below: the bf16 SPARSE kernel then emits candidates)
    if (!W_bf16 || !x_bf16) return tsc_fail(TSC_ERR_ARG, "bf16 handle without its operand buffers");
    for (long long r0 = 0; r0 < a.M; r0 += TSC_BF16_WS_ROWS) {
        const long long rows = a.M - r0 < TSC_BF16_WS_ROWS ? a.M - r0 : TSC_BF16_WS_ROWS;
        const long long n8 = rows * (a.K / 8);
        if (!(a.x_staged && a.M <= TSC_BF16_WS_ROWS)) {
            hipLaunchKernelGGL(convert_x_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream,
                               a.x + r0 * a.K, a.b_dec, n8, a.K / 8, (unsigned short *)x_bf16);
            TSC_HIP_CHECK(hipGetLastError());
        }
        EncodeArgs c = a;
        c.bf16 = 1;
        c.x = (const float *)x_bf16;
        c.Wt = W_bf16;
        c.bias = a.b_enc;  // b_dec is already subtracted
        c.M = rows;
        c.token_base = a.token_base + r0;
        if (a.out) c.out = a.out + r0 * a.N;
        if (a.screen_rows) c.screen_rows = a.screen_rows + 2 * r0;
        if (a.sp_count) {
            c.sp_feat = a.sp_feat + r0 * a.max_active;
            c.sp_val = a.sp_val + r0 * a.max_active;
            c.sp_count = a.sp_count + r0;
        }
        const int rc = tsc_launch_encode_f32(mode, c, stream);
        if (rc != TSC_OK) return rc;
    }
    return TSC_OK;
}
