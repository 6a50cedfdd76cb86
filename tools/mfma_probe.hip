// Micro-probe: how close to the fp32 MFMA issue rate do loops of increasing realism get on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o exp/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void probe(float *out, const float *in, int iters, unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 1], b0 = in[tid + 2], b1 = in[tid + 3];
    f32x4 fa = {a0, a1, b0, b1}, fb = fa;
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT >= 1) {  // LDS fragment reads at the GEMM's ratio: 2 b128 + 8 b32 per 16 MFMA
            const int o = (it & 63) * 64;
            fa = *(const f32x4 *)&lds[(o + tid * 4) & 16380];
            fb = *(const f32x4 *)&lds[(o + 2048 + tid * 4) & 16380];
#pragma unroll
            for (int s = 0; s < 4; ++s) { a0 += lds[(o + s * 128 + tid) & 16383]; b0 += lds[(o + 4096 + s * 128 + tid) & 16383]; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[3], 0, 0, 0);
            if (VARIANT >= 2) {  // ~3 VALU per MFMA
#pragma unroll
                for (int v = 0; v < 12; ++v) sink = sink * 1.0001f + a1;
            }
        }
        if (VARIANT >= 3 && (it & 3) == 3) __syncthreads();  // one barrier per 64 MFMA
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && stamps) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    float s = sink + a0 + b0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char *name, int blocks, float *out, float *in) {
    const int iters = 16384;
    static unsigned long long *stamps = nullptr;
    if (!stamps) hipMalloc(&stamps, 4096 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(out, in, 64, nullptr);
    hipEventRecord(e0);
    probe<V><<<blocks, 256>>>(out, in, iters, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 16 * 4096.0;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, 16 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz[blocks / 2];
    printf("%-34s blocks=%4d  %.3f ms  %.1f TFLOP/s  clock %.3f GHz  -> %.1f%% of issue rate at that clock\n", name, blocks, ms,
           flop / ms / 1e9, clk, 100.0 * (flop / ms / 1e9) / (65.536 * clk));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&in, 4096 * 4);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i % 17) * 0.01f - 0.05f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<0>("mfma only", blocks, out, in);
        run<1>("+ LDS fragment reads", blocks, out, in);
        run<2>("+ LDS + 3 VALU/MFMA", blocks, out, in);
        run<3>("+ LDS + VALU + barrier/64 MFMA", blocks, out, in);
    }
    return 0;
}
