// Diagnostic micro-benchmark (not part of the product path): cycles per
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 under different numbers of
// independent accumulators and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void probe16(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ void probe32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int nacc, int threads, float* out,
         unsigned long long* cyc) {
    const int iters = 1024;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-10s acc=%d waves/SIMD=%d : %.1f cycles per MFMA per wave (%.1f per SIMD)\n", name,
           nacc, threads / 256, (double)c / (iters * nacc),
           (double)c / (iters * nacc) / (threads / 256));
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&cyc, 64);
    for (int threads : {256, 512, 1024}) {
        run("16x16x4", probe16<1>, 1, threads, out, cyc);
        run("16x16x4", probe16<2>, 2, threads, out, cyc);
        run("16x16x4", probe16<4>, 4, threads, out, cyc);
        run("32x32x2", probe32<1>, 1, threads, out, cyc);
        run("32x32x2", probe32<2>, 2, threads, out, cyc);
    }
    return 0;
}
