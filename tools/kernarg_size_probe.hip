// Diagnostic: cost of a launch as a function of its kernel-argument bytes, for the two
// geometries of the training step (208 x 1024 threads, 203 x 256 threads).  The kernel reads
// one word per workgroup from the arguments (as a real kernel reads its tables).
//   hipcc --offload-arch=gfx950 -O2 -o tools/kernarg_size_probe tools/kernarg_size_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { int v[N]; };
template <int N, int T> __global__ __launch_bounds__(T) void k_touch(Big<N> b, int* out) {
    if (threadIdx.x == 0 && b.v[blockIdx.x % N] == 12345) out[0] = 1;
}
// the same with kernels that RUN for a while (a launch's argument handling might hide behind
// the previous kernel): every workgroup spins `ticks` of the 100 MHz clock
template <int N, int T> __global__ __launch_bounds__(T) void k_spin(Big<N> b, int* out, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && b.v[blockIdx.x % N] == 12345) out[0] = 1;
}
template <class F> float timeit(F f, int n = 3000) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 300; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}
template <int N> void one(int* out) {
    static Big<N> big;
    for (int i = 0; i < N; ++i) big.v[i] = i;
    const float a = timeit([&] { hipLaunchKernelGGL((k_touch<N, 1024>), dim3(208), dim3(1024), 0, 0, big, out); });
    const float b = timeit([&] { hipLaunchKernelGGL((k_touch<N, 256>), dim3(203), dim3(256), 0, 0, big, out); });
    const float c = timeit([&] {
        hipLaunchKernelGGL((k_touch<N, 1024>), dim3(208), dim3(1024), 0, 0, big, out);
        hipLaunchKernelGGL((k_touch<N, 256>), dim3(203), dim3(256), 0, 0, big, out);
    });
    const float d = timeit([&] {
        hipLaunchKernelGGL((k_spin<N, 1024>), dim3(208), dim3(1024), 0, 0, big, out, 2000);   // 20 us
        hipLaunchKernelGGL((k_spin<N, 256>), dim3(203), dim3(256), 0, 0, big, out, 400);     // 4 us
    });
    printf("  %5d B of arguments: 208 x 1024 thr %5.2f us | 203 x 256 thr %5.2f us | the pair, alternating %5.2f us | "
           "the pair spinning 20 + 4 us %5.2f us\n", N * 4, a, b, c, d);
}
int main() {
    int* out; hipMalloc(&out, 4);
    printf("us per launch, back to back on one stream:\n");
    for (int rep = 0; rep < 2; ++rep) {
        one<16>(out); one<32>(out); one<64>(out); one<256>(out); one<2560>(out);
    }
    return 0;
}
