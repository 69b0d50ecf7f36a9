// Diagnostic: what an EMPTY launch costs on this chip as a function of its geometry --
// workgroups, threads per workgroup, dynamic LDS per workgroup, kernel-argument bytes.
// Back-to-back launches on one stream, timed with HIP events (steady-state cost per launch).
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[1800]; };   // 7.2 KB of kernel arguments
template <int T> __global__ __launch_bounds__(T) void k_empty(int x) { (void)x; }
template <int T> __global__ __launch_bounds__(T) void k_empty_big(Big b) { (void)b; }
template <int T> __global__ __launch_bounds__(T) void k_touch(Big b, int* out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0 && b.v[blockIdx.x % 1800] == 12345) out[0] = (int)lds[0];
}
template <class F> float timeit(F f, int n = 2000) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}
int main() {
    int* out; hipMalloc(&out, 4);
    Big big; for (int i = 0; i < 1800; ++i) big.v[i] = i;
    hipFuncSetAttribute((const void*)k_touch<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_touch<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("us per launch (back to back):\n");
    const int grids[] = {16, 64, 144, 224, 256, 512};
    for (int g : grids) {
        printf("  %3d WG x 1024 thr, no LDS, 4 B args      %6.2f\n", g, timeit([&] { hipLaunchKernelGGL(k_empty<1024>, dim3(g), dim3(1024), 0, 0, 1); }));
        printf("  %3d WG x 1024 thr, no LDS, 7.2 KB args   %6.2f\n", g, timeit([&] { hipLaunchKernelGGL(k_empty_big<1024>, dim3(g), dim3(1024), 0, 0, big); }));
        printf("  %3d WG x 1024 thr, 120 KB LDS, 7.2 KB    %6.2f\n", g, timeit([&] { hipLaunchKernelGGL(k_touch<1024>, dim3(g), dim3(1024), 120 * 1024, 0, big, out); }));
        printf("  %3d WG x 1024 thr, 36 KB LDS, 7.2 KB     %6.2f\n", g, timeit([&] { hipLaunchKernelGGL(k_touch<1024>, dim3(g), dim3(1024), 36 * 1024, 0, big, out); }));
        printf("  %3d WG x  256 thr, 36 KB LDS, 7.2 KB     %6.2f\n", 4 * g, timeit([&] { hipLaunchKernelGGL(k_touch<256>, dim3(4 * g), dim3(256), 36 * 1024, 0, big, out); }));
        printf("  %3d WG x  256 thr, no LDS, 4 B args      %6.2f\n", 4 * g, timeit([&] { hipLaunchKernelGGL(k_empty<256>, dim3(4 * g), dim3(256), 0, 0, 1); }));
    }
    return 0;
}
