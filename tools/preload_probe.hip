// Probe: kernel-argument preload (-mllvm -amdgpu-kernarg-preload-count=N) on gfx950: does it
// run, and how much sooner does a kernel's first global load return when the pointer it
// needs arrives in SGPRs instead of through a (cold) scalar load of the kernarg segment?
// Build twice (with / without the flag) and compare the printed medians.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
__global__ void k(const float* p, unsigned* out, float* sink) {
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    const float v = p[threadIdx.x + blockIdx.x * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 123.456f) sink[0] = v;
}
__global__ void dirty(float* p, int n) {   // something else runs in between (cold caches)
    for (int i = threadIdx.x + blockIdx.x * 256; i < n; i += 256 * gridDim.x) p[i] += 1.f;
}
int main() {
    const int B = 64, N = 1 << 22;
    float *p, *sink, *junk;
    unsigned* out;
    hipMalloc(&p, B * 64 * 4); hipMalloc(&sink, 4); hipMalloc(&out, B * 4); hipMalloc(&junk, N * 4);
    hipMemset(p, 0, B * 64 * 4); hipMemset(junk, 0, N * 4);
    std::vector<unsigned> h(B), all;
    for (int it = 0; it < 200; ++it) {
        hipLaunchKernelGGL(dirty, dim3(256), dim3(256), 0, 0, junk, N);
        hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, p, out, sink);
        hipMemcpy(h.data(), out, B * 4, hipMemcpyDeviceToHost);
        if (it >= 20) all.insert(all.end(), h.begin(), h.end());
    }
    std::sort(all.begin(), all.end());
    printf("entry -> first load back: median %.2f us, p10 %.2f us (100 MHz ticks)\n",
           all[all.size() / 2] / 100.0, all[all.size() / 10] / 100.0);
    return 0;
}
