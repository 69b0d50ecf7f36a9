// Diagnostic: what does it cost a kernel that needs its argument block FIRST THING (tables ->
// pointer -> first load: the shape of k_fused's producers and of k_wgrad's tiles) that the block
// arrives through the kernarg segment -- written by the host a moment ago, so every first touch
// of a line misses all caches -- rather than sitting in a device buffer that the previous step's
// launch has already pulled into the L2s?  (tools/kernarg_size_probe.hip reads its arguments at
// the END of a spinning kernel: the miss hides behind the spin there, which is not what the real
// kernels do.)  Two launches per "step" as in training: 208 x 1024 threads with ~10 KB, then
// 203 x 512 threads with ~10 KB; each block: args -> table entry -> pointer -> load -> spin -> store.
//   hipcc --offload-arch=gfx950 -O2 -o tools/kernarg_resident_probe tools/kernarg_resident_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int kWords = 2560;   // 10 KB
struct Big {
    int table[kWords - 4];
    const int* src;
    int pad[2];
};
template <int T>
__device__ __forceinline__ void body(const Big& b, int* out, int ticks) {
    // one word of every 64-byte line, requested together (the real kernels' prefetch), then the chain
    const int wave = threadIdx.x >> 6;
    unsigned sink = 0;
    for (int line = wave; line < (int)(sizeof(Big) / 64); line += T / 64) sink |= ((const unsigned*)&b)[line * 16];
    const int idx = b.table[(blockIdx.x * 7) % (kWords - 4)];   // "which tile am I"
    const int v = b.src[(idx + threadIdx.x) & 1023];             // "my first operand load"
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
    if (v == 12345 || sink == 0xdeadbeefu) out[0] = v;
}
template <int T> __global__ __launch_bounds__(T) void k_byvalue(const Big b, int* out, int ticks) { body<T>(b, out, ticks); }
template <int T> __global__ __launch_bounds__(T) void k_resident(const Big* __restrict__ b, int* out, int ticks) {
    body<T>(*b, out, ticks);
}
// the shape of a launch that CARRIES the block by value as well but never reads it (head +
// resident pointer in front): is it the bytes passed or the bytes read that cost?
template <int T> __global__ __launch_bounds__(T) void k_both(const Big* __restrict__ b, const Big unused, int* out, int ticks) {
    body<T>(*b, out, ticks);
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
int main() {
    int *out, *src;
    Big* dev;
    hipMalloc(&out, 4);
    hipMalloc(&src, 4096);
    hipMemset(src, 0, 4096);
    hipMalloc(&dev, sizeof(Big));
    static Big big;
    for (int i = 0; i < kWords - 4; ++i) big.table[i] = i & 1023;
    big.src = src;
    hipMemcpy(dev, &big, sizeof(Big), hipMemcpyHostToDevice);
    printf("us per step (two launches back to back on one stream), arguments needed first thing:\n");
    for (int rep = 0; rep < 3; ++rep)
        for (int ticks : {0, 300, 1500}) {   // 0 / 3 / 15 us of 'work' per block (100 MHz ticks)
            const int t2 = ticks / 3;
            const float a = timeit([&] {
                hipLaunchKernelGGL((k_byvalue<1024>), dim3(208), dim3(1024), 0, 0, big, out, ticks);
                hipLaunchKernelGGL((k_byvalue<512>), dim3(203), dim3(512), 0, 0, big, out, t2);
            });
            const float b = timeit([&] {
                hipLaunchKernelGGL((k_resident<1024>), dim3(208), dim3(1024), 0, 0, (const Big*)dev, out, ticks);
                hipLaunchKernelGGL((k_resident<512>), dim3(203), dim3(512), 0, 0, (const Big*)dev, out, t2);
            });
            const float c = timeit([&] {
                hipLaunchKernelGGL((k_both<1024>), dim3(208), dim3(1024), 0, 0, (const Big*)dev, big, out, ticks);
                hipLaunchKernelGGL((k_both<512>), dim3(203), dim3(512), 0, 0, (const Big*)dev, big, out, t2);
            });
            printf("  work %4.1f + %4.1f us: 10 KB by value %6.2f us | resident block + pointer %6.2f us | saves %5.2f us per step"
                   " | pointer + 10 KB passed but not read %6.2f us\n",
                   ticks / 100.0, t2 / 100.0, a, b, a - b, c);
        }
    return 0;
}
