// Probe: does a 12 KB by-value kernel argument block launch and read back correctly on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int a[3000]; };  // 12 KB
__global__ void k(Big b_by_value, int* out) {
    const Big& b = *(const Big*)__builtin_amdgcn_kernarg_segment_ptr();
    int s = 0;
    for (int i = threadIdx.x; i < 3000; i += 64) s += b.a[i];
    atomicAdd(out, s);
}
int main() {
    Big b;
    long want = 0;
    for (int i = 0; i < 3000; ++i) { b.a[i] = i * 3 + 1; want += b.a[i]; }
    int* o;
    hipMalloc(&o, 4);
    hipMemset(o, 0, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, b, o);
    hipError_t e = hipDeviceSynchronize();
    int got = 0;
    hipMemcpy(&got, o, 4, hipMemcpyDeviceToHost);
    printf("launch %s; sum %d, want %ld\n", hipGetErrorString(e), got, want);
    return got == want ? 0 : 1;
}
