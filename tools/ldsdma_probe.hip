// Probe (gfx950): buffer_load_dwordx4 ... lds -- where does lane l's 16 bytes land, and what
// does an out-of-range lane write?   hipcc --offload-arch=gfx950 -O2 -o tools/ldsdma_probe tools/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* __restrict__ src, float* dst, int n_valid) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) lds[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_valid * 4, 0x27000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + wave * 256), 16, (wave * 256 + lane * 4) * 4, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) dst[i] = lds[i];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n), out(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int valid : {1024, 1000}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, o, valid);
        hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, firstbad = -1;
        for (int i = 0; i < n; ++i) {
            const float want = i < valid ? (float)i : 0.f;
            if (out[i] != want) { if (firstbad < 0) firstbad = i; ++bad; }
        }
        printf("valid %d: mismatches %d first %d (out[%d]=%g) out[1000..1003]=%g %g %g %g\n", valid, bad, firstbad,
               firstbad < 0 ? 0 : firstbad, firstbad < 0 ? 0.f : out[firstbad], out[1000], out[1001], out[1002], out[1003]);
    }
    return 0;
}
