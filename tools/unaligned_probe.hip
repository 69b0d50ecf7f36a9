// Diagnostic: are 4-byte-aligned buffer_load_dwordx4 / global dwordx4 loads and stores exact?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, int n, float* out, float* st) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n * 4, 0x00020000);
    int t = threadIdx.x;
    unsigned off = (unsigned)(t * 7) * 4u;   // rows of 7 floats: every alignment class
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    for (int i = 0; i < 4; ++i) out[t * 4 + i] = v[i];
    *reinterpret_cast<f32x4*>(st + t * 5 + 1) = v;   // unaligned 16-byte store
}
int main() {
    const int n = 64 * 7 + 2;
    float h[n + 8];
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *a, *out, *st;
    hipMalloc(&a, sizeof(h)); hipMalloc(&out, 64 * 4 * 4); hipMalloc(&st, 64 * 8 * 4);
    hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
    hipMemset(st, 0, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, n, out, st);
    float o[256], s[512];
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    hipMemcpy(s, st, sizeof(s), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t)
        for (int i = 0; i < 4; ++i) {
            int idx = t * 7 + i;
            float want = idx < n ? (float)idx : 0.f;
            if (o[t * 4 + i] != want) { if (bad < 8) printf("t%d i%d got %g want %g\n", t, i, o[t*4+i], want); ++bad; }
        }
    printf("unaligned buffer_load_dwordx4: %s (%d bad)\n", bad ? "WRONG" : "exact", bad);
    int bads = 0;
    for (int i = 0; i < 4; ++i) if (s[63 * 5 + 1 + i] != o[63 * 4 + i]) ++bads;
    printf("unaligned global store dwordx4: %s\n", bads ? "WRONG" : "exact");
    return 0;
}
