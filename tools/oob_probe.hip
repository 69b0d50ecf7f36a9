// Diagnostic: what does a 16-byte (and 8-byte) raw buffer load return when only PART of it
// lies inside the descriptor's num_records?  The input matrices are read with 16-byte loads
// whose last one may run past the end of the tensor; if the range check is per dword, a
// descriptor sized to the tensor masks the tail by itself and the caller owes no slack.
// The probe never touches unmapped memory: the descriptor is SHORTER than the allocation.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, int nrec_bytes, float* out4, float* out2, float* dst, int dst_bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nrec_bytes, 0x00020000);
    const int t = threadIdx.x;
    const unsigned off = (unsigned)t * 4u;   // lane t starts at float t: every overlap with the end
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    const f32x2 w = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    for (int i = 0; i < 4; ++i) out4[t * 4 + i] = v[i];
    for (int i = 0; i < 2; ++i) out2[t * 2 + i] = w[i];
    // stores: a 16-byte store that straddles the end of its descriptor
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, dst_bytes, 0x00020000);
    if (t == 0) {
        const f32x4 s = {1.f, 2.f, 3.f, 4.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), rd, (unsigned)(dst_bytes - 8), 0, 0);
    }
}
int main() {
    const int total = 128, nrec = 37;   // descriptor covers floats [0, 37)
    float h[total];
    for (int i = 0; i < total; ++i) h[i] = (float)(i + 1);
    float *a, *o4, *o2, *dst;
    hipMalloc(&a, sizeof(h)); hipMalloc(&o4, 64 * 4 * 4); hipMalloc(&o2, 64 * 2 * 4); hipMalloc(&dst, 64 * 4);
    hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(dst, 0, 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, nrec * 4, o4, o2, dst, 32);
    float r4[256], r2[128], d[16];
    hipMemcpy(r4, o4, sizeof(r4), hipMemcpyDeviceToHost);
    hipMemcpy(r2, o2, sizeof(r2), hipMemcpyDeviceToHost);
    hipMemcpy(d, dst, sizeof(d), hipMemcpyDeviceToHost);
    int per_dword = 1, whole = 1;
    for (int t = 0; t < 64; ++t)
        for (int i = 0; i < 4; ++i) {
            const int idx = t + i;
            const float pd = idx < nrec ? h[idx] : 0.f;              // per-dword check
            const float wh = t + 3 < nrec ? h[idx] : 0.f;            // all-or-nothing check
            if (r4[t * 4 + i] != pd) per_dword = 0;
            if (r4[t * 4 + i] != wh) whole = 0;
        }
    printf("dwordx4 straddling num_records: %s\n", per_dword ? "PER-DWORD (in-range dwords real, the rest 0)"
                                                  : whole ? "ALL-OR-NOTHING" : "OTHER");
    for (int t = 33; t < 39; ++t)
        printf("  lane %d (floats %d..%d, end %d): %g %g %g %g\n", t, t, t + 3, nrec, r4[t * 4], r4[t * 4 + 1],
               r4[t * 4 + 2], r4[t * 4 + 3]);
    int pd2 = 1;
    for (int t = 0; t < 64; ++t)
        for (int i = 0; i < 2; ++i)
            if (r2[t * 2 + i] != (t + i < nrec ? h[t + i] : 0.f)) pd2 = 0;
    printf("dwordx2 straddling num_records: %s\n", pd2 ? "PER-DWORD" : "NOT per-dword");
    printf("16-byte store at dst_bytes-8 (descriptor 32 B): words 5..9 = %g %g %g %g %g  (per-dword: 0 1 2 0 0)\n",
           d[5], d[6], d[7], d[8], d[9]);
    return 0;
}
