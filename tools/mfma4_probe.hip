// Diagnostic: register layout of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, K = 1).
// Lane l gives A = 1000*l + 1 ... we decode which (lane of A, lane of B) pair each output holds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A carries the lane id in its value, B = 1: D = sum_k A*B = A of the contributing lane
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, c, 0, 0, 0);
    f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) { out[l * 8 + i] = d1[i]; out[l * 8 + 4 + i] = d2[i]; }
}
int main() {
    float* o; hipMalloc(&o, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    float h[512]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            // hypothesis: D[vgpr i][lane l] = A(lane 4*(l/4)+i) * B(lane l)
            if (h[l * 8 + i] != (float)(4 * (l / 4) + i)) ok = 0;
            if (h[l * 8 + 4 + i] != (float)l) ok = 0;
        }
    printf("v_mfma_f32_4x4x1_16B_f32: D[vgpr i][lane l] = A[lane 4*(l/4)+i] * B[lane l]: %s\n", ok ? "CONFIRMED" : "NO");
    for (int l = 0; l < 8; ++l)
        printf("  lane %d: A-lanes %g %g %g %g | B-lanes %g %g %g %g\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    return 0;
}
