// Diagnostic: operand layout of v_mfma_f32_16x16x16_bf16 (the _1k form) on gfx950, as the bf16-operand
// variant of k_linear_big assumes it: lane (c = l & 15, q = l >> 4) supplies A[row c][k = 4 q .. 4 q + 3]
// and B[k = 4 q .. 4 q + 3][col c] as four bf16 each; D[row 4 q + r][col c] in register r.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ void probe(const float* A, const float* B, float* D) {   // A (16, 16) row-major [row][k], B (16, 16) [k][col]
    const int l = threadIdx.x, c = l & 15, q = l >> 4;
    s4 a, b;
    for (int i = 0; i < 4; ++i) {
        a[i] = (short)bf16_rne(A[c * 16 + 4 * q + i]);
        b[i] = (short)bf16_rne(B[(4 * q + i) * 16 + c]);
    }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + c] = acc[r];
}
static float rne(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; memcpy(&f, &u, 4); return f; }
int main() {
    float hA[256], hB[256], hD[256];
    for (int i = 0; i < 256; ++i) { hA[i] = sinf(0.37f * i) * 1.7f; hB[i] = cosf(0.11f * i + 1.f) * 0.9f; }
    float *A, *B, *D;
    hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D, 1024);
    hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, D);
    hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += (double)rne(hA[r * 16 + k]) * (double)rne(hB[k * 16 + c]);
            worst = fmax(worst, fabs(s - hD[r * 16 + c]));
        }
    printf("max |D - sum_k bf16(A) bf16(B)| = %.3g (layout %s)\n", worst, worst < 1e-5 ? "as assumed" : "DIFFERENT");
    return worst < 1e-5 ? 0 : 1;
}
