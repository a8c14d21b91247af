// Does v_mfma_f32_16x16x32_f16 keep fp16 DENORMAL inputs (|x| < 2^-14) or flush them to zero?  Decides whether a
// two-part fp16 operand split (a = a1 + a2, a2 ~ 2^-11 a) needs pre-scaling.  Also checks v_cvt to fp16 (RNE, denormal
// results) and the operand layout of the f16 form (same as bf16: lane l holds row l%16, k = 8 (l/16) + i).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* av, const float* bv, float* D, float* conv) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)av[(l % 16) * 32 + 8 * (l / 16) + i];
        b[i] = (_Float16)bv[(8 * (l / 16) + i) * 16 + l % 16];
    }
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
    if (l < 8) conv[l] = (float)a[l];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], hc[8];
    // row m of A: all 32 entries = 2^-(10 + m)  (rows 5.. are fp16 denormals: 2^-15 .. 2^-25); B = 1 -> D[m][n] = 32 * 2^-(10+m)
    for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) hA[m * 32 + k] = ldexpf(1.f, -(10 + m));
    for (int i = 0; i < 512; ++i) hB[i] = 1.f;
    float *dA, *dB, *dD, *dc;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD)); hipMalloc(&dc, sizeof(hc));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dc);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m)
        printf("A = 2^-%d (%s): D = %.6e, expected %.6e -> %s\n", 10 + m, (10 + m) > 14 ? "fp16 denormal" : "normal", hD[m * 16],
               32.0 * ldexp(1.0, -(10 + m)), fabs(hD[m * 16] - 32.0 * ldexp(1.0, -(10 + m))) < 1e-12 ? "kept" : (hD[m * 16] == 0.f ? "FLUSHED" : "other"));
    // denormal on the B side too
    for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) hA[m * 32 + k] = 1.f;
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) hB[k * 16 + n] = ldexpf(1.f, -(10 + n));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dc);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    for (int n = 0; n < 16; ++n) printf("B = 2^-%d: D = %.6e (%s)\n", 10 + n, hD[n], fabs(hD[n] - 32.0 * ldexp(1.0, -(10 + n))) < 1e-12 ? "kept" : "FLUSHED/other");
    return 0;
}
