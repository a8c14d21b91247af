// v_mfma_f32_16x16x4f32: operand layout check + how much VALU / LDS work hides under its 32-cycle issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void layout(const float* A, const float* B, float* D) {   // A [16][4], B [4][16], D [16][16]
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];       // hypothesis: lane l holds A[m = l%16][k = l/16]
    const float b = B[(l / 16) * 16 + l % 16];      //             lane l holds B[k = l/16][n = l%16]
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];   // hypothesis: D[4*(l/16)+r][l%16]
}
template <int MODE, int NT>
__global__ void __launch_bounds__(NT) rate(float* out, int iters, float a0) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += NT) lds[i] = 0.001f * i;
    __syncthreads();
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
    float opa[4], opb[4];
    for (int i = 0; i < 4; ++i) { opa[i] = a0 + i; opb[i] = a0 * i; }
    const float* lp = lds + (threadIdx.x & 63);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        float na[4], nb[4];
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { na[i] = lp[i * 80 + (it & 7) * 64]; nb[i] = lp[1024 + i * 72 + (it & 7) * 64]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[i], opb[j], acc[i * 4 + j], 0, 0, 0);
                if (MODE == 1 || MODE == 3) {            // 2 VALU per MFMA, interleaved
                    v[(i * 4 + j) & 7] = fmaf(v[(i * 4 + j) & 7], 1.0001f, 0.5f);
                    v[(i * 4 + j + 3) & 7] = fmaf(v[(i * 4 + j + 3) & 7], 0.9999f, 0.25f);
                }
            }
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { opa[i] = na[i]; opb[i] = nb[i]; }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MODE, int NT> void run(const char* name) {
    const int blocks = 256, iters = 20000;
    float* out; (void)hipMalloc(&out, 4 * NT * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<MODE, NT><<<blocks, NT>>>(out, 100, 1.f);
    (void)hipEventRecord(e0);
    rate<MODE, NT><<<blocks, NT>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s waves/SIMD=%d: %.3f ms  %.1f TF\n", name, NT / 256, ms, (double)blocks * (NT / 64) * iters * 16.0 * 2048 / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    float hA[64], hB[64], hD[256], *A, *B, *D;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(i % 7) - 3.f + 0.125f * (i / 7); hB[i] = (float)((i * 5) % 11) - 5.f; }
    (void)hipMalloc(&A, 256); (void)hipMalloc(&B, 256); (void)hipMalloc(&D, 1024);
    (void)hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(A, B, D);
    (void)hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double r = 0; for (int k = 0; k < 4; ++k) r += (double)hA[m * 4 + k] * hB[k * 16 + n];
        err = fmax(err, fabs(r - hD[m * 16 + n]));
    }
    printf("layout hypothesis max error: %g\n", err);
    run<0, 256>("16 mfma16x16x4"); run<0, 512>("16 mfma16x16x4");
    run<1, 256>("16 mfma + 32 valu interleaved"); run<1, 512>("16 mfma + 32 valu interleaved");
    run<2, 256>("16 mfma + 8 ds_read"); run<2, 512>("16 mfma + 8 ds_read");
    run<3, 256>("16 mfma + 8 ds_read + 32 valu"); run<3, 512>("16 mfma + 8 ds_read + 32 valu");
    return 0;
}
