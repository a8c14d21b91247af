// v_mfma_f32_16x16x32_bf16: layout check (A = [16 m][32 k], B = [32 k][16 n]) and the issue rate of the planned
// conv K-step: MB*NB*6 MFMAs fed by (MB+NB)*3 ds_read_b128.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef float f4 __attribute__((ext_vector_type(4)));
inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__global__ void layout(const uint16_t* A, const uint16_t* B, float* D) {   // A [16][32], B [32][16] bf16 bits
    const int l = threadIdx.x;
    union { bf8 v; uint16_t s[8]; } a, b;
    for (int i = 0; i < 8; ++i) {
        a.s[i] = A[(l % 16) * 32 + 8 * (l / 16) + i];       // hypothesis: lane l holds A[m = l%16][k = 8*(l/16) + i]
        b.s[i] = B[(8 * (l / 16) + i) * 16 + l % 16];       //             lane l holds B[k = 8*(l/16) + i][n = l%16]
    }
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];   // D[m = 4*(l/16)+r][n = l%16]
}
template <int MB, int NB, int NT>
__global__ void __launch_bounds__(NT) rate(float* out, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    f4 acc[MB][NB];
    for (int m = 0; m < MB; ++m) for (int n = 0; n < NB; ++n) acc[m][n] = f4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        union U { uint4 u; bf8 v; };
        U a[MB][3], b[NB][3];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[m][p].u = lds[((it & 3) * 15 + m * 3 + p) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[n][p].u = lds[1024 + (lane & 15) * 3 + (lane >> 4) + ((it & 3) * 12 + n * 3 + p) * 50];
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb < 3 - pa; ++pb)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][pa].v, b[n][pb].v, acc[m][n], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < MB; ++m) for (int n = 0; n < NB; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MB, int NB, int NT> void run(const char* name) {
    const int blocks = 256, iters = 4000;
    float* out; (void)hipMalloc(&out, 4 * NT * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<MB, NB, NT><<<blocks, NT, 65536>>>(out, 50);
    (void)hipEventRecord(e0);
    rate<MB, NB, NT><<<blocks, NT, 65536>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * (NT / 64) * iters * 6.0 * MB * NB;
    printf("%-34s waves/SIMD=%d: %.3f ms  %.0f TF bf16 = %.0f TF fp32-equivalent (6 products)\n", name, NT / 256, ms,
           mfma * 16384 / ms / 1e9, mfma * 16384 / 6 / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    uint16_t hA[512], hB[512]; float hD[256]; uint16_t *A, *B; float* D;
    for (int i = 0; i < 512; ++i) { hA[i] = f2bf((float)(i % 7) - 3.f + 0.125f * (i / 37)); hB[i] = f2bf((float)((i * 5) % 11) - 5.f + 0.25f * (i / 53)); }
    (void)hipMalloc(&A, 1024); (void)hipMalloc(&B, 1024); (void)hipMalloc(&D, 1024);
    (void)hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(A, B, D);
    (void)hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double r = 0; for (int k = 0; k < 32; ++k) r += (double)bf2f(hA[m * 32 + k]) * bf2f(hB[k * 16 + n]);
        err = fmax(err, fabs(r - hD[m * 16 + n]));
    }
    printf("layout hypothesis max error: %g\n", err);
    run<5, 4, 256>("K-step MB=5 NB=4"); run<5, 4, 512>("K-step MB=5 NB=4");
    run<3, 4, 256>("K-step MB=3 NB=4"); run<4, 4, 256>("K-step MB=4 NB=4");
    run<2, 8, 256>("K-step MB=2 NB=8");
    return 0;
}
