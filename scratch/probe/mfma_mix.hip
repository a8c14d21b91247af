// How fast does one wave per SIMD issue 4x4x1 MFMAs when LDS reads / VALU work are interleaved?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NDS, int NVALU, int NACC, int NT>
__global__ void __launch_bounds__(NT) k(float* out, unsigned long long* ts, int iters, float a0) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = 0.001f * i;
    __syncthreads();
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
    float opa[5], opb[9];
    for (int i = 0; i < 5; ++i) opa[i] = a0 + i;
    for (int i = 0; i < 9; ++i) opb[i] = a0 * i;
    const float* lp = lds + (threadIdx.x & 63) * 9 % 1024;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        float na[5], nb[9];
        if (NDS > 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) na[i] = i < NDS ? lp[i * 264 + (it & 15) * 16] : opa[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) nb[i] = (5 + i) < NDS ? lp[1500 + i * 67 + (it & 15)] : opb[i];
        }
#pragma unroll
        for (int i = 0; i < NVALU; ++i) v[i & 7] = fmaf(v[i & 7], 1.0001f, v[(i + 1) & 7]);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int idx = (c * 9 + t) % NACC;
                acc[idx] = __builtin_amdgcn_mfma_f32_4x4x1f32(opa[c], opb[t], acc[idx], 0, 0, 0);
            }
        if (NDS > 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) opa[i] = na[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) opb[i] = nb[i];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ts[0] = t1 - t0;
}
template <int NDS, int NVALU, int NACC, int NT> void run(const char* name, int blocks) {
    float* out; unsigned long long* ts;
    (void)hipMalloc(&out, 4 * NT * blocks); (void)hipMalloc(&ts, 8);
    int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NDS, NVALU, NACC, NT><<<blocks, NT>>>(out, ts, 100, 1.f);
    (void)hipEventRecord(e0);
    k<NDS, NVALU, NACC, NT><<<blocks, NT>>>(out, ts, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; (void)hipMemcpy(&t, ts, 8, hipMemcpyDeviceToHost);
    double mf = (double)iters * 45;
    printf("%-28s NT=%d blocks=%4d: %.3f ms  ticks/MFMA %.2f  tick rate %.2f GHz  %.1f TF\n", name, NT, blocks, ms, t / mf,
           t / (ms * 1e6), (double)blocks * (NT / 64) * mf * 512 / ms / 1e9);
    (void)hipFree(out); (void)hipFree(ts);
}
int main() {
    run<0, 0, 45, 256>("mfma only", 256);
    run<0, 0, 45, 512>("mfma only", 256);
    run<14, 30, 45, 256>("mfma + 14 ds + 30 valu", 256);
    run<14, 30, 45, 512>("mfma + 14 ds + 30 valu", 256);
    run<14, 30, 27, 512>("27acc: mfma + 14 ds + 30 valu", 256);
    run<14, 0, 45, 512>("mfma + 14 ds", 256);
    run<0, 45, 45, 512>("mfma + 45 valu", 256);
    run<0, 90, 45, 512>("mfma + 90 valu", 256);
    run<14, 30, 45, 1024>("mfma + 14 ds + 30 valu", 256);
    return 0;
}
