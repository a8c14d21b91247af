// Does a VALU / LDS instruction hide under a 4x4x1 MFMA when the two are finely interleaved (same wave / partner wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: 45 mfma; 1: 45 mfma + 45 valu interleaved 1:1; 2: 45 mfma then 45 valu (blocked); 3: 45 mfma + 15 ds_read interleaved 3:1
// 4: 45 mfma + 45 valu + 15 ds_read interleaved
template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k(float* out, int iters, float a0) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += NT) lds[i] = 0.001f * i;
    __syncthreads();
    f4 acc[45];
    for (int i = 0; i < 45; ++i) acc[i] = f4{0, 0, 0, 0};
    float v[9];
    for (int i = 0; i < 9; ++i) v[i] = a0 + i + threadIdx.x;
    float opa[5], opb[9];
    for (int i = 0; i < 5; ++i) opa[i] = a0 + i;
    for (int i = 0; i < 9; ++i) opb[i] = a0 * i;
    const float* lp = lds + (threadIdx.x & 63);
    float ld[15];
    for (int i = 0; i < 15; ++i) ld[i] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 15; ++i) ld[i] += lp[i * 72 + (it & 7) * 64];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c * 9 + t] = __builtin_amdgcn_mfma_f32_4x4x1f32(opa[c], opb[t], acc[c * 9 + t], 0, 0, 0);
            if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
                for (int c = 0; c < 5; ++c) v[t] = fmaf(v[t], 1.0001f, 0.5f);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 45; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0); }
        } else if (MODE == 2) {
            __builtin_amdgcn_sched_group_barrier(0x8, 45, 0); __builtin_amdgcn_sched_group_barrier(0x2, 45, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 15; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); }
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 15; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0); }
        }
    }
    float s = 0;
    for (int i = 0; i < 45; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 9; ++i) s += v[i];
    for (int i = 0; i < 15; ++i) s += ld[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MODE, int NT> void run(const char* name) {
    const int blocks = 256;
    float* out; (void)hipMalloc(&out, 4 * NT * blocks);
    int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, NT><<<blocks, NT>>>(out, 100, 1.f);
    (void)hipEventRecord(e0);
    k<MODE, NT><<<blocks, NT>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)iters * 45;
    printf("%-44s waves/SIMD=%d: %.3f ms  %.1f TF (MFMA flops)\n", name, NT / 256, ms, (double)blocks * (NT / 64) * mf * 512 / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    run<0, 256>("mfma only"); run<0, 512>("mfma only");
    run<1, 256>("mfma+valu 1:1 interleaved"); run<1, 512>("mfma+valu 1:1 interleaved");
    run<2, 256>("45 mfma then 45 valu"); run<2, 512>("45 mfma then 45 valu");
    run<3, 256>("mfma + ds_read 3:1 interleaved"); run<3, 512>("mfma + ds_read 3:1 interleaved");
    run<4, 256>("mfma+valu 1:1 + ds_read"); run<4, 512>("mfma+valu 1:1 + ds_read");
    return 0;
}
