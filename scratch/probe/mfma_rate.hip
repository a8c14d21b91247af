#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f4 acc[20];
    for (int i = 0; i < 20; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        }
    } else if (MODE == 1) {
        f4 c4[20];
        for (int i = 0; i < 20; ++i) c4[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[i], 0, 0, 0);
        }
        for (int i = 0; i < 20; ++i) acc[i] += c4[i];
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                acc[i][0] = fmaf(a, b, acc[i][0]); acc[i][1] = fmaf(a, b, acc[i][1]);
                acc[i][2] = fmaf(a, b, acc[i][2]); acc[i][3] = fmaf(a, b, acc[i][3]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double flop_per_inst, int blocks) {
    float* out; (void)hipMalloc(&out, 4 * 256 * blocks);
    int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.f, 2.f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * iters * 20;
    printf("%s blocks=%d: %.3f ms, %.1f TFLOP/s, %.2f ns per wave-inst per SIMD-equivalent\n", name, blocks, ms,
           insts * flop_per_inst / ms / 1e9, ms * 1e6 / (insts / 1024.0));
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 1024}) {
        run<0>("mfma_4x4x1_16b", 512.0, blocks);
        run<1>("mfma_16x16x4  ", 2048.0, blocks);
        run<2>("v_fma x4      ", 512.0, blocks);
    }
    return 0;
}
