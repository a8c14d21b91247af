// Issue rate of the two fp16 MFMA shapes, one wave per SIMD (256 workgroups of 256 threads) and two (512, 60 KB LDS each).
//   hipcc -O3 -w --offload-arch=gfx950 scratch/probe/mfma_shapes.hip -o /tmp/shapes && /tmp/shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
extern __shared__ char smem[];
template <int SHAPE, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 3); }
    float s = 0;
    if constexpr (SHAPE == 0) {
        f4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 48; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % NACC], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f16v acc[NACC];
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 24; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (smem[threadIdx.x] == 77 && s == 1.f) out[0] = 1;
}
template <int SHAPE, int NACC>
void run(const char* name, float* out) {
    const int lds = 60 * 1024;
    hipFuncSetAttribute((const void*)k<SHAPE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int wgs : {256, 512}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(wgs), dim3(256), lds, 0, out, 200);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(wgs), dim3(256), lds, 0, out, 200);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 * 200 * 48 * 16384.0;       // both loops: 48 x 16384 flop-units per iteration
        printf("%s, %d accumulators, %d workgroups: %.1f us per launch = %.2f PFLOP/s\n", name, NACC, wgs, ms / 5 * 1e3, flop / (ms / 5 * 1e-3) / 1e15);
    }
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<0, 12>("16x16x32 f16", out);
    run<0, 4>("16x16x32 f16", out);
    run<1, 6>("32x32x16 f16", out);
    run<1, 2>("32x32x16 f16", out);
    return 0;
}
