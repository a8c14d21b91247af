// What does a workgroup cost before it does any work?  256 threads, 12,800 workgroups (the 18->18 @320^2 convolution at N = 32),
// variants: empty / 33 KB of dynamic LDS / + two barriers / + a scalar prologue of integer divisions / + 32 KB of LDS writes.
//   hipcc -O3 --offload-arch=gfx950 scratch/probe/wg_floor.hip -o /tmp/wg_floor && /tmp/wg_floor
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ void __launch_bounds__(256) k(int* out, int a, int b, int c, int d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int acc = 0;
    if (V >= 3) {   // a prologue of dependent scalar divisions (the convolution decodes its tile with eight of them)
        int lin = blockIdx.x;
        int g = lin / a, sk = lin - g * a;
        int q = g / b, cg = g - q * b;
        int n = g / (b * c), tile = q - n * c;
        int ty = tile / d, tx = tile - ty * d;
        int c0 = (13 * sk) / a, c1 = (13 * (sk + 1)) / a;
        acc = sk + cg + n + ty + tx + c0 + c1;
    }
    if (V >= 2) __syncthreads();
    if (V >= 4) {
        uint4* p = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[threadIdx.x + 256 * i] = make_uint4(acc, i, threadIdx.x, 0);
    }
    if (V >= 2) __syncthreads();
    if (V >= 4) acc += reinterpret_cast<int*>(smem)[(threadIdx.x * 37) & 8191];
    if (acc == 123456789) out[0] = acc;
}
template <int V>
float run(int wgs, size_t lds, int* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), lds, 0, out, 1, 1, 400, 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(256), lds, 0, out, 1, 1, 400, 10);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    int* out;
    hipMalloc(&out, 64);
    for (int wgs : {3200, 12800}) {
        printf("%5d workgroups: empty %.1f us | 33 KB LDS %.1f | + 2 barriers %.1f | + scalar prologue %.1f | + 32 KB LDS writes %.1f | 49 KB LDS + all %.1f\n", wgs,
               run<0>(wgs, 0, out), run<1>(wgs, 33 * 1024, out), run<2>(wgs, 33 * 1024, out), run<3>(wgs, 33 * 1024, out),
               run<4>(wgs, 33 * 1024, out), run<4>(wgs, 49 * 1024, out));
    }
    return 0;
}
