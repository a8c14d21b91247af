// Round 4 probe: inline-asm LDS-DMA (buffer_load_dword / dwordx4 ... lds): M0 handling, exec mask, soffset, out-of-range lanes,
// LDS destinations beyond 64 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ void dma4(uint32_t lds_addr, uint32_t voff, v4i rs, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, v4i rs, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

__global__ void __launch_bounds__(256) k(const float* x, int n, float* out, int lds_off) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* l = reinterpret_cast<float*>(smem + lds_off);
    for (int i = tid; i < 2048; i += 256) l[i] = -7.f;
    __syncthreads();
    const uint64_t xa = reinterpret_cast<uint64_t>(x);
    const v4i rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)xa), __builtin_amdgcn_readfirstlane((int)(uint32_t)(xa >> 32)), n * 4, 0x00020000};
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + lds_off;     // LDS byte address of l
    // wave 0: plain, lanes 0..63 read x[100 + lane] -> l[0..63]
    if (wave == 0) dma4(base, (100 + lane) * 4, rs, 0);
    // wave 1: soffset 4000 bytes, lanes < 20 only -> l[64..83]; lanes >= 20 inactive
    if (wave == 1) { if (lane < 20) dma4(base + 256, lane * 4, rs, 4000); }
    // wave 2: out-of-range lanes (odd lanes offset beyond n*4) -> l[128..191]
    if (wave == 2) dma4(base + 512, (lane & 1) ? 0xfffffff0u : (uint32_t)lane * 4, rs, 0);
    // wave 3: dwordx4: lane reads x[4*(lane^1) .. +3] -> l[256 + 4 lane ..]
    if (wave == 3) dma16(base + 1024, (uint32_t)(lane ^ 1) * 16, rs, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048; i += 256) out[i] = l[i];
}

int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 2048 * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int lds_off : {0, 100 * 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 150 * 1024, 0, x, n, out, lds_off);
        std::vector<float> o(2048);
        hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; ++i) bad += o[i] != 100 + i;
        for (int i = 0; i < 64; ++i) bad += o[64 + i] != (i < 20 ? 1000 + i : -7.f);
        printf("lds_off %d: wave0/1 mismatches %d; wave1 tail sample l[84]=%g\n", lds_off, bad, o[84]);
        printf("  wave2 (odd lanes out of range): ");
        for (int i = 0; i < 8; ++i) printf("%g ", o[128 + i]);
        bad = 0;
        for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) bad += o[256 + 4 * i + j] != 4 * (i ^ 1) + j;
        printf("\n  wave3 dwordx4 mismatches %d (l[256..263] = %g %g %g %g %g %g %g %g)\n", bad, o[256], o[257], o[258], o[259], o[260], o[261], o[262], o[263]);
    }
    return 0;
}
