// Round 4 probe: does the VALU work of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?
// One 512-thread workgroup per CU (100 KB of LDS): roles by wave.  Per-wave cycle counts (s_memtime) and HW_ID are recorded.
//   hipcc -O3 --offload-arch=gfx950 scratch/probe/r4_overlap.hip -o scratch/probe/r4_overlap && scratch/probe/r4_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float fl2 __attribute__((ext_vector_type(2)));
extern __shared__ char smem[];

// ROLEMAP 0: waves 0-3 matrix, 4-7 vector; 1: even waves matrix, odd vector.  KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 cvt + fma_mix (split)
// 3: ds_write_b128 + ds_read_b128 stream
template <int KIND>
__device__ __forceinline__ void valu_block(float (&v)[16], int iters, char* lds) {
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) v[j % 16] = __builtin_fmaf(v[j % 16], 1.0001f, 0.5f);
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                fl2 t = {v[(2 * j) % 16], v[(2 * j + 1) % 16]};
                t = __builtin_elementwise_fma(t, fl2{1.0001f, 1.0001f}, fl2{0.5f, 0.5f});
                v[(2 * j) % 16] = t[0];
                v[(2 * j + 1) % 16] = t[1];
            }
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
                const fl2 t = {v[j], v[(j + 1) % 16]};
                const hf2 h = __builtin_convertvector(t, hf2);
                const float r0 = __builtin_fmaf((float)h[0], -1.f, t[0]), r1 = __builtin_fmaf((float)h[1], -1.f, t[1]);
                const hf2 h2 = __builtin_convertvector(fl2{r0, r1}, hf2);
                v[j] = (float)h2[0] + (float)h[1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                *reinterpret_cast<f4*>(lds + (threadIdx.x * 48 + j * 16 * 1024) % 65536) = f4{v[j], v[j + 1], v[j + 2], v[j + 3]};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4 r = *reinterpret_cast<const f4*>(lds + (threadIdx.x * 16 + j * 16 * 1024) % 65536);
                v[j] += r[0];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int m_iters, int v_iters, int rolemap, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool matrix = rolemap == 0 ? wave < 4 : (wave & 1) == 0;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 3); }
    f4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (matrix) {
        if (mode & 1)
            for (int it = 0; it < m_iters; ++it) {
#pragma unroll
                for (int m = 0; m < 48; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 12], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    } else {
        if (mode & 2) { __builtin_amdgcn_s_setprio(3); valu_block<KIND>(v, v_iters, smem); __builtin_amdgcn_s_setprio(0); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4) {
        cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        cyc[(blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc, int m_iters, int v_iters) {
    const int lds = 100 * 1024;
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rolemap = 0; rolemap < 2; ++rolemap)
        for (int mode = 1; mode <= 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), lds, 0, out, cyc, m_iters, v_iters, rolemap, mode);
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), lds, 0, out, cyc, m_iters, v_iters, rolemap, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(64);
            hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
            printf("%-10s rolemap %d mode %s: %7.1f us   wave cycles:", name, rolemap, mode == 1 ? "MFMA " : mode == 2 ? "VALU " : "BOTH ", ms / 5 * 1e3);
            for (int w = 0; w < 8; ++w) printf(" %llu(simd %llu)", h[w * 2], (h[w * 2 + 1] >> 4) & 3);
            printf("\n");
        }
}

int main() {
    float* out; hipMalloc(&out, 4096);
    unsigned long long* cyc; hipMalloc(&cyc, 64 * 8);
    run<0>("v_fma", out, cyc, 100, 300);
    run<1>("v_pk_fma", out, cyc, 100, 300);
    run<2>("cvt_split", out, cyc, 100, 300);
    run<3>("lds_rw", out, cyc, 100, 300);
    return 0;
}
