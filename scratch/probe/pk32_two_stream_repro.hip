// Stand-alone reproducer of the packed-fp32 operand misread: no library, no model, inline asm only.
//   hipcc -O2 --offload-arch=gfx950 pk32_two_stream_repro.hip -o pk32_repro && ./pk32_repro [launches]
// VICTIM (stream A, 512-thread workgroups): every lane repeats, on registers written long before,
//     v_pk_mul_f32 d[0:1], a[0:1], m[0:1] op_sel:[S0,S1] op_sel_hi:[H0,H1]      // d0 = a[S0] * m[S1], d1 = a[H0] * m[H1]
// and compares both halves with scalar v_mul_f32 of the same registers.  Nothing in the wave can create a hazard: the sources are
// loop-invariant registers, the compiler sees opaque asm blocks, form 5 pads the instruction with s_nop on both sides.
// AGGRESSOR (stream B, 256-thread workgroups, two per compute unit): one of five instruction streams, see below.
// Measured on MI355X (profiles/r05_pk32_repro.txt): see DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define PK(text) asm volatile(text : "=v"(d) : "v"(a), "v"(m))

template <int FORM> __global__ void __launch_bounds__(512) victim(const float* __restrict__ in, unsigned* __restrict__ rec, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    float a0 = in[4 * t], a1 = in[4 * t + 1], m0 = in[4 * t + 2], m1 = in[4 * t + 3];
    unsigned bad_lo = 0, bad_hi = 0;
    for (int i = 0; i < iters; ++i) {
        const u64 a = (u64)__builtin_bit_cast(unsigned, a1) << 32 | __builtin_bit_cast(unsigned, a0);    // (register pairs as 64-bit
        const u64 m = (u64)__builtin_bit_cast(unsigned, m1) << 32 | __builtin_bit_cast(unsigned, m0);    //  integers for the asm)
        u64 d; float lo, hi;
        if (FORM == 0) { PK("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]"); lo = a0 * m1; hi = a1 * m1; }
        if (FORM == 1) { PK("v_pk_mul_f32 %0, %1, %2"); lo = a0 * m0; hi = a1 * m1; }
        if (FORM == 2) { PK("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]"); lo = a1 * m0; hi = a1 * m1; }
        if (FORM == 3) { PK("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]"); lo = a0 * m0; hi = a1 * m0; }
        if (FORM == 4) { PK("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]"); lo = a0 + m1; hi = a1 + m1; }
        if (FORM == 5) { PK("s_nop 7\n s_nop 7\n v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n s_nop 7"); lo = a0 * m1; hi = a1 * m1; }
        if (FORM == 6) {        // the only other op_sel the library still has (fp8 path): destination-half select, 32-bit sources
            unsigned q = 0x12345678u, e = 0u;
            asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(q) : "v"(a0), "v"(m1));
            asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(e) : "v"(a0), "v"(m1));
            d = q; lo = __builtin_bit_cast(float, (e & 0xffffu) << 16 | 0x5678u); hi = 0.f;
        }
        asm volatile("" : "+v"(lo), "+v"(hi));                               // (scalar references: never re-packed)
        bad_lo += (unsigned)d != __builtin_bit_cast(unsigned, lo);
        bad_hi += (unsigned)(d >> 32) != __builtin_bit_cast(unsigned, hi);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(m0), "+v"(m1));          // keep the loop a loop
    }
    if (bad_lo | bad_hi) {
        atomicAdd(rec, bad_lo); atomicAdd(rec + 1, bad_hi);
        atomicAdd(rec + 2 + ((threadIdx.x & 63) >> 4), 1u);                  // which quarter of the wave (lanes 0-15 .. 48-63)
    }
}

// Aggressor modes (inline asm, so that the instruction mix is what is written here):
//   1  back-to-back MFMAs accumulating in AGPRs           2  a v_fma_f32 stream (no matrix, no AGPR)
//   3  v_accvgpr_read_b32 / v_accvgpr_write_b32 only      4  MFMAs with VGPR accumulators (no AGPR anywhere)
//   5  MFMAs on AGPR accumulators with v_accvgpr_read / v_accvgpr_mov between them (what hipcc makes of a loop that carries
//      its accumulators through a branch -- and what a convolution's epilogue beside another wave's K-loop looks like)
__global__ void __launch_bounds__(256) aggressor(float* out, int iters, int mode) {
    extern __shared__ float lds[];
    f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    h8 x, w;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); w[i] = (_Float16)(0.5f - i * 0.01f); }
    float v = threadIdx.x, r = 0.f;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n"
                 "v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0"
                 :: "v"(r) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    for (int i = 0; i < iters; ++i) {
        if (mode == 1)
            asm volatile("v_mfma_f32_16x16x32_f16 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x32_f16 a[4:7], %0, %1, a[4:7]\n"
                         "v_mfma_f32_16x16x32_f16 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x32_f16 a[4:7], %0, %1, a[4:7]"
                         :: "v"(x), "v"(w) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        else if (mode == 2)
            asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                         "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(r));
        else if (mode == 3)
            asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_mov_b32 a2, a1\n v_accvgpr_mov_b32 a7, a6\n"
                         "v_accvgpr_read_b32 %0, a3\n v_accvgpr_read_b32 %1, a4\n v_accvgpr_mov_b32 a1, a2\n v_accvgpr_mov_b32 a6, a7"
                         : "=v"(r), "=v"(v) :: "a1", "a2", "a6", "a7");
        else if (mode == 4) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, w, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, w, acc2, 0, 0, 0);
            asm volatile("" : "+v"(acc), "+v"(acc2));                       // (accumulators pinned in VGPRs)
        } else
            asm volatile("v_mfma_f32_16x16x32_f16 a[0:3], %2, %3, a[0:3]\n v_accvgpr_read_b32 %0, a5\n v_accvgpr_mov_b32 a6, a7\n"
                         "v_mfma_f32_16x16x32_f16 a[4:7], %2, %3, a[4:7]\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_mov_b32 a2, a3"
                         : "=v"(r), "=v"(v) : "v"(x), "v"(w) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    }
    asm volatile("s_nop 7\n s_nop 7\n v_accvgpr_read_b32 %0, a0" : "=v"(r));
    if (threadIdx.x == 0) lds[0] = v;
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc2.y + v + r + lds[0];
}

typedef void (*victim_fn)(const float*, unsigned*, int);
int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200, vblocks = 480, viters = 4000;
    const victim_fn forms[7] = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>};
    const char* names[7] = {"v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 (plain)", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel_hi:[1,0]",
                            "v_pk_add_f32 op_sel:[0,1]", "op_sel:[0,1] between s_nops", "v_cvt_pk_fp8_f32 op_sel:[0,0,1]"};
    const char* modes[6] = {"nothing", "MFMA (AGPR acc)", "v_fma_f32", "v_accvgpr_read/mov", "MFMA (VGPR acc)", "MFMA + v_accvgpr_*"};
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::vector<float> h(4 * 512 * vblocks);
    srand(1); for (auto& f : h) f = 0.5f + (rand() % 100000) * 1e-5f;
    float *in, *out; unsigned* rec;
    CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&rec, 6 * 4));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // 64 KB of LDS per aggressor workgroup: two per compute unit (2 waves per SIMD), so that the victim's waves find free slots BESIDE them
    // (with a small allocation the aggressor fills all 32 wave slots, the victim runs after it, and nothing goes wrong)
    CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int form = 0; form < 7; ++form)
        for (int mode = 0; mode < 6; ++mode) {
            unsigned long long tot[6] = {0, 0, 0, 0, 0, 0}; int bad_launches = 0;
            for (int l = 0; l < launches; ++l) {
                CK(hipMemsetAsync(rec, 0, 24, sa));
                if (mode) aggressor<<<1024, 256, 64 * 1024, sb>>>(out, 6000, mode);
                forms[form]<<<vblocks, 512, 0, sa>>>(in, rec, viters);
                unsigned r[6]; CK(hipMemcpyAsync(r, rec, 24, hipMemcpyDeviceToHost, sa)); CK(hipDeviceSynchronize());
                if (r[0] | r[1]) ++bad_launches;
                for (int k = 0; k < 6; ++k) tot[k] += r[k];
            }
            printf("%-32s beside %-22s: %3d of %d launches wrong; wrong low halves %llu, high halves %llu of %.3g; threads hit in lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
                   names[form], modes[mode], bad_launches, launches, tot[0], tot[1],
                   (double)launches * vblocks * 512 * viters, tot[2], tot[3], tot[4], tot[5]);
            fflush(stdout);
        }
    return 0;
}
