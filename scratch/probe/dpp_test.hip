#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// total of all 64 lanes, returned wave-uniform (SGPR)
__device__ __forceinline__ float wave_total(float v) {
    v += dpp_get<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);   // row_mirror        -> every lane holds its row's sum
    v += dpp_get<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v += dpp_get<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__global__ void k(const float* in, float* out) { out[threadIdx.x] = wave_total(in[threadIdx.x]); }
int main() {
    float h[64], o[64]; double ref = 0;
    for (int i = 0; i < 64; ++i) { h[i] = (float)((i * 37) % 11) + 0.25f * i; ref += h[i]; }
    float *a, *b; (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256);
    (void)hipMemcpy(a, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b);
    (void)hipMemcpy(o, b, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (o[i] != (float)ref) ++bad;
    printf("wave_total: ref %g got %g, bad lanes %d\n", ref, o[0], bad);
    return 0;
}
