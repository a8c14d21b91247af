#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + l] = c[i];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + 3 * l; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
        int blk = l / 4, j = l % 4;
        float want = ha[4 * blk + i] * hb[4 * blk + j];
        if (hd[i * 64 + l] != want) { if (bad < 5) printf("mismatch reg %d lane %d: got %g want %g\n", i, l, hd[i*64+l], want); ++bad; }
    }
    printf("hypothesis D[reg i][lane 4b+j] = A[4b+i]*B[4b+j]: %s (%d bad)\n", bad ? "WRONG" : "OK", bad);
    if (bad) { for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l], hd[64+l], hd[128+l], hd[192+l]); }
    return 0;
}
