// which lane does row_shl:2 / row_shr:2 read?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x102, 0xf, 0xf, false);        // row_shl:2
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x112, 0xf, 0xf, false);   // row_shr:2
}
int main() {
    int* d; (void)hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    int h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("row_shl:2 :"); for (int i = 0; i < 20; ++i) printf(" %d", h[i]); printf("\nrow_shr:2 :"); for (int i = 0; i < 20; ++i) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}
