// What does ds_read_b64_tr_b16 return?  LDS element i holds the value i; every lane passes its own address.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int a = 0;
    if (mode == 0) a = l * 4;                                  // lane-linear 8-byte pieces
    if (mode == 1) a = (l & 15) * 64 + (l >> 4) * 4;           // 16 rows of 64 elements, 4 column groups
    if (mode == 2) a = (l & 15) * 16 + (l >> 4) * 256;         // [16 rows][16 cols] blocks, one block per 16-lane group
    if (mode == 3) a = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 1024;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    unsigned short h[256];
    for (int mode = 0; mode < 4; ++mode) {
        k<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : " | ");
    }
    return 0;
}
