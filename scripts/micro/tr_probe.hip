// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS holds a [rows][16 cols] uint16 image with value row*100+col;
// every lane supplies the address documented in cdna_hip_programming.md T10 and we print what it receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __align__(16) unsigned short img[64 * 32];   // 64 rows x 32 cols (64-byte rows)
    for (int i = threadIdx.x; i < 64 * 32; i += 64) img[i] = (unsigned short)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int colblock = (g & 1) * 16, kbase = (g >> 1) * 8;
    const int row = kbase + q;                // read 0: rows kbase..kbase+3
    unsigned addr = (unsigned)(size_t)(&img[row * 32 + colblock + 4 * p]);
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
