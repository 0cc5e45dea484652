// Calibration: achievable v_mfma_f32_32x32x16_bf16 rate on this chip in the occupancy shapes the conv kernels use.
// build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed * 0.5f + j * 0.25f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks, int threads, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = (double)blocks * (threads / 64) * iters * NACC * 3;
        if (rep == 2) printf("%-28s blocks %5d thr %4d: %8.3f ms  %.1f TF bf16  (%.1f cyc/MFMA/SIMD at 2.4GHz assuming even spread)\n", name, blocks, threads, ms,
               mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
    }
    hipFree(out);
}

int main() {
    run<5>("1 wave/SIMD, 5 acc", 256, 256, 2000);
    run<5>("2 waves/SIMD, 5 acc", 256, 512, 2000);
    run<5>("2 waves/SIMD, 5 acc, short", 384, 512, 36);   // the dark3 conv: 540 MFMAs per wave
    run<5>("2 waves/SIMD, 5 acc, short", 256, 512, 36);
    run<5>("2 waves/SIMD, 5 acc, 1ms", 256, 512, 600);
    run<10>("1 wave/SIMD, 10 acc", 256, 256, 1000);
    return 0;
}
