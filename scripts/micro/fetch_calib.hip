// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the load widths the kernels of this library use (MI355X_MICROARCH.md, HBM: the
// counter reports half the bytes of 16-B-per-lane streaming reads; "other access widths are uncalibrated: calibrate on a known byte
// count").  Streams a 1 GiB buffer (past the 256 MiB Infinity Cache) once per kernel with 4-, 8- and 16-byte loads per lane; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
// and divide the known byte count by the counter (scripts/pmc_summary.py --calibration reads the result).
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename V>
__global__ __launch_bounds__(256) void stream_read(const V* __restrict__ p, size_t n, float* __restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = p[i];
        const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int j = 0; j < (int)(sizeof(V) / 4); ++j) acc += f[j];
    }
    if (acc == 123456.789f) out[0] = acc;      // never true: keeps the loads alive
}

// the same bytes read as 8 strided dword rows per thread (the access pattern of the plane-writing BN+LIF kernel: one dword per lane
// and row, rows `stride` elements apart)
__global__ __launch_bounds__(256) void stream_read_rows(const float* __restrict__ p, size_t rowlen, float* __restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += p[(size_t)r * rowlen + i];
    }
    if (acc == 123456.789f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read<float>, grid, block, 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(stream_read<float2>, grid, block, 0, 0, (const float2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(stream_read<float4>, grid, block, 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(stream_read_rows, grid, block, 0, 0, (const float*)buf, bytes / 32, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
