// development: which XCD (HW_REG_XCC_ID) runs work-group i of a 1-D grid and of a dim3(8, ny, nz) grid -- the assumption behind EAS_CHAN_GRID
// (csrc/eas_common.h) and behind any scheme that wants two blocks on the same L2.   hipcc --offload-arch=gfx950 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void who(int* out) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11));
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0) out[lin] = (int)xcc;
}

static void run(dim3 grid, const char* name) {
    const int n = grid.x * grid.y * grid.z;
    int* d;
    hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(who, grid, dim3(256), 0, 0, d);
    std::vector<int> h(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int match = 0;
    for (int i = 0; i < n; ++i) match += h[i] == (i % 8);
    printf("%s: %d blocks, XCC_ID == linear index %% 8 for %d of them; first 24:", name, n, match);
    for (int i = 0; i < 24 && i < n; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}

int main() {
    run(dim3(4096), "1-D grid");
    run(dim3(16128), "1-D grid (16128)");
    run(dim3(8, 20, 16), "dim3(8, 20, 16)");
    run(dim3(8, 63, 16), "dim3(8, 63, 16)");
    return 0;
}
