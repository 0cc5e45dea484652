// Does the range check of a raw buffer load (stride 0) on gfx950 include the scalar offset?  And does voffset = 0xFFFFFFFF return zeros
// whatever the scalar offset is?  (conv_wgrad_mfma.hip relies on: check = voffset + inst_offset against num_records, soffset excluded.)
// build: hipcc --offload-arch=gfx950 -O2 -o buffer_oob buffer_oob.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* buf, unsigned num_records, unsigned voff, unsigned soff, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, num_records, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + threadIdx.x * 16, soff, 0);
    out[threadIdx.x] = __builtin_bit_cast(float, v[0]);
}

int main() {
    const size_t n = 2u << 20;   // 2 MiB allocation, 1 MiB declared to the descriptor
    float *buf, *out;
    hipMalloc(&buf, n);
    hipMalloc(&out, 64 * 4);
    float* h = new float[n / 4];
    for (size_t i = 0; i < n / 4; ++i) h[i] = 1.0f + (float)(i >> 16);
    hipMemcpy(buf, h, n, hipMemcpyHostToDevice);
    struct { unsigned voff, soff; const char* what; } cases[] = {
        {0, 0, "in range"},
        {600u << 10, 512u << 10, "voffset < num_records, voffset + soffset > num_records (memory exists)"},
        {1100u << 10, 0, "voffset > num_records"},
        {0xFFFFFFF0u - 1024, 512u << 10, "voffset = sentinel, soffset large"},
        {0xFFFFFFFFu - 2048, 0, "voffset = sentinel, soffset 0"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, 1u << 20, c.voff, c.soff, out);
        float o[64];
        hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        printf("%-80s -> lane0 %.1f lane63 %.1f\n", c.what, o[0], o[63]);
    }
    return 0;
}
