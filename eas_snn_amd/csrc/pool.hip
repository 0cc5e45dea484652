// SPP pooling block fused: out = cat[x, maxpool_k0(x), maxpool_k1(x), maxpool_k2(x)] along channels (stride 1, padding k/2)
// and its backward (reference: SPPBottleneck.forward, yolox/models/network_blocks.py:143-147, on the spike tensors of
// dark5; ATen max_pool2d semantics: the first maximum in row-major scan order wins ties, which matters for 0/1 inputs).
// One block per (image, channel) plane held in LDS; the backward recomputes the arg-max from x (nothing is saved) and
// gathers in a fixed order (deterministic, no atomics).
#include "eas_common.h"

namespace {

constexpr int kMaxPlane = 1024;

__device__ __forceinline__ int argmax_window(const float* plane, int H, int W, int h, int w, int r) {
    const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
    const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
    int best = h0 * W + w0;
    float bv = plane[best];
    for (int a = h0; a <= h1; ++a)
        for (int b = w0; b <= w1; ++b) {
            const float v = plane[a * W + b];
            if (v > bv || v != v) {   // strictly greater (first maximum wins), NaN propagates like ATen
                bv = v;
                best = a * W + b;
            }
        }
    return best;
}

__global__ __launch_bounds__(EAS_BLOCK) void spp_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int k0,
                                                            int k1, int k2) {
    __shared__ float plane[kMaxPlane];
    const int HW = H * W;
    const int64_t pc = blockIdx.x;          // n*C + c
    const int64_t n = pc / C, c = pc - n * C;
    const float* src = x + pc * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = src[i];
    __syncthreads();
    float* dst = out + (n * 4 * C + c) * HW;
    const int ks[3] = {k0, k1, k2};
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const int h = i / W, w = i - h * W;
        dst[i] = plane[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[(int64_t)(j + 1) * C * HW + i] = plane[argmax_window(plane, H, W, h, w, ks[j] / 2)];
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void spp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ gx,
                                                            int C, int H, int W, int k0, int k1, int k2) {
    __shared__ float plane[kMaxPlane];
    __shared__ short amax[3][kMaxPlane];
    const int HW = H * W;
    const int64_t pc = blockIdx.x;
    const int64_t n = pc / C, c = pc - n * C;
    const float* src = x + pc * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = src[i];
    __syncthreads();
    const int ks[3] = {k0, k1, k2};
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const int h = i / W, w = i - h * W;
#pragma unroll
        for (int j = 0; j < 3; ++j) amax[j][i] = (short)argmax_window(plane, H, W, h, w, ks[j] / 2);
    }
    __syncthreads();
    const float* g = gout + (n * 4 * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const int h = i / W, w = i - h * W;
        float s = g[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = ks[j] / 2;
            const float* gj = g + (int64_t)(j + 1) * C * HW;
            const int h0 = h - r < 0 ? 0 : h - r, h1 = h + r >= H ? H - 1 : h + r;
            const int w0 = w - r < 0 ? 0 : w - r, w1 = w + r >= W ? W - 1 : w + r;
            for (int a = h0; a <= h1; ++a)
                for (int b = w0; b <= w1; ++b)
                    if (amax[j][a * W + b] == i) s += gj[a * W + b];
        }
        gx[pc * HW + i] = s;
    }
}

}  // namespace

extern "C" {

int eas_spp_pool_fwd(const float* x, float* out, int64_t N, int C, int H, int W, int k0, int k1, int k2, eas_stream_t stream) {
    if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return EAS_ERR_INVALID_ARG;
    if (H * W > kMaxPlane || !(k0 & 1) || !(k1 & 1) || !(k2 & 1) || N * C > 0x7fffffff) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    hipLaunchKernelGGL(spp_fwd_kernel, dim3((unsigned)(N * C)), dim3(H * W >= 192 ? EAS_BLOCK : 128), 0, eas_s(stream), x, out, C, H, W, k0, k1, k2);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_spp_pool_bwd(const float* x, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                     eas_stream_t stream) {
    if (!x || !grad_out || !grad_x || N <= 0 || C <= 0 || H <= 0 || W <= 0) return EAS_ERR_INVALID_ARG;
    if (H * W > kMaxPlane || !(k0 & 1) || !(k1 & 1) || !(k2 & 1) || N * C > 0x7fffffff) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    hipLaunchKernelGGL(spp_bwd_kernel, dim3((unsigned)(N * C)), dim3(H * W >= 192 ? EAS_BLOCK : 128), 0, eas_s(stream), x, grad_out, grad_x, C, H, W, k0, k1, k2);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
