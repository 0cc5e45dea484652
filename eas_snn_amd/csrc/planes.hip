// Spike planes: the storage form of spike tensors between the fused layers of the spiking backbone / neck / head.
//
// A spike tensor [NI][C][HW] (NI = T*N images; values 0/1 and SEW sums 0..4: exact in bf16) is kept as bf16 in blocks of 8 channels,
//     planes[NI][C/8][HW][8],
// the layout in which every consumer reads its matrix-core operand directly: the MFMA B fragment of a convolution is 8 consecutive input
// channels of ONE pixel = one 16-byte load (eas_conv_fwd_planes; NCHW fp32 needs 8 strided 4-byte loads and a conversion), and the
// pixel-major LDS images of the 3x3 forward and of the weight-gradient kernels are filled by plain 16-byte copies.  Half the HBM bytes of
// fp32 (2 B instead of 4 B per neuron-step written by eas_bn_lif_fwd_ex, read by the next convolution and by its weight gradient).
// This file: conversions at the borders of the fused path (API callers, hooks, statistics taps, kernels that read fp32 NCHW).
// Reference: the tensors are the outputs of ParametricLIFNode inside BaseConv, yolox/models/network_blocks.py:52-53 after utils_snn.py:16-58.
#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// thread = 4 consecutive pixels of one 8-channel group of one image
__global__ __launch_bounds__(EAS_BLOCK) void planes_from_f32_kernel(const float* __restrict__ x, bf16x8* __restrict__ planes, int64_t NI, int C, int HW,
                                                                    int src_ctot, int dst_gtot, int* __restrict__ inexact) {
    const int G = C / 8, hw4 = HW / 4;
    const int64_t total = NI * G * hw4;
    unsigned bad = 0;      // OR of the bit patterns: a value is exact in bf16 <=> the low 16 bits of its fp32 pattern are zero
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % hw4);
        const int64_t r = i / hw4;
        const int g = (int)(r % G);
        const int64_t n = r / G;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(x + ((n * src_ctot + 8 * g + j) * (int64_t)HW + 4 * q));
        bf16x8* dst = planes + (n * dst_gtot + g) * (int64_t)HW + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = reinterpret_cast<const float*>(&v[j])[e];
                bad |= __float_as_uint(f);
                o[j] = (__bf16)f;
            }
            dst[e] = o;
        }
    }
    if (inexact && (bad & 0xffffu)) atomicOr(inexact, 1);      // a tensor promised to hold spikes / small integers does not
}

__global__ __launch_bounds__(EAS_BLOCK) void planes_to_f32_kernel(const bf16x8* __restrict__ planes, float* __restrict__ x, int64_t NI, int C, int HW,
                                                                  int src_gtot, int dst_ctot) {
    const int G = C / 8, hw4 = HW / 4;
    const int64_t total = NI * G * hw4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % hw4);
        const int64_t r = i / hw4;
        const int g = (int)(r % G);
        const int64_t n = r / G;
        const bf16x8* src = planes + (n * src_gtot + g) * (int64_t)HW + 4 * q;
        bf16x8 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = src[e];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<float4*>(x + ((n * dst_ctot + 8 * g + j) * (int64_t)HW + 4 * q)) =
                make_float4((float)v[0][j], (float)v[1][j], (float)v[2][j], (float)v[3][j]);
    }
}

// cat[nearest-upsample(a, x up), b] along the channels, planes in / planes out: every output pixel of every 8-channel group is one 16-byte
// copy (groups [0, Ca/8) from a at (h / up, w / up), the rest from b)
__global__ __launch_bounds__(EAS_BLOCK) void upcat_planes_kernel(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, bf16x8* __restrict__ out,
                                                                 int64_t M, int Ga, int Gb, int H, int W, int up) {
    const int Ho = H * up, Wo = W * up, G = Ga + Gb;
    const int64_t total = M * G * (int64_t)Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wo);
        int64_t r = i / Wo;
        const int h = (int)(r % Ho);
        r /= Ho;
        const int g = (int)(r % G);
        const int64_t n = r / G;
        out[i] = g < Ga ? a[((n * Ga + g) * H + h / up) * (int64_t)W + w / up] : b[((n * Gb + (g - Ga)) * Ho + h) * (int64_t)Wo + w];
    }
}

}  // namespace

extern "C" {

// x [NI][C][HW] fp32 (channels of a [NI][src_ctot][HW] tensor, pointer at the first; 0 = C) -> planes [NI][C/8][HW][8] bf16 (groups of a
// [NI][dst_ctot/8][HW][8] tensor, pointer at the first group).  Values must be exact in bf16 (spikes, small integers):
// *inexact_flag, if not NULL, is OR-ed with 1 when one is not.
int eas_spike_planes_from_f32(const float* x, int src_ctot, void* planes, int dst_ctot, int64_t NI, int C, int HW, int* inexact_flag,
                              eas_stream_t stream) {
    if (!x || !planes || NI < 1 || C < 1 || HW < 1) return EAS_ERR_INVALID_ARG;
    if (C % 8 != 0 || HW % 4 != 0 || (src_ctot && src_ctot < C) || (dst_ctot && (dst_ctot < C || dst_ctot % 8 != 0))) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)planes) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(planes_from_f32_kernel, dim3(eas_grid_1d(NI * (C / 8) * (HW / 4))), dim3(EAS_BLOCK), 0, eas_s(stream), x, (bf16x8*)planes, NI, C, HW,
                       src_ctot ? src_ctot : C, (dst_ctot ? dst_ctot : C) / 8, inexact_flag);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// the inverse: planes -> fp32 NCHW (what callers outside the fused path, hooks and the fp32-reading kernels see)
int eas_spike_planes_to_f32(const void* planes, int src_ctot, float* x, int dst_ctot, int64_t NI, int C, int HW, eas_stream_t stream) {
    if (!x || !planes || NI < 1 || C < 1 || HW < 1) return EAS_ERR_INVALID_ARG;
    if (C % 8 != 0 || HW % 4 != 0 || (dst_ctot && dst_ctot < C) || (src_ctot && (src_ctot < C || src_ctot % 8 != 0))) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)planes) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(planes_to_f32_kernel, dim3(eas_grid_1d(NI * (C / 8) * (HW / 4))), dim3(EAS_BLOCK), 0, eas_s(stream), (const bf16x8*)planes, x, NI, C,
                       HW, (src_ctot ? src_ctot : C) / 8, dst_ctot ? dst_ctot : C);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// a [M][Ca/8][H*W][8], b [M][Cb/8][(H*up)*(W*up)][8] -> out [M][(Ca+Cb)/8][(H*up)*(W*up)][8]: eas_upcat_fwd on spike planes (the neck of the
// converted YOLOPAFPN: torch.cat([self.upsample(f), x], 1), yolox/models/yolo_pafpn.py:101-121; up = 1: plain concatenation).  The backward is
// eas_upcat_bwd on the fp32 gradient.
int eas_upcat_planes_fwd(const void* a, const void* b, void* out, int64_t M, int Ca, int Cb, int H, int W, int up, eas_stream_t stream) {
    if (!a || !b || !out || M < 1 || Ca < 1 || Cb < 1 || H < 1 || W < 1 || up < 1) return EAS_ERR_INVALID_ARG;
    if (Ca % 8 != 0 || Cb % 8 != 0) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    const int64_t total = M * ((Ca + Cb) / 8) * (int64_t)H * up * W * up;
    EAS_LAUNCH(upcat_planes_kernel, dim3(eas_grid_1d(total)), dim3(EAS_BLOCK), 0, eas_s(stream), (const bf16x8*)a, (const bf16x8*)b,
                       (bf16x8*)out, M, Ca / 8, Cb / 8, H, W, up);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
