// Config-4 input: RVT "stacked histogram" representation -> per-polarity event counts on the model canvas.
// Replaces RVTGEN4Dataset.generate_slices(..., method='event_sum') + the zero padding of the validation letterbox
// (yolox/data/datasets/rvt_gen4.py:109-125 and :516-533 with scale 1).
//
//   hist  u8  [B][Tm][2*nbins][H][W]   channel = polarity * nbins + bin  (the reference's reshape(n, 2, -1, H, W))
//   out   f32 [B][Tm][2][Hc][Wc]       out[b][j][p] = sum over bins of hist[b][i][p*nbins + bin], zero outside H x W
// A sample whose sequence is younger than Tm representations supplies only its first n_valid[b] slices; they are the
// LAST n_valid[b] output slices (j = Tm - n_valid[b] + i), the leading ones are zero (rvt_gen4.py:122-123).
//
// HBM-bound integer work: per output element nbins bytes read + 4 bytes written (14 B at nbins = 10).  One thread owns 16
// consecutive pixels of one (b, j, p, row): nbins 16-byte loads issued up front (64 lanes x 16 B = 1 KiB contiguous per plane
// row), byte-wise unpack-accumulate in 16 integer registers, four float4 stores.  Sums <= 255 * nbins are exact in fp32.
#include "eas_common.h"

namespace {


template <int NB>
__global__ __launch_bounds__(EAS_BLOCK) void stacked_hist_sum_kernel(const uint8_t* __restrict__ hist, const int32_t* __restrict__ n_valid,
                                                                     int Tm, int nbins_rt, int H, int W, int Hc, int Wc,
                                                                     float* __restrict__ out, int64_t total_groups) {
    const int nbins = NB > 0 ? NB : nbins_rt;
    const int wg = Wc / 16;                              // 16-pixel groups per output row
    const int64_t plane = (int64_t)H * W;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int xg = (int)(g % wg);
        int64_t r = g / wg;
        const int y = (int)(r % Hc);
        r /= Hc;
        const int p = (int)(r & 1);
        r >>= 1;
        const int j = (int)(r % Tm);
        const int b = (int)(r / Tm);
        const int x0 = xg * 16;
        float4* dst = reinterpret_cast<float4*>(out + ((((int64_t)b * Tm + j) * 2 + p) * Hc + y) * (int64_t)Wc + x0);
        const int nv = n_valid ? n_valid[b] : Tm;
        const int i = j - (Tm - nv);                     // input slice feeding output slice j
        if (i < 0 || y >= H || x0 >= W) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[0] = z; dst[1] = z; dst[2] = z; dst[3] = z;
            continue;
        }
        const uint8_t* src = hist + ((((int64_t)b * Tm + i) * 2 + p) * nbins) * plane + (int64_t)y * W + x0;
        uint32_t acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0;
        if (x0 + 16 <= W && (W & 15) == 0) {
            uint4 v[NB > 0 ? NB : 1];
            if (NB > 0) {
#pragma unroll
                for (int k = 0; k < NB; ++k) v[k] = *reinterpret_cast<const uint4*>(src + (int64_t)k * plane);
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const uint32_t wds[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[4 * q + 0] += wds[q] & 0xffu;
                        acc[4 * q + 1] += (wds[q] >> 8) & 0xffu;
                        acc[4 * q + 2] += (wds[q] >> 16) & 0xffu;
                        acc[4 * q + 3] += wds[q] >> 24;
                    }
                }
            } else {
                for (int k = 0; k < nbins; ++k) {
                    const uint4 u = *reinterpret_cast<const uint4*>(src + (int64_t)k * plane);
                    const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[4 * q + 0] += wds[q] & 0xffu;
                        acc[4 * q + 1] += (wds[q] >> 8) & 0xffu;
                        acc[4 * q + 2] += (wds[q] >> 16) & 0xffu;
                        acc[4 * q + 3] += wds[q] >> 24;
                    }
                }
            }
        } else {                                         // ragged row end or rows that are not 16-byte aligned: bytes one by one
            for (int k = 0; k < nbins; ++k) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (x0 + e < W) acc[e] += src[(int64_t)k * plane + e];
            }
        }
        dst[0] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
        dst[1] = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
        dst[2] = make_float4((float)acc[8], (float)acc[9], (float)acc[10], (float)acc[11]);
        dst[3] = make_float4((float)acc[12], (float)acc[13], (float)acc[14], (float)acc[15]);
    }
}

}  // namespace

extern "C" int eas_stacked_hist_event_sum(const uint8_t* hist, const int32_t* n_valid, int B, int Tm, int nbins, int H, int W, int Hc,
                                          int Wc, float* out, eas_stream_t stream) {
    if (!hist || !out || B < 0 || Tm < 1 || nbins < 1 || H < 1 || W < 1 || Hc < H || Wc < W) return EAS_ERR_INVALID_ARG;
    if (Wc % 16 != 0 || nbins > 255) return EAS_ERR_UNSUPPORTED;      // model canvases are multiples of 32
    if (((uintptr_t)hist | (uintptr_t)out) & 15) return EAS_ERR_INVALID_ARG;
    if (B == 0) return EAS_OK;
    const int64_t groups = (int64_t)B * Tm * 2 * Hc * (Wc / 16);
    const int grid = eas_grid_1d(groups, EAS_BLOCK, 1 << 20);
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (nbins == 10)
        EAS_LAUNCH((stacked_hist_sum_kernel<10>), dim3(grid), dim3(EAS_BLOCK), 0, st, hist, n_valid, Tm, nbins, H, W, Hc, Wc, out, groups);
    else
        EAS_LAUNCH((stacked_hist_sum_kernel<0>), dim3(grid), dim3(EAS_BLOCK), 0, st, hist, n_valid, Tm, nbins, H, W, Hc, Wc, out, groups);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}
