// Epilogue of the fused eval-mode step  conv -> BatchNorm (running statistics) -> (P)LIF over T  (eas_conv_bn_lif_eval):
// BaseConv.forward of a converted block in eval mode, yolox/models/network_blocks.py:52-53 after yolox/utils/utils_snn.py:16-58.
//
// The matrix-core kernels (conv_mfma.hip, conv1x1_mfma.hip) leave a wave with 32-channel x 32-pixel accumulator tiles.  Here the tiles of
// a wave are either
//   LM = 1 (time-major): tile j = time step j of the wave's 32 pixels (the convolution's N index enumerates (time, pixel): the staged patch /
//           the per-lane pixel addresses of tile j point into image j * N + n), so a lane holds z_0 .. z_{T-1} of its 16 (channel, pixel)
//           neurons in registers and walks the neuron over T without any state outside the loop's own four registers;
//   LM = 2 (shared input): ONE convolution result for all T steps (the first spiking layer behind the stateless stem sees T identical frames,
//           yolox/models/spiking_yolox.py:52-57): tile j = 32 more pixels, every neuron steps T times on the same z.
// The convolution output y never reaches HBM: per neuron-step the kernel writes 2 bytes (spike planes) instead of the 4 + 4 + 2 of
// eas_conv_fwd + eas_bn_lif_fwd_ex, with the SAME arithmetic (z = fma(y, gamma * invstd, beta - mean * gamma * invstd), eas_lif_step), so
// the spikes are bit-identical to the two-kernel path.
#pragma once
#include "eas_common.h"

typedef __bf16 eas_bf16x4 __attribute__((ext_vector_type(4)));

typedef EasLifRange EasLifRangeDev;      // one neuron layer of the call (include/eas_hip.h)

struct EasLifEpiDev {
    int T, N, HW, Cout, csplit;                  // csplit = Cout: one range
    EasLifRangeDev r[2];
};

// z tile(s) -> neurons.  ACC = f32x16 accumulator type; acc[j] tile j; row e of a tile = channel mt*32 + (e&3) + 8*(e>>2) + 4*h, column = the
// lane's pixel (img[j], pix[j]); ok[j]: the lane's column of tile j is a real pixel.
template <int WN, int LM, typename ACC>
__device__ __forceinline__ void eas_lif_epilogue(const ACC (&acc)[WN], const EasLifEpiDev& L, int mt, int h, const int (&img)[WN],
                                                 const int (&pix)[WN], const bool (&ok)[WN]) {
    const int T = L.T;
    const float Tf = (float)T;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = mt * 32 + 8 * q + 4 * h;           // first of the lane's 4 consecutive channels of this quad
        if (c >= L.Cout) continue;                       // (Cout % 8 == 0: a quad is inside or outside as a whole)
        const int ri = c >= L.csplit ? 1 : 0;
        const EasLifRangeDev& R = L.r[ri];
        const int cr = c - (ri ? L.csplit : 0);          // channel inside the range
        const int Cr = ri ? L.Cout - L.csplit : L.csplit;
        float scale[4], shift[4];
        {
            const float4 ga = *reinterpret_cast<const float4*>(R.gamma + cr), be = *reinterpret_cast<const float4*>(R.beta + cr);
            const float4 mu = *reinterpret_cast<const float4*>(R.mean + cr), is = *reinterpret_cast<const float4*>(R.invstd + cr);
            scale[0] = ga.x * is.x; scale[1] = ga.y * is.y; scale[2] = ga.z * is.z; scale[3] = ga.w * is.w;
            shift[0] = be.x - mu.x * scale[0]; shift[1] = be.y - mu.y * scale[1]; shift[2] = be.z - mu.z * scale[2]; shift[3] = be.w - mu.w * scale[3];
        }
        const float k = R.w_logit ? eas_sigmoidf(*R.w_logit) : R.k_const;
        const float omk = 1.0f - k;
        constexpr int NJ = LM == 1 ? 1 : WN;             // pixel columns the lane owns
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            if (!ok[jj]) continue;
            const int64_t n = img[jj], p = pix[jj];
            const int64_t nbase = (n * Cr + cr) * (int64_t)L.HW + p;          // of rate / v_in / v_out, channel cr
            float v[4], cnt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = R.v_in ? R.v_in[nbase + (int64_t)e * L.HW] : 0.0f;
                cnt[e] = 0.0f;
            }
            auto step = [&](int t, const ACC& zt) {
                const int64_t im = (int64_t)t * L.N + n;
                float s[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float hh;
                    eas_lif_step<false, false, false>(v[e], fmaf(zt[4 * q + e], scale[e], shift[e]), k, omk, R.v_th, 0.0f, hh, s[e]);
                    cnt[e] += s[e];
                }
                if (R.res_planes) {
                    const eas_bf16x4 rp = *reinterpret_cast<const eas_bf16x4*>(
                        reinterpret_cast<const unsigned char*>(R.res_planes) + ((im * (R.res_ctot / 8) + cr / 8) * (int64_t)L.HW + p) * 16 + 8 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += (float)rp[e];
                } else if (R.res_f32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += R.res_f32[((im * R.res_ctot + cr + e) * (int64_t)L.HW) + p];
                }
                if (R.planes) {
                    eas_bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)s[e];
                    *reinterpret_cast<eas_bf16x4*>(reinterpret_cast<unsigned char*>(R.planes) +
                                                   ((im * (R.out_ctot / 8) + (R.out_c0 + cr) / 8) * (int64_t)L.HW + p) * 16 + 8 * h) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) R.out_f32[((im * R.out_ctot + R.out_c0 + cr + e) * (int64_t)L.HW) + p] = s[e];
                }
            };
            if constexpr (LM == 1) {
#pragma unroll
                for (int t = 0; t < WN; ++t) step(t, acc[t]);
            } else {
                for (int t = 0; t < T; ++t) step(t, acc[jj]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (R.v_out) R.v_out[nbase + (int64_t)e * L.HW] = v[e];
                if (R.rate) R.rate[nbase + (int64_t)e * L.HW] = cnt[e] / Tf;
            }
        }
    }
}
