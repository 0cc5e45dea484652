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

// activation of the plain convolution epilogue (eas_conv_fwd_act): SiLU as nn.SiLU computes it, x * sigmoid(x) (bn_act.hip uses the same form)
__device__ __forceinline__ float eas_epi_silu(float v) { return v * eas_sigmoidf(v); }

typedef EasLifRange EasLifRangeDev;      // one neuron layer of the call (include/eas_hip.h)

struct EasLifEpiDev {
    int T, N, HW, Cout, csplit;                  // csplit = Cout: one range
    EasLifRangeDev r[2];
};

// z tile(s) -> neurons.  ACC = f32x16 accumulator type; acc[j] tile j; row e of a tile = channel mt*32 + (e&3) + 8*(e>>2) + 4*h, column = the
// lane's pixel (img[j], pix[j]); ok[j]: the lane's column of tile j is a real pixel.
// LM = 1 with WN = M2 * T (M2 = 1 or 2 spatial tiles per wave): tile j = time step j / M2 of spatial tile j % M2; img / pix / ok are indexed by
// the spatial tile (entries 0 .. M2-1).
template <int WN, int LM, typename ACC>
__device__ __forceinline__ void eas_lif_epilogue(const ACC (&acc)[WN], const EasLifEpiDev& L, int mt, int h, const int (&img)[WN],
                                                 const int (&pix)[WN], const bool (&ok)[WN]) {
    const int T = L.T;
    const int M2 = LM == 1 ? WN / T : 1;             // (wave-uniform; the loops below are unrolled over WN and masked)
    const float Tf = (float)T;
    // the BatchNorm constants of the lane's four channel quads, requested together (one exposed load latency instead of four)
    float4 ga[4], be[4], mu[4], is[4];
    float kq[4];
    bool qok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = mt * 32 + 8 * q + 4 * h;           // first of the lane's 4 consecutive channels of this quad
        qok[q] = c < L.Cout;                             // (Cout % 8 == 0: a quad is inside or outside as a whole)
        const int cc = qok[q] ? c : 0;
        const int ri = cc >= L.csplit ? 1 : 0;
        const EasLifRangeDev& R = L.r[ri];
        const int cr = cc - (ri ? L.csplit : 0);
        ga[q] = *reinterpret_cast<const float4*>(R.gamma + cr);
        be[q] = *reinterpret_cast<const float4*>(R.beta + cr);
        mu[q] = *reinterpret_cast<const float4*>(R.mean + cr);
        is[q] = *reinterpret_cast<const float4*>(R.invstd + cr);
        kq[q] = R.w_logit ? *R.w_logit : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!qok[q]) continue;
        const int c = mt * 32 + 8 * q + 4 * h;
        const int ri = c >= L.csplit ? 1 : 0;
        const EasLifRangeDev& R = L.r[ri];
        const int cr = c - (ri ? L.csplit : 0);          // channel inside the range
        const int Cr = ri ? L.Cout - L.csplit : L.csplit;
        float scale[4], shift[4];
        scale[0] = ga[q].x * is[q].x; scale[1] = ga[q].y * is[q].y; scale[2] = ga[q].z * is[q].z; scale[3] = ga[q].w * is[q].w;
        shift[0] = be[q].x - mu[q].x * scale[0]; shift[1] = be[q].y - mu[q].y * scale[1];
        shift[2] = be[q].z - mu[q].z * scale[2]; shift[3] = be[q].w - mu[q].w * scale[3];
        const float k = R.w_logit ? eas_sigmoidf(kq[q]) : R.k_const;
        const float omk = 1.0f - k;
        // byte strides of one time step in the destination / shortcut tensors (wave-uniform)
        const int64_t out_tstep = (int64_t)L.N * R.out_ctot * L.HW * (R.planes ? 2 : 4);
        const int64_t res_tstep = (int64_t)L.N * R.res_ctot * L.HW * (R.res_planes ? 2 : 4);
#pragma unroll
        for (int jj = 0; jj < (LM == 1 ? 2 : WN); ++jj) {       // pixel columns the lane owns (time-major: one or two)
            if (LM == 1 && jj >= M2) continue;
            if (!ok[jj]) continue;
            const int64_t n = img[jj], p = pix[jj];
            const int64_t nbase = (n * Cr + cr) * (int64_t)L.HW + p;          // of rate / v_in / v_out, channel cr
            float v[4], cnt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = R.v_in ? R.v_in[nbase + (int64_t)e * L.HW] : 0.0f;
                cnt[e] = 0.0f;
            }
            // step-0 addresses of the lane's quad; every later step adds the tensor's time-step stride
            unsigned char* po = R.planes ? reinterpret_cast<unsigned char*>(R.planes) + ((n * (R.out_ctot / 8) + (R.out_c0 + cr) / 8) * (int64_t)L.HW + p) * 16 + 8 * h
                                         : reinterpret_cast<unsigned char*>(R.out_f32 + (n * R.out_ctot + R.out_c0 + cr) * (int64_t)L.HW + p);
            const unsigned char* pr = R.res_planes ? reinterpret_cast<const unsigned char*>(R.res_planes) + ((n * (R.res_ctot / 8) + cr / 8) * (int64_t)L.HW + p) * 16 + 8 * h
                                      : (R.res_f32 ? reinterpret_cast<const unsigned char*>(R.res_f32 + (n * R.res_ctot + cr) * (int64_t)L.HW + p) : nullptr);
            auto step = [&](const ACC& zt) {
                float s[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float hh;
                    eas_lif_step<false, false, false>(v[e], fmaf(zt[4 * q + e], scale[e], shift[e]), k, omk, R.v_th, 0.0f, hh, s[e]);
                    cnt[e] += s[e];
                }
                if (R.res_planes) {
                    const eas_bf16x4 rp = *reinterpret_cast<const eas_bf16x4*>(pr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += (float)rp[e];
                } else if (R.res_f32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += reinterpret_cast<const float*>(pr)[(int64_t)e * L.HW];
                }
                if (R.planes) {
                    eas_bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)s[e];
                    *reinterpret_cast<eas_bf16x4*>(po) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(po)[(int64_t)e * L.HW] = s[e];
                }
                po += out_tstep;
                if (pr) pr += res_tstep;
            };
            if constexpr (LM == 1) {
                if (M2 == 1) {
#pragma unroll
                    for (int t = 0; t < WN; ++t) step(acc[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < WN / 2; ++t) step(acc[2 * t + (jj & 1)]);
                }
            } else {
                for (int t = 0; t < T; ++t) step(acc[jj]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (R.v_out) R.v_out[nbase + (int64_t)e * L.HW] = v[e];
                if (R.rate) R.rate[nbase + (int64_t)e * L.HW] = cnt[e] / Tf;
            }
        }
    }
}

// ---- BatchNorm (running statistics) + activation in the epilogue of the PLAIN convolution kernels (eas_conv_bn_act_eval): the eval-mode
// form of a real-valued BaseConv, "act(bn(conv(x)))" network_blocks.py:52-53 -- the arithmetic of bn_silu_fwd_kernel (bn_act.hip) on the
// accumulators: scale = gamma * invstd, shift = beta - mean * scale, out = act(fma(y, scale, shift)).
struct EasBnActEpiDev {
    int on, act, csplit, Cout;
    EasBnActRange r[2];
};

// acc[j]: tile j of the wave (row e = channel mt*32 + (e&3) + 8*(e>>2) + 4*h, column = the lane's pixel); yb[j] = img * Cout * HW + pixel of
// the lane's column (the offset the plain epilogue stores at), < 0: no pixel.
template <int WN, typename ACC>
__device__ __forceinline__ void eas_bnact_epilogue(const ACC (&acc)[WN], const EasBnActEpiDev& B, int mt, int h, const long (&yb)[WN], long HW) {
    float scale[16], shift[16];
    float* ob[4];
    long oct[4];
    bool qok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = mt * 32 + 8 * q + 4 * h;           // first of the lane's 4 consecutive channels of this quad (a quad lies in one range)
        qok[q] = c < B.Cout;
        const int cc = qok[q] ? c : 0;
        const int ri = cc >= B.csplit ? 1 : 0;
        const EasBnActRange& R = B.r[ri];
        const int cr = cc - (ri ? B.csplit : 0);
        const float4 ga = *reinterpret_cast<const float4*>(R.gamma + cr), be = *reinterpret_cast<const float4*>(R.beta + cr);
        const float4 mu = *reinterpret_cast<const float4*>(R.mean + cr), is = *reinterpret_cast<const float4*>(R.invstd + cr);
        scale[4 * q + 0] = ga.x * is.x; scale[4 * q + 1] = ga.y * is.y; scale[4 * q + 2] = ga.z * is.z; scale[4 * q + 3] = ga.w * is.w;
        shift[4 * q + 0] = be.x - mu.x * scale[4 * q + 0]; shift[4 * q + 1] = be.y - mu.y * scale[4 * q + 1];
        shift[4 * q + 2] = be.z - mu.z * scale[4 * q + 2]; shift[4 * q + 3] = be.w - mu.w * scale[4 * q + 3];
        ob[q] = R.out + (long)(R.out_c0 + cr) * HW;
        oct[q] = (long)R.out_ctot * HW;
    }
    const long img_stride = (long)B.Cout * HW;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        if (yb[j] < 0) continue;
        const long img = yb[j] / img_stride, pix = yb[j] - img * img_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!qok[q]) continue;
            float* po = ob[q] + img * oct[q] + pix;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaf(acc[j][4 * q + e], scale[4 * q + e], shift[4 * q + e]);
                po[(long)e * HW] = B.act ? eas_epi_silu(z) : z;
            }
        }
    }
}

