// K4 (second half): BatchNorm(step_mode='m') + multi-step LIF fused, forward and backward.
//
// Data: y = conv output, [T][N][C][HW] fp32.  BN statistics couple all of T*N*HW per channel, so
// training needs a statistics pass before the fused apply pass (SURVEY "Hard parts").
// Work decomposition for every kernel here: blockIdx.y = channel c, blockIdx.x = chunk of that
// channel's N*HW/4 float4 groups.  Per-channel constants (scale, shift, means) are block-uniform,
// reductions are per block -> per (channel, chunk) partial -> fixed-order finalize (deterministic).
// Reads are contiguous runs of HW floats (one (n,c) plane), 16 B per lane.
//
// HBM traffic per neuron-step: stats 4 B (read y); fwd 8 B (read y, write s);
// bwd 20 B (pass 1 reads y, grad_s; pass 2 reads them again and writes grad_y).  Only y is kept
// for backward: h_t is recomputed in registers from y (T <= 8).
#include <stdlib.h>

#include "eas_common.h"

namespace {

constexpr int VEC = 4;
constexpr int kMaxChunks = 64;
constexpr int NW = EAS_BLOCK / EAS_WAVE;

static inline int pick_chunks(int64_t groups_per_channel, int C) {
    int64_t want = (groups_per_channel + EAS_BLOCK - 1) / EAS_BLOCK;  // one group per thread
    int64_t cap = 8192 / (C > 0 ? C : 1);                                // keep the grid around <= 8k blocks
    if (cap < 1) cap = 1;
    if (cap > kMaxChunks) cap = kMaxChunks;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

// ------------------------------------------------------------------------------------------------ stats
__global__ __launch_bounds__(EAS_BLOCK) void bn_stats_partial(const float* __restrict__ y, int TN, int C, int HW,
                                                              double* __restrict__ part, int Cy) {
    __shared__ double red[NW];
    const int c = blockIdx.y;
    const int hw4 = HW / VEC;
    const int64_t groups = (int64_t)TN * hw4;
    double s = 0.0, ss = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    GroupWalk gw((int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride, hw4);
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // two groups in flight per thread (both loads issued before the double-precision accumulation of either); the order in
    // which a thread adds its groups is unchanged
    for (; g + stride < groups; g += 2 * stride) {
        const float4 v0 = reinterpret_cast<const float4*>(y + (((int64_t)gw.n * Cy + c) * (int64_t)HW))[gw.q];
        gw.next();
        const float4 v1 = reinterpret_cast<const float4*>(y + (((int64_t)gw.n * Cy + c) * (int64_t)HW))[gw.q];
        gw.next();
        s += (double)v0.x + (double)v0.y + (double)v0.z + (double)v0.w;
        ss += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
        s += (double)v1.x + (double)v1.y + (double)v1.z + (double)v1.w;
        ss += (double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w;
    }
    if (g < groups) {
        const float4 v = reinterpret_cast<const float4*>(y + (((int64_t)gw.n * Cy + c) * (int64_t)HW))[gw.q];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    const double ts = eas_block_sum<double, NW>(s, red);
    const double tss = eas_block_sum<double, NW>(ss, red);
    if (threadIdx.x == 0) {
        part[((int64_t)c * kMaxChunks + blockIdx.x) * 2 + 0] = ts;
        part[((int64_t)c * kMaxChunks + blockIdx.x) * 2 + 1] = tss;
    }
}

// scalar path for HW % 4 != 0
__global__ __launch_bounds__(EAS_BLOCK) void bn_stats_partial_scalar(const float* __restrict__ y, int TN, int C, int HW,
                                                                     double* __restrict__ part) {
    __shared__ double red[NW];
    const int c = blockIdx.y;
    const int64_t groups = (int64_t)TN * HW;
    double s = 0.0, ss = 0.0;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = g / HW;
        const int q = (int)(g - n * HW);
        const float v = y[(n * C + c) * (int64_t)HW + q];
        s += v;
        ss += (double)v * v;
    }
    const double ts = eas_block_sum<double, NW>(s, red);
    const double tss = eas_block_sum<double, NW>(ss, red);
    if (threadIdx.x == 0) {
        part[((int64_t)c * kMaxChunks + blockIdx.x) * 2 + 0] = ts;
        part[((int64_t)c * kMaxChunks + blockIdx.x) * 2 + 1] = tss;
    }
}

__global__ __launch_bounds__(EAS_WAVE) void bn_stats_finalize(const double* __restrict__ part, int nchunks, double count,
                                                              int replicas, float eps, float momentum, float* __restrict__ mean,
                                                              float* __restrict__ invstd, float* __restrict__ rmean,
                                                              float* __restrict__ rvar) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    if ((int)threadIdx.x < nchunks) {
        s = part[((int64_t)c * kMaxChunks + threadIdx.x) * 2 + 0];
        ss = part[((int64_t)c * kMaxChunks + threadIdx.x) * 2 + 1];
    }
    s = eas_wave_sum(s);
    ss = eas_wave_sum(ss);
    if (threadIdx.x == 0) {
        const double m = s / count;
        double var = ss / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            const double full = count * replicas;   // identical replicas leave mean/var unchanged, only the sample count grows
            const double unbiased = full > 1.0 ? var * full / (full - 1.0) : var;
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unbiased);
        }
    }
}

struct BnLifOut {
    const float* residual;   // nullable [T][N][C][HW]: spikes_out = spikes + residual (SEW shortcut, network_blocks.py:99-104)
    int out_ctot;            // 0: dense; else spikes are channels of a [T][N][out_ctot][HW] tensor (pointer already at channel 0 of the slice)
    int y_ctot;              // 0: dense; else y is such a channel slice of a [T][N][y_ctot][HW] tensor (one convolution feeding two BN+LIF layers)
};

// Spike planes (eas_hip.h "spike planes"): the output values (0/1 spikes, SEW sums: exact in bf16) as bf16 in blocks of 8 channels,
// [T*N][C/8][HW][8] -- the layout in which every consumer's matrix-core fragment (8 consecutive input channels of one pixel) is ONE
// 16-byte load, at half the bytes of fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ forward
template <int T_, bool HARD, bool DI, bool STRICT>
__global__ __launch_bounds__(EAS_BLOCK) void bn_lif_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* v_in, float* v_out,
                                                               EasLifParams p, float* __restrict__ spikes,
                                                               float* __restrict__ mean_out, int N, int C, int HW, int bcast,
                                                               BnFin fin, BnLifOut ox) {
    __shared__ float st[2];
    const int c = blockIdx.z * 8 + blockIdx.x;      // EAS_CHAN_GRID: the chunks of a channel share an XCD
    if (c >= C) return;
    const int chunk = blockIdx.y, nchunks_g = gridDim.y;
    float mu, istd;
    if (fin.part) {
        bn_finalize_in_block(fin, c, st, mu, istd, chunk == 0);
    } else {
        mu = mean[c];
        istd = invstd[c];
    }
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    const int64_t Mo = ox.out_ctot ? (int64_t)N * ox.out_ctot * HW : (int64_t)N * C * HW;
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const int hw4 = HW / VEC;
    const int64_t groups = (int64_t)N * hw4;
    const int64_t M = (int64_t)N * C * HW;
    const int Cy = ox.y_ctot ? ox.y_ctot : C;
    const int64_t yts = bcast ? 0 : (int64_t)N * Cy * HW;       // T identical input frames: one plane stands for all steps
    GroupWalk gw((int64_t)chunk * blockDim.x + threadIdx.x, (int64_t)nchunks_g * blockDim.x, hw4);
    for (int64_t g = (int64_t)chunk * blockDim.x + threadIdx.x; g < groups; g += (int64_t)nchunks_g * blockDim.x, gw.next()) {
        const int64_t n = gw.n;
        const int q = gw.q;
        const int64_t base = (n * C + c) * (int64_t)HW + (int64_t)q * VEC;
        float4 ys[T_];
        const int64_t ybase = (n * Cy + c) * (int64_t)HW + (int64_t)q * VEC;
#pragma unroll
        for (int t = 0; t < T_; ++t) ys[t] = *reinterpret_cast<const float4*>(y + (int64_t)t * yts + ybase);
        const float vr0 = HARD ? p.v_reset : 0.0f;
        float4 v = v_in ? *reinterpret_cast<const float4*>(v_in + base) : make_float4(vr0, vr0, vr0, vr0);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < T_; ++t) {
            float4 h, s;
            eas_lif_step<HARD, DI, STRICT>(v.x, fmaf(ys[t].x, scale, shift), k, omk, p.v_th, p.v_reset, h.x, s.x);
            eas_lif_step<HARD, DI, STRICT>(v.y, fmaf(ys[t].y, scale, shift), k, omk, p.v_th, p.v_reset, h.y, s.y);
            eas_lif_step<HARD, DI, STRICT>(v.z, fmaf(ys[t].z, scale, shift), k, omk, p.v_th, p.v_reset, h.z, s.z);
            eas_lif_step<HARD, DI, STRICT>(v.w, fmaf(ys[t].w, scale, shift), k, omk, p.v_th, p.v_reset, h.w, s.w);
            acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
            if (ox.residual) {
                const float4 r = *reinterpret_cast<const float4*>(ox.residual + (int64_t)t * M + base);
                s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
            }
            const int64_t obase = ox.out_ctot ? (n * ox.out_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base;
            *reinterpret_cast<float4*>(spikes + (int64_t)t * Mo + obase) = s;
        }
        if (v_out) *reinterpret_cast<float4*>(v_out + base) = v;
        if (mean_out) {
            const float Tf = (float)T_;   // sum / T, like ATen's mean
            *reinterpret_cast<float4*>(mean_out + base) = make_float4(acc.x / Tf, acc.y / Tf, acc.z / Tf, acc.w / Tf);
        }
    }
}

// mean / invstd (and the running statistics) of every channel from the chunk partials, one block per channel, in the summation order of
// bn_finalize_in_block: the statistics the plane-writing kernel below reads are bit-identical to the ones the fp32 kernel forms itself.
__global__ __launch_bounds__(EAS_BLOCK) void bn_finalize_kernel(BnFin fin) {
    __shared__ float st[2];
    float mu, istd;
    bn_finalize_in_block(fin, (int)blockIdx.x, st, mu, istd, true);
}

// The same layer writing its output as spike planes.  A thread owns the 8 channels of one group for PV consecutive pixels of an image
// (PV = 1 is what runs): 8 x T loads of PV floats (each channel row is contiguous over the lanes), the neuron over T in registers, and
// per step ONE 16-byte store per pixel.  blockIdx -> (channel group, chunk of the group's N * HW / PV pixel groups).  residual: planes as well.
template <int T_, bool HARD, bool DI, bool STRICT, int PV>
__global__ __launch_bounds__(EAS_BLOCK) void bn_lif_fwd_sp_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* v_in, float* v_out,
                                                                  EasLifParams p, bf16x8* __restrict__ planes, const bf16x8* __restrict__ res_planes,
                                                                  float* __restrict__ mean_out, int N, int C, int HW, int bcast, int y_ctot,
                                                                  int out_groups_tot, int res_groups_tot) {
    typedef float vecp __attribute__((ext_vector_type(PV)));
    const int grp = blockIdx.z * 8 + blockIdx.x;
    const int G = C / 8;
    if (grp >= G) return;
    const int chunk = blockIdx.y, nchunks_g = gridDim.y;
    float scale[8], shift[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * grp + j;
        scale[j] = gamma[c] * invstd[c];
        shift[j] = beta[c] - mean[c] * scale[j];
    }
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const int hwp = HW / PV;
    const int64_t groups = (int64_t)N * hwp;
    const int Cy = y_ctot ? y_ctot : C;
    const int64_t yts = bcast ? 0 : (int64_t)N * Cy * HW;
    const float vr0 = HARD ? p.v_reset : 0.0f;
    const float Tf = (float)T_;
    GroupWalk gw((int64_t)chunk * blockDim.x + threadIdx.x, (int64_t)nchunks_g * blockDim.x, hwp);
    for (int64_t g = (int64_t)chunk * blockDim.x + threadIdx.x; g < groups; g += (int64_t)nchunks_g * blockDim.x, gw.next()) {
        const int64_t n = gw.n;
        const int q = gw.q;
        vecp ys[8][T_];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t yb = (n * Cy + 8 * grp + j) * (int64_t)HW + (int64_t)q * PV;
#pragma unroll
            for (int t = 0; t < T_; ++t) ys[j][t] = *reinterpret_cast<const vecp*>(y + (int64_t)t * yts + yb);
        }
        float v[8][PV], acc[8][PV];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vecp v0;
            if (v_in) v0 = *reinterpret_cast<const vecp*>(v_in + (n * C + 8 * grp + j) * (int64_t)HW + (int64_t)q * PV);
#pragma unroll
            for (int e = 0; e < PV; ++e) { v[j][e] = v_in ? v0[e] : vr0; acc[j][e] = 0.f; }
        }
#pragma unroll
        for (int t = 0; t < T_; ++t) {
            bf16x8 rp[PV];
            if (res_planes) {
#pragma unroll
                for (int e = 0; e < PV; ++e) rp[e] = res_planes[(((int64_t)t * N + n) * res_groups_tot + grp) * (int64_t)HW + (int64_t)q * PV + e];
            }
#pragma unroll
            for (int e = 0; e < PV; ++e) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float h, s;
                    eas_lif_step<HARD, DI, STRICT>(v[j][e], fmaf(ys[j][t][e], scale[j], shift[j]), k, omk, p.v_th, p.v_reset, h, s);
                    acc[j][e] += s;
                    if (res_planes) s += (float)rp[e][j];
                    o[j] = (__bf16)s;
                }
                planes[(((int64_t)t * N + n) * out_groups_tot + grp) * (int64_t)HW + (int64_t)q * PV + e] = o;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t base = (n * C + 8 * grp + j) * (int64_t)HW + (int64_t)q * PV;
            if (v_out) {
                vecp o;
#pragma unroll
                for (int e = 0; e < PV; ++e) o[e] = v[j][e];
                *reinterpret_cast<vecp*>(v_out + base) = o;
            }
            if (mean_out) {
                vecp o;
#pragma unroll
                for (int e = 0; e < PV; ++e) o[e] = acc[j][e] / Tf;
                *reinterpret_cast<vecp*>(mean_out + base) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Recompute h_t from y, then walk the LIF backward: dz[t] = dL/dz_t (z = BN output), dk += dL/dk.
template <int T_, bool HARD, bool DI, bool STRICT>
__device__ __forceinline__ void recompute_dz(const float (&yv)[T_], const float (&gs)[T_], float v0, float scale,
                                             float shift, float k, float omk, const EasLifParams& p, bool detach,
                                             int sg_id, float alpha, float (&dz)[T_], float& dk, float& da) {
    float h[T_], vprev[T_], z[T_];
    float v = v0;
#pragma unroll
    for (int t = 0; t < T_; ++t) {
        float s;
        vprev[t] = v;
        z[t] = fmaf(yv[t], scale, shift);
        eas_lif_step<HARD, DI, STRICT>(v, z[t], k, omk, p.v_th, p.v_reset, h[t], s);
    }
    float gv = 0.f;
#pragma unroll
    for (int t = T_ - 1; t >= 0; --t) {
        float dkt, gx;
        eas_lif_step_bwd<HARD, DI, STRICT>(gs[t], gv, h[t], vprev[t], z[t], k, omk, p.v_th, p.v_reset, detach, sg_id,
                                           alpha, dkt, gx, da);
        dz[t] = gx;
        dk += dkt;
    }
}

template <int T_, bool HARD, bool DI, bool STRICT, bool APPLY>
__global__ __launch_bounds__(EAS_BLOCK) void bn_lif_bwd_kernel(
    const float* __restrict__ grad_s, const float* __restrict__ grad_mean, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ v_init, EasLifParams p, int sg_id, float alpha,
    int batch_stats, double* __restrict__ part, int nchunks, float* __restrict__ grad_y, float* __restrict__ grad_gamma,
    float* __restrict__ grad_beta, float* __restrict__ grad_w, int N, int C, int HW, int bcast, int gs_ctot, int y_ctot,
    const float* __restrict__ alpha_dev, float* __restrict__ grad_alpha, int c_begin, int c_end, int finalize) {
    __shared__ double red[NW];
    __shared__ float bc[2];
    if (alpha_dev) alpha = fabsf(*alpha_dev);    // learnable slope (EAS_SG_PATAN)
    // channels [c_begin, c_end) of the layer: a launch pair (sums, apply) may cover a channel group only, so that the apply pass finds the
    // group's y / grad_s still in the Infinity Cache (launch_bwd_t)
    const int c = c_begin + blockIdx.z * 8 + blockIdx.x;      // EAS_CHAN_GRID: the chunks of a channel share an XCD
    if (c >= c_end) return;
    const int chunk = blockIdx.y, nchunks_g = gridDim.y;
    const float mu = mean[c], istd = invstd[c];
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const bool detach = (p.flags & EAS_LIF_DETACH_RESET) != 0;
    const int hw4 = HW / VEC;
    const int64_t groups = (int64_t)N * hw4;
    const int64_t M = (int64_t)N * C * HW;
    const float invT = 1.0f / (float)T_;
    const int Cy = y_ctot ? y_ctot : C;                               // y / grad_y may be channel slices of a wider tensor
    const int64_t My = (int64_t)N * Cy * HW;
    const int64_t yts = bcast ? 0 : My;
    const int64_t Mg = gs_ctot ? (int64_t)N * gs_ctot * HW : M;      // grad_s may be a channel slice of a wider tensor
    float m1 = 0.f, m2 = 0.f;
    if (APPLY) {
        // fixed-order reduction of this channel's chunk partials (every block computes the same value)
        if (threadIdx.x < EAS_WAVE) {
            double s1 = 0.0, s2 = 0.0;
            if ((int)threadIdx.x < nchunks) {
                s1 = part[((int64_t)c * kMaxChunks + threadIdx.x) * 4 + 0];
                s2 = part[((int64_t)c * kMaxChunks + threadIdx.x) * 4 + 1];
            }
            s1 = eas_wave_sum(s1);
            s2 = eas_wave_sum(s2);
            if (threadIdx.x == 0) {
                const double cnt = (double)T_ * N * HW;
                bc[0] = (float)(s1 / cnt);
                bc[1] = (float)(s2 / cnt);
                if (chunk == 0) {
                    grad_beta[c] = (float)s1;
                    grad_gamma[c] = (float)s2;
                }
            }
        }
        __syncthreads();
        if (batch_stats) { m1 = bc[0]; m2 = bc[1]; }
    }
    float s1 = 0.f, s2 = 0.f, dk = 0.f, da = 0.f;
    GroupWalk gw((int64_t)chunk * blockDim.x + threadIdx.x, (int64_t)nchunks_g * blockDim.x, hw4);
    for (int64_t g = (int64_t)chunk * blockDim.x + threadIdx.x; g < groups; g += (int64_t)nchunks_g * blockDim.x, gw.next()) {
        const int64_t n = gw.n;
        const int q = gw.q;
        const int64_t base = (n * C + c) * (int64_t)HW + (int64_t)q * VEC;
        float4 ys[T_], gsv[T_];
        const int64_t ybase = (n * Cy + c) * (int64_t)HW + (int64_t)q * VEC;
#pragma unroll
        for (int t = 0; t < T_; ++t) ys[t] = *reinterpret_cast<const float4*>(y + (int64_t)t * yts + ybase);
        float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grad_mean) {
            gm = *reinterpret_cast<const float4*>(grad_mean + base);
            gm.x *= invT; gm.y *= invT; gm.z *= invT; gm.w *= invT;
        }
#pragma unroll
        for (int t = 0; t < T_; ++t) {
            gsv[t] = gm;
            if (grad_s) {
                const int64_t gbase = gs_ctot ? (n * gs_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base;
                const float4 g4 = *reinterpret_cast<const float4*>(grad_s + (int64_t)t * Mg + gbase);
                gsv[t].x += g4.x; gsv[t].y += g4.y; gsv[t].z += g4.z; gsv[t].w += g4.w;
            }
        }
        const float vr0 = HARD ? p.v_reset : 0.0f;
        float4 v0 = make_float4(vr0, vr0, vr0, vr0);
        if (v_init) v0 = *reinterpret_cast<const float4*>(v_init + base);
        float4 outv[T_];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float yv[T_], gs[T_], dz[T_];
#pragma unroll
            for (int t = 0; t < T_; ++t) {
                yv[t] = reinterpret_cast<const float*>(&ys[t])[e];
                gs[t] = reinterpret_cast<const float*>(&gsv[t])[e];
            }
            const float v0e = reinterpret_cast<const float*>(&v0)[e];
            float dke = 0.f;
            recompute_dz<T_, HARD, DI, STRICT>(yv, gs, v0e, scale, shift, k, omk, p, detach, sg_id, alpha, dz, dke, da);
#pragma unroll
            for (int t = 0; t < T_; ++t) {
                const float xhat = (yv[t] - mu) * istd;
                if (APPLY) {
                    reinterpret_cast<float*>(&outv[t])[e] = scale * (dz[t] - m1 - xhat * m2);
                } else {
                    s1 += dz[t];
                    s2 += dz[t] * xhat;
                }
            }
            dk += dke;
        }
        if (APPLY) {
            if (bcast) {                            // the T steps share one input plane: its gradient is their sum
                float4 a = outv[0];
#pragma unroll
                for (int t = 1; t < T_; ++t) { a.x += outv[t].x; a.y += outv[t].y; a.z += outv[t].z; a.w += outv[t].w; }
                *reinterpret_cast<float4*>(grad_y + ybase) = a;
            } else {
#pragma unroll
                for (int t = 0; t < T_; ++t) *reinterpret_cast<float4*>(grad_y + (int64_t)t * My + ybase) = outv[t];
            }
        }
    }
    if (APPLY && grad_w && finalize && chunk == 0 && c == c_begin) {
        // dL/dw of the (scalar) PLIF decay: fixed-order sum of every (channel, chunk) partial of pass 1, by one block
        double acc = 0.0;
        for (int i = threadIdx.x; i < C * nchunks; i += blockDim.x) {
            const int cc = i / nchunks, j = i - cc * nchunks;
            acc += part[((int64_t)cc * kMaxChunks + j) * 4 + 2];
        }
        const double tot = eas_block_sum<double, NW>(acc, red);
        if (threadIdx.x == 0) *grad_w = (float)tot * (k * (1.0f - k));
    }
    if (APPLY && grad_alpha && finalize && chunk == 0 && c == c_begin) {
        // dL/dalpha of the learnable surrogate slope: same fixed-order sum over slot 3 of the partials
        double acc = 0.0;
        for (int i = threadIdx.x; i < C * nchunks; i += blockDim.x) {
            const int cc = i / nchunks, j = i - cc * nchunks;
            acc += part[((int64_t)cc * kMaxChunks + j) * 4 + 3];
        }
        const double tot = eas_block_sum<double, NW>(acc, red);
        if (threadIdx.x == 0) {
            const float a = *alpha_dev;
            *grad_alpha = (float)tot * (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f));
        }
    }
    if (!APPLY) {
        const double t1 = eas_block_sum<double, NW>((double)s1, red);
        const double t2 = eas_block_sum<double, NW>((double)s2, red);
        const double t3 = eas_block_sum<double, NW>((double)dk, red);
        double t4 = 0.0;
        if (sg_id == EAS_SG_PATAN) t4 = eas_block_sum<double, NW>((double)da, red);
        if (threadIdx.x == 0) {
            double* o = part + ((int64_t)c * kMaxChunks + chunk) * 4;
            o[0] = t1; o[1] = t2; o[2] = t3; o[3] = t4;
        }
    }
}

// One pass for the layers whose channel fits the registers of one block (8x10 maps at batch 64: 1280 float4 groups per channel and time
// step): one block per channel keeps y and grad_s of the channel in registers, sums dz and dz * xhat over the block (fixed order), and
// applies the BatchNorm backward from the registers -- 12 B per neuron-step instead of 20 and one real launch instead of two of ~17 us
// each (these layers are bound by the launches, not by their 30 MB).  Per element the arithmetic of the two-pass kernels; the per-channel
// dL/dk / dL/dalpha partials go to the same workspace slots (chunk 0) and a one-block kernel adds them over the channels in fixed order.
// (first half.)
template <int T_, bool HARD, bool DI, bool STRICT, int NT_, int GPT>
__global__ __launch_bounds__(NT_) void bn_lif_bwd_small_kernel(
    const float* __restrict__ grad_s, const float* __restrict__ grad_mean, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ v_init, EasLifParams p, int sg_id, float alpha,
    int batch_stats, double* __restrict__ part, float* __restrict__ grad_y, float* __restrict__ grad_gamma,
    float* __restrict__ grad_beta, int N, int C, int HW, int gs_ctot, int y_ctot, const float* __restrict__ alpha_dev) {
    constexpr int NWV = NT_ / EAS_WAVE;
    __shared__ double red[4][NWV];
    __shared__ float bc[2];
    if (alpha_dev) alpha = fabsf(*alpha_dev);
    const int c = blockIdx.x;
    const float mu = mean[c], istd = invstd[c];
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const bool detach = (p.flags & EAS_LIF_DETACH_RESET) != 0;
    const int hw4 = HW / VEC;
    const int groups = N * hw4;
    const int64_t M = (int64_t)N * C * HW;
    const int Cy = y_ctot ? y_ctot : C;
    const int64_t My = (int64_t)N * Cy * HW;
    const int64_t Mg = gs_ctot ? (int64_t)N * gs_ctot * HW : M;
    const float invT = 1.0f / (float)T_;
    float4 ys[GPT][T_], gsv[GPT][T_], v0[GPT];
    int64_t ybase[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g = threadIdx.x + i * NT_;
        const int gg = g < groups ? g : groups - 1;          // surplus threads re-read the last group (their results are not used)
        const int n = gg / hw4, q = gg - n * hw4;
        const int64_t base = ((int64_t)n * C + c) * (int64_t)HW + (int64_t)q * VEC;
        ybase[i] = ((int64_t)n * Cy + c) * (int64_t)HW + (int64_t)q * VEC;
#pragma unroll
        for (int t = 0; t < T_; ++t) ys[i][t] = *reinterpret_cast<const float4*>(y + (int64_t)t * My + ybase[i]);
        float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grad_mean) {
            gm = *reinterpret_cast<const float4*>(grad_mean + base);
            gm.x *= invT; gm.y *= invT; gm.z *= invT; gm.w *= invT;
        }
#pragma unroll
        for (int t = 0; t < T_; ++t) {
            gsv[i][t] = gm;
            if (grad_s) {
                const int64_t gbase = gs_ctot ? ((int64_t)n * gs_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base;
                const float4 g4 = *reinterpret_cast<const float4*>(grad_s + (int64_t)t * Mg + gbase);
                gsv[i][t].x += g4.x; gsv[i][t].y += g4.y; gsv[i][t].z += g4.z; gsv[i][t].w += g4.w;
            }
        }
        const float vr0 = HARD ? p.v_reset : 0.0f;
        v0[i] = make_float4(vr0, vr0, vr0, vr0);
        if (v_init) v0[i] = *reinterpret_cast<const float4*>(v_init + base);
    }
    // pass 1 (registers): the channel's sums
    float s1 = 0.f, s2 = 0.f, dk = 0.f, da = 0.f;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const bool live = threadIdx.x + i * NT_ < groups;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float yv[T_], gs[T_], dz[T_];
#pragma unroll
            for (int t = 0; t < T_; ++t) {
                yv[t] = reinterpret_cast<const float*>(&ys[i][t])[e];
                gs[t] = reinterpret_cast<const float*>(&gsv[i][t])[e];
            }
            float dke = 0.f, dae = 0.f;
            recompute_dz<T_, HARD, DI, STRICT>(yv, gs, reinterpret_cast<const float*>(&v0[i])[e], scale, shift, k, omk, p, detach, sg_id, alpha, dz, dke, dae);
            if (live) {
#pragma unroll
                for (int t = 0; t < T_; ++t) {
                    s1 += dz[t];
                    s2 += dz[t] * ((yv[t] - mu) * istd);
                }
                dk += dke;
                da += dae;
            }
        }
    }
    double t1 = eas_wave_sum((double)s1), t2 = eas_wave_sum((double)s2), t3 = eas_wave_sum((double)dk), t4 = eas_wave_sum((double)da);
    if ((threadIdx.x & (EAS_WAVE - 1)) == 0) {
        const int w = threadIdx.x / EAS_WAVE;
        red[0][w] = t1; red[1][w] = t2; red[2][w] = t3; red[3][w] = t4;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t1 = t2 = t3 = t4 = 0.0;
        for (int w = 0; w < NWV; ++w) { t1 += red[0][w]; t2 += red[1][w]; t3 += red[2][w]; t4 += red[3][w]; }
        const double cnt = (double)T_ * N * HW;
        bc[0] = (float)(t1 / cnt);
        bc[1] = (float)(t2 / cnt);
        grad_beta[c] = (float)t1;
        grad_gamma[c] = (float)t2;
        double* o = part + ((int64_t)c * kMaxChunks) * 4;      // chunk 0 of the two-pass layout: read by bn_lif_bwd_scalars_kernel
        o[0] = t1; o[1] = t2; o[2] = t3; o[3] = sg_id == EAS_SG_PATAN ? t4 : 0.0;
    }
    __syncthreads();
    const float m1 = batch_stats ? bc[0] : 0.f, m2 = batch_stats ? bc[1] : 0.f;
    // pass 2 (registers): the neuron again, grad_y
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        if (threadIdx.x + i * NT_ >= groups) continue;
        float4 outv[T_];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float yv[T_], gs[T_], dz[T_];
#pragma unroll
            for (int t = 0; t < T_; ++t) {
                yv[t] = reinterpret_cast<const float*>(&ys[i][t])[e];
                gs[t] = reinterpret_cast<const float*>(&gsv[i][t])[e];
            }
            float dke = 0.f, dae = 0.f;
            recompute_dz<T_, HARD, DI, STRICT>(yv, gs, reinterpret_cast<const float*>(&v0[i])[e], scale, shift, k, omk, p, detach, sg_id, alpha, dz, dke, dae);
#pragma unroll
            for (int t = 0; t < T_; ++t) reinterpret_cast<float*>(&outv[t])[e] = scale * (dz[t] - m1 - ((yv[t] - mu) * istd) * m2);
        }
#pragma unroll
        for (int t = 0; t < T_; ++t) *reinterpret_cast<float4*>(grad_y + (int64_t)t * My + ybase[i]) = outv[t];
    }
}

// dL/dw of the scalar PLIF decay and dL/dalpha of a learnable surrogate slope from the per-channel partials of bn_lif_bwd_small_kernel
// (slot 2 / 3 of chunk 0), added over the channels in fixed order by one block -- what the apply pass of the two-pass form does at its end
__global__ __launch_bounds__(EAS_BLOCK) void bn_lif_bwd_scalars_kernel(const double* __restrict__ part, int C, EasLifParams p, float* __restrict__ grad_w,
                                                                       const float* __restrict__ alpha_dev, float* __restrict__ grad_alpha) {
    __shared__ double red[NW];
    const float k = eas_lif_k(p);
    if (grad_w) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < C; i += blockDim.x) acc += part[((int64_t)i * kMaxChunks) * 4 + 2];
        const double tot = eas_block_sum<double, NW>(acc, red);
        if (threadIdx.x == 0) *grad_w = (float)tot * (k * (1.0f - k));
    }
    if (grad_alpha) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < C; i += blockDim.x) acc += part[((int64_t)i * kMaxChunks) * 4 + 3];
        const double tot = eas_block_sum<double, NW>(acc, red);
        if (threadIdx.x == 0) {
            const float a = *alpha_dev;
            *grad_alpha = (float)tot * (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f));
        }
    }
}

template <int T_, bool HARD, bool DI, bool STRICT>
int launch_fwd_t(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                 const float* v_in, float* v_out, EasLifParams p, float* spikes, float* mean_out, int N, int C, int HW,
                 int bcast, const BnFin& fin, const BnLifOut& ox, bf16x8* planes, const bf16x8* res_planes, int out_groups, int res_groups,
                 hipStream_t st) {
    if (planes) {
        // spike planes out: the statistics are finalized by their own small launch (a block of this kernel owns 8 channels: finalizing
        // them inside every block would cost 8x the partial-sum traffic), then one thread = 8 channels x PV pixels
        if (fin.part) {              // publishes mean / invstd (the arrays this launch was given) and updates the running statistics
            EAS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(EAS_BLOCK), 0, st, fin);
            EAS_CHECK_LAUNCH();
        }
        // one pixel per thread: 8 x T dword loads (each channel row contiguous over the lanes) and T 16-byte stores, few registers, many
        // waves in flight (measured on config 2: 1.02 ms per step for all layers against 1.10 / 1.24 ms with 2 / 4 pixels per thread and
        // 1.10 ms for the fp32-writing kernel)
        constexpr int PV = 1;
        const int chunks = pick_chunks((int64_t)N * (HW / PV), C / 8);
        EAS_LAUNCH((bn_lif_fwd_sp_kernel<T_, HARD, DI, STRICT, PV>), EAS_CHAN_GRID(chunks, C / 8), dim3(EAS_BLOCK), 0, st, y,
                           mean, invstd, gamma, beta, v_in,
                           v_out, p, planes, res_planes, mean_out, N, C, HW, bcast, ox.y_ctot, out_groups, res_groups);
        EAS_CHECK_LAUNCH();
        return EAS_OK;
    }
    const int chunks = pick_chunks((int64_t)N * (HW / VEC), C);
    EAS_LAUNCH((bn_lif_fwd_kernel<T_, HARD, DI, STRICT>), EAS_CHAN_GRID(chunks, C), dim3(EAS_BLOCK), 0, st, y, mean,
                       invstd, gamma, beta, v_in, v_out, p, spikes, mean_out, N, C, HW, bcast, fin, ox);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

template <bool HARD, bool DI, bool STRICT>
int launch_fwd(int T, const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
               const float* v_in, float* v_out, EasLifParams p, float* spikes, float* mean_out, int N, int C, int HW,
               int bcast, const BnFin& fin, const BnLifOut& ox, bf16x8* planes, const bf16x8* res_planes, int out_groups, int res_groups,
               hipStream_t st) {
#define EAS_CASE(TT) \
    case TT: return launch_fwd_t<TT, HARD, DI, STRICT>(y, mean, invstd, gamma, beta, v_in, v_out, p, spikes, mean_out, N, C, HW, bcast, fin, ox, planes, res_planes, out_groups, res_groups, st);
    switch (T) {
        EAS_CASE(1) EAS_CASE(2) EAS_CASE(3) EAS_CASE(4) EAS_CASE(5) EAS_CASE(6) EAS_CASE(7) EAS_CASE(8)
        default: return EAS_ERR_UNSUPPORTED;
    }
#undef EAS_CASE
}

template <int T_, bool HARD, bool DI, bool STRICT>
int launch_bwd_t(const float* grad_s, const float* grad_mean, const float* y, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, const float* v_init, EasLifParams p, int sg, float alpha,
                 int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta, float* grad_w, double* ws, int N,
                 int C, int HW, int bcast, int gs_ctot, int y_ctot, const float* alpha_dev, float* grad_alpha, hipStream_t st) {
    const int chunks = pick_chunks((int64_t)N * (HW / VEC), C);
    {
        // one block per channel where a channel fits the registers of a block (bn_lif_bwd_small_kernel); EAS_BNLIF_BWD=two keeps the
        // two-pass launches (development)
        const char* form = getenv("EAS_BNLIF_BWD");      // (read per call: a test compares the two forms in one process)
        const bool small_ok = !(form && form[0] == 't');
        constexpr int GPT = T_ <= 3 ? 3 : (T_ <= 5 ? 2 : 1);         // 8 * T floats per group in registers
        const int64_t groups = (int64_t)N * (HW / VEC);
        if (small_ok && !bcast && C >= 64 && groups <= 512 * GPT) {       // (1024-thread blocks are capped at 128 registers: they spill)
#define EAS_SMALL(NT_) EAS_LAUNCH((bn_lif_bwd_small_kernel<T_, HARD, DI, STRICT, NT_, GPT>), dim3(C), dim3(NT_), 0, st, grad_s, grad_mean, y, mean, \
                                   invstd, gamma, beta, v_init, p, sg, alpha, batch_stats, ws, grad_y, grad_gamma, grad_beta, N, C, HW, gs_ctot, y_ctot, alpha_dev)
            if (groups <= 256 * GPT) EAS_SMALL(256);
            else EAS_SMALL(512);
#undef EAS_SMALL
            EAS_CHECK_LAUNCH();
            if (grad_w || grad_alpha) {
                EAS_LAUNCH(bn_lif_bwd_scalars_kernel, dim3(1), dim3(EAS_BLOCK), 0, st, ws, C, p, grad_w, alpha_dev, grad_alpha);
                EAS_CHECK_LAUNCH();
            }
            return EAS_OK;
        }
    }
    // two passes: the BatchNorm backward needs the channel's sums of dz and dz * xhat before any grad_y; pass 2 recomputes the neuron
    // (cheaper than parking dz: measured).
    // Channel groups (EAS_BNLIF_BWD_GROUP_MB, development; 0 = one pair of launches for the whole layer): the pair runs group by group with
    // at most that many MB of y + grad_s per group, so that the apply pass of a group re-reads what its sum pass has just read while it is
    // still in the 256 MB Infinity Cache.
    static const int group_mb = eas_dev_env("EAS_BNLIF_BWD_GROUP_MB") ? atoi(eas_dev_env("EAS_BNLIF_BWD_GROUP_MB")) : 0;
    int gc = C;
    if (group_mb > 0) {
        const double per_channel = (double)N * HW * 4.0 * ((bcast ? 1 : T_) + T_);
        gc = (int)(group_mb * 1048576.0 / per_channel) / 8 * 8;
        if (gc < 8) gc = 8;
        if (gc > C) gc = C;
    }
    for (int c0 = 0; c0 < C; c0 += gc) {
        const int c1 = c0 + gc < C ? c0 + gc : C;
        const int last = c1 == C;
        EAS_LAUNCH((bn_lif_bwd_kernel<T_, HARD, DI, STRICT, false>), EAS_CHAN_GRID(chunks, c1 - c0), dim3(EAS_BLOCK), 0, st, grad_s,
                   grad_mean, y, mean, invstd, gamma, beta, v_init, p, sg, alpha, batch_stats, ws, chunks, grad_y,
                   grad_gamma, grad_beta, (float*)nullptr, N, C, HW, bcast, gs_ctot, y_ctot, alpha_dev, (float*)nullptr, c0, c1, 0);
        EAS_CHECK_LAUNCH();
        EAS_LAUNCH((bn_lif_bwd_kernel<T_, HARD, DI, STRICT, true>), EAS_CHAN_GRID(chunks, c1 - c0), dim3(EAS_BLOCK), 0, st, grad_s,
                   grad_mean, y, mean, invstd, gamma, beta, v_init, p, sg, alpha, batch_stats, ws, chunks, grad_y,
                   grad_gamma, grad_beta, grad_w, N, C, HW, bcast, gs_ctot, y_ctot, alpha_dev, grad_alpha, c0, c1, last);
        EAS_CHECK_LAUNCH();
    }
    return EAS_OK;
}

template <bool HARD, bool DI, bool STRICT>
int launch_bwd(int T, const float* grad_s, const float* grad_mean, const float* y, const float* mean,
               const float* invstd, const float* gamma, const float* beta, const float* v_init, EasLifParams p, int sg,
               float alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta, float* grad_w,
               double* ws, int N, int C, int HW, int bcast, int gs_ctot, int y_ctot, const float* alpha_dev, float* grad_alpha,
               hipStream_t st) {
#define EAS_CASE(TT)                                                                                               \
    case TT:                                                                                                       \
        return launch_bwd_t<TT, HARD, DI, STRICT>(grad_s, grad_mean, y, mean, invstd, gamma, beta, v_init, p, sg, alpha, \
                                                  batch_stats, grad_y, grad_gamma, grad_beta, grad_w, ws, N, C, HW, bcast, gs_ctot, y_ctot,         \
                                                  alpha_dev, grad_alpha, st);
    switch (T) {
        EAS_CASE(1) EAS_CASE(2) EAS_CASE(3) EAS_CASE(4) EAS_CASE(5) EAS_CASE(6) EAS_CASE(7) EAS_CASE(8)
        default: return EAS_ERR_UNSUPPORTED;
    }
#undef EAS_CASE
}

}  // namespace

extern "C" {

int64_t eas_bn_workspace_doubles(int C) { return (int64_t)C * kMaxChunks * 4; }

// launches the partial-sum kernel; returns the number of chunks per channel (> 0) or a negative status
static int stats_partial(const float* y, int TN, int C, int HW, double* workspace, int y_ctot, hipStream_t st) {
    const int Cy = y_ctot ? y_ctot : C;
    int chunks;
    if (HW % VEC == 0 && (((uintptr_t)y) & 15) == 0) {
        chunks = pick_chunks((int64_t)TN * (HW / VEC), C);
        EAS_LAUNCH(bn_stats_partial, dim3(chunks, C), dim3(EAS_BLOCK), 0, st, y, TN, C, HW, workspace, Cy);
    } else {
        if (Cy != C) return EAS_ERR_UNSUPPORTED;
        chunks = pick_chunks((int64_t)TN * HW, C);
        EAS_LAUNCH(bn_stats_partial_scalar, dim3(chunks, C), dim3(EAS_BLOCK), 0, st, y, TN, C, HW, workspace);
    }
    if (hipGetLastError() != hipSuccess) return EAS_ERR_LAUNCH;
    return chunks;
}

int eas_bn_stats(const float* y, int TN, int C, int HW, int replicas, float eps, float momentum, float* mean,
                 float* invstd, float* running_mean, float* running_var, double* workspace, eas_stream_t stream) {
    if (!y || !mean || !invstd || !workspace || TN < 1 || C < 1 || HW < 1 || replicas < 1) return EAS_ERR_INVALID_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EAS_ERR_INVALID_ARG;
    if (C > 65535) return EAS_ERR_UNSUPPORTED;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int chunks = stats_partial(y, TN, C, HW, workspace, 0, st);
    if (chunks < 0) return chunks;
    EAS_LAUNCH(bn_stats_finalize, dim3(C), dim3(EAS_WAVE), 0, st, workspace, chunks, (double)TN * HW, replicas, eps,
                       momentum, mean, invstd, running_mean, running_var);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_bn_stats_partial(const float* y, int y_ctot, int TN, int C, int HW, double* workspace, eas_stream_t stream) {
    if (!y || !workspace || TN < 1 || C < 1 || HW < 1 || (y_ctot != 0 && y_ctot < C)) return EAS_ERR_INVALID_ARG;
    if (C > 65535) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    return stats_partial(y, TN, C, HW, workspace, y_ctot == C ? 0 : y_ctot, eas_s(stream));
}

static int fin_from(const EasBnPending* pend, float* mean, float* invstd, BnFin& fin) {
    fin = BnFin{};
    if (!pend || !pend->partial) return EAS_OK;
    if (pend->chunks < 1 || pend->chunks > (pend->pitch ? pend->pitch : kMaxChunks) || pend->pitch < 0 || !(pend->count >= 1.0) || pend->replicas < 1) return EAS_ERR_INVALID_ARG;
    if ((pend->running_mean == nullptr) != (pend->running_var == nullptr)) return EAS_ERR_INVALID_ARG;
    fin.part = pend->partial; fin.nchunks = pend->chunks; fin.pitch = pend->pitch ? pend->pitch : kMaxChunks; fin.replicas = pend->replicas; fin.count = pend->count;
    fin.eps = pend->eps; fin.momentum = pend->momentum; fin.mean_out = mean; fin.invstd_out = invstd;
    fin.rmean = pend->running_mean; fin.rvar = pend->running_var;
    return EAS_OK;
}

int eas_bn_lif_fwd_ex(const float* y, int y_ctot, float* mean, float* invstd, const float* gamma, const float* beta,
                      const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th, float v_reset,
                      int flags, float* spikes, float* mean_out, int T, int N, int C, int HW, int y_bcast,
                      const EasBnPending* pending, const float* residual, int out_ctot, void* spikes_planes, const void* residual_planes,
                      int residual_ctot, eas_stream_t stream) {
    if (!y || !mean || !invstd || !gamma || !beta || (!spikes && !spikes_planes) || (spikes && spikes_planes) || T < 1 || N < 1 || C < 1 || HW < 1)
        return EAS_ERR_INVALID_ARG;
    if (HW % VEC != 0 || C > 65535 || T > 8) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)y | (uintptr_t)v_in | (uintptr_t)v_out | (uintptr_t)spikes | (uintptr_t)mean_out | (uintptr_t)residual |
         (uintptr_t)spikes_planes | (uintptr_t)residual_planes) & 15)
        return EAS_ERR_INVALID_ARG;
    if ((out_ctot != 0 && out_ctot < C) || (y_ctot != 0 && (y_ctot < C || y_bcast))) return EAS_ERR_INVALID_ARG;
    if (spikes_planes) {
        // planes are blocks of 8 channels: the layer's channels, its slice of a concatenation and a residual must be whole blocks
        if (C % 8 != 0 || (out_ctot % 8) != 0 || residual || (residual_planes && (residual_ctot % 8 != 0 || (residual_ctot != 0 && residual_ctot < C))))
            return EAS_ERR_UNSUPPORTED;
    } else if (residual_planes) {
        return EAS_ERR_UNSUPPORTED;
    }
    BnFin fin;
    if (int rc = fin_from(pending, mean, invstd, fin)) return rc;
    const BnLifOut ox{residual, out_ctot == C ? 0 : out_ctot, y_ctot == C ? 0 : y_ctot};
    EasLifParams p{w_logit, k_const, v_th, v_reset, flags};
    const bool hard = flags & EAS_LIF_HARD_RESET, di = flags & EAS_LIF_DECAY_INPUT, strict = flags & EAS_LIF_FIRE_STRICT;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int out_groups = (out_ctot ? out_ctot : C) / 8, res_groups = (residual_ctot ? residual_ctot : C) / 8;
#define EAS_DISPATCH(H, D, S)                                                                                                          \
    return launch_fwd<H, D, S>(T, y, mean, invstd, gamma, beta, v_in, v_out, p, spikes, mean_out, N, C, HW, y_bcast, fin, ox,           \
                               (bf16x8*)spikes_planes, (const bf16x8*)residual_planes, out_groups, res_groups, st)
    if (!hard && !di && !strict) EAS_DISPATCH(false, false, false);
    if (hard && !di && !strict) EAS_DISPATCH(true, false, false);
    if (!hard && di && !strict) EAS_DISPATCH(false, true, false);
    if (hard && di && !strict) EAS_DISPATCH(true, true, false);
    return EAS_ERR_UNSUPPORTED;  // strict '>' firing only exists in the sampler / in-repo LIFCell
#undef EAS_DISPATCH
}

int eas_bn_lif_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th, float v_reset,
                   int flags, float* spikes, float* mean_out, int T, int N, int C, int HW, int y_bcast,
                   eas_stream_t stream) {
    return eas_bn_lif_fwd_ex(y, 0, const_cast<float*>(mean), const_cast<float*>(invstd), gamma, beta, v_in, v_out, w_logit, k_const, v_th,
                             v_reset, flags, spikes, mean_out, T, N, C, HW, y_bcast, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

static int bn_lif_bwd_impl(const float* grad_s, int grad_s_ctot, const float* grad_mean, const float* y, int y_ctot, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, const float* v_init,
                           const float* w_logit, float k_const, float v_th, float v_reset, int flags, int surrogate,
                           float alpha, const float* alpha_dev, float* grad_alpha, int batch_stats, float* grad_y, float* grad_gamma,
                           float* grad_beta, float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast,
                           eas_stream_t stream) {
    if (!y || !mean || !invstd || !gamma || !beta || !grad_y || !grad_gamma || !grad_beta || !workspace ||
        (!grad_s && !grad_mean) || T < 1 || N < 1 || C < 1 || HW < 1)
        return EAS_ERR_INVALID_ARG;
    if (surrogate < EAS_SG_ATAN || surrogate > EAS_SG_PATAN) return EAS_ERR_INVALID_ARG;
    if (surrogate == EAS_SG_PATAN ? !alpha_dev : !(alpha > 0.f)) return EAS_ERR_INVALID_ARG;
    if (grad_w && !w_logit) return EAS_ERR_INVALID_ARG;
    if (HW % VEC != 0 || C > 65535 || T > 8) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)y | (uintptr_t)grad_s | (uintptr_t)grad_mean | (uintptr_t)v_init | (uintptr_t)grad_y) & 15)
        return EAS_ERR_INVALID_ARG;
    if ((grad_s_ctot != 0 && grad_s_ctot < C) || (y_ctot != 0 && (y_ctot < C || y_bcast))) return EAS_ERR_INVALID_ARG;
    const int gs_ctot = grad_s_ctot == C ? 0 : grad_s_ctot;
    const int yc = y_ctot == C ? 0 : y_ctot;
    EasLifParams p{w_logit, k_const, v_th, v_reset, flags};
    const bool hard = flags & EAS_LIF_HARD_RESET, di = flags & EAS_LIF_DECAY_INPUT, strict = flags & EAS_LIF_FIRE_STRICT;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
#define EAS_DISPATCH(H, D, S)                                                                                       \
    return launch_bwd<H, D, S>(T, grad_s, grad_mean, y, mean, invstd, gamma, beta, v_init, p, surrogate, alpha,     \
                               batch_stats, grad_y, grad_gamma, grad_beta, grad_w, workspace, N, C, HW, y_bcast, gs_ctot, yc,    \
                               alpha_dev, grad_alpha, st)
    if (!hard && !di && !strict) EAS_DISPATCH(false, false, false);
    if (hard && !di && !strict) EAS_DISPATCH(true, false, false);
    if (!hard && di && !strict) EAS_DISPATCH(false, true, false);
    if (hard && di && !strict) EAS_DISPATCH(true, true, false);
    return EAS_ERR_UNSUPPORTED;
#undef EAS_DISPATCH
}

int eas_bn_lif_bwd_ex(const float* grad_s, int grad_s_ctot, const float* grad_mean, const float* y, int y_ctot, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* v_init,
                      const float* w_logit, float k_const, float v_th, float v_reset, int flags, int surrogate,
                      float alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                      float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream) {
    if (surrogate == EAS_SG_PATAN) return EAS_ERR_INVALID_ARG;      // learnable slope: eas_bn_lif_bwd_patan
    return bn_lif_bwd_impl(grad_s, grad_s_ctot, grad_mean, y, y_ctot, mean, invstd, gamma, beta, v_init, w_logit, k_const, v_th, v_reset,
                           flags, surrogate, alpha, nullptr, nullptr, batch_stats, grad_y, grad_gamma, grad_beta, grad_w, workspace, T, N, C,
                           HW, y_bcast, stream);
}

int eas_bn_lif_bwd_patan(const float* grad_s, int grad_s_ctot, const float* grad_mean, const float* y, int y_ctot, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, const float* v_init,
                         const float* w_logit, float k_const, float v_th, float v_reset, int flags, const float* alpha,
                         float* grad_alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                         float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream) {
    return bn_lif_bwd_impl(grad_s, grad_s_ctot, grad_mean, y, y_ctot, mean, invstd, gamma, beta, v_init, w_logit, k_const, v_th, v_reset,
                           flags, EAS_SG_PATAN, 0.f, alpha, grad_alpha, batch_stats, grad_y, grad_gamma, grad_beta, grad_w, workspace, T, N, C,
                           HW, y_bcast, stream);
}

int eas_bn_lif_bwd(const float* grad_s, const float* grad_mean, const float* y, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, const float* v_init,
                   const float* w_logit, float k_const, float v_th, float v_reset, int flags, int surrogate,
                   float alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                   float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream) {
    return eas_bn_lif_bwd_ex(grad_s, 0, grad_mean, y, 0, mean, invstd, gamma, beta, v_init, w_logit, k_const, v_th, v_reset, flags,
                             surrogate, alpha, batch_stats, grad_y, grad_gamma, grad_beta, grad_w, workspace, T, N, C, HW, y_bcast, stream);
}

}  // extern "C"
