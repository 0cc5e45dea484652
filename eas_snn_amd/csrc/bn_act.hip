// BatchNorm2d + SiLU fused, forward and backward, for the real-valued (ANN) BaseConv blocks of the stem, PAFPN neck
// and head (yolox/models/network_blocks.py:52-53 with act = nn.SiLU).  Same decomposition as bn_lif.hip:
// blockIdx.y = channel, blockIdx.x = chunk of that channel's N*HW/4 float4 groups; statistics come from
// eas_bn_stats.  HBM traffic per element: fwd 8 B (read y, write out); bwd 20 B (two passes over y and grad_out
// because the BN backward needs sum(dz) and sum(dz*xhat) first, then writes grad_y); nothing but y is saved.
#include <stdlib.h>

#include "eas_common.h"

namespace {

constexpr int VEC = 4;
constexpr int kMaxChunks = 64;
constexpr int NW = EAS_BLOCK / EAS_WAVE;

static inline int pick_chunks(int64_t groups_per_channel, int C) {
    int64_t want = (groups_per_channel + EAS_BLOCK - 1) / EAS_BLOCK;
    int64_t cap = 8192 / (C > 0 ? C : 1);
    if (cap < 1) cap = 1;
    if (cap > kMaxChunks) cap = kMaxChunks;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

__device__ __forceinline__ float silu(float z) { return z * eas_sigmoidf(z); }
__device__ __forceinline__ float silu_grad(float z) {
    const float sg = eas_sigmoidf(z);
    return sg * (1.0f + z * (1.0f - sg));
}

// c / chunk / nchunks_g: the block's channel, its chunk of the channel's float4 groups and the chunks per channel (from blockIdx of a
// launch of one layer; a grouped launch -- bn_silu_fwd_group_kernel -- maps its flat grid onto (layer, channel, chunk) first)
__device__ __forceinline__ void bn_silu_fwd_body(const float* __restrict__ y, const float* __restrict__ mean,
                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ out, int N, int C, int HW, const BnFin& fin,
                                                 int out_ctot, int y_ctot, const int c, const int chunk, const int nchunks_g) {
    __shared__ float st[2];
    if (c >= C) return;
    const int hw4 = HW / VEC;
    const int64_t groups = (int64_t)N * hw4;
    float mu, istd;
    if (fin.part) {
        bn_finalize_in_block(fin, c, st, mu, istd, chunk == 0);
    } else {
        mu = mean[c];
        istd = invstd[c];
    }
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    GroupWalk gw((int64_t)chunk * blockDim.x + threadIdx.x, (int64_t)nchunks_g * blockDim.x, hw4);
    for (int64_t g = (int64_t)chunk * blockDim.x + threadIdx.x; g < groups; g += (int64_t)nchunks_g * blockDim.x, gw.next()) {
        const int64_t n = gw.n;
        const int q = gw.q;
        const int64_t base = (n * C + c) * (int64_t)HW + (int64_t)q * VEC;
        const float4 v = *reinterpret_cast<const float4*>(y + (y_ctot ? (n * y_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base));
        float4 o;
        o.x = silu(fmaf(v.x, scale, shift)); o.y = silu(fmaf(v.y, scale, shift));
        o.z = silu(fmaf(v.z, scale, shift)); o.w = silu(fmaf(v.w, scale, shift));
        *reinterpret_cast<float4*>(out + (out_ctot ? (n * out_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base)) = o;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void bn_silu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ out, int N,
                                                                int C, int HW, BnFin fin, int out_ctot, int y_ctot) {
    // EAS_CHAN_GRID: the chunks of a channel share an XCD
    bn_silu_fwd_body(y, mean, invstd, gamma, beta, out, N, C, HW, fin, out_ctot, y_ctot, (int)(blockIdx.z * 8 + blockIdx.x), (int)blockIdx.y,
                     (int)gridDim.y);
}

template <bool APPLY>
__device__ __forceinline__ void bn_silu_bwd_body(const float* __restrict__ grad_out, const float* __restrict__ y,
                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 int batch_stats, double* __restrict__ part, int nchunks,
                                                 float* __restrict__ grad_y, float* __restrict__ grad_gamma,
                                                 float* __restrict__ grad_beta, int N, int C, int HW, int go_ctot, int y_ctot, const int c,
                                                 const int chunk, const int nchunks_g) {
    __shared__ double red[NW];
    __shared__ float bc[2];
    if (c >= C) return;
    const float mu = mean[c], istd = invstd[c];
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    const int hw4 = HW / VEC;
    const int64_t groups = (int64_t)N * hw4;
    float m1 = 0.f, m2 = 0.f;
    if (APPLY) {
        if (threadIdx.x < EAS_WAVE) {
            double s1 = 0.0, s2 = 0.0;
            if ((int)threadIdx.x < nchunks) {
                s1 = part[((int64_t)c * kMaxChunks + threadIdx.x) * 2 + 0];
                s2 = part[((int64_t)c * kMaxChunks + threadIdx.x) * 2 + 1];
            }
            s1 = eas_wave_sum(s1);
            s2 = eas_wave_sum(s2);
            if (threadIdx.x == 0) {
                const double cnt = (double)N * HW;
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
    float s1 = 0.f, s2 = 0.f;
    GroupWalk gw((int64_t)chunk * blockDim.x + threadIdx.x, (int64_t)nchunks_g * blockDim.x, hw4);
    for (int64_t g = (int64_t)chunk * blockDim.x + threadIdx.x; g < groups; g += (int64_t)nchunks_g * blockDim.x, gw.next()) {
        const int64_t n = gw.n;
        const int q = gw.q;
        const int64_t base = (n * C + c) * (int64_t)HW + (int64_t)q * VEC;
        const int64_t ybase = y_ctot ? (n * y_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base;       // y and grad_y: the same slice
        const float4 yv = *reinterpret_cast<const float4*>(y + ybase);
        const float4 gv = *reinterpret_cast<const float4*>(grad_out + (go_ctot ? (n * go_ctot + c) * (int64_t)HW + (int64_t)q * VEC : base));
        const float ys[4] = {yv.x, yv.y, yv.z, yv.w};
        const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dz = gs[e] * silu_grad(fmaf(ys[e], scale, shift));
            const float xhat = (ys[e] - mu) * istd;
            if (APPLY) {
                o[e] = scale * (dz - m1 - xhat * m2);
            } else {
                s1 += dz;
                s2 += dz * xhat;
            }
        }
        if (APPLY) *reinterpret_cast<float4*>(grad_y + ybase) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (!APPLY) {
        const double t1 = eas_block_sum<double, NW>((double)s1, red);
        const double t2 = eas_block_sum<double, NW>((double)s2, red);
        if (threadIdx.x == 0) {
            part[((int64_t)c * kMaxChunks + chunk) * 2 + 0] = t1;
            part[((int64_t)c * kMaxChunks + chunk) * 2 + 1] = t2;
        }
    }
}

template <bool APPLY>
__global__ __launch_bounds__(EAS_BLOCK) void bn_silu_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int batch_stats, double* __restrict__ part, int nchunks,
                                                                float* __restrict__ grad_y, float* __restrict__ grad_gamma,
                                                                float* __restrict__ grad_beta, int N, int C, int HW, int go_ctot, int y_ctot) {
    // EAS_CHAN_GRID: the chunks of a channel share an XCD
    bn_silu_bwd_body<APPLY>(grad_out, y, mean, invstd, gamma, beta, batch_stats, part, nchunks, grad_y, grad_gamma, grad_beta, N, C, HW, go_ctot,
                            y_ctot, (int)(blockIdx.z * 8 + blockIdx.x), (int)blockIdx.y, (int)gridDim.y);
}

// ---- grouped launches (include/eas_hip.h eas_bn_silu_fwd_group / eas_bn_silu_bwd_group): layer p owns blocks [first[p], first[p + 1]) of
// the flat grid, 8 * chunks[p] * ceil(C / 8) of them in the order of EAS_CHAN_GRID (channel-in-octet fastest, then chunk, then octet)
constexpr int kMaxBnGroup = 12;
struct BnSiluFwdGroup {
    const float* y[kMaxBnGroup];
    const float* mean[kMaxBnGroup];
    const float* invstd[kMaxBnGroup];
    const float* gamma[kMaxBnGroup];
    const float* beta[kMaxBnGroup];
    float* out[kMaxBnGroup];
    BnFin fin[kMaxBnGroup];
    int N[kMaxBnGroup], C[kMaxBnGroup], HW[kMaxBnGroup], out_ctot[kMaxBnGroup], y_ctot[kMaxBnGroup], chunks[kMaxBnGroup];
    int first[kMaxBnGroup + 1];
    int n;
};

__global__ __launch_bounds__(EAS_BLOCK) void bn_silu_fwd_group_kernel(const BnSiluFwdGroup a) {
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int lb = (int)blockIdx.x - a.first[p];
    const int r = lb >> 3, chunk = r % a.chunks[p], c = (r / a.chunks[p]) * 8 + (lb & 7);
    bn_silu_fwd_body(a.y[p], a.mean[p], a.invstd[p], a.gamma[p], a.beta[p], a.out[p], a.N[p], a.C[p], a.HW[p], a.fin[p], a.out_ctot[p],
                     a.y_ctot[p], c, chunk, a.chunks[p]);
}

struct BnSiluBwdGroup {
    const float* grad_out[kMaxBnGroup];
    const float* y[kMaxBnGroup];
    const float* mean[kMaxBnGroup];
    const float* invstd[kMaxBnGroup];
    const float* gamma[kMaxBnGroup];
    const float* beta[kMaxBnGroup];
    double* part[kMaxBnGroup];
    float* grad_y[kMaxBnGroup];
    float* grad_gamma[kMaxBnGroup];
    float* grad_beta[kMaxBnGroup];
    int batch_stats[kMaxBnGroup], N[kMaxBnGroup], C[kMaxBnGroup], HW[kMaxBnGroup], go_ctot[kMaxBnGroup], y_ctot[kMaxBnGroup], chunks[kMaxBnGroup];
    int first[kMaxBnGroup + 1];
    int n;
};

template <bool APPLY>
__global__ __launch_bounds__(EAS_BLOCK) void bn_silu_bwd_group_kernel(const BnSiluBwdGroup a) {
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int lb = (int)blockIdx.x - a.first[p];
    const int r = lb >> 3, chunk = r % a.chunks[p], c = (r / a.chunks[p]) * 8 + (lb & 7);
    bn_silu_bwd_body<APPLY>(a.grad_out[p], a.y[p], a.mean[p], a.invstd[p], a.gamma[p], a.beta[p], a.batch_stats[p], a.part[p], a.chunks[p],
                            a.grad_y[p], a.grad_gamma[p], a.grad_beta[p], a.N[p], a.C[p], a.HW[p], a.go_ctot[p], a.y_ctot[p], c, chunk,
                            a.chunks[p]);
}

// Small maps (the 8x10 / 16x20 levels of neck and head: a channel's N*HW values fit the registers of ONE block): both passes in one
// launch without any exchange between blocks -- block = channel, every thread keeps its <= GPT float4 groups of grad_out and y in
// registers (all loads issued up front: one memory latency), the block adds (sum dz, sum dz*xhat) in a fixed order and writes grad_y
// from the registers: 12 B per element instead of 20 and one launch instead of two (these layers are launch-bound: ~6 us per launch
// for 2.6 MB).  Same per-element arithmetic as the two-pass kernels.
template <int NT, int GPT>
__global__ __launch_bounds__(NT) void bn_silu_bwd_small_kernel(const float* __restrict__ grad_out, const float* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int batch_stats, float* __restrict__ grad_y, float* __restrict__ grad_gamma,
                                                              float* __restrict__ grad_beta, int N, int C, int HW, int go_ctot, int y_ctot) {
    constexpr int NWV = NT / EAS_WAVE;
    __shared__ double red[2][NWV];
    __shared__ float bc[2];
    const int c = blockIdx.x;
    const float mu = mean[c], istd = invstd[c];
    const float scale = gamma[c] * istd;
    const float shift = beta[c] - mu * scale;
    const int hw4 = HW / VEC;
    const int groups = N * hw4;
    float4 yv[GPT], gv[GPT];
    int64_t base[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g = threadIdx.x + i * NT;
        const int gg = g < groups ? g : groups - 1;          // surplus threads re-read the last group (not used)
        const int n = gg / hw4, q = gg - n * hw4;
        base[i] = ((int64_t)n * (y_ctot ? y_ctot : C) + c) * (int64_t)HW + (int64_t)q * VEC;      // of y and grad_y
        yv[i] = *reinterpret_cast<const float4*>(y + base[i]);
        gv[i] = *reinterpret_cast<const float4*>(grad_out + ((int64_t)n * (go_ctot ? go_ctot : C) + c) * (int64_t)HW + (int64_t)q * VEC);
    }
    float s1 = 0.f, s2 = 0.f;
    float dz[GPT][4], xh[GPT][4];
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const float ys[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w};
        const float gs[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
        const bool live = threadIdx.x + i * NT < groups;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dz[i][e] = gs[e] * silu_grad(fmaf(ys[e], scale, shift));
            xh[i][e] = (ys[e] - mu) * istd;
            if (live) {
                s1 += dz[i][e];
                s2 += dz[i][e] * xh[i][e];
            }
        }
    }
    double t1 = eas_wave_sum((double)s1), t2 = eas_wave_sum((double)s2);
    if ((threadIdx.x & (EAS_WAVE - 1)) == 0) {
        red[0][threadIdx.x / EAS_WAVE] = t1;
        red[1][threadIdx.x / EAS_WAVE] = t2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t1 = 0.0; t2 = 0.0;
        for (int w = 0; w < NWV; ++w) { t1 += red[0][w]; t2 += red[1][w]; }
        const double cnt = (double)N * HW;
        bc[0] = (float)(t1 / cnt);
        bc[1] = (float)(t2 / cnt);
        grad_beta[c] = (float)t1;
        grad_gamma[c] = (float)t2;
    }
    __syncthreads();
    const float m1 = batch_stats ? bc[0] : 0.f, m2 = batch_stats ? bc[1] : 0.f;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        if (threadIdx.x + i * NT < groups) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = scale * (dz[i][e] - m1 - xh[i][e] * m2);
            *reinterpret_cast<float4*>(grad_y + base[i]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

}  // namespace

extern "C" {

int eas_bn_silu_fwd_ex(const float* y, float* mean, float* invstd, const float* gamma, const float* beta,
                       float* out, int N, int C, int HW, const EasBnPending* pend, int out_ctot, int y_ctot, eas_stream_t stream) {
    if (!y || !mean || !invstd || !gamma || !beta || !out || N < 1 || C < 1 || HW < 1 || (out_ctot != 0 && out_ctot < C) || (y_ctot != 0 && y_ctot < C))
        return EAS_ERR_INVALID_ARG;
    if (out_ctot == C) out_ctot = 0;
    if (y_ctot == C) y_ctot = 0;
    if (HW % VEC != 0 || C > 65535) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)y | (uintptr_t)out) & 15) return EAS_ERR_INVALID_ARG;
    BnFin fin{};
    if (pend && pend->partial) {
        if (pend->chunks < 1 || pend->chunks > (pend->pitch ? pend->pitch : kMaxChunks) || pend->pitch < 0 || !(pend->count >= 1.0) || pend->replicas < 1) return EAS_ERR_INVALID_ARG;
        if ((pend->running_mean == nullptr) != (pend->running_var == nullptr)) return EAS_ERR_INVALID_ARG;
        fin.part = pend->partial; fin.nchunks = pend->chunks; fin.pitch = pend->pitch ? pend->pitch : kMaxChunks; fin.replicas = pend->replicas; fin.count = pend->count;
        fin.eps = pend->eps; fin.momentum = pend->momentum; fin.mean_out = mean; fin.invstd_out = invstd;
        fin.rmean = pend->running_mean; fin.rvar = pend->running_var;
    }
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int chunks = pick_chunks((int64_t)N * (HW / VEC), C);
    EAS_LAUNCH(bn_silu_fwd_kernel, EAS_CHAN_GRID(chunks, C), dim3(EAS_BLOCK), 0, st, y, mean, invstd, gamma, beta, out, N, C, HW, fin,
                       out_ctot, y_ctot);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_bn_silu_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    float* out, int N, int C, int HW, eas_stream_t stream) {
    return eas_bn_silu_fwd_ex(y, const_cast<float*>(mean), const_cast<float*>(invstd), gamma, beta, out, N, C, HW, nullptr, 0, 0, stream);
}

int eas_bn_silu_bwd(const float* grad_out, const float* y, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                    double* workspace, int N, int C, int HW, int go_ctot, int y_ctot, eas_stream_t stream) {
    if (!grad_out || !y || !mean || !invstd || !gamma || !beta || !grad_y || !grad_gamma || !grad_beta || !workspace ||
        N < 1 || C < 1 || HW < 1 || (go_ctot != 0 && go_ctot < C) || (y_ctot != 0 && y_ctot < C))
        return EAS_ERR_INVALID_ARG;
    if (go_ctot == C) go_ctot = 0;
    if (y_ctot == C) y_ctot = 0;
    if (HW % VEC != 0 || C > 65535) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)y | (uintptr_t)grad_out | (uintptr_t)grad_y) & 15) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int chunks = pick_chunks((int64_t)N * (HW / VEC), C);
    {
        // one block per channel where a channel fits the registers of a block (see bn_silu_bwd_small_kernel); EAS_BNSILU_BWD=two keeps
        // the two-pass launches (development)
        static const bool small_ok = !(eas_dev_env("EAS_BNSILU_BWD") && eas_dev_env("EAS_BNSILU_BWD")[0] == 't');
        constexpr int GPT = 6;
        const int64_t groups = (int64_t)N * (HW / VEC);
        if (small_ok && C >= 64 && groups <= 1024 * GPT) {
#define EAS_SMALL(NT_) EAS_LAUNCH((bn_silu_bwd_small_kernel<NT_, GPT>), dim3(C), dim3(NT_), 0, st, grad_out, y, mean, invstd, gamma, beta, \
                                           batch_stats, grad_y, grad_gamma, grad_beta, N, C, HW, go_ctot, y_ctot)
            if (groups <= 256 * GPT) EAS_SMALL(256);
            else if (groups <= 512 * GPT) EAS_SMALL(512);
            else EAS_SMALL(1024);
#undef EAS_SMALL
            EAS_CHECK_LAUNCH();
            return EAS_OK;
        }
    }
    EAS_LAUNCH(bn_silu_bwd_kernel<false>, EAS_CHAN_GRID(chunks, C), dim3(EAS_BLOCK), 0, st, grad_out, y, mean, invstd, gamma, beta,
                       batch_stats, workspace, chunks, grad_y, grad_gamma, grad_beta, N, C, HW, go_ctot, y_ctot);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(bn_silu_bwd_kernel<true>, EAS_CHAN_GRID(chunks, C), dim3(EAS_BLOCK), 0, st, grad_out, y, mean, invstd, gamma, beta,
                       batch_stats, workspace, chunks, grad_y, grad_gamma, grad_beta, N, C, HW, go_ctot, y_ctot);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_bn_silu_fwd_group(const EasBnSiluFwdProblem* pr, int n, eas_stream_t stream) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (n > kMaxBnGroup) return EAS_ERR_UNSUPPORTED;
    BnSiluFwdGroup a{};
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        const EasBnSiluFwdProblem& q = pr[p];
        int out_ctot = q.out_ctot, y_ctot = q.y_ctot;
        if (!q.y || !q.mean || !q.invstd || !q.gamma || !q.beta || !q.out || q.N < 1 || q.C < 1 || q.HW < 1 || (out_ctot != 0 && out_ctot < q.C) ||
            (y_ctot != 0 && y_ctot < q.C))
            return EAS_ERR_INVALID_ARG;
        if (out_ctot == q.C) out_ctot = 0;
        if (y_ctot == q.C) y_ctot = 0;
        if (q.HW % VEC != 0 || q.C > 65535) return EAS_ERR_UNSUPPORTED;
        if (((uintptr_t)q.y | (uintptr_t)q.out) & 15) return EAS_ERR_INVALID_ARG;
        BnFin fin{};
        const EasBnPending* pend = &q.pending;
        if (pend->partial) {
            if (pend->chunks < 1 || pend->chunks > (pend->pitch ? pend->pitch : kMaxChunks) || pend->pitch < 0 || !(pend->count >= 1.0) || pend->replicas < 1) return EAS_ERR_INVALID_ARG;
            if ((pend->running_mean == nullptr) != (pend->running_var == nullptr)) return EAS_ERR_INVALID_ARG;
            fin.part = pend->partial; fin.nchunks = pend->chunks; fin.pitch = pend->pitch ? pend->pitch : kMaxChunks; fin.replicas = pend->replicas; fin.count = pend->count;
            fin.eps = pend->eps; fin.momentum = pend->momentum; fin.mean_out = q.mean; fin.invstd_out = q.invstd;
            fin.rmean = pend->running_mean; fin.rvar = pend->running_var;
        }
        a.y[p] = q.y; a.mean[p] = q.mean; a.invstd[p] = q.invstd; a.gamma[p] = q.gamma; a.beta[p] = q.beta; a.out[p] = q.out;
        a.fin[p] = fin;
        a.N[p] = q.N; a.C[p] = q.C; a.HW[p] = q.HW; a.out_ctot[p] = out_ctot; a.y_ctot[p] = y_ctot;
        a.chunks[p] = pick_chunks((int64_t)q.N * (q.HW / VEC), q.C);
        a.first[p] = blocks;
        blocks += 8 * a.chunks[p] * ((q.C + 7) / 8);
    }
    a.first[n] = blocks;
    a.n = n;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(bn_silu_fwd_group_kernel, dim3(blocks), dim3(EAS_BLOCK), 0, eas_s(stream), a);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_bn_silu_bwd_group(const EasBnSiluBwdProblem* pr, int n, eas_stream_t stream) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (n > kMaxBnGroup) return EAS_ERR_UNSUPPORTED;
    BnSiluBwdGroup a{};
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        const EasBnSiluBwdProblem& q = pr[p];
        int go_ctot = q.grad_out_ctot, y_ctot = q.y_ctot;
        if (!q.grad_out || !q.y || !q.mean || !q.invstd || !q.gamma || !q.beta || !q.grad_y || !q.grad_gamma || !q.grad_beta || !q.workspace ||
            q.N < 1 || q.C < 1 || q.HW < 1 || (go_ctot != 0 && go_ctot < q.C) || (y_ctot != 0 && y_ctot < q.C))
            return EAS_ERR_INVALID_ARG;
        if (go_ctot == q.C) go_ctot = 0;
        if (y_ctot == q.C) y_ctot = 0;
        if (q.HW % VEC != 0 || q.C > 65535) return EAS_ERR_UNSUPPORTED;
        if (((uintptr_t)q.y | (uintptr_t)q.grad_out | (uintptr_t)q.grad_y) & 15) return EAS_ERR_INVALID_ARG;
        a.grad_out[p] = q.grad_out; a.y[p] = q.y; a.mean[p] = q.mean; a.invstd[p] = q.invstd; a.gamma[p] = q.gamma; a.beta[p] = q.beta;
        a.part[p] = q.workspace; a.grad_y[p] = q.grad_y; a.grad_gamma[p] = q.grad_gamma; a.grad_beta[p] = q.grad_beta;
        a.batch_stats[p] = q.batch_stats; a.N[p] = q.N; a.C[p] = q.C; a.HW[p] = q.HW; a.go_ctot[p] = go_ctot; a.y_ctot[p] = y_ctot;
        a.chunks[p] = pick_chunks((int64_t)q.N * (q.HW / VEC), q.C);
        a.first[p] = blocks;
        blocks += 8 * a.chunks[p] * ((q.C + 7) / 8);
    }
    a.first[n] = blocks;
    a.n = n;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(bn_silu_bwd_group_kernel<false>, dim3(blocks), dim3(EAS_BLOCK), 0, eas_s(stream), a);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(bn_silu_bwd_group_kernel<true>, dim3(blocks), dim3(EAS_BLOCK), 0, eas_s(stream), a);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
