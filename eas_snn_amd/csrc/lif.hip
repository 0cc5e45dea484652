// K2: multi-step (P)LIF integrate-fire-reset, forward and backward, and the time-mean readout.
// HBM-bound elementwise kernels.  One thread owns 4 consecutive neurons (16-B loads/stores) and
// walks the T time steps with the membrane potential in registers, so each [T][M] tensor is
// touched exactly once: fwd reads x (4 B) and writes s (4 B) [+ h (4 B) when training];
// bwd reads grad_s, h (8 B) and writes grad_x (4 B) per neuron-step.
#include "eas_common.h"

namespace {

constexpr int VEC = 4;

template <int T_, bool HARD, bool DI, bool STRICT>
__global__ __launch_bounds__(EAS_BLOCK) void lif_fwd_kernel(const float* __restrict__ x, const float* v_in, float* v_out,
                                                            EasLifParams p, float* __restrict__ spikes,
                                                            float* __restrict__ h_save, float* __restrict__ mean_out,
                                                            int T_rt, int64_t M) {
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const int T = T_ > 0 ? T_ : T_rt;
    const int64_t nvec = (M % VEC == 0 || T == 1) ? M / VEC : 0;   // rows of [T][M] must be 16-B aligned for float4
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const float vr0 = HARD ? p.v_reset : 0.0f;
        float4 v = v_in ? reinterpret_cast<const float4*>(v_in)[i] : make_float4(vr0, vr0, vr0, vr0);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (T_ > 0) {
            float4 xs[T_ > 0 ? T_ : 1];
#pragma unroll
            for (int t = 0; t < T_; ++t) xs[t] = reinterpret_cast<const float4*>(x + (int64_t)t * M)[i];
#pragma unroll
            for (int t = 0; t < T_; ++t) {
                float4 h, s;
                eas_lif_step<HARD, DI, STRICT>(v.x, xs[t].x, k, omk, p.v_th, p.v_reset, h.x, s.x);
                eas_lif_step<HARD, DI, STRICT>(v.y, xs[t].y, k, omk, p.v_th, p.v_reset, h.y, s.y);
                eas_lif_step<HARD, DI, STRICT>(v.z, xs[t].z, k, omk, p.v_th, p.v_reset, h.z, s.z);
                eas_lif_step<HARD, DI, STRICT>(v.w, xs[t].w, k, omk, p.v_th, p.v_reset, h.w, s.w);
                reinterpret_cast<float4*>(spikes + (int64_t)t * M)[i] = s;
                if (h_save) reinterpret_cast<float4*>(h_save + (int64_t)t * M)[i] = h;
                acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
            }
        } else {
            for (int t = 0; t < T; ++t) {
                const float4 xt = reinterpret_cast<const float4*>(x + (int64_t)t * M)[i];
                float4 h, s;
                eas_lif_step<HARD, DI, STRICT>(v.x, xt.x, k, omk, p.v_th, p.v_reset, h.x, s.x);
                eas_lif_step<HARD, DI, STRICT>(v.y, xt.y, k, omk, p.v_th, p.v_reset, h.y, s.y);
                eas_lif_step<HARD, DI, STRICT>(v.z, xt.z, k, omk, p.v_th, p.v_reset, h.z, s.z);
                eas_lif_step<HARD, DI, STRICT>(v.w, xt.w, k, omk, p.v_th, p.v_reset, h.w, s.w);
                reinterpret_cast<float4*>(spikes + (int64_t)t * M)[i] = s;
                if (h_save) reinterpret_cast<float4*>(h_save + (int64_t)t * M)[i] = h;
                acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
            }
        }
        if (v_out) reinterpret_cast<float4*>(v_out)[i] = v;
        if (mean_out) {
            const float Tf = (float)T;   // sum / T, like ATen's mean
            reinterpret_cast<float4*>(mean_out)[i] = make_float4(acc.x / Tf, acc.y / Tf, acc.z / Tf, acc.w / Tf);
        }
    }
    // scalar remainder (M % 4 elements, or everything when rows are unaligned)
    {
        for (int64_t j = nvec * VEC + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
            float v = v_in ? v_in[j] : (HARD ? p.v_reset : 0.0f), acc = 0.f;
            for (int t = 0; t < T; ++t) {
                float h, s;
                eas_lif_step<HARD, DI, STRICT>(v, x[(int64_t)t * M + j], k, omk, p.v_th, p.v_reset, h, s);
                spikes[(int64_t)t * M + j] = s;
                if (h_save) h_save[(int64_t)t * M + j] = h;
                acc += s;
            }
            if (v_out) v_out[j] = v;
            if (mean_out) mean_out[j] = acc / (float)T;
        }
    }
}

// v_{t-1} recomputed from h_{t-1} (spike = H(h - v_th)), so only h is stored.
template <bool HARD, bool STRICT>
__device__ __forceinline__ float v_after(float h, float v_th, float v_reset) {
    const float u = h - v_th;
    const float s = STRICT ? (u > 0.0f ? 1.0f : 0.0f) : (u >= 0.0f ? 1.0f : 0.0f);
    return HARD ? (1.0f - s) * h + s * v_reset : h - s * v_th;
}

template <int T_, bool HARD, bool DI, bool STRICT>
__global__ __launch_bounds__(EAS_BLOCK) void lif_bwd_kernel(const float* __restrict__ grad_s,
                                                            const float* __restrict__ grad_mean,
                                                            const float* __restrict__ h_save,
                                                            const float* __restrict__ v_init,
                                                            const float* __restrict__ x, EasLifParams p, int sg_id,
                                                            float alpha, const float* __restrict__ alpha_dev,
                                                            float* __restrict__ grad_x, float* __restrict__ partial,
                                                            float* __restrict__ partial_a, int T_rt, int64_t M) {
    __shared__ float red[EAS_BLOCK / EAS_WAVE];
    if (alpha_dev) alpha = fabsf(*alpha_dev);    // learnable slope (EAS_SG_PATAN)
    float da = 0.f;
    const float k = eas_lif_k(p);
    const float omk = 1.0f - k;
    const bool detach = (p.flags & EAS_LIF_DETACH_RESET) != 0;
    const int T = T_ > 0 ? T_ : T_rt;
    const float invT = 1.0f / (float)T;
    const int64_t nvec = (M % VEC == 0 || T == 1) ? M / VEC : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float dk = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float gv[VEC] = {0.f, 0.f, 0.f, 0.f};
        float gm[VEC] = {0.f, 0.f, 0.f, 0.f};
        if (grad_mean) {
            const float4 g = reinterpret_cast<const float4*>(grad_mean)[i];
            gm[0] = g.x * invT; gm[1] = g.y * invT; gm[2] = g.z * invT; gm[3] = g.w * invT;
        }
        float4 hn = reinterpret_cast<const float4*>(h_save + (int64_t)(T - 1) * M)[i];
        for (int t = T - 1; t >= 0; --t) {
            const float4 hc = hn;
            float vp[VEC];
            if (t > 0) {
                hn = reinterpret_cast<const float4*>(h_save + (int64_t)(t - 1) * M)[i];
                vp[0] = v_after<HARD, STRICT>(hn.x, p.v_th, p.v_reset);
                vp[1] = v_after<HARD, STRICT>(hn.y, p.v_th, p.v_reset);
                vp[2] = v_after<HARD, STRICT>(hn.z, p.v_th, p.v_reset);
                vp[3] = v_after<HARD, STRICT>(hn.w, p.v_th, p.v_reset);
            } else if (v_init) {
                const float4 v0 = reinterpret_cast<const float4*>(v_init)[i];
                vp[0] = v0.x; vp[1] = v0.y; vp[2] = v0.z; vp[3] = v0.w;
            } else {
                vp[0] = vp[1] = vp[2] = vp[3] = HARD ? p.v_reset : 0.f;
            }
            float gs[VEC] = {gm[0], gm[1], gm[2], gm[3]};
            if (grad_s) {
                const float4 g = reinterpret_cast<const float4*>(grad_s + (int64_t)t * M)[i];
                gs[0] += g.x; gs[1] += g.y; gs[2] += g.z; gs[3] += g.w;
            }
            float xv[VEC] = {0.f, 0.f, 0.f, 0.f};
            if (DI && x) {
                const float4 xt = reinterpret_cast<const float4*>(x + (int64_t)t * M)[i];
                xv[0] = xt.x; xv[1] = xt.y; xv[2] = xt.z; xv[3] = xt.w;
            }
            const float hh[VEC] = {hc.x, hc.y, hc.z, hc.w};
            float gx[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float dkt;
                eas_lif_step_bwd<HARD, DI, STRICT>(gs[e], gv[e], hh[e], vp[e], xv[e], k, omk, p.v_th, p.v_reset,
                                                   detach, sg_id, alpha, dkt, gx[e], da);
                dk += dkt;
            }
            reinterpret_cast<float4*>(grad_x + (int64_t)t * M)[i] = make_float4(gx[0], gx[1], gx[2], gx[3]);
        }
    }
    {
        for (int64_t j = nvec * VEC + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
            float gv = 0.f;
            const float gm = grad_mean ? grad_mean[j] * invT : 0.f;
            for (int t = T - 1; t >= 0; --t) {
                const float h = h_save[(int64_t)t * M + j];
                const float vp = t > 0 ? v_after<HARD, STRICT>(h_save[(int64_t)(t - 1) * M + j], p.v_th, p.v_reset)
                                       : (v_init ? v_init[j] : (HARD ? p.v_reset : 0.f));
                const float gs = gm + (grad_s ? grad_s[(int64_t)t * M + j] : 0.f);
                const float xv = (DI && x) ? x[(int64_t)t * M + j] : 0.f;
                float dkt, gx;
                eas_lif_step_bwd<HARD, DI, STRICT>(gs, gv, h, vp, xv, k, omk, p.v_th, p.v_reset, detach, sg_id,
                                                   alpha, dkt, gx, da);
                dk += dkt;
                grad_x[(int64_t)t * M + j] = gx;
            }
        }
    }
    if (partial) {
        const float tot = eas_block_sum<float, EAS_BLOCK / EAS_WAVE>(dk, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    }
    if (partial_a) {
        const float tot = eas_block_sum<float, EAS_BLOCK / EAS_WAVE>(da, red);
        if (threadIdx.x == 0) partial_a[blockIdx.x] = tot;
    }
}

// final stage of the grad_w reduction: sum the per-block partials in a fixed order (deterministic)
// and apply dk/dw = k (1 - k).
// and, for the learnable surrogate slope, d|alpha|/dalpha = sign(alpha) (0 at 0, like torch.abs).
__global__ __launch_bounds__(EAS_BLOCK) void lif_gradw_finalize(const float* __restrict__ partial, int n,
                                                                const float* __restrict__ w_logit,
                                                                float* __restrict__ grad_w,
                                                                const float* __restrict__ partial_a,
                                                                const float* __restrict__ alpha_dev,
                                                                float* __restrict__ grad_alpha) {
    __shared__ double red[EAS_BLOCK / EAS_WAVE];
    if (grad_w) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)partial[i];
        const double tot = eas_block_sum<double, EAS_BLOCK / EAS_WAVE>(acc, red);
        if (threadIdx.x == 0) {
            const float k = eas_sigmoidf(*w_logit);
            *grad_w = (float)tot * (k * (1.0f - k));
        }
    }
    if (grad_alpha) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)partial_a[i];
        const double tot = eas_block_sum<double, EAS_BLOCK / EAS_WAVE>(acc, red);
        if (threadIdx.x == 0) {
            const float a = *alpha_dev;
            *grad_alpha = (float)tot * (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f));
        }
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void time_mean_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int T, int64_t M) {
    const int64_t nvec = (M % VEC == 0 || T == 1) ? M / VEC : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float Tf = (float)T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < T; ++t) {
            const float4 v = reinterpret_cast<const float4*>(x + (int64_t)t * M)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x / Tf, a.y / Tf, a.z / Tf, a.w / Tf);
    }
    {
        for (int64_t j = nvec * VEC + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += stride) {
            float a = 0.f;
            for (int t = 0; t < T; ++t) a += x[(int64_t)t * M + j];
            out[j] = a / Tf;
        }
    }
}

constexpr int kReduceBlocks = 2048;

template <bool HARD, bool DI, bool STRICT>
int launch_fwd(const float* x, const float* v_in, float* v_out, EasLifParams p, float* spikes, float* h_save, float* mean_out, int T,
               int64_t M, hipStream_t st) {
    const int grid = eas_grid_1d(M % VEC == 0 ? M / VEC : M);
#define EAS_CASE(TT)                                                                                              \
    case TT:                                                                                                      \
        EAS_LAUNCH((lif_fwd_kernel<TT, HARD, DI, STRICT>), dim3(grid), dim3(EAS_BLOCK), 0, st, x, v_in, v_out, \
                           p, spikes, h_save, mean_out, T, M);                                                       \
        break;
    switch (T) {
        EAS_CASE(1) EAS_CASE(2) EAS_CASE(3) EAS_CASE(4) EAS_CASE(5) EAS_CASE(6) EAS_CASE(7) EAS_CASE(8)
        default:
            EAS_LAUNCH((lif_fwd_kernel<0, HARD, DI, STRICT>), dim3(grid), dim3(EAS_BLOCK), 0, st, x, v_in, v_out, p,
                               spikes, h_save, mean_out, T, M);
    }
#undef EAS_CASE
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

template <bool HARD, bool DI, bool STRICT>
int launch_bwd(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init, const float* x,
               EasLifParams p, int sg, float alpha, const float* alpha_dev, float* grad_alpha, float* grad_x, float* grad_w,
               float* workspace, int T, int64_t M, hipStream_t st) {
    int grid = eas_grid_1d(M % VEC == 0 ? M / VEC : M);
    if (grid > kReduceBlocks) grid = kReduceBlocks;
    float* partial = grad_w ? workspace : nullptr;
    float* partial_a = grad_alpha ? workspace + kReduceBlocks : nullptr;
    EAS_LAUNCH((lif_bwd_kernel<0, HARD, DI, STRICT>), dim3(grid), dim3(EAS_BLOCK), 0, st, grad_s, grad_mean,
                       h_save, v_init, x, p, sg, alpha, alpha_dev, grad_x, partial, partial_a, T, M);
    EAS_CHECK_LAUNCH();
    if (grad_w || grad_alpha) {
        EAS_LAUNCH(lif_gradw_finalize, dim3(1), dim3(EAS_BLOCK), 0, st, partial, grid, p.w_logit, grad_w, partial_a, alpha_dev,
                           grad_alpha);
        EAS_CHECK_LAUNCH();
    }
    return EAS_OK;
}

}  // namespace

extern "C" {

int64_t eas_reduce_workspace_floats(int64_t) { return 2 * kReduceBlocks; }   // grad_w partials | grad_alpha partials

int eas_lif_fwd(const float* x, const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th,
                float v_reset, int flags, float* spikes, float* h_save, float* mean_out, int T, int64_t M, eas_stream_t stream) {
    if (!x || !spikes || T < 1 || M < 0) return EAS_ERR_INVALID_ARG;
    if (M == 0) return EAS_OK;
    if (((uintptr_t)x | (uintptr_t)v_in | (uintptr_t)v_out | (uintptr_t)spikes | (uintptr_t)h_save | (uintptr_t)mean_out) & 15)
        return EAS_ERR_INVALID_ARG;
    EasLifParams p{w_logit, k_const, v_th, v_reset, flags};
    const bool hard = flags & EAS_LIF_HARD_RESET, di = flags & EAS_LIF_DECAY_INPUT, strict = flags & EAS_LIF_FIRE_STRICT;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
#define EAS_DISPATCH(H, D, S) return launch_fwd<H, D, S>(x, v_in, v_out, p, spikes, h_save, mean_out, T, M, st)
    if (!hard && !di && !strict) EAS_DISPATCH(false, false, false);
    if (!hard && !di && strict) EAS_DISPATCH(false, false, true);
    if (!hard && di && !strict) EAS_DISPATCH(false, true, false);
    if (!hard && di && strict) EAS_DISPATCH(false, true, true);
    if (hard && !di && !strict) EAS_DISPATCH(true, false, false);
    if (hard && !di && strict) EAS_DISPATCH(true, false, true);
    if (hard && di && !strict) EAS_DISPATCH(true, true, false);
    EAS_DISPATCH(true, true, true);
#undef EAS_DISPATCH
}

static int lif_bwd_impl(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init,
                        const float* x, const float* w_logit, float k_const, float v_th, float v_reset, int flags,
                        int surrogate, float alpha, const float* alpha_dev, float* grad_alpha, float* grad_x, float* grad_w,
                        float* workspace, int T, int64_t M, eas_stream_t stream) {
    if (!h_save || !grad_x || T < 1 || M < 0 || (!grad_s && !grad_mean)) return EAS_ERR_INVALID_ARG;
    if (surrogate < EAS_SG_ATAN || surrogate > EAS_SG_PATAN) return EAS_ERR_INVALID_ARG;
    if (surrogate == EAS_SG_PATAN ? !alpha_dev : !(alpha > 0.f)) return EAS_ERR_INVALID_ARG;
    if ((grad_w && (!workspace || !w_logit)) || (grad_alpha && (!workspace || !alpha_dev))) return EAS_ERR_INVALID_ARG;
    if (M == 0) return EAS_OK;
    if (((uintptr_t)grad_s | (uintptr_t)grad_mean | (uintptr_t)h_save | (uintptr_t)v_init | (uintptr_t)x |
         (uintptr_t)grad_x) & 15)
        return EAS_ERR_INVALID_ARG;
    const bool hard = flags & EAS_LIF_HARD_RESET, di = flags & EAS_LIF_DECAY_INPUT, strict = flags & EAS_LIF_FIRE_STRICT;
    if (di && w_logit && grad_w && !x) return EAS_ERR_INVALID_ARG;
    EasLifParams p{w_logit, k_const, v_th, v_reset, flags};
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
#define EAS_DISPATCH(H, D, S) \
    return launch_bwd<H, D, S>(grad_s, grad_mean, h_save, v_init, x, p, surrogate, alpha, alpha_dev, grad_alpha, grad_x, grad_w, workspace, T, M, st)
    if (!hard && !di && !strict) EAS_DISPATCH(false, false, false);
    if (!hard && !di && strict) EAS_DISPATCH(false, false, true);
    if (!hard && di && !strict) EAS_DISPATCH(false, true, false);
    if (!hard && di && strict) EAS_DISPATCH(false, true, true);
    if (hard && !di && !strict) EAS_DISPATCH(true, false, false);
    if (hard && !di && strict) EAS_DISPATCH(true, false, true);
    if (hard && di && !strict) EAS_DISPATCH(true, true, false);
    EAS_DISPATCH(true, true, true);
#undef EAS_DISPATCH
}

int eas_lif_bwd(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init,
                const float* x, const float* w_logit, float k_const, float v_th, float v_reset, int flags,
                int surrogate, float alpha, float* grad_x, float* grad_w, float* workspace, int T, int64_t M,
                eas_stream_t stream) {
    if (surrogate == EAS_SG_PATAN) return EAS_ERR_INVALID_ARG;      // learnable slope: eas_lif_bwd_patan
    return lif_bwd_impl(grad_s, grad_mean, h_save, v_init, x, w_logit, k_const, v_th, v_reset, flags, surrogate, alpha, nullptr, nullptr,
                        grad_x, grad_w, workspace, T, M, stream);
}

int eas_lif_bwd_patan(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init,
                      const float* x, const float* w_logit, float k_const, float v_th, float v_reset, int flags,
                      const float* alpha, float* grad_alpha, float* grad_x, float* grad_w, float* workspace, int T, int64_t M,
                      eas_stream_t stream) {
    return lif_bwd_impl(grad_s, grad_mean, h_save, v_init, x, w_logit, k_const, v_th, v_reset, flags, EAS_SG_PATAN, 0.f, alpha, grad_alpha,
                        grad_x, grad_w, workspace, T, M, stream);
}

int eas_time_mean(const float* x, float* out, int T, int64_t M, eas_stream_t stream) {
    if (!x || !out || T < 1 || M < 0) return EAS_ERR_INVALID_ARG;
    if (M == 0) return EAS_OK;
    if (((uintptr_t)x | (uintptr_t)out) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(time_mean_kernel, dim3(eas_grid_1d(M % VEC == 0 ? M / VEC : M)), dim3(EAS_BLOCK), 0, eas_s(stream), x, out, T, M);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
