// Layout glue of the PAFPN neck: out = cat[upsample_nearest(a, x up), b] along channels in one kernel, and its backward
// (grad_a = sum over each up x up cell, grad_b = contiguous copy) in one kernel.  Replaces nn.Upsample + torch.cat
// (yolox/models/yolo_pafpn.py:101-113: `torch.cat([self.upsample(fpn_out0), x1], 1)`) and, with up = 1, the plain
// two-tensor concatenations of the bottom-up path (:115-121); backward replaces upsample_nearest2d_backward + two
// slice copies.  HBM-bound: every element read / written once, 16-byte accesses along W.
#include "eas_common.h"

namespace {

struct UpcatGeom { long long M; int Ca, Cb, H, W, up; };   // a [M][Ca][H][W]; b [M][Cb][H*up][W*up]; out [M][Ca+Cb][H*up][W*up]

__global__ __launch_bounds__(EAS_BLOCK) void upcat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                              UpcatGeom g) {
    const int Ho = g.H * g.up, Wo = g.W * g.up, C = g.Ca + g.Cb, w4 = Wo / 4;
    const long long total = g.M * C * Ho * w4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % w4);
        long long r = i / w4;
        const int y = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C);
        const long long m = r / C;
        float4 v;
        if (c < g.Ca) {
            const float* src = a + ((m * g.Ca + c) * g.H + y / g.up) * (long long)g.W;
            if (g.up == 2) {
                const float2 s = *reinterpret_cast<const float2*>(src + 2 * x4);
                v = make_float4(s.x, s.x, s.y, s.y);
            } else {
                v = *reinterpret_cast<const float4*>(src + 4 * x4);
            }
        } else {
            v = *reinterpret_cast<const float4*>(b + ((m * g.Cb + (c - g.Ca)) * Ho + y) * (long long)Wo + 4 * x4);
        }
        *reinterpret_cast<float4*>(out + ((m * C + c) * Ho + y) * (long long)Wo + 4 * x4) = v;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void upcat_bwd_kernel(const float* __restrict__ go, float* __restrict__ ga, float* __restrict__ gb,
                                                              UpcatGeom g) {
    const int Ho = g.H * g.up, Wo = g.W * g.up, C = g.Ca + g.Cb;
    const int wa2 = g.W / 2, wb4 = Wo / 4;
    const long long na = g.M * g.Ca * g.H * wa2, nb = g.M * g.Cb * Ho * wb4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (long long)gridDim.x * blockDim.x) {
        if (i < na) {                                   // two grad_a elements: a 2 x 4 (up = 2) or 1 x 2 (up = 1) patch of grad_out
            const int x2 = (int)(i % wa2);
            long long r = i / wa2;
            const int h = (int)(r % g.H);
            r /= g.H;
            const int c = (int)(r % g.Ca);
            const long long m = r / g.Ca;
            const float* src = go + ((m * C + c) * Ho + h * g.up) * (long long)Wo;
            float2 o;
            if (g.up == 2) {
                const float4 t = *reinterpret_cast<const float4*>(src + 4 * x2), u = *reinterpret_cast<const float4*>(src + Wo + 4 * x2);
                o.x = ((t.x + t.y) + u.x) + u.y;        // row-major accumulation like upsample_nearest2d_backward
                o.y = ((t.z + t.w) + u.z) + u.w;
            } else {
                o = *reinterpret_cast<const float2*>(src + 2 * x2);
            }
            *reinterpret_cast<float2*>(ga + ((m * g.Ca + c) * g.H + h) * (long long)g.W + 2 * x2) = o;
        } else {
            const long long k = i - na;
            const int x4 = (int)(k % wb4);
            long long r = k / wb4;
            const int y = (int)(r % Ho);
            r /= Ho;
            const int c = (int)(r % g.Cb);
            const long long m = r / g.Cb;
            *reinterpret_cast<float4*>(gb + ((m * g.Cb + c) * Ho + y) * (long long)Wo + 4 * x4) =
                *reinterpret_cast<const float4*>(go + ((m * C + g.Ca + c) * Ho + y) * (long long)Wo + 4 * x4);
        }
    }
}

// Focus (space to depth, yolox/models/network_blocks.py:198-213): out[m][k*C + c][h][w] = x[m][c][2h + dy_k][2w + dx_k] with
// (dy, dx) = (0,0), (1,0), (0,1), (1,1) for k = 0..3 -- four strided slices + torch.cat in the reference; a permutation, so
// the backward is the inverse permutation.  One thread handles a 2 x 4 input patch (two float4 loads, four float2 stores).
__global__ __launch_bounds__(EAS_BLOCK) void focus_kernel(const float* __restrict__ src, float* __restrict__ dst, long long M, int C, int Ho,
                                                          int Wo, int inverse) {
    const int w2 = Wo / 2;
    const long long total = M * C * Ho * w2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x2 = (int)(i % w2);
        long long r = i / w2;
        const int h = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C);
        const long long m = r / C;
        // full-resolution patch rows 2h, 2h+1, columns 4*x2 .. 4*x2+3 ; packed outputs (k, h, 2*x2 .. 2*x2+1)
        const long long full = ((m * C + c) * (2LL * Ho) + 2 * h) * (2LL * Wo) + 4 * x2;
        const long long packed = ((m * 4 * C + c) * Ho + h) * (long long)Wo + 2 * x2;
        const long long kstride = (long long)C * Ho * Wo;
        if (!inverse) {
            const float4 t = *reinterpret_cast<const float4*>(src + full), u = *reinterpret_cast<const float4*>(src + full + 2 * Wo);
            *reinterpret_cast<float2*>(dst + packed) = make_float2(t.x, t.z);                  // k = 0: (0, 0)
            *reinterpret_cast<float2*>(dst + packed + kstride) = make_float2(u.x, u.z);        // k = 1: (1, 0)
            *reinterpret_cast<float2*>(dst + packed + 2 * kstride) = make_float2(t.y, t.w);    // k = 2: (0, 1)
            *reinterpret_cast<float2*>(dst + packed + 3 * kstride) = make_float2(u.y, u.w);    // k = 3: (1, 1)
        } else {
            const float2 k0 = *reinterpret_cast<const float2*>(src + packed), k1 = *reinterpret_cast<const float2*>(src + packed + kstride);
            const float2 k2 = *reinterpret_cast<const float2*>(src + packed + 2 * kstride), k3 = *reinterpret_cast<const float2*>(src + packed + 3 * kstride);
            *reinterpret_cast<float4*>(dst + full) = make_float4(k0.x, k2.x, k0.y, k2.y);
            *reinterpret_cast<float4*>(dst + full + 2 * Wo) = make_float4(k1.x, k3.x, k1.y, k3.y);
        }
    }
}

int check_geom(const UpcatGeom& g) {
    if (g.M < 1 || g.Ca < 1 || g.Cb < 1 || g.H < 1 || g.W < 1) return EAS_ERR_INVALID_ARG;
    if ((g.up != 1 && g.up != 2) || (g.W * g.up) % 4 != 0 || g.W % 2 != 0) return EAS_ERR_UNSUPPORTED;
    return EAS_OK;
}


// Bias gradient of a convolution: out[c] = sum over (n, pixel) of g[n][c][pixel] (ATen: grad_y.sum((0, 2, 3)), a reduce kernel that takes
// 12-27 us for the 1 / 4 / num_classes-channel prediction maps of the head).  One block per channel, fixed order: every thread sums its
// strided share in double, the waves and the block add in a fixed tree -- deterministic.
__device__ __forceinline__ void channel_sum_body(const float* __restrict__ g, float* __restrict__ out, int N, int C, int HW, const int c) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t total = (int64_t)N * HW;
    if ((HW & 3) == 0) {
        const int hw4 = HW / 4;
        for (int64_t i = threadIdx.x; i < total / 4; i += blockDim.x) {
            const int64_t n = i / hw4, q = i - n * hw4;
            const float4 v = *reinterpret_cast<const float4*>(g + (n * C + c) * (int64_t)HW + 4 * q);
            s += (double)((v.x + v.y) + (v.z + v.w));
        }
    } else {
        for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
            const int64_t n = i / HW, p = i - n * HW;
            s += (double)g[(n * C + c) * (int64_t)HW + p];
        }
    }
    const double t = eas_block_sum<double, 4>(s, red);
    if (threadIdx.x == 0) out[c] = (float)t;
}

__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int C, int HW) {
    channel_sum_body(g, out, N, C, HW, (int)blockIdx.x);
}

// several tensors in one launch (eas_channel_sum_group): tensor p owns blocks [first[p], first[p] + C[p])
constexpr int kMaxSumGroup = 16;
struct ChannelSumGroup {
    const float* g[kMaxSumGroup];
    float* out[kMaxSumGroup];
    int N[kMaxSumGroup], C[kMaxSumGroup], HW[kMaxSumGroup], first[kMaxSumGroup + 1];
    int n;
};
__global__ __launch_bounds__(256) void channel_sum_group_kernel(const ChannelSumGroup a) {
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    channel_sum_body(a.g[p], a.out[p], a.N[p], a.C[p], a.HW[p], (int)blockIdx.x - a.first[p]);
}

// Input gradient of the head's 1x1 prediction convolutions (eas_pred_dgrad_group): gx[n][c][p] = sum_k wa[k][c] * ga[n][k][p] (+ the same for a
// second reader of the input: obj_preds next to reg_preds, yolo_head.py:161-163 of the reference) with 1 / 4 / num_classes reduction channels.
// As matrix-core "convolutions" with K padded to 16 these took two launches and 100 us per step for 110 MB of stores; here every thread
// holds the (<= 8) grad_y values of four pixels and writes 32 channels of them: plain fp32 FMAs in the fixed order k = 0 .. K-1 (first reader,
// then second), store-bound.
constexpr int kMaxPredGroup = 8, kPredK = 8, kPredCg = 32;
struct PredDgradGroup {
    const float* ga[kMaxPredGroup];
    const float* gb[kMaxPredGroup];
    const float* wa[kMaxPredGroup];
    const float* wb[kMaxPredGroup];
    float* gx[kMaxPredGroup];
    int Ka[kMaxPredGroup], Kb[kMaxPredGroup], N[kMaxPredGroup], C[kMaxPredGroup], HW[kMaxPredGroup], qblocks[kMaxPredGroup], first[kMaxPredGroup + 1];
    int n;
};
__global__ __launch_bounds__(256) void pred_dgrad_group_kernel(const PredDgradGroup a) {
    __shared__ float ws[kPredK][kPredCg];
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int b = (int)blockIdx.x - a.first[p];
    const int cg = b / a.qblocks[p], qb = b - cg * a.qblocks[p];
    const int Ka = a.Ka[p], Kb = a.Kb[p], C = a.C[p], HW = a.HW[p], K = Ka + Kb;
    const int c0 = cg * kPredCg;
    for (int i = threadIdx.x; i < K * kPredCg; i += blockDim.x) {
        const int k = i / kPredCg, c = c0 + i % kPredCg;
        ws[k][i % kPredCg] = c < C ? (k < Ka ? a.wa[p][(size_t)k * C + c] : a.wb[p][(size_t)(k - Ka) * C + c]) : 0.0f;
    }
    __syncthreads();
    const int hw4 = HW / 4;
    const long long i = (long long)qb * 256 + threadIdx.x;
    if (i >= (long long)a.N[p] * hw4) return;
    const int n = (int)(i / hw4), q = (int)(i - (long long)n * hw4);
    float4 g[kPredK];
#pragma unroll
    for (int k = 0; k < kPredK; ++k) {
        if (k < Ka) g[k] = *reinterpret_cast<const float4*>(a.ga[p] + ((size_t)n * Ka + k) * HW + 4 * q);
        else if (k < K) g[k] = *reinterpret_cast<const float4*>(a.gb[p] + ((size_t)n * Kb + (k - Ka)) * HW + 4 * q);
        else g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* out = a.gx[p] + ((size_t)n * C + c0) * HW + 4 * q;
    const int nc = C - c0 < kPredCg ? C - c0 : kPredCg;
    for (int c = 0; c < nc; ++c) {
        float4 v = make_float4(ws[0][c] * g[0].x, ws[0][c] * g[0].y, ws[0][c] * g[0].z, ws[0][c] * g[0].w);
#pragma unroll
        for (int k = 1; k < kPredK; ++k)
            if (k < K) {
                const float wk = ws[k][c];
                v.x = fmaf(wk, g[k].x, v.x); v.y = fmaf(wk, g[k].y, v.y); v.z = fmaf(wk, g[k].z, v.z); v.w = fmaf(wk, g[k].w, v.w);
            }
        *reinterpret_cast<float4*>(out + (size_t)c * HW) = v;
    }
}

}  // namespace

extern "C" {

int eas_upcat_fwd(const float* a, const float* b, float* out, int64_t M, int Ca, int Cb, int H, int W, int up, eas_stream_t stream) {
    if (!a || !b || !out) return EAS_ERR_INVALID_ARG;
    const UpcatGeom g{M, Ca, Cb, H, W, up};
    if (int rc = check_geom(g)) return rc;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(upcat_fwd_kernel, dim3(eas_grid_1d(M * (Ca + Cb) * H * up * (W * up / 4))), dim3(EAS_BLOCK), 0, eas_s(stream), a, b, out, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_upcat_bwd(const float* grad_out, float* grad_a, float* grad_b, int64_t M, int Ca, int Cb, int H, int W, int up, eas_stream_t stream) {
    if (!grad_out || !grad_a || !grad_b) return EAS_ERR_INVALID_ARG;
    const UpcatGeom g{M, Ca, Cb, H, W, up};
    if (int rc = check_geom(g)) return rc;
    if (((uintptr_t)grad_out | (uintptr_t)grad_a | (uintptr_t)grad_b) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(upcat_bwd_kernel, dim3(eas_grid_1d(M * Ca * H * (W / 2) + M * Cb * H * up * (W * up / 4))), dim3(EAS_BLOCK), 0,
                       eas_s(stream), grad_out, grad_a, grad_b, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// x [M][C][2*Ho][2*Wo] -> out [M][4*C][Ho][Wo] (inverse = 0), or the inverse permutation (inverse = 1: the backward).  Wo % 2 == 0.
int eas_focus(const float* src, float* dst, int64_t M, int C, int Ho, int Wo, int inverse, eas_stream_t stream) {
    if (!src || !dst || M < 1 || C < 1 || Ho < 1 || Wo < 1) return EAS_ERR_INVALID_ARG;
    if (Wo % 2 != 0) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(focus_kernel, dim3(eas_grid_1d(M * C * Ho * (Wo / 2))), dim3(EAS_BLOCK), 0, eas_s(stream), src, dst, (long long)M, C, Ho, Wo,
                       inverse);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// out[c] = sum_{n, pixel} g[n][c][pixel] for g [N][C][HW] (fp32, 16-byte aligned when HW % 4 == 0): the bias gradient of a convolution.
int eas_channel_sum(const float* g, float* out, int N, int C, int HW, eas_stream_t stream) {
    if (!g || !out || N < 1 || C < 1 || HW < 1) return EAS_ERR_INVALID_ARG;
    if ((HW & 3) == 0 && ((uintptr_t)g & 15)) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(channel_sum_kernel, dim3(C), dim3(256), 0, eas_s(stream), g, out, N, C, HW);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_channel_sum_group(const EasChannelSumProblem* pr, int n, eas_stream_t stream) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (n > kMaxSumGroup) return EAS_ERR_UNSUPPORTED;
    ChannelSumGroup a{};
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        const EasChannelSumProblem& q = pr[p];
        if (!q.g || !q.out || q.N < 1 || q.C < 1 || q.HW < 1) return EAS_ERR_INVALID_ARG;
        if ((q.HW & 3) == 0 && ((uintptr_t)q.g & 15)) return EAS_ERR_INVALID_ARG;
        a.g[p] = q.g; a.out[p] = q.out; a.N[p] = q.N; a.C[p] = q.C; a.HW[p] = q.HW;
        a.first[p] = blocks;
        blocks += q.C;
    }
    a.first[n] = blocks;
    a.n = n;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(channel_sum_group_kernel, dim3(blocks), dim3(256), 0, eas_s(stream), a);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_pred_dgrad_group(const EasPredDgradProblem* pr, int n, eas_stream_t stream) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (n > kMaxPredGroup) return EAS_ERR_UNSUPPORTED;
    PredDgradGroup a{};
    long long blocks = 0;
    for (int p = 0; p < n; ++p) {
        const EasPredDgradProblem& q = pr[p];
        if (!q.gy_a || !q.w_a || !q.gx || q.Ka < 1 || q.Kb < 0 || (q.Kb > 0 && (!q.gy_b || !q.w_b)) || q.N < 1 || q.C < 1 || q.HW < 1) return EAS_ERR_INVALID_ARG;
        if (q.Ka + q.Kb > kPredK || (q.HW & 3) != 0) return EAS_ERR_UNSUPPORTED;
        if (((uintptr_t)q.gy_a | (uintptr_t)q.gy_b | (uintptr_t)q.gx) & 15) return EAS_ERR_INVALID_ARG;
        const long long quads = (long long)q.N * (q.HW / 4);
        a.ga[p] = q.gy_a; a.gb[p] = q.Kb ? q.gy_b : q.gy_a; a.wa[p] = q.w_a; a.wb[p] = q.Kb ? q.w_b : q.w_a; a.gx[p] = q.gx;
        a.Ka[p] = q.Ka; a.Kb[p] = q.Kb; a.N[p] = q.N; a.C[p] = q.C; a.HW[p] = q.HW;
        a.qblocks[p] = (int)((quads + 255) / 256);
        a.first[p] = (int)blocks;
        blocks += (long long)a.qblocks[p] * ((q.C + kPredCg - 1) / kPredCg);
        if (blocks >= (1LL << 31)) return EAS_ERR_UNSUPPORTED;
    }
    a.first[n] = (int)blocks;
    a.n = n;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(pred_dgrad_group_kernel, dim3((unsigned)blocks), dim3(256), 0, eas_s(stream), a);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
