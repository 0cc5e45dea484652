// SPP pooling block fused: out = cat[x, maxpool_k0(x), maxpool_k1(x), maxpool_k2(x)] along channels (stride 1, padding k/2)
// and its backward (reference: SPPBottleneck.forward, yolox/models/network_blocks.py:143-147, on the spike tensors of
// dark5; ATen max_pool2d semantics: the first maximum in row-major scan order wins ties, which matters for 0/1 inputs).
//
// A block holds PPB (image, channel) planes in LDS (dark5 planes are 8x10: three of them fill a 256-thread block).
// The arg-max is separable under the first-maximum rule: the leftmost maximum of each row window (one scan over the
// widest window serves all three pool sizes), then the first row whose row-maximum is strictly greater.  The three
// arg-max indices of a pixel are packed into one word (10 bits each, planes <= 1024 pixels).  The backward recomputes
// the arg-max from x (nothing is saved) and gathers separably (column pass, then row pass) in a fixed order (deterministic,
// no atomics), with the pooled gradients staged in LDS.
#include "eas_common.h"

namespace {

constexpr int kMaxPlane = 1024;   // pixels per plane (10-bit indices) and LDS floats per block

struct SppGeom {
    int C, H, W, HW, ppb, r0, r1, r2;
    long long planes;
};

__device__ __forceinline__ bool better(float v, float bv) { return v > bv || v != v; }   // NaN propagates like ATen

// plane[ppb*HW] loaded; fills amax[ppb*HW] with the packed plane-local arg-max indices of the three windows.
__device__ __forceinline__ void spp_argmax(const float* plane, unsigned* rowarg, unsigned* amax, const SppGeom& g, int work) {
    const int rs[3] = {g.r0, g.r1, g.r2};
    const int rmax = max(g.r0, max(g.r1, g.r2));
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, w = i - h * g.W;
        const float* row = plane + lp * g.HW + h * g.W;
        int best[3];
        float bv[3];
        bool have[3] = {false, false, false};
        const int b0 = w - rmax < 0 ? 0 : w - rmax, b1 = w + rmax >= g.W ? g.W - 1 : w + rmax;
        for (int b = b0; b <= b1; ++b) {
            const float v = row[b];
            const int d = b > w ? b - w : w - b;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (d <= rs[j] && (!have[j] || better(v, bv[j]))) {
                    have[j] = true;
                    bv[j] = v;
                    best[j] = b;
                }
        }
        rowarg[idx] = (unsigned)best[0] | ((unsigned)best[1] << 10) | ((unsigned)best[2] << 20);   // column of the row-window maximum
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, w = i - h * g.W;
        const float* pl = plane + lp * g.HW;
        const unsigned* ra = rowarg + lp * g.HW;
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = rs[j];
            const int a0 = h - r < 0 ? 0 : h - r, a1 = h + r >= g.H ? g.H - 1 : h + r;
            int best = a0 * g.W + (int)((ra[a0 * g.W + w] >> (10 * j)) & 1023u);
            float bv = pl[best];
            for (int a = a0 + 1; a <= a1; ++a) {
                const int q = a * g.W + (int)((ra[a * g.W + w] >> (10 * j)) & 1023u);
                const float v = pl[q];
                if (better(v, bv)) {
                    bv = v;
                    best = q;
                }
            }
            packed |= (unsigned)best << (10 * j);
        }
        amax[idx] = packed;
    }
}

// LDS is sized by the launch (ppb * HW words per array, not the 1024-pixel maximum): three 8x10 planes per block are 2.9 KB for the forward and
// 9.6 KB for the backward instead of 12 / 40 KB, so the CU holds its eight 4-wave blocks instead of four (the backward: 0.21 -> 0.1 ms)
__global__ __launch_bounds__(EAS_BLOCK) void spp_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, SppGeom g) {
    extern __shared__ __align__(16) unsigned char spp_smem[];
    const int cap = g.ppb * g.HW;
    float* plane = reinterpret_cast<float*>(spp_smem);
    unsigned* rowarg = reinterpret_cast<unsigned*>(spp_smem) + cap;
    unsigned* amax = reinterpret_cast<unsigned*>(spp_smem) + 2 * cap;
    const long long p0 = (long long)blockIdx.x * g.ppb;
    const int np = (int)(g.planes - p0 < g.ppb ? g.planes - p0 : g.ppb);
    const int work = np * g.HW;
    const float* src = x + p0 * g.HW;
    for (int i = threadIdx.x; i < work; i += blockDim.x) plane[i] = src[i];
    __syncthreads();
    spp_argmax(plane, rowarg, amax, g, work);
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW;
        const long long pc = p0 + lp, n = pc / g.C, c = pc - n * g.C;
        float* dst = out + (n * 4 * g.C + c) * g.HW + i;
        const float* pl = plane + lp * g.HW;
        const unsigned pk = amax[idx];
        dst[0] = pl[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[(long long)(j + 1) * g.C * g.HW] = pl[(pk >> (10 * j)) & 1023u];
    }
}

// XPL: x is given as spike planes (bf16 [N][C/8][HW][8]): the fp32 copy of the spikes that only this kernel (and the forward) read is gone
template <bool XPL>
__global__ __launch_bounds__(EAS_BLOCK) void spp_bwd_kernel(const void* __restrict__ x_, const float* __restrict__ gout, float* __restrict__ gx,
                                                            SppGeom g) {
    extern __shared__ __align__(16) unsigned char spp_smem[];
    const int cap = g.ppb * g.HW;
    float* plane = reinterpret_cast<float*>(spp_smem);
    unsigned* rowarg = reinterpret_cast<unsigned*>(spp_smem) + cap;
    unsigned* amax = reinterpret_cast<unsigned*>(spp_smem) + 2 * cap;
    float* gp[3] = {plane + 3 * cap, plane + 4 * cap, plane + 5 * cap};
    float* tcol[3] = {plane + 6 * cap, plane + 7 * cap, plane + 8 * cap};
    unsigned* brow = reinterpret_cast<unsigned*>(spp_smem) + 9 * cap;
    const long long p0 = (long long)blockIdx.x * g.ppb;
    const int np = (int)(g.planes - p0 < g.ppb ? g.planes - p0 : g.ppb);
    const int work = np * g.HW;
    const float* src = reinterpret_cast<const float*>(x_) + p0 * g.HW;
    const __bf16* xp = reinterpret_cast<const __bf16*>(x_);
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW;
        const long long pc = p0 + lp, n = pc / g.C, c = pc - n * g.C;
        if constexpr (XPL) plane[idx] = (float)xp[((n * (g.C / 8) + c / 8) * g.HW + i) * 8 + (c & 7)];
        else plane[idx] = src[idx];
        const float* gsrc = gout + (n * 4 * g.C + c) * g.HW + i;
#pragma unroll
        for (int j = 0; j < 3; ++j) gp[j][idx] = gsrc[(long long)(j + 1) * g.C * g.HW];
    }
    __syncthreads();
    spp_argmax(plane, rowarg, amax, g, work);
    __syncthreads();
    // row of each pool's arg-max, worked out once per pixel (the column pass below compares it 5 + 9 + 13 times per pixel: the integer
    // division inside that loop was most of the kernel's instructions)
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const unsigned pk = amax[idx];
        brow[idx] = ((pk & 1023u) / g.W) | ((((pk >> 10) & 1023u) / g.W) << 10) | ((((pk >> 20) & 1023u) / g.W) << 20);
    }
    __syncthreads();
    // Separable gather.  Output (a, b) of pool j sends its gradient to (row, col) = (bestrow_j(a, b), rowarg_j(bestrow, b)):
    //   tcol_j(h, b) = sum over a in [h - r, h + r] of g_j(a, b) where bestrow_j(a, b) == h          (column pass)
    //   gx(h, w)    += sum over b in [w - r, w + r] of tcol_j(h, b) where rowarg_j(h, b) == w         (row pass)
    // 2 * (2r + 1) reads per pixel and pool instead of (2r + 1)^2, fixed order (deterministic, no atomics).
    const int rs[3] = {g.r0, g.r1, g.r2};
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, b = i - h * g.W;
        const int base = lp * g.HW;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = rs[j];
            const int a0 = h - r < 0 ? 0 : h - r, a1 = h + r >= g.H ? g.H - 1 : h + r;
            float t = 0.f;
            for (int a = a0; a <= a1; ++a) {
                const int q = base + a * g.W + b;
                if ((int)((brow[q] >> (10 * j)) & 1023u) == h) t += gp[j][q];
            }
            tcol[j][idx] = t;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, w = i - h * g.W;
        const long long pc = p0 + lp, n = pc / g.C, c = pc - n * g.C;
        const int rowbase = lp * g.HW + h * g.W;
        float s = gout[(n * 4 * g.C + c) * g.HW + i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int r = rs[j];
            const int b0 = w - r < 0 ? 0 : w - r, b1 = w + r >= g.W ? g.W - 1 : w + r;
            for (int b = b0; b <= b1; ++b)
                if ((int)((rowarg[rowbase + b] >> (10 * j)) & 1023u) == w) s += tcol[j][rowbase + b];
        }
        gx[pc * g.HW + i] = s;
    }
}

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// block = ppb (image, 8-channel group) maps of HW pixels x 16 bytes; LDS: the maps and, per pool size, their row-window maxima
__global__ __launch_bounds__(EAS_BLOCK) void spp_planes_fwd_kernel(const u16x8* __restrict__ x, u16x8* __restrict__ out, SppGeom g) {
    extern __shared__ __align__(16) unsigned char spp_smem[];
    const int cap = g.ppb * g.HW;
    u16x8* map = reinterpret_cast<u16x8*>(spp_smem);
    u16x8* rowmax = map + cap;                       // [3][cap]
    const long long p0 = (long long)blockIdx.x * g.ppb;
    const int np = (int)(g.planes - p0 < g.ppb ? g.planes - p0 : g.ppb);
    const int work = np * g.HW;
    const int G = g.C;                               // channel groups of x (make_geom was given C / 8)
    for (int i = threadIdx.x; i < work; i += blockDim.x) map[i] = x[p0 * g.HW + i];
    __syncthreads();
    const int rs[3] = {g.r0, g.r1, g.r2};
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, w = i - h * g.W;
        const u16x8* row = map + lp * g.HW + h * g.W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int b0 = w - rs[j] < 0 ? 0 : w - rs[j], b1 = w + rs[j] >= g.W ? g.W - 1 : w + rs[j];
            u16x8 m = row[b0];
            for (int b = b0 + 1; b <= b1; ++b) m = __builtin_elementwise_max(m, row[b]);
            rowmax[j * cap + idx] = m;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        const int lp = idx / g.HW, i = idx - lp * g.HW, h = i / g.W, w = i - h * g.W;
        const long long pg = p0 + lp, n = pg / G, grp = pg - n * G;
        u16x8* dst = out + (n * 4 * G + grp) * g.HW + i;
        dst[0] = map[idx];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int a0 = h - rs[j] < 0 ? 0 : h - rs[j], a1 = h + rs[j] >= g.H ? g.H - 1 : h + rs[j];
            const u16x8* col = rowmax + j * cap + lp * g.HW + w;
            u16x8 m = col[a0 * g.W];
            for (int a = a0 + 1; a <= a1; ++a) m = __builtin_elementwise_max(m, col[a * g.W]);
            dst[(long long)(j + 1) * G * g.HW] = m;
        }
    }
}

int make_geom(SppGeom& g, int64_t N, int C, int H, int W, int k0, int k1, int k2) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return EAS_ERR_INVALID_ARG;
    if (H * W > kMaxPlane || !(k0 & 1) || !(k1 & 1) || !(k2 & 1) || k0 < 1 || k1 < 1 || k2 < 1 || N * C > 0x7fffffff) return EAS_ERR_UNSUPPORTED;
    g.C = C; g.H = H; g.W = W; g.HW = H * W;
    g.r0 = k0 / 2; g.r1 = k1 / 2; g.r2 = k2 / 2;
    g.planes = N * C;
    g.ppb = EAS_BLOCK / g.HW < 1 ? 1 : EAS_BLOCK / g.HW;
    return EAS_OK;
}

}  // namespace

extern "C" {

int eas_spp_pool_fwd(const float* x, float* out, int64_t N, int C, int H, int W, int k0, int k1, int k2, eas_stream_t stream) {
    if (!x || !out) return EAS_ERR_INVALID_ARG;
    SppGeom g;
    if (int rc = make_geom(g, N, C, H, W, k0, k1, k2)) return rc;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(spp_fwd_kernel, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), (size_t)3 * g.ppb * g.HW * 4, eas_s(stream), x, out, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_spp_pool_bwd(const float* x, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                     eas_stream_t stream) {
    if (!x || !grad_out || !grad_x) return EAS_ERR_INVALID_ARG;
    SppGeom g;
    if (int rc = make_geom(g, N, C, H, W, k0, k1, k2)) return rc;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(spp_bwd_kernel<false>, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), (size_t)10 * g.ppb * g.HW * 4, eas_s(stream),
               (const void*)x, grad_out, grad_x, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// The same block on SPIKE PLANES: x_planes [N][C/8][HW][8] (bf16 spikes / small integers) -> out_planes [N][4C/8][HW][8], channel groups
// [0, C/8) = x, then the three pooled copies.  Values are non-negative and exact in bf16, so their order is the order of their bit patterns:
// the window maximum is a packed 16-bit unsigned maximum over 16-byte pixels, separably (row window, then column window).
int eas_spp_pool_planes_fwd(const void* x_planes, void* out_planes, int64_t N, int C, int H, int W, int k0, int k1, int k2, eas_stream_t stream) {
    if (!x_planes || !out_planes) return EAS_ERR_INVALID_ARG;
    if (C % 8 != 0 || (((uintptr_t)x_planes | (uintptr_t)out_planes) & 15)) return EAS_ERR_UNSUPPORTED;
    SppGeom g;
    if (int rc = make_geom(g, N, C / 8, H, W, k0, k1, k2)) return rc;           // "planes" here = (image, 8-channel group) maps
    EAS_CLEAR_ERR();
    EAS_LAUNCH(spp_planes_fwd_kernel, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), (size_t)4 * g.ppb * g.HW * 16, eas_s(stream),
               (const u16x8*)x_planes, (u16x8*)out_planes, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// eas_spp_pool_bwd with x given as spike planes (grad_out fp32 [N][4C][HW], grad_x fp32 [N][C][HW])
int eas_spp_pool_planes_bwd(const void* x_planes, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                            eas_stream_t stream) {
    if (!x_planes || !grad_out || !grad_x) return EAS_ERR_INVALID_ARG;
    if (C % 8 != 0) return EAS_ERR_UNSUPPORTED;
    SppGeom g;
    if (int rc = make_geom(g, N, C, H, W, k0, k1, k2)) return rc;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(spp_bwd_kernel<true>, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), (size_t)10 * g.ppb * g.HW * 4, eas_s(stream),
               x_planes, grad_out, grad_x, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
