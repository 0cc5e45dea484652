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

__global__ __launch_bounds__(EAS_BLOCK) void spp_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, SppGeom g) {
    __shared__ float plane[kMaxPlane];
    __shared__ unsigned rowarg[kMaxPlane];
    __shared__ unsigned amax[kMaxPlane];
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

__global__ __launch_bounds__(EAS_BLOCK) void spp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ gx,
                                                            SppGeom g) {
    __shared__ float plane[kMaxPlane];
    __shared__ unsigned rowarg[kMaxPlane];
    __shared__ unsigned amax[kMaxPlane];
    __shared__ float gp[3][kMaxPlane];
    __shared__ float tcol[3][kMaxPlane];
    __shared__ unsigned brow[kMaxPlane];
    const long long p0 = (long long)blockIdx.x * g.ppb;
    const int np = (int)(g.planes - p0 < g.ppb ? g.planes - p0 : g.ppb);
    const int work = np * g.HW;
    const float* src = x + p0 * g.HW;
    for (int idx = threadIdx.x; idx < work; idx += blockDim.x) {
        plane[idx] = src[idx];
        const int lp = idx / g.HW, i = idx - lp * g.HW;
        const long long pc = p0 + lp, n = pc / g.C, c = pc - n * g.C;
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
    EAS_LAUNCH(spp_fwd_kernel, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), 0, eas_s(stream), x, out, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_spp_pool_bwd(const float* x, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                     eas_stream_t stream) {
    if (!x || !grad_out || !grad_x) return EAS_ERR_INVALID_ARG;
    SppGeom g;
    if (int rc = make_geom(g, N, C, H, W, k0, k1, k2)) return rc;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(spp_bwd_kernel, dim3((unsigned)((g.planes + g.ppb - 1) / g.ppb)), dim3(EAS_BLOCK), 0, eas_s(stream), x, grad_out, grad_x, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
