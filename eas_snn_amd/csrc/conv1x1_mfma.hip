// K4 (conv half): 1x1 convolutions (CSP conv1/conv2/conv3, bottleneck conv1, SPP, head stems; forward and input gradient)
// on the matrix cores, without LDS: a 1x1 convolution is y[co][p] = sum_ci w[co][ci] * x[ci][p] with no spatial reuse,
// so every wave streams its own pixels once.  For the MFMA B operand lane (pixel r, half h) needs 8 consecutive input
// channels of ONE pixel: 8 dword loads, each a 128-byte line per 32 lanes (pixels are contiguous in NCHW), converted in
// registers to exact bf16 term(s).  A = weights pre-packed in fragment order (conv_pack_weights_kernel, taps = 1), read
// from L1/L2.  HBM-bound (<= 64 flop/byte): x is read once per 32*WM output channels, y written once.
// (reference: nn.Conv2d(k=1) inside BaseConv, yolox/models/network_blocks.py:31-56.)
#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// loads of lanes without a pixel / channel groups past Cin read these zeros (stride 0) instead of being masked
__device__ __attribute__((aligned(32))) float eas_c1_zero_page[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct C1Geom {
    int NI, Cin, Cout, HW;
    int tiles_per_img, total_tiles;   // 32-pixel tiles
    int MT, KSTEPS;
};

template <int XT>
__device__ __forceinline__ void to_terms(const float (&v)[8], bf16x8 (&b)[XT]) {
    if constexpr (XT == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b[0][j] = (__bf16)v[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hi = (__bf16)v[j];
            const float r1 = v[j] - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const float r2 = r1 - (float)mid;
            b[0][j] = hi;
            b[1][j] = mid;
            b[2][j] = (__bf16)r2;
        }
    }
}

template <int XT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_kernel(const float* __restrict__ x, const bf16x8* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = blockIdx.y * WM;
    const int tile0 = (blockIdx.x * 4 + wave) * WN;
    if (tile0 >= g.total_tiles) return;

    // per-lane pixel of each N-tile
    long xoff[WN];    // element offset of (img, channel 8h, pixel) ; -1: no pixel
    long yoff[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int t = tile0 + n;
        const int img = t / g.tiles_per_img, p = (t - img * g.tiles_per_img) * 32 + r;
        const bool ok = t < g.total_tiles && p < g.HW;
        xoff[n] = ok ? ((long)img * g.Cin + 8 * h) * g.HW + p : -1;
        yoff[n] = ok ? (long)img * g.Cout * g.HW + p : -1;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    const size_t a_term = (size_t)g.MT * g.KSTEPS * 64;
    const bf16x8* ap[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int mt = (mt0 + m) < g.MT ? (mt0 + m) : g.MT - 1;
        ap[m] = wp + (size_t)mt * g.KSTEPS * 64 + lane;
    }

    // raw x of k-step ks: 8 channels (ks*16 + 8h + j) of this lane's pixel per N-tile.  Lanes without a pixel and channel
    // groups past Cin (Cin % 8 == 0) read the zero page with stride 0: unconditional loads, no per-element masking
    auto fetch = [&](float (&raw)[WN][8], int ks) {
        const bool ch_ok = ks * 16 + 8 * h < g.Cin;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const bool ok = ch_ok && xoff[n] >= 0;
            const float* src = ok ? x + xoff[n] + (long)ks * 16 * g.HW : eas_c1_zero_page;
            const long cs = ok ? (long)g.HW : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[n][j] = src[j * cs];
        }
    };
    auto step = [&](float (&raw)[WN][8], int ks) {
        bf16x8 a[WM][3];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[m][t] = ap[m][t * a_term + (size_t)ks * 64];
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            bf16x8 b[XT];
            to_terms<XT>(raw[n], b);
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                if constexpr (XT == 1) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[0], acc[m][n], 0, 0, 0);
                } else {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[2], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b[0], acc[m][n], 0, 0, 0);
                }
            }
        }
    };

    // channel loop, two raw buffers: the loads of k-step s+1 are in flight while k-step s is converted and multiplied
    float r0[WN][8], r1[WN][8];
    fetch(r0, 0);
    int ks = 0;
    for (; ks + 1 < g.KSTEPS; ks += 2) {
        fetch(r1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(r0, ks);
        fetch(r0, ks + 2 < g.KSTEPS ? ks + 2 : ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(r1, ks + 1);
    }
    if (ks < g.KSTEPS) step(r0, ks);

    // lean epilogue: one pointer per (M-tile, pixel tile), rows by multiples of the channel stride, bias once per M-tile,
    // per-element channel check only for a ragged last M-tile
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        if (mt0 + m >= g.MT) continue;
        const int co0 = (mt0 + m) * 32 + 4 * h;
        const bool full = (mt0 + m) * 32 + 32 <= g.Cout;
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = 0.0f;
        if (bias) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (e & 3) + 8 * (e >> 2);
                bv[e] = co < g.Cout ? bias[co] : 0.0f;
            }
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            if (yoff[n] < 0) continue;
            float* yp = y + yoff[n] + (long)co0 * g.HW;
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) yp[(long)((e & 3) + 8 * (e >> 2)) * g.HW] = acc[m][n][e] + bv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[(long)((e & 3) + 8 * (e >> 2)) * g.HW] = acc[m][n][e] + bv[e];
            }
        }
    }
}

template <int XT, int WM, int WN>
int launch_c1(const float* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    hipLaunchKernelGGL((conv1x1_mfma_kernel<XT, WM, WN>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

}  // namespace

// called by eas_conv_fwd (conv_mfma.hip) for ksize == 1, stride == 1
int eas_conv1x1_dispatch(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                         hipStream_t st) {
    C1Geom g{};
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.HW = HW;
    g.tiles_per_img = (HW + 31) / 32;
    g.total_tiles = NI * g.tiles_per_img;
    g.MT = (Cout + 31) / 32;
    g.KSTEPS = (Cin + 15) / 16;
    const bf16x8* wp = (const bf16x8*)packed_w;
    // all output channels in one wave when they fit (x is then read exactly once); 2 pixel tiles per wave
    if (x_terms == 1) {
        if (g.MT >= 4) return launch_c1<1, 4, 2>(x, wp, bias, y, g, st);
        if (g.MT >= 2) return launch_c1<1, 2, 2>(x, wp, bias, y, g, st);
        return launch_c1<1, 1, 2>(x, wp, bias, y, g, st);
    }
    if (g.MT >= 4) return launch_c1<3, 4, 2>(x, wp, bias, y, g, st);
    if (g.MT >= 2) return launch_c1<3, 2, 2>(x, wp, bias, y, g, st);
    return launch_c1<3, 1, 2>(x, wp, bias, y, g, st);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of a 1x1 convolution: dW[co][ci] = sum_{n,p} gy[n][co][p] * x[n][ci][p]   (reduction index = pixel).
// Both operands are read straight from global memory in MFMA fragment shape: lane (channel r, half h) needs 8 consecutive
// pixels of one channel = 32 contiguous bytes of NCHW.  grad_y is split into three exact bf16 terms in registers, x into
// one (spikes) or three.  A wave owns a 32 (co) x 128 (ci) tile; the four waves of a block take four co tiles over the
// same pixels (their x loads hit in L1); pixel ranges are split over blockIdx.x and reduced through slabs in fixed order.
namespace {

struct W1Geom {
    int NI, Cin, Cout, HW;
    int ksteps_img, total_ksteps, per_slice;
    int ci_blocks;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int XT>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ slabs,
                                                               W1Geom g) {
    constexpr int WN = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cog = blockIdx.y / g.ci_blocks, cib = blockIdx.y - cog * g.ci_blocks;
    const int co = (cog * 4 + wave) * 32 + r;          // A row of this lane
    const int ci0 = cib * 32 * WN;
    if ((cog * 4 + wave) * 32 >= g.Cout) return;
    const bool co_ok = co < g.Cout;
    const long a_row = (long)(co_ok ? co : g.Cout - 1) * g.HW + 8 * h;
    long b_row[WN];
    bool ci_ok[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ci = ci0 + 32 * n + r;
        ci_ok[n] = ci < g.Cin;
        b_row[n] = (long)(ci_ok[n] ? ci : g.Cin - 1) * g.HW + 8 * h;
    }
    f32x16 acc[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;

    const int ks_begin = blockIdx.x * g.per_slice;
    const int ks_end = ks_begin + g.per_slice < g.total_ksteps ? ks_begin + g.per_slice : g.total_ksteps;

    // rows past Cout / Cin read the zero page (no offsets added): unconditional loads, no masking of the values
    auto fetch = [&](f32x4 (&ra)[2], f32x4 (&rb)[WN][2], int ks) {
        const int img = ks / g.ksteps_img, p0 = (ks - img * g.ksteps_img) * 16;
        const float* pa = co_ok ? gy + (long)img * g.Cout * g.HW + a_row + p0 : eas_c1_zero_page;
        ra[0] = *(const f32x4*)pa;
        ra[1] = *(const f32x4*)(pa + 4);
        const float* pb = x + (long)img * g.Cin * g.HW + p0;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const float* q = ci_ok[n] ? pb + b_row[n] : eas_c1_zero_page;
            rb[n][0] = *(const f32x4*)q;
            rb[n][1] = *(const f32x4*)(q + 4);
        }
    };
    auto step = [&](f32x4 (&ra)[2], f32x4 (&rb)[WN][2]) {
        float va[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) va[j] = ra[j >> 2][j & 3];
        bf16x8 a[3];
        to_terms<3>(va, a);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            float vb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) vb[j] = rb[n][j >> 2][j & 3];
            bf16x8 b[XT];
            to_terms<XT>(vb, b);
            if constexpr (XT == 1) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[n], 0, 0, 0);
            } else {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[n], 0, 0, 0);
            }
        }
    };

    f32x4 a0[2], a1[2], b0[WN][2], b1[WN][2];
    int ks = ks_begin;
    if (ks < ks_end) fetch(a0, b0, ks);
    for (; ks + 1 < ks_end; ks += 2) {
        fetch(a1, b1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(a0, b0);
        fetch(a0, b0, ks + 2 < ks_end ? ks + 2 : ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(a1, b1);
    }
    if (ks < ks_end) step(a0, b0);

    float* slab = slabs + (size_t)blockIdx.x * g.Cout * g.Cin;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ci = ci0 + 32 * n + r;
        if (ci >= g.Cin) continue;
        const int row0 = (cog * 4 + wave) * 32 + 4 * h;
        float* sp = slab + (size_t)row0 * g.Cin + ci;
        if (row0 - 4 * h + 32 <= g.Cout) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sp[(size_t)((e & 3) + 8 * (e >> 2)) * g.Cin] = acc[n][e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (row0 + (e & 3) + 8 * (e >> 2) < g.Cout) sp[(size_t)((e & 3) + 8 * (e >> 2)) * g.Cin] = acc[n][e];
        }
    }
}

bool w1_geom(W1Geom& g, int NI, int Cin, int Cout, int HW) {
    if (HW % 16 != 0) return false;
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.HW = HW;
    g.ksteps_img = HW / 16;
    g.total_ksteps = NI * g.ksteps_img;
    g.ci_blocks = (Cin + 127) / 128;
    const int yz = ((Cout + 127) / 128) * g.ci_blocks;
    int slices = (512 + yz - 1) / yz;                       // two 4-wave blocks per CU
    if (slices > g.total_ksteps) slices = g.total_ksteps;
    g.per_slice = (g.total_ksteps + slices - 1) / slices;
    return true;
}

}  // namespace

int eas_conv1x1_wgrad_slices(int NI, int Cin, int Cout, int HW) {
    W1Geom g{};
    if (!w1_geom(g, NI, Cin, Cout, HW)) return 0;
    return (g.total_ksteps + g.per_slice - 1) / g.per_slice;
}

// slabs: eas_conv1x1_wgrad_slices(...) * Cout * Cin floats; the caller reduces them (conv_wgrad_reduce_kernel)
int eas_conv1x1_wgrad_dispatch(const float* x, const float* gy, float* slabs, int NI, int Cin, int Cout, int HW, int x_terms, hipStream_t st) {
    W1Geom g{};
    if (!w1_geom(g, NI, Cin, Cout, HW)) return EAS_ERR_UNSUPPORTED;
    const int slices = (g.total_ksteps + g.per_slice - 1) / g.per_slice;
    dim3 grid(slices, ((Cout + 127) / 128) * g.ci_blocks);
    if (x_terms == 1) hipLaunchKernelGGL(conv1x1_wgrad_kernel<1>, grid, dim3(256), 0, st, x, gy, slabs, g);
    else hipLaunchKernelGGL(conv1x1_wgrad_kernel<3>, grid, dim3(256), 0, st, x, gy, slabs, g);
    return EAS_OK;
}
