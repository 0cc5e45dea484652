// K4 (conv half): 1x1 convolutions (CSP conv1/conv2/conv3, bottleneck conv1, SPP, head stems; forward and input gradient)
// on the matrix cores, without LDS: a 1x1 convolution is y[co][p] = sum_ci w[co][ci] * x[ci][p] with no spatial reuse,
// so every wave streams its own pixels once.  For the MFMA B operand lane (pixel r, half h) needs 8 consecutive input
// channels of ONE pixel: 8 dword loads, each a 128-byte line per 32 lanes (pixels are contiguous in NCHW), converted in
// registers to exact bf16 term(s).  A = weights pre-packed in fragment order (conv_pack_weights_kernel, taps = 1), read
// from L1/L2.  HBM-bound (<= 64 flop/byte): x is read once per 32*WM output channels, y written once.
// (reference: nn.Conv2d(k=1) inside BaseConv, yolox/models/network_blocks.py:31-56.)
#include <stdlib.h>

#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// loads of lanes without a pixel / channel groups past Cin read these zeros (stride 0) instead of being masked
__device__ __attribute__((aligned(32))) float eas_c1_zero_page[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// pixel blocks (grid.x) of the last launch / geometry query of this thread: the statistics partials per channel
thread_local int tl_c1_blocks = 0;

struct C1Geom {
    int NI, Cin, Cout, HW;
    int tiles_per_img, total_tiles;   // 32-pixel tiles
    int MT, KSTEPS;
    double* stats;                    // BatchNorm statistics in the epilogue (eas_conv_fwd_stats): [Cout][stats_nb][2], stats_nb = gridDim.x
    int stats_nb;
};

// statistics epilogue of the 1x1 kernels: 4 waves of a block = 4 pixel groups of the same WM * 32 channels
template <int WM, int WN, typename ACC>
__device__ __forceinline__ void c1_stats(const ACC (&acc)[WM][WN], const long (&yoff)[WN], double* red, const C1Geom& g, int mt0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool valid[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) valid[n] = yoff[n] >= 0;
    eas_conv_stats_wave<WM, WN>(acc, valid, reinterpret_cast<float*>(red + 4 * WM * 64) + wave * EAS_STATS_SCRATCH, red + wave * (WM * 64), lane);
    __syncthreads();
    if ((int)threadIdx.x < WM * 32)
        eas_conv_stats_store(red, WM * 64, 4, (int)threadIdx.x, mt0 * 32 + (int)threadIdx.x, g.Cout, g.stats, g.stats_nb, (int)blockIdx.x);
}

template <int XT, typename TIN = float>
__device__ __forceinline__ void to_terms(const TIN (&v)[8], bf16x8 (&b)[XT]) {
    if constexpr (XT == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b[0][j] = (__bf16)(float)v[j];      // TIN = uint8_t: spike bytes 0..255 are exact in bf16
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

// RAGK: Cin is not a multiple of 8 (the input gradient of the 1/4/num_classes-channel prediction convolutions, whose "input" is
// grad_y): channel validity is then tested per channel instead of per 8-channel group.
template <int XT, int WM, int WN, bool RAGK = false, typename TIN = float>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_kernel(const TIN* __restrict__ x, const bf16x8* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = blockIdx.y * WM;
    const int tile0 = (blockIdx.x * 4 + wave) * WN;
    if (tile0 >= g.total_tiles && !g.stats) return;      // with statistics every wave reaches the block reduction (its lanes hold no pixel)

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
    auto fetch = [&](TIN (&raw)[WN][8], int ks) {
        const int ch0 = ks * 16 + 8 * h;
        const bool ch_ok = ch0 < g.Cin;
        const TIN* zero = reinterpret_cast<const TIN*>(eas_c1_zero_page);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const bool ok = ch_ok && xoff[n] >= 0;
            const TIN* src = ok ? x + xoff[n] + (long)ks * 16 * g.HW : zero;
            const long cs = ok ? (long)g.HW : 0;
            if constexpr (RAGK) {
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[n][j] = *((ok && ch0 + j < g.Cin) ? src + j * cs : zero);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[n][j] = src[j * cs];
            }
        }
    };
    auto step = [&](TIN (&raw)[WN][8], int ks) {
        bf16x8 a[WM][3];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[m][t] = ap[m][t * a_term + (size_t)ks * 64];
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            bf16x8 b[XT];
            to_terms<XT, TIN>(raw[n], b);
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
    TIN r0[WN][8], r1[WN][8];
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

    if (g.stats) {
        __shared__ __align__(16) double red[4 * WM * 64 + 2 * EAS_STATS_SCRATCH];      // [wave][WM * 32][2] doubles, then the waves' float patches
        c1_stats<WM, WN>(acc, yoff, red, g, mt0);
    }

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

// Spike-byte input (eas_conv_fwd_u8), direct form.  Per-lane byte loads would issue as many load instructions as the fp32 form
// (the texture path processes instructions, not bytes), so the quarter of the bytes would buy nothing.  Instead every lane
// fetches 16 consecutive pixels of ONE channel with a single 16-byte load (a k-step of a wave = 16 channels x 32*WN pixels =
// 512*WN bytes = 32*WN lanes), the wave parks them in its private LDS slice ([tile][channel][32 pixels], double-buffered, no
// block barrier: a wave only reads what it wrote) and each lane picks the 8 channel bytes of its pixel from there
// (8 ds_read_u8 per tile and k-step against 3*WM MFMAs).  HW % 16 == 0.
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1x1_u8_kernel(const uint8_t* __restrict__ x, const bf16x8* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][2][WN * 16 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = blockIdx.y * WM;
    const int tile0 = (blockIdx.x * 4 + wave) * WN;
    if (tile0 >= g.total_tiles) return;

    long yoff[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int t = tile0 + n;
        const int img = t / g.tiles_per_img, p = (t - img * g.tiles_per_img) * 32 + r;
        yoff[n] = (t < g.total_tiles && p < g.HW) ? (long)img * g.Cout * g.HW + p : -1;
    }
    // staging role of this lane: channel j of the k-step, 16-pixel chunk c of tile sn
    const int sj = lane & 15, sc = (lane >> 4) & 1, sn = lane >> 5;
    long soff = -1;
    if (sn < WN) {
        const int t = tile0 + sn;
        const int img = t / g.tiles_per_img, p0 = (t - img * g.tiles_per_img) * 32 + 16 * sc;
        if (t < g.total_tiles && p0 < g.HW) soff = ((long)img * g.Cin + sj) * g.HW + p0;
    }
    const int s_lds = (sn * 16 + sj) * 32 + 16 * sc;

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
    auto fetch = [&](int ks) -> uint4 {
        const bool ok = soff >= 0 && ks * 16 + sj < g.Cin;
        const uint8_t* src = ok ? x + soff + (long)ks * 16 * g.HW : reinterpret_cast<const uint8_t*>(eas_c1_zero_page);
        return *reinterpret_cast<const uint4*>(src);
    };
    auto commit = [&](int buf, const uint4& v) {
        if (sn < WN) *reinterpret_cast<uint4*>(&stage[wave][buf][s_lds]) = v;
    };
    auto step = [&](int buf, int ks) {
        bf16x8 a[WM][3];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[m][t] = ap[m][t * a_term + (size_t)ks * 64];
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            bf16x8 b;
            const uint8_t* q = &stage[wave][buf][(n * 16 + 8 * h) * 32 + r];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (__bf16)(float)q[j * 32];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], b, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], b, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], b, acc[m][n], 0, 0, 0);
            }
        }
    };

    uint4 v = fetch(0);
    commit(0, v);
    for (int ks = 0; ks < g.KSTEPS; ++ks) {
        const bool more = ks + 1 < g.KSTEPS;
        if (more) v = fetch(ks + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // this wave's LDS writes of buffer ks & 1 are visible to its reads
        __builtin_amdgcn_wave_barrier();
        step(ks & 1, ks);
        if (more) commit((ks + 1) & 1, v);
    }

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

template <int WM, int WN>
int launch_c1_u8(const uint8_t* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    hipLaunchKernelGGL((conv1x1_u8_kernel<WM, WN>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

// Variant for layers with many input channels and few pixels (dark5: 256..1024 channels on 15 360 pixels).  There the direct
// kernel is bound by the texture path, not by HBM or the matrix cores: every wave fetches its own copy of the weight fragments
// (WM x 3 KB per k-step) -- 64 KB per k-step and CU against 768 cycles of MFMA work.  Here the four waves of a block (same
// output channels, different pixels) share them: the block loads each k-step's WM x 3 fragments once (coalesced 16-byte
// loads), double-buffers them in LDS and every wave reads its operands with conflict-free ds_read_b128.  One barrier per k-step.
template <int XT, int WM, int WN, typename TIN = float>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_sharedA_kernel(const TIN* __restrict__ x, const bf16x8* __restrict__ wp,
                                                                      const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    constexpr int NFRAG = WM * 3;                        // 1 KB fragments per k-step
    constexpr int NLD = (NFRAG * 64 + 255) / 256;        // 16-byte staging loads per thread and k-step
    __shared__ bf16x8 As[2][NFRAG][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = blockIdx.y * WM;
    const int tile0 = (blockIdx.x * 4 + wave) * WN;

    long xoff[WN], yoff[WN];
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

    // staging piece q = it*256 + tid -> fragment f = q / 64 (= term * WM + m), lane q % 64
    const size_t a_term = (size_t)g.MT * g.KSTEPS * 64;
    const bf16x8* asrc[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        int q = it * 256 + tid;
        q = q < NFRAG * 64 ? q : NFRAG * 64 - 1;
        const int f = q >> 6, t = f / WM, m = f - t * WM;
        const int mt = (mt0 + m) < g.MT ? (mt0 + m) : g.MT - 1;
        asrc[it] = wp + t * a_term + (size_t)mt * g.KSTEPS * 64 + (q & 63);
    }
    bf16x8 areg[NLD];
    auto a_fetch = [&](int ks) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) areg[it] = asrc[it][(size_t)ks * 64];
    };
    auto a_commit = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = it * 256 + tid;
            if (q < NFRAG * 64) (&As[buf][0][0])[q] = areg[it];
        }
    };
    auto x_fetch = [&](TIN (&raw)[WN][8], int ks) {
        const bool ch_ok = ks * 16 + 8 * h < g.Cin;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const bool ok = ch_ok && xoff[n] >= 0;
            const TIN* src = ok ? x + xoff[n] + (long)ks * 16 * g.HW : reinterpret_cast<const TIN*>(eas_c1_zero_page);
            const long cs = ok ? (long)g.HW : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[n][j] = src[j * cs];
        }
    };
    auto step = [&](TIN (&raw)[WN][8], int buf) {
        bf16x8 a[WM][3];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[m][t] = As[buf][t * WM + m][lane];
        bf16x8 b[WN][XT];
#pragma unroll
        for (int n = 0; n < WN; ++n) to_terms<XT, TIN>(raw[n], b[n]);
        if constexpr (XT == 1) {
#pragma unroll
            for (int ta = 2; ta >= 0; --ta)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ta], b[n][0], acc[m][n], 0, 0, 0);
        } else {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[q]], b[n][PB[q]], acc[m][n], 0, 0, 0);
        }
    };

    TIN r0[WN][8], r1[WN][8];
    a_fetch(0);
    x_fetch(r0, 0);
    a_commit(0);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < g.KSTEPS; ks += 2) {
        a_fetch(ks + 1);
        x_fetch(r1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(r0, 0);
        a_commit(1);
        __syncthreads();
        const int kn = ks + 2 < g.KSTEPS ? ks + 2 : ks + 1;
        a_fetch(kn);
        x_fetch(r0, kn);
        __builtin_amdgcn_sched_barrier(0);
        step(r1, 1);
        a_commit(0);
        __syncthreads();
    }
    if (ks < g.KSTEPS) step(r0, 0);

    if (g.stats) {
        __shared__ __align__(16) double red[4 * WM * 64 + 2 * EAS_STATS_SCRATCH];      // [wave][WM * 32][2] doubles, then the waves' float patches
        c1_stats<WM, WN>(acc, yoff, red, g, mt0);
    }

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

template <int XT, int WM, int WN, typename TIN = float>
int launch_c1_shared(const TIN* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    tl_c1_blocks = (int)grid.x;
    if (!y) return EAS_OK;            // geometry query
    if (g.stats && (int)grid.x != g.stats_nb) return EAS_ERR_INVALID_ARG;
    hipLaunchKernelGGL((conv1x1_mfma_sharedA_kernel<XT, WM, WN, TIN>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

template <int XT, int WM, int WN, bool RAGK = false, typename TIN = float>
int launch_c1(const TIN* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    tl_c1_blocks = (int)grid.x;
    if (!y) return EAS_OK;            // geometry query
    if (g.stats && (int)grid.x != g.stats_nb) return EAS_ERR_INVALID_ARG;
    hipLaunchKernelGGL((conv1x1_mfma_kernel<XT, WM, WN, RAGK, TIN>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

}  // namespace

// wave-tile choice and launch for a 1x1 convolution; TIN = float (x_terms 1 or 3) or uint8_t (spike bytes, one term)
template <typename TIN>
static int conv1x1_dispatch_t(const TIN* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                              hipStream_t st, double* stats = nullptr, int stats_nb = 0) {
    constexpr bool U8 = sizeof(TIN) == 1;
    if (U8 && stats) return EAS_ERR_UNSUPPORTED;
    C1Geom g{};
    g.stats = stats; g.stats_nb = stats_nb;
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.HW = HW;
    g.tiles_per_img = (HW + 31) / 32;
    g.total_tiles = NI * g.tiles_per_img;
    g.MT = (Cout + 31) / 32;
    g.KSTEPS = (Cin + 15) / 16;
    const bf16x8* wp = (const bf16x8*)packed_w;
    // wave tile = WM channel tiles x WN pixel tiles.  Large layers: all output channels in one wave when they fit (x is then read
    // once per 128 output channels), 2 pixel tiles per wave.  Layers with few pixels and many channels (dark5: 15 360 pixels, up
    // to 1024 -> 512 channels) would leave most CUs with one 4-wave block or none: shrink the wave tile until the grid has at
    // least ~2 blocks per CU (x is re-read from L2 by more channel groups, which these small layers can afford).
    int wm = g.MT >= 4 ? 4 : (g.MT >= 2 ? 2 : 1), wn = 2;
    auto blocks = [&](int m, int n) { return (long)((g.total_tiles + 4 * n - 1) / (4 * n)) * ((g.MT + m - 1) / m); };
    static const long want = getenv("EAS_C1_BLOCKS") ? atol(getenv("EAS_C1_BLOCKS")) : 512;
    if (blocks(wm, wn) < want) wn = 1;
    while (blocks(wm, wn) < want && wm > 1) wm >>= 1;
    if constexpr (U8) {
        if (Cin % 8 != 0 || x_terms != 1) return EAS_ERR_UNSUPPORTED;
    } else if (Cin % 8 != 0) {     // ragged input channels: the direct kernel with per-channel validity (few channels: one pixel tile per wave)
        if (x_terms == 1) {
            if (g.MT >= 4) return launch_c1<1, 4, 1, true>(x, wp, bias, y, g, st);
            return g.MT >= 2 ? launch_c1<1, 2, 1, true>(x, wp, bias, y, g, st) : launch_c1<1, 1, 1, true>(x, wp, bias, y, g, st);
        }
        if (g.MT >= 4) return launch_c1<3, 4, 1, true>(x, wp, bias, y, g, st);
        return g.MT >= 2 ? launch_c1<3, 2, 1, true>(x, wp, bias, y, g, st) : launch_c1<3, 1, 1, true>(x, wp, bias, y, g, st);
    }
    // many input channels (>= 256): weight-fragment traffic, not HBM, bounds the direct kernel -> block-shared fragments
    static const int shared_min = getenv("EAS_C1_SHARED_MIN_CIN") ? atoi(getenv("EAS_C1_SHARED_MIN_CIN")) : 256;
    if (Cin >= shared_min && g.MT >= 4) {
        // wave tile of the shared form: shrink it while the grid has fewer than ~2 blocks per CU
        int sm = 4, sn = 2;
        static const char* force = getenv("EAS_C1_SHARED_SHAPE");     // development: "42", "41", "22", "21"
        if (force) { sm = force[0] - '0'; sn = force[1] - '0'; }
        else {
            constexpr long want_s = 384;       // swept on the dark4 / dark5 shapes (scripts/dev_conv.py c1time)
            if (blocks(sm, sn) < want_s) sn = 1;
            if (blocks(sm, sn) < want_s) sm = 2;
        }
#define EAS_C1S(XT_)                                                                                 \
    do {                                                                                             \
        if (sm == 4 && sn == 2) return launch_c1_shared<XT_, 4, 2, TIN>(x, wp, bias, y, g, st);      \
        if (sm == 4) return launch_c1_shared<XT_, 4, 1, TIN>(x, wp, bias, y, g, st);                 \
        if (sn == 2) return launch_c1_shared<XT_, 2, 2, TIN>(x, wp, bias, y, g, st);                 \
        return launch_c1_shared<XT_, 2, 1, TIN>(x, wp, bias, y, g, st);                              \
    } while (0)
        if (x_terms == 1) EAS_C1S(1);
        if constexpr (!U8) EAS_C1S(3);
#undef EAS_C1S
    }
#define EAS_C1(XT_)                                                                                  \
    do {                                                                                             \
        if (wm == 4 && wn == 2) return launch_c1<XT_, 4, 2, false, TIN>(x, wp, bias, y, g, st);      \
        if (wm == 2 && wn == 2) return launch_c1<XT_, 2, 2, false, TIN>(x, wp, bias, y, g, st);      \
        if (wm == 1 && wn == 2) return launch_c1<XT_, 1, 2, false, TIN>(x, wp, bias, y, g, st);      \
        if (wm == 4) return launch_c1<XT_, 4, 1, false, TIN>(x, wp, bias, y, g, st);                 \
        if (wm == 2) return launch_c1<XT_, 2, 1, false, TIN>(x, wp, bias, y, g, st);                 \
        return launch_c1<XT_, 1, 1, false, TIN>(x, wp, bias, y, g, st);                              \
    } while (0)
    if constexpr (U8) {
        static const bool staged = !(getenv("EAS_C1_U8_FORM") && getenv("EAS_C1_U8_FORM")[0] == 'b');     // development: "bytes" = per-lane byte loads
        if (staged && HW % 16 == 0) {
            if (wm == 4 && wn == 2) return launch_c1_u8<4, 2>(x, wp, bias, y, g, st);
            if (wm == 2 && wn == 2) return launch_c1_u8<2, 2>(x, wp, bias, y, g, st);
            if (wm == 1 && wn == 2) return launch_c1_u8<1, 2>(x, wp, bias, y, g, st);
            if (wm == 4) return launch_c1_u8<4, 1>(x, wp, bias, y, g, st);
            if (wm == 2) return launch_c1_u8<2, 1>(x, wp, bias, y, g, st);
            return launch_c1_u8<1, 1>(x, wp, bias, y, g, st);
        }
    }
    if (x_terms == 1) EAS_C1(1);
    if constexpr (!U8) EAS_C1(3);
#undef EAS_C1
    return EAS_ERR_UNSUPPORTED;
}

// called by eas_conv_fwd (conv_mfma.hip) for ksize == 1, stride == 1
int eas_conv1x1_dispatch(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                         hipStream_t st, double* stats, int stats_nb, int* nb_out) {
    tl_c1_blocks = 0;
    const int rc = conv1x1_dispatch_t<float>(x, packed_w, bias, y, NI, Cin, Cout, HW, x_terms, st, stats, stats_nb);
    if (nb_out) *nb_out = tl_c1_blocks;
    return rc;
}

// the same convolution reading spikes as bytes (eas_conv_fwd_u8)
int eas_conv1x1_dispatch_u8(const uint8_t* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, hipStream_t st) {
    return conv1x1_dispatch_t<uint8_t>(x, packed_w, bias, y, NI, Cin, Cout, HW, 1, st);
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

// LDS-staged form.  The direct form above hands the texture unit 64 different cache lines per load instruction (lanes =
// channels, 32 bytes each), which bounds it near 2 TB/s.  Here the block stages a [channels] x [KC pixels] slab of grad_y
// and x with coalesced 16-byte loads (consecutive lanes = consecutive pixels of one channel row), converts every element ONCE
// to its exact bf16 term(s) (the direct form converts x in each of its four waves) and the waves read MFMA fragments from
// LDS: lane (channel r, half h) reads 16 bytes at row r, pixels 8h..8h+7 of a k-step.  Row pitch = KC*2 + 16 bytes (an odd
// multiple of 16: the 16 lanes of a b128 pass cover all banks).  Block tile = (32*WVM) co x (32*WVN*NT) ci, 4 waves;
// double-buffered: the next chunk's global loads are issued before the current chunk's MFMAs and written to the other
// buffer after them (one barrier per chunk).
// XU8: x is given as spike bytes (uint8, XT = 1): an x staging item is then 4 pixels = ONE 32-bit load instead of 16 bytes.
template <int XT, int WVM, int WVN, int NT, int KS, bool XU8 = false>
__global__ __launch_bounds__(256) void conv1x1_wgrad_lds_kernel(const void* __restrict__ x_, const float* __restrict__ gy,
                                                                float* __restrict__ slabs, W1Geom g) {
    static_assert(!XU8 || XT == 1, "spike bytes are one exact bf16 term");
    const float* x = reinterpret_cast<const float*>(x_);
    const uint8_t* x8 = reinterpret_cast<const uint8_t*>(x_);
    constexpr int KC = 16 * KS;                 // pixels per chunk
    constexpr int PITCH = KC * 2 + 16;          // bytes per channel row and term
    constexpr int RA = 32 * WVM, RB = 32 * WVN * NT;
    constexpr int A_TERM = RA * PITCH, B_TERM = RB * PITCH;
    constexpr int BUF = 3 * A_TERM + XT * B_TERM;
    constexpr int V4R = KC / 4;                 // float4 loads per channel row and chunk
    constexpr int ITEMS = (RA + RB) * V4R;      // float4 staging items per chunk
    constexpr int NIT = (ITEMS + 255) / 256;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WVN, wn = wave % WVN;
    const int cob = blockIdx.y / g.ci_blocks, cib = blockIdx.y - cob * g.ci_blocks;
    const int co0 = cob * RA, ci0 = cib * RB;

    // staging items: item -> (row, float4 index in the row); rows [0, RA) grad_y channels, [RA, RA+RB) x channels
    int it_goff[NIT];        // element offset inside the image's tensor (channel * HW + 4 * v), -1: zero row
    int it_lofs[NIT];        // LDS byte offset of the first term
    bool it_isa[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int item = it * 256 + tid;
        item = item < ITEMS ? item : ITEMS - 1;
        const int row = item / V4R, v = item - row * V4R;
        const bool isa = row < RA;
        const int ch = isa ? co0 + row : ci0 + row - RA;
        const bool ok = isa ? ch < g.Cout : ch < g.Cin;
        it_isa[it] = isa;
        it_goff[it] = ok ? ch * g.HW + 4 * v : -1;
        it_lofs[it] = isa ? row * PITCH + 8 * v : 3 * A_TERM + (row - RA) * PITCH + 8 * v;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;

    const int chunks_img = g.HW / KC;
    const int total_chunks = g.NI * chunks_img;
    const int c_begin = blockIdx.x * g.per_slice;
    const int c_end = c_begin + g.per_slice < total_chunks ? c_begin + g.per_slice : total_chunks;

    f32x4 L[NIT];
    auto fetch = [&](int c) {
        const int img = c / chunks_img, p0 = (c - img * chunks_img) * KC;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it * 256 >= ITEMS) continue;
            if (XU8 && !it_isa[it]) {
                const uint8_t* src8 = it_goff[it] >= 0 ? x8 + (long)img * g.Cin * g.HW + it_goff[it] + p0 : reinterpret_cast<const uint8_t*>(eas_c1_zero_page);
                L[it][0] = __builtin_bit_cast(float, *(const uint32_t*)src8);
                continue;
            }
            const float* base = it_isa[it] ? gy + (long)img * g.Cout * g.HW : x + (long)img * g.Cin * g.HW;
            const float* src = it_goff[it] >= 0 ? base + it_goff[it] + p0 : eas_c1_zero_page;
            L[it] = *(const f32x4*)src;
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it * 256 >= ITEMS || it * 256 + tid >= ITEMS) continue;
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 t0, t1, t2;
            if (XU8 && !it_isa[it]) {
                const uint32_t w4 = __builtin_bit_cast(uint32_t, L[it][0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) t0[j] = (__bf16)(float)((w4 >> (8 * j)) & 0xffu);
                *(bf16x4*)(buf + it_lofs[it]) = t0;
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = L[it][j];
                const __bf16 hi = (__bf16)v;
                const float r1 = v - (float)hi;
                const __bf16 mid = (__bf16)r1;
                t0[j] = hi;
                t1[j] = mid;
                t2[j] = (__bf16)(r1 - (float)mid);
            }
            unsigned char* dst = buf + it_lofs[it];
            *(bf16x4*)dst = t0;
            if (it_isa[it]) {
                *(bf16x4*)(dst + A_TERM) = t1;
                *(bf16x4*)(dst + 2 * A_TERM) = t2;
            } else if (XT == 3) {
                *(bf16x4*)(dst + B_TERM) = t1;
                *(bf16x4*)(dst + 2 * B_TERM) = t2;
            }
        }
    };

    const int a_frag = (wm * 32 + r) * PITCH + 16 * h;                        // + term * A_TERM + ks * 32
    const int b_frag = 3 * A_TERM + (wn * NT * 32 + r) * PITCH + 16 * h;      // + n * 32 * PITCH + term * B_TERM + ks * 32

    int c = c_begin;
    if (c < c_end) {
        fetch(c);
        commit(smem);
    }
    __syncthreads();
    int par = 0;
    for (; c < c_end; ++c, par ^= 1) {
        const unsigned char* cur = smem + par * BUF;
        const bool more = c + 1 < c_end;
        if (more) fetch(c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = *(const bf16x8*)(cur + a_frag + t * A_TERM + ks * 32);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bf16x8 b[XT];
#pragma unroll
                for (int t = 0; t < XT; ++t) b[t] = *(const bf16x8*)(cur + b_frag + n * 32 * PITCH + t * B_TERM + ks * 32);
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
        }
        if (more) commit(smem + (par ^ 1) * BUF);
        __syncthreads();
    }

    // partial sums of this block -> slab [slice][co][ci]; D: col = ci = lane&31, row = co = (e&3) + 8*(e>>2) + 4*h
    float* slab = slabs + (size_t)blockIdx.x * g.Cout * g.Cin;
    const int row0 = co0 + wm * 32 + 4 * h;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ci = ci0 + (wn * NT + n) * 32 + r;
        if (ci >= g.Cin) continue;
        float* sp = slab + (size_t)row0 * g.Cin + ci;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (row0 + (e & 3) + 8 * (e >> 2) < g.Cout) sp[(size_t)((e & 3) + 8 * (e >> 2)) * g.Cin] = acc[n][e];
    }
}

struct W1Plan { int wvm, wvn, nt, ks, slices; };

// block shape for the LDS form: all four waves busy whatever Cout is
W1Plan w1_plan(W1Geom& g, int NI, int Cin, int Cout, int HW) {
    W1Plan p;
    if (Cout > 64) { p.wvm = 4; p.wvn = 1; p.nt = 4; }
    else if (Cout > 32) { p.wvm = 2; p.wvn = 2; p.nt = 2; }
    else { p.wvm = 1; p.wvn = 4; p.nt = 1; }
    p.ks = HW % 32 == 0 ? 2 : 1;
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.HW = HW;
    const int RA = 32 * p.wvm, RB = 32 * p.wvn * p.nt;
    g.ci_blocks = (Cin + RB - 1) / RB;
    const int yz = ((Cout + RA - 1) / RA) * g.ci_blocks;
    const int total_chunks = NI * (HW / (16 * p.ks));
    static const int target = getenv("EAS_W1_BLOCKS") ? atoi(getenv("EAS_W1_BLOCKS")) : 512;
    int slices = (target + yz - 1) / yz;
    if (slices > total_chunks) slices = total_chunks;
    g.per_slice = (total_chunks + slices - 1) / slices;
    p.slices = (total_chunks + g.per_slice - 1) / g.per_slice;
    return p;
}

template <int XT, int WVM, int WVN, int NT, int KS, bool XU8 = false>
int launch_w1_lds(const void* x, const float* gy, float* slabs, W1Geom g, int slices, hipStream_t st) {
    auto kern = conv1x1_wgrad_lds_kernel<XT, WVM, WVN, NT, KS, XU8>;
    constexpr int PITCH = 16 * KS * 2 + 16;
    const size_t lds = (size_t)2 * (3 * 32 * WVM + XT * 32 * WVN * NT) * PITCH;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(slices, ((g.Cout + 32 * WVM - 1) / (32 * WVM)) * g.ci_blocks);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, gy, slabs, g);
    return EAS_OK;
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

static bool w1_use_lds() {
    static const int v = getenv("EAS_W1_FORM") ? (getenv("EAS_W1_FORM")[0] == 'd' ? 0 : 1) : 1;   // development switch: "direct" / "lds"
    return v != 0;
}

int eas_conv1x1_wgrad_slices(int NI, int Cin, int Cout, int HW) {
    W1Geom g{};
    if (HW % 16 != 0) return 0;
    if (w1_use_lds()) return w1_plan(g, NI, Cin, Cout, HW).slices;
    if (!w1_geom(g, NI, Cin, Cout, HW)) return 0;
    return (g.total_ksteps + g.per_slice - 1) / g.per_slice;
}

// slabs: eas_conv1x1_wgrad_slices(...) * Cout * Cin floats; the caller reduces them (conv_wgrad_reduce_kernel)
// x as spike bytes (eas_conv_wgrad_u8): the LDS-staged form only
int eas_conv1x1_wgrad_dispatch_u8(const uint8_t* x, const float* gy, float* slabs, int NI, int Cin, int Cout, int HW, hipStream_t st) {
    W1Geom g{};
    if (HW % 16 != 0 || !w1_use_lds()) return EAS_ERR_UNSUPPORTED;
    const W1Plan p = w1_plan(g, NI, Cin, Cout, HW);
#define EAS_W1(M_, N_, T_) (p.ks == 2 ? launch_w1_lds<1, M_, N_, T_, 2, true>(x, gy, slabs, g, p.slices, st) : launch_w1_lds<1, M_, N_, T_, 1, true>(x, gy, slabs, g, p.slices, st))
    return p.wvm == 4 ? EAS_W1(4, 1, 4) : (p.wvm == 2 ? EAS_W1(2, 2, 2) : EAS_W1(1, 4, 1));
#undef EAS_W1
}

int eas_conv1x1_wgrad_dispatch(const float* x, const float* gy, float* slabs, int NI, int Cin, int Cout, int HW, int x_terms, hipStream_t st) {
    W1Geom g{};
    if (HW % 16 != 0) return EAS_ERR_UNSUPPORTED;
    if (w1_use_lds()) {
        const W1Plan p = w1_plan(g, NI, Cin, Cout, HW);
#define EAS_W1(XT_, M_, N_, T_) (p.ks == 2 ? launch_w1_lds<XT_, M_, N_, T_, 2>(x, gy, slabs, g, p.slices, st) : launch_w1_lds<XT_, M_, N_, T_, 1>(x, gy, slabs, g, p.slices, st))
#define EAS_W1_SHAPE(XT_) (p.wvm == 4 ? EAS_W1(XT_, 4, 1, 4) : (p.wvm == 2 ? EAS_W1(XT_, 2, 2, 2) : EAS_W1(XT_, 1, 4, 1)))
        return x_terms == 1 ? EAS_W1_SHAPE(1) : EAS_W1_SHAPE(3);
#undef EAS_W1_SHAPE
#undef EAS_W1
    }
    if (!w1_geom(g, NI, Cin, Cout, HW)) return EAS_ERR_UNSUPPORTED;
    const int slices = (g.total_ksteps + g.per_slice - 1) / g.per_slice;
    dim3 grid(slices, ((Cout + 127) / 128) * g.ci_blocks);
    if (x_terms == 1) hipLaunchKernelGGL(conv1x1_wgrad_kernel<1>, grid, dim3(256), 0, st, x, gy, slabs, g);
    else hipLaunchKernelGGL(conv1x1_wgrad_kernel<3>, grid, dim3(256), 0, st, x, gy, slabs, g);
    return EAS_OK;
}
