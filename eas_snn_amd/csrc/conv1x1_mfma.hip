// K4 (conv half): 1x1 convolutions (CSP conv1/conv2/conv3, bottleneck conv1, SPP, head stems; forward and input gradient)
// on the matrix cores, without LDS: a 1x1 convolution is y[co][p] = sum_ci w[co][ci] * x[ci][p] with no spatial reuse,
// so every wave streams its own pixels once.  For the MFMA B operand lane (pixel r, half h) needs 8 consecutive input
// channels of ONE pixel: 8 dword loads, each a 128-byte line per 32 lanes (pixels are contiguous in NCHW), converted in
// registers to exact bf16 term(s).  A = weights pre-packed in fragment order (conv_pack_weights_kernel, taps = 1), read
// from L1/L2.  HBM-bound (<= 64 flop/byte): x is read once per 32*WM output channels, y written once.
// (reference: nn.Conv2d(k=1) inside BaseConv, yolox/models/network_blocks.py:31-56.)
#include <stdlib.h>

#include "eas_common.h"
#include "conv_lif_epi.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// loads of lanes without a pixel / channel groups past Cin read these zeros (stride 0) instead of being masked
__device__ __attribute__((aligned(32))) float eas_c1_zero_page[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// pixel blocks (grid.x) of the last launch / geometry query of this thread: the statistics partials per channel
thread_local int tl_c1_blocks = 0;

// C1GeomCore: what the plain tiles read (the grouped launch carries up to twelve by value); C1Geom adds the fused eval epilogues' descriptors
struct C1GeomCore {
    int NI, Cin, Cout, HW;
    int tiles_per_img, total_tiles;   // 32-pixel tiles
    int MT, KSTEPS;
    double* stats;                    // BatchNorm statistics in the epilogue (eas_conv_fwd_stats): [Cout][stats_nb][2], stats_nb = gridDim.x
    int stats_nb;
    int* inexact;                     // x_terms == 1 on fp32 input: OR-ed with 1 when a value is not exact in bf16 (NULL: not reported)
    int act;                          // epilogue activation on (acc + bias): 0 none, 1 SiLU (eas_conv_fwd_act)
    int accum;                        // y += conv(x) + bias instead of y = (grouped launches: the second reader's input gradient, EasConvProblem.accumulate)
};
struct C1GeomBna : C1GeomCore {      // + the BatchNorm / activation epilogue (LM = 3) of a grouped eas_conv_bn_act_eval
    EasBnActEpiDev bna;
};
struct C1Geom : C1GeomCore {
    // fused eval step conv -> BatchNorm (running statistics) -> LIF over T (kernel template LM = 1, conv_lif_epi.h): NI = samples N, total_tiles
    // counts SPATIAL 32-pixel tiles, pixel tile n of a wave = time step n (WN == lif.T): image n * N + sample
    EasLifEpiDev lif;
    EasBnActEpiDev bna;               // BatchNorm (running statistics) + activation in the plain epilogue (eas_conv_bn_act_eval); on = 0: plain store
};

// statistics epilogue of the 1x1 kernels: 4 waves of a block = 4 pixel groups of the same WM * 32 channels
template <int WM, int WN, typename ACC, typename G>
__device__ __forceinline__ void c1_stats(const ACC (&acc)[WM][WN], const long (&yoff)[WN], double* red, const G& g, int mt0, int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool valid[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) valid[n] = yoff[n] >= 0;
    eas_conv_stats_wave<WM, WN>(acc, valid, reinterpret_cast<float*>(red + 4 * WM * 64) + wave * EAS_STATS_SCRATCH, red + wave * (WM * 64), lane);
    __syncthreads();
    if ((int)threadIdx.x < WM * 32)
        eas_conv_stats_store(red, WM * 64, 4, (int)threadIdx.x, mt0 * 32 + (int)threadIdx.x, g.Cout, g.stats, g.stats_nb, bx);
}

template <int XT, typename TIN = float>
__device__ __forceinline__ void to_terms(const TIN (&v)[8], bf16x8 (&b)[XT]) {
    if constexpr (XT == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b[0][j] = (__bf16)(float)v[j];
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
// PL: x is given as spike planes (bf16 [NI][Cin/8][HW][8], one exact term): a lane's 8 channels of its pixel are ONE 16-byte load and
// already the MFMA operand -- no conversion, an eighth of the load instructions, half the bytes.
// bx / by: the block's pixel-block and channel-group index (blockIdx.x / .y of a launch of one problem; a grouped launch maps its
// flat grid onto (problem, block) first -- conv1x1_group_kernel)
template <int XT, int WM, int WN, bool RAGK, bool PL, int LM, typename G>
__device__ __forceinline__ void c1_body(const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias,
                                        float* __restrict__ y, const G& g, const int bx, const int by) {
    static_assert(!PL || (XT == 1 && !RAGK), "spike planes are one exact bf16 term in whole 8-channel groups");
    static_assert(LM == 0 || LM == 1 || LM == 3, "1x1: plain, time-major fused neuron epilogue, or BatchNorm + activation epilogue");
    constexpr bool TM = LM == 1;
    typedef float TIN;
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = by * WM;
    const int tile0 = (bx * 4 + wave) * (TM ? 1 : WN);
    if (tile0 >= g.total_tiles && !g.stats) return;      // with statistics every wave reaches the block reduction (its lanes hold no pixel)

    // per-lane pixel of each N-tile
    long xoff[WN];    // element offset of (img, channel 8h, pixel) ; -1: no pixel
    long yoff[WN];
    int l_img[LM == 1 ? WN : 1], l_pix[LM == 1 ? WN : 1];
    bool l_ok[LM == 1 ? WN : 1];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int t = TM ? tile0 : tile0 + n;             // time-major: tile n = time step n of the same spatial tile
        const int smp = t / g.tiles_per_img, p = (t - smp * g.tiles_per_img) * 32 + r;
        const int img = TM ? n * g.NI + smp : smp;
        const bool ok = t < g.total_tiles && p < g.HW;
        xoff[n] = ok ? (PL ? ((long)img * (g.Cin / 8) + h) * g.HW + p : ((long)img * g.Cin + 8 * h) * g.HW + p) : -1;
        yoff[n] = ok ? (long)img * g.Cout * g.HW + p : -1;
        if constexpr (LM == 1) {
            l_img[n] = smp;
            l_pix[n] = p;
            l_ok[n] = ok;
        }
    }

    f32x16 acc[WM][WN];
    unsigned bad = 0;      // XT == 1 on fp32 input: OR of the bit patterns read (tag check, see the end of the channel loop)
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

    if constexpr (PL) {
        // planes: group 2 * ks + h of this lane's pixel per N-tile; lanes without a pixel / groups past Cin read the zero page
        auto pfetch = [&](bf16x8 (&raw)[WN], int ks) {
            const bool ch_ok = ks * 16 + 8 * h < g.Cin;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const bool ok = ch_ok && xoff[n] >= 0;
                raw[n] = *(ok ? xp + xoff[n] + (long)ks * 2 * g.HW : reinterpret_cast<const bf16x8*>(eas_c1_zero_page));
            }
        };
        auto pstep = [&](const bf16x8 (&raw)[WN], int ks) {
            bf16x8 a[WM][3];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int t = 0; t < 3; ++t) a[m][t] = ap[m][t * a_term + (size_t)ks * 64];
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][2], raw[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], raw[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], raw[n], acc[m][n], 0, 0, 0);
                }
        };
        bf16x8 p0[WN], p1[WN];
        pfetch(p0, 0);
        int ks = 0;
        for (; ks + 1 < g.KSTEPS; ks += 2) {
            pfetch(p1, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            pstep(p0, ks);
            pfetch(p0, ks + 2 < g.KSTEPS ? ks + 2 : ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            pstep(p1, ks + 1);
        }
        if (ks < g.KSTEPS) pstep(p0, ks);
    } else {
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
            if constexpr (XT == 1 && !PL) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bad |= __float_as_uint(raw[n][j]);       // exact in bf16 <=> low 16 pattern bits zero
            }
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
    }

    if constexpr (XT == 1 && !PL) {
        if (g.inexact && (bad & 0xffffu)) atomicOr(g.inexact, 1);      // a tensor tagged "spikes / small integers" was not
    }

    if (g.stats) {
        __shared__ __align__(16) double red[4 * WM * 64 + 2 * EAS_STATS_SCRATCH];      // [wave][WM * 32][2] doubles, then the waves' float patches
        c1_stats<WM, WN>(acc, yoff, red, g, mt0, bx);
    }

    if constexpr (LM == 1) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
            if (mt0 + m < g.MT) eas_lif_epilogue<WN, LM>(acc[m], g.lif, mt0 + m, h, l_img, l_pix, l_ok);
        return;
    }

    if constexpr (LM == 3) {      // BatchNorm (running statistics) + activation on the accumulators (eas_conv_bn_act_eval)
#pragma unroll
        for (int m = 0; m < WM; ++m)
            if (mt0 + m < g.MT) eas_bnact_epilogue<WN>(acc[m], g.bna, mt0 + m, h, yoff, (long)g.HW);
        return;
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
            if (g.accum) {        // block-uniform: y += (the second reader of a tensor adds its input gradient to the first one's)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[(long)((e & 3) + 8 * (e >> 2)) * g.HW] += acc[m][n][e] + bv[e];
                continue;
            }
            if (g.act) {          // block-uniform: the activation's transcendental never runs for the plain convolution
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[(long)((e & 3) + 8 * (e >> 2)) * g.HW] = eas_epi_silu(acc[m][n][e] + bv[e]);
            } else if (full) {
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

template <int XT, int WM, int WN, bool RAGK = false, bool PL = false, int LM = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_kernel(const float* __restrict__ x, const bf16x8* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    c1_body<XT, WM, WN, RAGK, PL, LM, C1Geom>(x, wp, bias, y, g, (int)blockIdx.x, (int)blockIdx.y);
}

// Grouped launch (include/eas_hip.h eas_conv_fwd_group, ksize 1): problem p owns pixel blocks [first[p], first[p + 1]) of grid.x
constexpr int kMaxGroup1 = 12;
template <typename G>
struct C1GroupArgsT {
    G g[kMaxGroup1];
    const float* x[kMaxGroup1];
    const bf16x8* wp[kMaxGroup1];
    const float* bias[kMaxGroup1];
    float* y[kMaxGroup1];
    int first[kMaxGroup1 + 1];
    int n;
};
typedef C1GroupArgsT<C1GeomCore> C1GroupArgs;

template <int XT, int WM, int WN, bool RAGK, int LM = 0, typename G = C1GeomCore>
__global__ __launch_bounds__(256, 2) void conv1x1_group_kernel(const C1GroupArgsT<G> a) {
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    if ((int)blockIdx.y * WM >= a.g[p].MT) return;        // grid.y covers the widest problem
    c1_body<XT, WM, WN, RAGK, false, LM, G>(a.x[p], a.wp[p], a.bias[p], a.y[p], a.g[p], (int)blockIdx.x - a.first[p], (int)blockIdx.y);
}

// Variant for layers with many input channels and few pixels (dark5: 256..1024 channels on 15 360 pixels).  There the direct
// kernel is bound by the texture path, not by HBM or the matrix cores: every wave fetches its own copy of the weight fragments
// (WM x 3 KB per k-step) -- 64 KB per k-step and CU against 768 cycles of MFMA work.  Here the four waves of a block (same
// output channels, different pixels) share them: the block loads each k-step's WM x 3 fragments once (coalesced 16-byte
// loads), double-buffers them in LDS and every wave reads its operands with conflict-free ds_read_b128.  One barrier per k-step.
template <int XT, int WM, int WN, bool PL = false, int LM = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_sharedA_kernel(const float* __restrict__ x, const bf16x8* __restrict__ wp,
                                                                      const float* __restrict__ bias, float* __restrict__ y, C1Geom g) {
    static_assert(!PL || XT == 1, "spike planes are one exact bf16 term");
    static_assert(LM == 0 || LM == 1 || LM == 3, "1x1: plain, time-major fused neuron epilogue, or BatchNorm + activation epilogue");
    constexpr bool TM = LM == 1;
    typedef float TIN;
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(x);
    constexpr int NFRAG = WM * 3;                        // 1 KB fragments per k-step
    constexpr int NLD = (NFRAG * 64 + 255) / 256;        // 16-byte staging loads per thread and k-step
    __shared__ bf16x8 As[2][NFRAG][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mt0 = blockIdx.y * WM;
    const int tile0 = (blockIdx.x * 4 + wave) * (TM ? 1 : WN);

    long xoff[WN], yoff[WN];
    int l_img[LM == 1 ? WN : 1], l_pix[LM == 1 ? WN : 1];
    bool l_ok[LM == 1 ? WN : 1];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int t = TM ? tile0 : tile0 + n;             // time-major: tile n = time step n of the same spatial tile
        const int smp = t / g.tiles_per_img, p = (t - smp * g.tiles_per_img) * 32 + r;
        const int img = TM ? n * g.NI + smp : smp;
        const bool ok = t < g.total_tiles && p < g.HW;
        xoff[n] = ok ? (PL ? ((long)img * (g.Cin / 8) + h) * g.HW + p : ((long)img * g.Cin + 8 * h) * g.HW + p) : -1;
        yoff[n] = ok ? (long)img * g.Cout * g.HW + p : -1;
        if constexpr (LM == 1) {
            l_img[n] = smp;
            l_pix[n] = p;
            l_ok[n] = ok;
        }
    }
    f32x16 acc[WM][WN];
    unsigned bad = 0;
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
    // two register sets: the fragments of k-step s + 2 are requested while step s multiplies and are written to LDS a whole step later --
    // with one set the commit waited for the load issued just before the step's few MFMAs (WM * WN * 3 or 6 of them: 200-800 cycles
    // against an L2 / HBM round trip), i.e. every k-step of these many-channel, few-pixel layers ended in an exposed load latency
    bf16x8 areg0[NLD], areg1[NLD];
    auto a_fetch = [&](bf16x8 (&areg)[NLD], int ks) {
        const int kc = ks < g.KSTEPS ? ks : g.KSTEPS - 1;
#pragma unroll
        for (int it = 0; it < NLD; ++it) areg[it] = asrc[it][(size_t)kc * 64];
    };
    auto a_commit = [&](const bf16x8 (&areg)[NLD], int buf) {
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
        for (int n = 0; n < WN; ++n) {
            if constexpr (XT == 1 && !PL) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bad |= __float_as_uint(raw[n][j]);       // exact in bf16 <=> low 16 pattern bits zero
            }
            to_terms<XT, TIN>(raw[n], b[n]);
        }
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

    if constexpr (PL) {
        auto p_fetch = [&](bf16x8 (&raw)[WN], int ks) {
            const bool ch_ok = ks * 16 + 8 * h < g.Cin;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const bool ok = ch_ok && xoff[n] >= 0;
                raw[n] = *(ok ? xp + xoff[n] + (long)ks * 2 * g.HW : reinterpret_cast<const bf16x8*>(eas_c1_zero_page));
            }
        };
        auto p_step = [&](const bf16x8 (&raw)[WN], int buf) {
            bf16x8 a[WM][3];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int t = 0; t < 3; ++t) a[m][t] = As[buf][t * WM + m][lane];
#pragma unroll
            for (int ta = 2; ta >= 0; --ta)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][ta], raw[n], acc[m][n], 0, 0, 0);
        };
        bf16x8 q0[WN], q1[WN];
        a_fetch(areg0, 0);
        p_fetch(q0, 0);
        a_commit(areg0, 0);
        a_fetch(areg1, 1);
        __syncthreads();
        int ks = 0;
        for (; ks + 1 < g.KSTEPS; ks += 2) {
            a_fetch(areg0, ks + 2);
            p_fetch(q1, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            p_step(q0, 0);
            a_commit(areg1, 1);                          // step ks + 1 (requested a whole step ago)
            __syncthreads();
            const int kn = ks + 2 < g.KSTEPS ? ks + 2 : ks + 1;
            a_fetch(areg1, ks + 3);
            p_fetch(q0, kn);
            __builtin_amdgcn_sched_barrier(0);
            p_step(q1, 1);
            a_commit(areg0, 0);                          // step ks + 2
            __syncthreads();
        }
        if (ks < g.KSTEPS) p_step(q0, 0);
    } else {
    TIN r0[WN][8], r1[WN][8];
    a_fetch(areg0, 0);
    x_fetch(r0, 0);
    a_commit(areg0, 0);
    a_fetch(areg1, 1);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < g.KSTEPS; ks += 2) {
        a_fetch(areg0, ks + 2);
        x_fetch(r1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        step(r0, 0);
        a_commit(areg1, 1);                              // step ks + 1 (requested a whole step ago)
        __syncthreads();
        const int kn = ks + 2 < g.KSTEPS ? ks + 2 : ks + 1;
        a_fetch(areg1, ks + 3);
        x_fetch(r0, kn);
        __builtin_amdgcn_sched_barrier(0);
        step(r1, 1);
        a_commit(areg0, 0);                              // step ks + 2
        __syncthreads();
    }
    if (ks < g.KSTEPS) step(r0, 0);
    }

    if constexpr (XT == 1 && !PL) {
        if (g.inexact && (bad & 0xffffu)) atomicOr(g.inexact, 1);      // a tensor tagged "spikes / small integers" was not
    }

    if (g.stats) {
        __shared__ __align__(16) double red[4 * WM * 64 + 2 * EAS_STATS_SCRATCH];      // [wave][WM * 32][2] doubles, then the waves' float patches
        c1_stats<WM, WN>(acc, yoff, red, g, mt0, (int)blockIdx.x);
    }

    if constexpr (LM == 1) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
            if (mt0 + m < g.MT) eas_lif_epilogue<WN, LM>(acc[m], g.lif, mt0 + m, h, l_img, l_pix, l_ok);
        return;
    }

    if constexpr (LM == 3) {      // BatchNorm (running statistics) + activation on the accumulators (eas_conv_bn_act_eval)
#pragma unroll
        for (int m = 0; m < WM; ++m)
            if (mt0 + m < g.MT) eas_bnact_epilogue<WN>(acc[m], g.bna, mt0 + m, h, yoff, (long)g.HW);
        return;
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
            if (g.act) {          // block-uniform: the activation's transcendental never runs for the plain convolution
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[(long)((e & 3) + 8 * (e >> 2)) * g.HW] = eas_epi_silu(acc[m][n][e] + bv[e]);
            } else if (full) {
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

template <int XT, int WM, int WN, bool PL = false, int LM = 0>
int launch_c1_shared(const float* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    tl_c1_blocks = (int)grid.x;
    if (!y) return EAS_OK;            // geometry query
    if (g.stats && (int)grid.x != g.stats_nb) return EAS_ERR_INVALID_ARG;
    EAS_LAUNCH((conv1x1_mfma_sharedA_kernel<XT, WM, WN, PL, LM>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

template <int XT, int WM, int WN, bool RAGK = false, bool PL = false, int LM = 0>
int launch_c1(const float* x, const bf16x8* wp, const float* bias, float* y, C1Geom g, hipStream_t st) {
    dim3 grid((g.total_tiles + 4 * WN - 1) / (4 * WN), (g.MT + WM - 1) / WM);
    tl_c1_blocks = (int)grid.x;
    if (!y) return EAS_OK;            // geometry query
    if (g.stats && (int)grid.x != g.stats_nb) return EAS_ERR_INVALID_ARG;
    EAS_LAUNCH((conv1x1_mfma_kernel<XT, WM, WN, RAGK, PL, LM>), grid, dim3(256), 0, st, x, wp, bias, y, g);
    return EAS_OK;
}

// fused eval step on spike planes, time-major: a block's four waves = four SPATIAL 32-pixel tiles, each wave all T = WN time steps of its tile
template <int WM, int WN, bool SHARED>
int launch_c1_lif(const float* x, const bf16x8* wp, C1Geom g, hipStream_t st, bool query) {
    if (query) return EAS_OK;
    dim3 grid((g.total_tiles + 3) / 4, (g.MT + WM - 1) / WM);
    if constexpr (SHARED) EAS_LAUNCH((conv1x1_mfma_sharedA_kernel<1, WM, WN, true, 1>), grid, dim3(256), 0, st, x, wp, (const float*)nullptr, (float*)nullptr, g);
    else EAS_LAUNCH((conv1x1_mfma_kernel<1, WM, WN, false, true, 1>), grid, dim3(256), 0, st, x, wp, (const float*)nullptr, (float*)nullptr, g);
    return EAS_OK;
}

}  // namespace

// wave-tile choice and launch for a 1x1 convolution; PL: x is a spike-plane tensor (x_terms 1)
template <bool PL>
static int conv1x1_dispatch_t(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                              hipStream_t st, double* stats = nullptr, int stats_nb = 0, int* inexact = nullptr, int act = 0,
                              const EasBnActEpiDev* bna = nullptr) {
    C1Geom g{};
    g.stats = stats; g.stats_nb = stats_nb;
    g.inexact = inexact;
    g.act = act;
    if (bna) g.bna = *bna;
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
    static const long want = eas_dev_env("EAS_C1_BLOCKS") ? atol(eas_dev_env("EAS_C1_BLOCKS")) : 512;
    if (blocks(wm, wn) < want) wn = 1;
    while (blocks(wm, wn) < want && wm > 1) wm >>= 1;
    if (bna && (PL || x_terms != 3 || Cin % 8 != 0)) return EAS_ERR_UNSUPPORTED;     // the BatchNorm + activation epilogue: real-valued fp32 inputs
    if constexpr (PL) {
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
    static const int shared_min = eas_dev_env("EAS_C1_SHARED_MIN_CIN") ? atoi(eas_dev_env("EAS_C1_SHARED_MIN_CIN")) : 256;
    if (Cin >= shared_min && g.MT >= 4) {
        // wave tile of the shared form: shrink it while the grid has fewer than ~2 blocks per CU
        int sm = 4, sn = 2;
        static const char* force = eas_dev_env("EAS_C1_SHARED_SHAPE");     // development: "42", "41", "22", "21"
        if (force) { sm = force[0] - '0'; sn = force[1] - '0'; }
        else {
            constexpr long want_s = 384;       // swept on the dark4 / dark5 shapes (scripts/dev_conv.py c1time)
            if (blocks(sm, sn) < want_s) sn = 1;
            if (blocks(sm, sn) < want_s) sm = 2;
        }
#define EAS_C1S(XT_)                                                                                \
    do {                                                                                            \
        if (sm == 4 && sn == 2) return launch_c1_shared<XT_, 4, 2, PL>(x, wp, bias, y, g, st);      \
        if (sm == 4) return launch_c1_shared<XT_, 4, 1, PL>(x, wp, bias, y, g, st);                 \
        if (sn == 2) return launch_c1_shared<XT_, 2, 2, PL>(x, wp, bias, y, g, st);                 \
        return launch_c1_shared<XT_, 2, 1, PL>(x, wp, bias, y, g, st);                              \
    } while (0)
        if constexpr (!PL) {
            if (bna) {
                if (sm == 4 && sn == 2) return launch_c1_shared<3, 4, 2, false, 3>(x, wp, bias, y, g, st);
                if (sm == 4) return launch_c1_shared<3, 4, 1, false, 3>(x, wp, bias, y, g, st);
                if (sn == 2) return launch_c1_shared<3, 2, 2, false, 3>(x, wp, bias, y, g, st);
                return launch_c1_shared<3, 2, 1, false, 3>(x, wp, bias, y, g, st);
            }
        }
        if (x_terms == 1) EAS_C1S(1);
        if constexpr (!PL) EAS_C1S(3);
#undef EAS_C1S
    }
#define EAS_C1(XT_)                                                                                 \
    do {                                                                                            \
        if (wm == 4 && wn == 2) return launch_c1<XT_, 4, 2, false, PL>(x, wp, bias, y, g, st);      \
        if (wm == 2 && wn == 2) return launch_c1<XT_, 2, 2, false, PL>(x, wp, bias, y, g, st);      \
        if (wm == 1 && wn == 2) return launch_c1<XT_, 1, 2, false, PL>(x, wp, bias, y, g, st);      \
        if (wm == 4) return launch_c1<XT_, 4, 1, false, PL>(x, wp, bias, y, g, st);                 \
        if (wm == 2) return launch_c1<XT_, 2, 1, false, PL>(x, wp, bias, y, g, st);                 \
        return launch_c1<XT_, 1, 1, false, PL>(x, wp, bias, y, g, st);                              \
    } while (0)
    if constexpr (!PL) {
        if (bna) {
            if (wm == 4 && wn == 2) return launch_c1<3, 4, 2, false, false, 3>(x, wp, bias, y, g, st);
            if (wm == 2 && wn == 2) return launch_c1<3, 2, 2, false, false, 3>(x, wp, bias, y, g, st);
            if (wm == 1 && wn == 2) return launch_c1<3, 1, 2, false, false, 3>(x, wp, bias, y, g, st);
            if (wm == 4) return launch_c1<3, 4, 1, false, false, 3>(x, wp, bias, y, g, st);
            if (wm == 2) return launch_c1<3, 2, 1, false, false, 3>(x, wp, bias, y, g, st);
            return launch_c1<3, 1, 1, false, false, 3>(x, wp, bias, y, g, st);
        }
    }
    if (x_terms == 1) EAS_C1(1);
    if constexpr (!PL) EAS_C1(3);
#undef EAS_C1
    return EAS_ERR_UNSUPPORTED;
}

// called by eas_conv_fwd (conv_mfma.hip) for ksize == 1, stride == 1; planes != 0: x is a spike-plane tensor
int eas_conv1x1_dispatch(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                         hipStream_t st, double* stats, int stats_nb, int* nb_out, int planes, int* inexact, int act, const void* bna) {
    tl_c1_blocks = 0;
    const EasBnActEpiDev* b = (const EasBnActEpiDev*)bna;
    const int rc = planes ? conv1x1_dispatch_t<true>(x, packed_w, bias, y, NI, Cin, Cout, HW, 1, st, stats, stats_nb, nullptr, act, b)
                          : conv1x1_dispatch_t<false>(x, packed_w, bias, y, NI, Cin, Cout, HW, x_terms, st, stats, stats_nb, inexact, act, b);
    if (nb_out) *nb_out = tl_c1_blocks;
    return rc;
}

// 1x1 layer of the fused eval step (eas_conv_bn_lif_eval, conv_mfma.hip): spike planes in, distinct frames per step, T = 3 or 5
int eas_conv1x1_lif_dispatch(const void* x, const void* packed_w, const EasLifEpiDev& lif, int Cin, int x_terms, int x_shared, hipStream_t st, bool query) {
    if (x_terms != 2 || x_shared || Cin % 8 != 0 || (lif.T != 3 && lif.T != 5)) return EAS_ERR_UNSUPPORTED;
    C1Geom g{};
    g.NI = lif.N; g.Cin = Cin; g.Cout = lif.Cout; g.HW = lif.HW;
    g.tiles_per_img = (lif.HW + 31) / 32;
    g.total_tiles = lif.N * g.tiles_per_img;            // spatial tiles
    g.MT = (lif.Cout + 31) / 32;
    g.KSTEPS = (Cin + 15) / 16;
    g.lif = lif;
    const bf16x8* wp = (const bf16x8*)packed_w;
    const float* xf = (const float*)x;
    auto blocks = [&](int m) { return (long)((g.total_tiles + 3) / 4) * ((g.MT + m - 1) / m); };
    // channel tiles per wave: two (T = 3: 6 accumulator tiles, T = 5: 10; four M-tiles x 3 steps spill) while the grid keeps ~2 blocks per CU
    int wm = g.MT >= 2 ? 2 : 1;
    while (wm > 1 && blocks(wm) < 512) wm >>= 1;
    static const int shared_min = eas_dev_env("EAS_C1_SHARED_MIN_CIN") ? atoi(eas_dev_env("EAS_C1_SHARED_MIN_CIN")) : 256;
    if (Cin >= shared_min && g.MT >= 2 && wm >= 2)
        return lif.T == 3 ? launch_c1_lif<2, 3, true>(xf, wp, g, st, query) : launch_c1_lif<2, 5, true>(xf, wp, g, st, query);
    if (lif.T == 3) return wm == 2 ? launch_c1_lif<2, 3, false>(xf, wp, g, st, query) : launch_c1_lif<1, 3, false>(xf, wp, g, st, query);
    return wm == 2 ? launch_c1_lif<2, 5, false>(xf, wp, g, st, query) : launch_c1_lif<1, 5, false>(xf, wp, g, st, query);
}

// Grouped 1x1 launch (eas_conv_fwd_group, conv_group.hip): one wave-tile shape for all problems -- channel tiles per wave from the
// narrowest problem (a wave of a wider shape would multiply clamped duplicates), pixel tiles per wave from the group's total block count
// (the ~2 blocks per CU rule of the single launch).  The direct kernel only (the block-shared weight fragments of the >= 256-channel
// layers are a different block shape; in a group those layers are the small riders).
namespace {
template <int XT, int WM, int WN, bool RAGK>
int launch_c1_group(const C1GroupArgs& a, int grid_y, hipStream_t st) {
    EAS_LAUNCH((conv1x1_group_kernel<XT, WM, WN, RAGK>), dim3(a.first[a.n], grid_y), dim3(256), 0, st, a);
    return EAS_OK;
}
// the same launch with the BatchNorm / activation epilogue (grouped eas_conv_bn_act_eval)
template <int WM, int WN>
int launch_c1_group_bna(const C1GroupArgs& a, const EasBnActEpiDev* bna, const int* order, int grid_y, hipStream_t st) {
    C1GroupArgsT<C1GeomBna> b{};
    for (int i = 0; i < a.n; ++i) {
        static_cast<C1GeomCore&>(b.g[i]) = a.g[i];
        b.g[i].bna = bna[order[i]];
        b.x[i] = a.x[i]; b.wp[i] = a.wp[i]; b.bias[i] = nullptr; b.y[i] = a.y[i];
    }
    for (int i = 0; i <= a.n; ++i) b.first[i] = a.first[i];
    b.n = a.n;
    EAS_LAUNCH((conv1x1_group_kernel<3, WM, WN, false, 3, C1GeomBna>), dim3(b.first[b.n], grid_y), dim3(256), 0, st, b);
    return EAS_OK;
}
}  // namespace

int eas_conv1x1_group(const EasConvProblem* pr, int n, int x_terms, hipStream_t st, int* nb_out, bool query, const void* bna_) {
    const EasBnActEpiDev* bna = (const EasBnActEpiDev*)bna_;
    if (n > kMaxGroup1 || x_terms != 3) return EAS_ERR_UNSUPPORTED;
    C1GroupArgs a{};
    bool ragk = false;
    int mt_min = 1 << 30, mt_max = 0;
    for (int p = 0; p < n; ++p) {
        C1GeomCore& g = a.g[p];
        const EasConvProblem& q = pr[p];
        g.NI = q.NI; g.Cin = q.Cin; g.Cout = q.Cout; g.HW = q.Hi * q.Wi;
        g.tiles_per_img = (g.HW + 31) / 32;
        g.total_tiles = q.NI * g.tiles_per_img;
        g.MT = (q.Cout + 31) / 32;
        g.KSTEPS = (q.Cin + 15) / 16;
        g.stats = q.stats;
        g.accum = q.accumulate ? 1 : 0;
        if (q.stats && q.accumulate) return EAS_ERR_INVALID_ARG;
        ragk = ragk || q.Cin % 8 != 0;
        mt_min = g.MT < mt_min ? g.MT : mt_min;
        mt_max = g.MT > mt_max ? g.MT : mt_max;
        a.x[p] = (const float*)q.x; a.wp[p] = (const bf16x8*)q.packed_w; a.bias[p] = q.bias; a.y[p] = q.y;
    }
    auto blocks = [&](int m, int w) {
        long b = 0;
        for (int p = 0; p < n; ++p) b += (long)((a.g[p].total_tiles + 4 * w - 1) / (4 * w)) * ((a.g[p].MT + m - 1) / m);
        return b;
    };
    int wm = mt_min >= 4 ? 4 : (mt_min >= 2 ? 2 : 1), wn = ragk ? 1 : 2;
    if (blocks(wm, wn) < 512) wn = 1;
    while (blocks(wm, wn) < 512 && wm > 1) wm >>= 1;
    if (wm == 2 && wn == 2) wn = 1;          // instantiated shapes: (4,2) (4,1) (2,1) (1,1); ragged input channels: (4,1) (1,1)
    if (wm == 1) wn = 1;
    if (ragk && wm == 2) wm = 1;
    // Grid order: the problems with the longest channel loops first.  Blocks start in blockIdx order and a block's duration goes with its
    // k-steps (the 512-channel stem of the 8x10 level: 32 dependent steps against 8 on the 32x40 level); at the END of the grid those few
    // long blocks were the launch's tail (stems of the three levels: 121 us against 49 us for the large level alone).
    int order[kMaxGroup1];
    for (int p = 0; p < n; ++p) order[p] = p;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && a.g[order[j]].KSTEPS > a.g[order[j - 1]].KSTEPS; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    C1GroupArgs b = a;
    int bx = 0;
    for (int i = 0; i < n; ++i) {
        const int p = order[i];
        b.g[i] = a.g[p]; b.x[i] = a.x[p]; b.wp[i] = a.wp[p]; b.bias[i] = a.bias[p]; b.y[i] = a.y[p];
        b.first[i] = bx;
        const int gxp = (b.g[i].total_tiles + 4 * wn - 1) / (4 * wn);
        b.g[i].stats_nb = gxp;
        if (nb_out) nb_out[p] = gxp;
        bx += gxp;
    }
    b.first[n] = bx;
    b.n = n;
    a = b;
    if (query) return EAS_OK;
    const int gy = (mt_max + wm - 1) / wm;
    if (bna) {
        if (ragk) return EAS_ERR_UNSUPPORTED;
        if (wm == 4) return wn == 2 ? launch_c1_group_bna<4, 2>(a, bna, order, gy, st) : launch_c1_group_bna<4, 1>(a, bna, order, gy, st);
        return wm == 2 ? launch_c1_group_bna<2, 1>(a, bna, order, gy, st) : launch_c1_group_bna<1, 1>(a, bna, order, gy, st);
    }
    if (ragk) return wm == 4 ? launch_c1_group<3, 4, 1, true>(a, gy, st) : launch_c1_group<3, 1, 1, true>(a, gy, st);
    if (wm == 4) return wn == 2 ? launch_c1_group<3, 4, 2, false>(a, gy, st) : launch_c1_group<3, 4, 1, false>(a, gy, st);
    return wm == 2 ? launch_c1_group<3, 2, 1, false>(a, gy, st) : launch_c1_group<3, 1, 1, false>(a, gy, st);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of a 1x1 convolution: dW[co][ci] = sum_{n,p} gy[n][co][p] * x[n][ci][p]   (reduction index = pixel).
// The block stages a [channels] x [KC pixels] slab of grad_y and x with coalesced 16-byte loads (consecutive lanes = consecutive
// pixels of one channel row), converts every element ONCE to its exact bf16 term(s) and the waves read MFMA fragments from LDS:
// lane (channel r, half h) reads 16 bytes at row r, pixels 8h..8h+7 of a k-step.  Row pitch = KC*2 + 16 bytes (an odd multiple of
// 16: the 16 lanes of a b128 pass cover all banks).  Block tile = (32*WVM) co x (32*WVN*NT) ci, 4 waves; double-buffered: the next
// chunk's global loads are issued before the current chunk's MFMAs and written to the other buffer after them (one barrier per
// chunk).  Pixel ranges are split over blockIdx.x and reduced through slabs in fixed order.
// (A direct form -- lanes = channels, 32 bytes each, no LDS -- handed the texture unit 64 cache lines per load instruction and was
// bound near 2 TB/s: removed.)
// XPL: x is a spike-plane tensor (bf16 [NI][Cin/8][HW][8], XT = 1).  A staging item is then one 16-byte load = the 8 channels of a
// pixel, copied as it is into a pixel-major LDS image ([32-channel plane][pixel][64 B], the layout of the 3x3 weight-gradient
// kernel), and a lane's 8 consecutive pixels of one channel come from two ds_read_b64_tr_b16 (hardware transpose): no conversion.
namespace {

struct W1Geom {
    int NI, Cin, Cout, HW;
    int ksteps_img, total_ksteps, per_slice;
    int ci_blocks;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4w lds_s16x4w;

__device__ __forceinline__ bf16x8 w1_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p0);
    const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// bx / by: the block's pixel slice (and slab) and its (co, ci) block -- blockIdx.x / .y of a launch of one layer, or what the grouped
// launch (conv1x1_wgrad_group_kernel) derives from its flat grid
template <int XT, int WVM, int WVN, int NT, int KS, bool XPL>
__device__ __forceinline__ void conv1x1_wgrad_body(const void* __restrict__ x_, const float* __restrict__ gy, float* __restrict__ slabs,
                                                   const W1Geom& g, unsigned char* smem, const int bx, const int by) {
    static_assert(!XPL || XT == 1, "spike planes are one exact bf16 term");
    const float* x = reinterpret_cast<const float*>(x_);
    const bf16x8* xp = reinterpret_cast<const bf16x8*>(x_);
    constexpr int KC = 16 * KS;                 // pixels per chunk
    constexpr int PITCH = KC * 2 + 16;          // bytes per channel row and term
    constexpr int RA = 32 * WVM, RB = 32 * WVN * NT;
    constexpr int A_TERM = RA * PITCH;
    constexpr int B_TERM = XPL ? (RB / 32) * KC * 64 : RB * PITCH;      // planes: [32-channel plane][KC pixels][64 B]
    constexpr int BUF = 3 * A_TERM + XT * B_TERM;
    constexpr int V4R = KC / 4;                 // float4 loads per channel row and chunk
    constexpr int ITEMS_A = RA * V4R;
    constexpr int ITEMS_B = XPL ? (RB / 8) * KC : RB * V4R;             // planes: (8-channel group, pixel) pairs of 16 bytes
    constexpr int ITEMS = ITEMS_A + ITEMS_B;
    constexpr int NIT = (ITEMS + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WVN, wn = wave % WVN;
    const int cob = by / g.ci_blocks, cib = by - cob * g.ci_blocks;
    const int co0 = cob * RA, ci0 = cib * RB;

    // staging items: [0, ITEMS_A) grad_y (channel row, float4 index); then x -- fp32: (channel row, float4 index); planes: (group, pixel)
    int it_goff[NIT];        // element offset inside the image's tensor, -1: zero row / group past Cin
    int it_lofs[NIT];        // LDS byte offset of the first term
    bool it_isa[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int item = it * 256 + tid;
        item = item < ITEMS ? item : ITEMS - 1;
        const bool isa = item < ITEMS_A;
        it_isa[it] = isa;
        if (isa || !XPL) {
            const int row = item / V4R, v = item - row * V4R;
            const int ch = isa ? co0 + row : ci0 + row - RA;
            const bool ok = isa ? ch < g.Cout : ch < g.Cin;
            it_goff[it] = ok ? ch * g.HW + 4 * v : -1;
            it_lofs[it] = isa ? row * PITCH + 8 * v : 3 * A_TERM + (row - RA) * PITCH + 8 * v;
        } else {
            const int ib = item - ITEMS_A;
            const int gi = ib / KC, px = ib - gi * KC;          // consecutive lanes = consecutive pixels of one group: contiguous 16-byte pieces
            const bool ok = ci0 + gi * 8 < g.Cin;
            it_goff[it] = ok ? (ci0 / 8 + gi) * g.HW + px : -1;  // in 16-byte units
            it_lofs[it] = 3 * A_TERM + (gi >> 2) * (KC * 64) + px * 64 + (gi & 3) * 16;
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;

    const int chunks_img = g.HW / KC;
    const int total_chunks = g.NI * chunks_img;
    const int c_begin = bx * g.per_slice;
    const int c_end = c_begin + g.per_slice < total_chunks ? c_begin + g.per_slice : total_chunks;

    f32x4 L[NIT];
    auto fetch = [&](int c) {
        const int img = c / chunks_img, p0 = (c - img * chunks_img) * KC;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it * 256 >= ITEMS) continue;
            if (XPL && !it_isa[it]) {
                const bf16x8* src = it_goff[it] >= 0 ? xp + (long)img * (g.Cin / 8) * g.HW + it_goff[it] + p0 : reinterpret_cast<const bf16x8*>(eas_c1_zero_page);
                L[it] = __builtin_bit_cast(f32x4, *src);
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
            if (XPL && !it_isa[it]) {
                *(f32x4*)(buf + it_lofs[it]) = L[it];           // the 8 channels of a pixel, as they are
                continue;
            }
            bf16x4 t0, t1, t2;
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
    const int b_frag = 3 * A_TERM + (wn * NT * 32 + r) * PITCH + 16 * h;      // fp32 form: + n * 32 * PITCH + term * B_TERM + ks * 32
    // planes form: transposing-read geometry (conv_wgrad_mfma.hip): 16-lane group gq -> channel column block (gq & 1) * 16, pixels
    // (gq >> 1) * 8 + (li >> 2) and + 4 of the k-step
    const int gq = lane >> 4, li = lane & 15;
    const int b_tr = 3 * A_TERM + ((gq >> 1) * 8 + (li >> 2)) * 64 + ((gq & 1) * 16 + 4 * (li & 3)) * 2;   // + plane * KC * 64 + ks * 16 * 64

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
                if constexpr (XPL) {
                    const unsigned char* pb = cur + b_tr + (wn * NT + n) * (KC * 64) + ks * 16 * 64;
                    b[0] = w1_tr_frag(pb, pb + 4 * 64);
                } else {
#pragma unroll
                    for (int t = 0; t < XT; ++t) b[t] = *(const bf16x8*)(cur + b_frag + n * 32 * PITCH + t * B_TERM + ks * 32);
                }
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
    float* slab = slabs + (size_t)bx * g.Cout * g.Cin;
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

template <int XT, int WVM, int WVN, int NT, int KS, bool XPL = false>
__global__ __launch_bounds__(256) void conv1x1_wgrad_lds_kernel(const void* __restrict__ x_, const float* __restrict__ gy,
                                                                float* __restrict__ slabs, W1Geom g) {
    extern __shared__ __align__(16) unsigned char smem[];
    conv1x1_wgrad_body<XT, WVM, WVN, NT, KS, XPL>(x_, gy, slabs, g, smem, (int)blockIdx.x, (int)blockIdx.y);
}

// Grouped launch (eas_conv_wgrad_group_partial, ksize 1): layer p owns blocks [first[p], first[p + 1]) of the flat grid, slices[p] x
// (co, ci) blocks, slice fastest.  HW % 16 == 0; a layer's chunk size is its own (W1GroupArgs.ks2).
constexpr int kMaxW1Group = 12;
struct W1GroupArgs {
    W1Geom g[kMaxW1Group];
    const void* x[kMaxW1Group];
    const float* gy[kMaxW1Group];
    float* slabs[kMaxW1Group];
    int slices[kMaxW1Group];
    int ks2[kMaxW1Group];              // 1: 32-pixel chunks (KS = 2), else 16-pixel chunks
    int first[kMaxW1Group + 1];
    int n;
};

template <int XT, int WVM, int WVN, int NT>
__global__ __launch_bounds__(256) void conv1x1_wgrad_group_kernel(const W1GroupArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int lb = (int)blockIdx.x - a.first[p];
    const int by = lb / a.slices[p], bx = lb - by * a.slices[p];
    // 32-pixel chunks (two k-steps per barrier) where the map allows it, 16-pixel chunks for the others (the 8x10 level: 80 pixels)
    if (a.ks2[p]) conv1x1_wgrad_body<XT, WVM, WVN, NT, 2, false>(a.x[p], a.gy[p], a.slabs[p], a.g[p], smem, bx, by);
    else conv1x1_wgrad_body<XT, WVM, WVN, NT, 1, false>(a.x[p], a.gy[p], a.slabs[p], a.g[p], smem, bx, by);
}

struct W1Plan { int wvm, wvn, nt, ks, slices; };

// blocks of a kernel instance a CU holds at once (registers, waves and its dynamic LDS), cached per instance
template <int XT, int WVM, int WVN, int NT, int KS, bool XPL>
int resident_w1() {
    static int nb = 0;
    if (nb > 0) return nb;
    auto kern = conv1x1_wgrad_lds_kernel<XT, WVM, WVN, NT, KS, XPL>;
    constexpr int PITCH = 16 * KS * 2 + 16;
    constexpr int B_TERM = XPL ? WVN * NT * 16 * KS * 64 : 32 * WVN * NT * PITCH;
    const size_t lds = (size_t)2 * (3 * 32 * WVM * PITCH + XT * B_TERM);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, lds) != hipSuccess || n < 1) n = 2;
    nb = n;
    return nb;
}

// block shape: all four waves busy whatever Cout is.  Pixel slices: every block walks per_slice chunks and the launch takes
// ceil(slices * yz / slots) rounds of the chip's block slots (256 CUs x the blocks of the kernel instance a CU really holds); the slice
// count with the fewest chunk periods wins, ties go to fewer slabs (the old rule aimed at 512 blocks whatever the instance: 540 blocks on
// 512 slots are two rounds)
W1Plan w1_plan(W1Geom& g, int NI, int Cin, int Cout, int HW, int x_terms, bool planes) {
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
    int res = 2;
#define EAS_W1R(XT_, M_, N_, T_, PL_) (p.ks == 2 ? resident_w1<XT_, M_, N_, T_, 2, PL_>() : resident_w1<XT_, M_, N_, T_, 1, PL_>())
#define EAS_W1R_SHAPE(XT_, PL_) (p.wvm == 4 ? EAS_W1R(XT_, 4, 1, 4, PL_) : (p.wvm == 2 ? EAS_W1R(XT_, 2, 2, 2, PL_) : EAS_W1R(XT_, 1, 4, 1, PL_)))
    if (planes) res = EAS_W1R_SHAPE(1, true);
    else res = x_terms == 1 ? EAS_W1R_SHAPE(1, false) : EAS_W1R_SHAPE(3, false);
#undef EAS_W1R_SHAPE
#undef EAS_W1R
    static const int force = eas_dev_env("EAS_W1_BLOCKS") ? atoi(eas_dev_env("EAS_W1_BLOCKS")) : 0;      // development: the old rule with this target
    int best = 1;
    if (force > 0) {
        best = (force + yz - 1) / yz;
        if (best > total_chunks) best = total_chunks;
    } else {
        static const int slot_pct = eas_dev_env("EAS_WG_SLOT_PCT") ? atoi(eas_dev_env("EAS_WG_SLOT_PCT")) : 100;      // development (conv_wgrad_mfma.hip wg_plan)
        const long slots = 256L * res * slot_pct / 100;
        long best_cost = -1;
        for (int sl = 1; sl <= total_chunks && (long)sl * yz <= 2 * slots; ++sl) {
            const long rounds = ((long)sl * yz + slots - 1) / slots;
            const long cost = rounds * ((total_chunks + sl - 1) / sl);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sl; }
        }
    }
    g.per_slice = (total_chunks + best - 1) / best;
    p.slices = (total_chunks + g.per_slice - 1) / g.per_slice;
    return p;
}

template <int XT, int WVM, int WVN, int NT, int KS, bool XPL = false>
int launch_w1_lds(const void* x, const float* gy, float* slabs, W1Geom g, int slices, hipStream_t st) {
    auto kern = conv1x1_wgrad_lds_kernel<XT, WVM, WVN, NT, KS, XPL>;
    constexpr int PITCH = 16 * KS * 2 + 16;
    constexpr int B_TERM = XPL ? WVN * NT * 16 * KS * 64 : 32 * WVN * NT * PITCH;
    const size_t lds = (size_t)2 * (3 * 32 * WVM * PITCH + XT * B_TERM);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(slices, ((g.Cout + 32 * WVM - 1) / (32 * WVM)) * g.ci_blocks);
    EAS_LAUNCH(kern, grid, dim3(256), lds, st, x, gy, slabs, g);
    return EAS_OK;
}

}  // namespace

int eas_conv1x1_wgrad_slices(int NI, int Cin, int Cout, int HW, int x_terms, int planes) {
    W1Geom g{};
    if (HW % 16 != 0) return 0;
    return w1_plan(g, NI, Cin, Cout, HW, x_terms, planes != 0).slices;
}

// slabs: eas_conv1x1_wgrad_slices(...) * Cout * Cin floats; the caller reduces them (conv_wgrad_reduce_kernel)
// planes != 0: x is a spike-plane tensor (one term)
int eas_conv1x1_wgrad_dispatch(const void* x, const float* gy, float* slabs, int NI, int Cin, int Cout, int HW, int x_terms, hipStream_t st,
                               int planes) {
    W1Geom g{};
    if (HW % 16 != 0 || (planes && Cin % 8 != 0)) return EAS_ERR_UNSUPPORTED;
    const W1Plan p = w1_plan(g, NI, Cin, Cout, HW, x_terms, planes != 0);
#define EAS_W1(XT_, M_, N_, T_, PL_) (p.ks == 2 ? launch_w1_lds<XT_, M_, N_, T_, 2, PL_>(x, gy, slabs, g, p.slices, st) : launch_w1_lds<XT_, M_, N_, T_, 1, PL_>(x, gy, slabs, g, p.slices, st))
#define EAS_W1_SHAPE(XT_, PL_) (p.wvm == 4 ? EAS_W1(XT_, 4, 1, 4, PL_) : (p.wvm == 2 ? EAS_W1(XT_, 2, 2, 2, PL_) : EAS_W1(XT_, 1, 4, 1, PL_)))
    if (planes) return EAS_W1_SHAPE(1, true);
    return x_terms == 1 ? EAS_W1_SHAPE(1, false) : EAS_W1_SHAPE(3, false);
#undef EAS_W1_SHAPE
#undef EAS_W1
}

// the slab kernels of several 1x1 layers as one grid: one block shape (all layers on the same side of the 32 / 64 output-channel
// bounds), a common number of 16-pixel chunks per block (see group_tau in conv_wgrad_mfma.hip: same rule)
namespace {
template <int WVM, int WVN, int NT>
int launch_w1_group(const W1GroupArgs& a, int blocks, hipStream_t st, int* res_out) {
    auto kern = conv1x1_wgrad_group_kernel<3, WVM, WVN, NT>;
    constexpr int PITCH = 16 * 2 * 2 + 16;            // sized for the 32-pixel chunks (KS = 2); the 16-pixel bodies use less
    const size_t lds = (size_t)2 * (3 * 32 * WVM * PITCH + 3 * 32 * WVN * NT * PITCH);
    static int res = 0;
    if (res == 0) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 2;
        res = nb;
    }
    if (res_out) { *res_out = res; return EAS_OK; }
    EAS_LAUNCH(kern, dim3(blocks), dim3(256), lds, st, a);
    return EAS_OK;
}
}  // namespace

int eas_conv1x1_wgrad_group(const EasWgradProblem* pr, int n, int x_terms, hipStream_t st, int* slabs_out, bool query) {
    if (n > kMaxW1Group || x_terms != 3) return EAS_ERR_UNSUPPORTED;
    W1GroupArgs a{};
    int yz[kMaxW1Group], nc[kMaxW1Group];
    const int cls0 = pr[0].Cout > 64 ? 2 : (pr[0].Cout > 32 ? 1 : 0);
    const int wvm = cls0 == 2 ? 4 : (cls0 == 1 ? 2 : 1), wvn = cls0 == 2 ? 1 : (cls0 == 1 ? 2 : 4), nt = cls0 == 2 ? 4 : (cls0 == 1 ? 2 : 1);
    const int RA = 32 * wvm, RB = 32 * wvn * nt;
    for (int p = 0; p < n; ++p) {
        const EasWgradProblem& q = pr[p];
        if (q.NI <= 0 || q.Cin <= 0 || q.Cout <= 0 || q.Hi <= 0 || q.Wi <= 0) return EAS_ERR_INVALID_ARG;
        if (!query && (!q.x || !q.grad_y || !q.workspace)) return EAS_ERR_INVALID_ARG;
        const int HW = q.Hi * q.Wi;
        const int cls = q.Cout > 64 ? 2 : (q.Cout > 32 ? 1 : 0);
        if (HW % 16 != 0 || cls != cls0) return EAS_ERR_UNSUPPORTED;
        W1Geom& g = a.g[p];
        g.NI = q.NI; g.Cin = q.Cin; g.Cout = q.Cout; g.HW = HW;
        g.ci_blocks = (q.Cin + RB - 1) / RB;
        yz[p] = ((q.Cout + RA - 1) / RA) * g.ci_blocks;
        a.ks2[p] = HW % 32 == 0 ? 1 : 0;
        nc[p] = q.NI * (HW / (a.ks2[p] ? 32 : 16));
        a.x[p] = q.x; a.gy[p] = q.grad_y; a.slabs[p] = q.workspace;
    }
    int res = 2;
    int rc = cls0 == 2 ? launch_w1_group<4, 1, 4>(a, 0, st, &res) : (cls0 == 1 ? launch_w1_group<2, 2, 2>(a, 0, st, &res) : launch_w1_group<1, 4, 1>(a, 0, st, &res));
    if (rc != EAS_OK) return rc;
    // common chunks per block (the rule of group_tau, conv_wgrad_mfma.hip)
    static const int slot_pct_g = eas_dev_env("EAS_WG_SLOT_PCT") ? atoi(eas_dev_env("EAS_WG_SLOT_PCT")) : 100;      // development
    const long slots = 256L * res * slot_pct_g / 100;
    int max_nc = 0;
    for (int p = 0; p < n; ++p) max_nc = nc[p] > max_nc ? nc[p] : max_nc;
    int tau = max_nc;
    long best_cost = -1;
    for (int t = max_nc; t >= 1; --t) {
        long blocks = 0;
        for (int p = 0; p < n; ++p) blocks += (long)((nc[p] + t - 1) / t) * yz[p];
        if (blocks > 2 * slots && best_cost >= 0) break;
        const long cost = ((blocks + slots - 1) / slots) * t;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; tau = t; }
    }
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        a.g[p].per_slice = tau < nc[p] ? tau : nc[p];
        a.slices[p] = (nc[p] + a.g[p].per_slice - 1) / a.g[p].per_slice;
        if (slabs_out) slabs_out[p] = a.slices[p];
        a.first[p] = blocks;
        blocks += a.slices[p] * yz[p];
    }
    a.first[n] = blocks;
    a.n = n;
    if (query) return EAS_OK;
    return cls0 == 2 ? launch_w1_group<4, 1, 4>(a, blocks, st, nullptr) : (cls0 == 1 ? launch_w1_group<2, 2, 2>(a, blocks, st, nullptr) : launch_w1_group<1, 4, 1>(a, blocks, st, nullptr));
}
