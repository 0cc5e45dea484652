// Input gradient of a 3x3 stride-1 convolution with at most 8 input channels (the stem: 8 sampler channels -> 32; the gradient flows on
// into the event sampler).  eas_conv_fwd on grad_y with transposed weights computes it as a GEMM with M = Cin = 8 rows in a 32-row MFMA
// tile: three quarters of the matrix-core work are padding (168 us per step at config 2 against 42 us of HBM time).
//
// Here the nine taps are STACKED along M instead: P[(tap, c)][q] = sum_co w[co][c][tap] * grad_y[co][q] is one GEMM with M = 9 * 8 = 72 rows
// (three 32-row tiles, 75 % used), K = Cout and no spatial shift at all -- a lane's B operand is the Cout channel values of ITS pixel,
// loaded straight from grad_y (no LDS patch), split into three exact bf16 terms in registers.  The shifts come afterwards, on the fp32
// results: grad_x[c][y][x] = sum_(kh,kw) P[(kh,kw,c)][y + 1 - kh][x + 1 - kw].  A wave owns a 32-column slice (30 useful columns, one halo
// column on either side) of a strip of rows and walks down the rows: the column shifts are lane shifts (ds_bpermute, no memory), the row
// shifts are three running sums in registers (the rows above / at / below the grad_y row just processed).  Per grad_x element the 9 x Cout
// products are the same exact bf16-term products as before, summed per tap on the matrix cores (smallest term products first) and then
// over the taps in the fixed order kh = 2, 1, 0 / kw = 0, 1, 2.
//
// Weights: built per block from the fp32 tensor into LDS as MFMA A fragments ([M-tile][k-step][term][lane], 9 KB per k-step) -- 2304
// floats, not worth a pack launch.
#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSdCols = 30;      // useful columns of a 32-lane slice

__device__ __attribute__((aligned(16))) float eas_sd_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void sd_split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    lo = (__bf16)r2;
}

struct SdGeom {
    int NI, Cin, Cout, H, W;
    int R;                  // rows of grad_x per strip
    int nslices, nstrips;
    int nwaves;
};

template <int KS>      // k-steps of 16 grad_y channels: Cout <= 16 * KS
__global__ __launch_bounds__(256) void conv_dgrad_small_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                                const SdGeom g) {
    __shared__ bf16x8 wa[3][KS][3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < 3 * KS * 64; idx += 256) {
        const int l = idx & 63;
        const int s = (idx >> 6) % KS, i = (idx >> 6) / KS;
        const int m = 32 * i + (l & 31), tap = m >> 3, c = m & 7;
        bf16x8 t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * (l >> 5) + j;
            const float v = (tap < 9 && c < g.Cin && k < g.Cout) ? w[((size_t)k * g.Cin + c) * 9 + tap] : 0.0f;
            __bf16 a, b, d;
            sd_split3(v, a, b, d);
            t0[j] = a; t1[j] = b; t2[j] = d;
        }
        wa[i][s][0][l] = t0; wa[i][s][1][l] = t1; wa[i][s][2][l] = t2;
    }
    __syncthreads();
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= g.nwaves) return;
    const int slice = gw % g.nslices, rest = gw / g.nslices;
    const int strip = rest % g.nstrips, n = rest / g.nstrips;
    const int r = lane & 31, hh = lane >> 5;
    const int col = slice * kSdCols - 1 + r;
    const bool colok = col >= 0 && col < g.W;
    const int y0 = strip * g.R;
    const int y1 = y0 + g.R < g.H ? y0 + g.R : g.H;           // rows [y0, y1) of grad_x
    const size_t plane = (size_t)g.H * g.W;
    const float* src = gy + ((size_t)n * g.Cout + 8 * hh) * plane + (colok ? col : 0);
    const int nb_up = ((lane + 1) & 63) * 4, nb_dn = ((lane + 63) & 63) * 4;      // ds_bpermute addresses of the lanes x + 1 / x - 1

    float cur[KS * 8], nxt[KS * 8];
    // the loads are unconditional: a column outside the image or a channel past Cout reads the zero page (channel stride 0), so the 8 * KS
    // loads of a row are in flight together (as conditional loads every one of them was a branch of its own)
    // (element offsets of the lane's channels inside its image, computed once: NI * Cout * H * W < 2^31; -1 = the zero page)
    int choff[KS * 8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) choff[s * 8 + j] = (colok && 16 * s + 8 * hh + j < g.Cout) ? (16 * s + j) * (int)plane : -1;
    auto load_row = [&](int q, float (&v)[KS * 8]) {
        if (q >= 0 && q < g.H) {                               // wave-uniform
            const float* row = src + (size_t)q * g.W;
#pragma unroll
            for (int k = 0; k < KS * 8; ++k) {
                const float* p = choff[k] >= 0 ? row + choff[k] : eas_sd_zero_page;
                v[k] = *p;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KS * 8; ++k) v[k] = 0.0f;
        }
    };
    float sp[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, sn[4] = {0.f, 0.f, 0.f, 0.f};      // rows q - 1, q, q + 1 of grad_x
    load_row(y0 - 1, cur);
    for (int q = y0 - 1; q <= y1; ++q) {
        if (q < y1) load_row(q + 1, nxt);
        if (q >= 0 && q < g.H) {
            bf16x8 b[KS][3];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 a0, a1, a2;
                    sd_split3(cur[s * 8 + j], a0, a1, a2);
                    b[s][0][j] = a0; b[s][1][j] = a1; b[s][2][j] = a2;
                }
            f32x16 acc[3];
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // smallest term products first; consecutive MFMAs go to different accumulators; the first product of a tile starts from the
            // constant zero operand (no 48 register writes per row)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][s][PA[p]][lane], b[s][PB[p]], (s == 0 && p == 0) ? zero : acc[i], 0, 0, 0);
            // accumulator element e of tile i: tap 4 * i + (e >> 2), channel 4 * hh + (e & 3) of the lane's pixel
#pragma unroll
            for (int kh = 2; kh >= 0; --kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tap = kh * 3 + kw, i = tap >> 2, eb = (tap & 3) * 4;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        float v = acc[i][eb + cc];
                        if (kw == 0) v = __int_as_float(__builtin_amdgcn_ds_bpermute(nb_up, __float_as_int(v)));        // P at column x + 1
                        if (kw == 2) v = __int_as_float(__builtin_amdgcn_ds_bpermute(nb_dn, __float_as_int(v)));        // P at column x - 1
                        if (kh == 0) sp[cc] += v;
                        else if (kh == 1) sc[cc] += v;
                        else sn[cc] += v;
                    }
                }
        }
        const int yo = q - 1;
        if (yo >= y0 && yo < y1 && colok && r >= 1 && r <= kSdCols) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                if (4 * hh + cc < g.Cin) gx[(((size_t)n * g.Cin + 4 * hh + cc) * g.H + yo) * g.W + col] = sp[cc];
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            sp[cc] = sc[cc];
            sc[cc] = sn[cc];
            sn[cc] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < KS * 8; ++k) cur[k] = nxt[k];
    }
}

}  // namespace

extern "C" {

// 1 when eas_conv_dgrad_small takes this geometry
int eas_conv_dgrad_small_supported(int NI, int Cin, int Cout, int H, int W) {
    return NI > 0 && Cin >= 1 && Cin <= 8 && Cout >= 1 && Cout <= 64 && H >= 1 && W >= 1 && (long long)NI * Cout * H * W < (1LL << 31) ? 1 : 0;
}

// grad_x [NI, Cin, H, W] of a 3x3 stride-1 convolution (padding 1) with Cin <= 8 from grad_y [NI, Cout, H, W] (Cout <= 64) and the fp32
// weights [Cout, Cin, 3, 3] (no packed form needed).  Replaces eas_conv_fwd on grad_y with mode-1 weights for the stem (yolox/models/
// darknet.py: the first BaseConv of CSPDarknet), where Cin = 8 rows fill a quarter of a 32-row matrix-core tile.
int eas_conv_dgrad_small(const float* grad_y, const float* w, float* grad_x, int NI, int Cin, int Cout, int H, int W, eas_stream_t stream) {
    if (!grad_y || !w || !grad_x) return EAS_ERR_INVALID_ARG;
    if (!eas_conv_dgrad_small_supported(NI, Cin, Cout, H, W)) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    SdGeom g{};
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W;
    g.nslices = (W + kSdCols - 1) / kSdCols;
    // about three waves per SIMD (1024 SIMDs), strips of at least 8 rows (a strip recomputes two halo rows)
    long want = (3072 + (long)NI * g.nslices - 1) / ((long)NI * g.nslices);
    if (want < 1) want = 1;
    int R = (int)((H + want - 1) / want);
    if (R < 8) R = H < 8 ? H : 8;
    g.R = R;
    g.nstrips = (H + R - 1) / R;
    const long nwaves = (long)NI * g.nstrips * g.nslices;
    if (nwaves >= (1L << 30)) return EAS_ERR_UNSUPPORTED;
    g.nwaves = (int)nwaves;
    const dim3 grid((unsigned)((nwaves + 3) / 4));
    hipStream_t st = eas_s(stream);
    const int ks = (Cout + 15) / 16;
    if (ks == 1) EAS_LAUNCH(conv_dgrad_small_kernel<1>, grid, dim3(256), 0, st, grad_y, w, grad_x, g);
    else if (ks == 2) EAS_LAUNCH(conv_dgrad_small_kernel<2>, grid, dim3(256), 0, st, grad_y, w, grad_x, g);
    else if (ks == 3) EAS_LAUNCH(conv_dgrad_small_kernel<3>, grid, dim3(256), 0, st, grad_y, w, grad_x, g);
    else EAS_LAUNCH(conv_dgrad_small_kernel<4>, grid, dim3(256), 0, st, grad_y, w, grad_x, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
