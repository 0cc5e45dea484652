// Direct convolution for the adaptive sampler's conv stacks: Conv2d(2->4, k, pad k/2) [+ ReLU + Conv2d(4->4, k)]
// (yolox/models/embedding.py:106-111), stride 1, "same" zero padding, NCHW fp32, k in {3,5,7}.
//
// Generic GEMM-shaped conv kernels waste the machine here: 2..4 channels cannot fill an MFMA tile, and MIOpen's
// igemm path also pays NCHW<->NHWC transposes.  This is a VALU kernel: one 256-thread block owns a 16x64 output
// tile; the input tile + halo for ALL input channels is staged once in LDS (rows padded to 80 floats so the four
// rows a wave touches fall on different bank quarters); each thread produces 4 consecutive pixels of EVERY output
// channel from registers (16..  accumulators), weights come from scalar loads (uniform addresses -> SGPRs).
// The same kernel computes the data gradient (weights flipped + channel-transposed, optional ReLU mask epilogue).
// The weight gradient kernel keeps x (with halo) and grad_y tiles in LDS, gives each thread one (ci, ky) row of
// the filter for all (co, kx) and a share of the tile's pixels, and reduces block partials in fixed order; for k >= 5 the
// matrix-core form of smallconv_wgrad_mfma.hip produces the block partials instead.
#include <stdlib.h>

#include "eas_common.h"

namespace {

constexpr int TH = 16, TW = 64, LWS = 80, NT = 256;

template <int K>
struct Geo {
    static constexpr int PAD = K / 2;
    static constexpr int LH = TH + K - 1;
    static constexpr int LW = TW + K - 1;
    static constexpr int NV = 4 + K - 1;          // input values a thread needs per row
    static constexpr int NV4 = (NV + 3) / 4;
};

// Tile staging with register prefetch (issue-early / write-late): every thread owns PER elements of the
// [CIN][LH][LW] input window (origin (y0-PAD, x0-PAD), zero filled outside the image).  `load` issues all PER
// global loads back to back (out-of-range lanes read element 0 and are zeroed by a select, so there are no
// branches and the loads stay in flight across the compute phase of the previous tile); `store` writes them to
// LDS (channel planes PLANE floats apart, rows LWS floats apart).
template <int CIN, int K, int PLANE, int PITCH = LWS>
struct Stager {
    using G = Geo<K>;
    static constexpr int NE = CIN * G::LH * G::LW;
    static constexpr int PER = (NE + NT - 1) / NT;
    // Which window element a thread stages does not depend on the tile: (channel, row, column) -> image-relative offset, packed
    // (row, column) and LDS offset are worked out ONCE per thread (the divisions by the window width / height per element and
    // tile were as many vector-ALU instructions as the convolution's FMAs); per tile only the bounds test and one add remain.
    int rel[PER];        // (c * H + r) * W + col, relative to the window origin
    int rc[PER];         // r | col << 8, or -1 for the surplus elements of the last slice
    int lofs[PER];       // c * PLANE + r * PITCH + col

    __device__ __forceinline__ void init(int H, int W) {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int col = idx % G::LW;
            const int r = (idx / G::LW) % G::LH;
            const int c = idx / (G::LW * G::LH);
            rel[it] = (c * H + r) * W + col;
            rc[it] = idx < NE ? (r | (col << 8)) : -1;
            lofs[it] = c * PLANE + r * PITCH + col;
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ x, float (&pre)[PER], int n, int H, int W, int y0, int x0) const {
        const int oy = y0 - G::PAD, ox = x0 - G::PAD;
        const float* win = x + ((int64_t)n * CIN * H + oy) * W + ox;       // window origin (may lie outside the image)
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int r = rc[it] & 255, col = rc[it] >> 8;
            const unsigned gy = (unsigned)(oy + r), gx = (unsigned)(ox + col);
            const bool ok = rc[it] >= 0 && gy < (unsigned)H && gx < (unsigned)W;
            const float v = *(ok ? win + rel[it] : x);
            pre[it] = ok ? v : 0.f;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, const float (&pre)[PER]) const {
#pragma unroll
        for (int it = 0; it < PER; ++it)
            if (rc[it] >= 0) lds[lofs[it]] = pre[it];
    }
};

// FWD:   y[n][o] = b[o] + sum_{i,ky,kx} w[o][i][ky][kx] * x[n][i][. + ky - PAD][. + kx - PAD]          (CIN=Cin, COUT=Cout)
// DGRAD: gx[n][o] = sum_{i,ky,kx} w[i][o][K-1-ky][K-1-kx] * gy[n][i][. + ky - PAD][. + kx - PAD]       (CIN=Cout, COUT=Cin)
template <int CIN, int COUT, int K, bool DGRAD>
__global__ __launch_bounds__(NT, 2) void smallconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, const float* __restrict__ mask,
                                                          float* __restrict__ y, int N, int H, int W, int relu) {
    using G = Geo<K>;
    // Row pitch 128 floats (a multiple of the 64-bank row): ds_read_b128 serves the lanes in the groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31}, ...; with lane = (row << 4) | column-quad a group then touches column quads 0-3 and 12-15 of one row and
    // 4-11 of the next -- every bank once.  (The earlier pitch of 80 floats shifted the second row by 16 banks onto the first
    // row's quads 12-15: PMC showed 62 % of the LDS cycles as bank conflicts; the kernel gained 3 %, it is not LDS-bound.)
    constexpr int LWF = 128;
    constexpr int PLANE = G::LH * LWF;
    using St = Stager<CIN, K, PLANE, LWF>;
    St st;
    st.init(H, W);
    __shared__ __attribute__((aligned(16))) float lds[CIN * PLANE];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float pre[St::PER];
    int tile = blockIdx.x;
    if (tile < ntiles)
        st.load(x, pre, tile / (tiles_x * tiles_y), H, W, ((tile / tiles_x) % tiles_y) * TH, (tile % tiles_x) * TW);
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int ty0 = ((tile / tiles_x) % tiles_y) * TH, tx0 = (tile % tiles_x) * TW;
        __syncthreads();                               // readers of the previous tile are done
        st.store(lds, pre);
        __syncthreads();
        const int nxt = tile + gridDim.x;              // next tile's loads fly during this tile's FMAs
        if (nxt < ntiles)
            st.load(x, pre, nxt / (tiles_x * tiles_y), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
        float acc[COUT][4];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float bv = (!DGRAD && b) ? b[o] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[o][j] = bv;
        }
#pragma unroll 1
        for (int i = 0; i < CIN; ++i) {                // not unrolled: keeps the live set small
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float v[G::NV4 * 4];
                const float4* row = reinterpret_cast<const float4*>(lds + i * PLANE + (ty + ky) * LWF + 4 * tx);
#pragma unroll
                for (int q = 0; q < G::NV4; ++q) {
                    const float4 t = row[q];
                    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
#pragma unroll
                    for (int o = 0; o < COUT; ++o) {
                        const float wv = DGRAD ? w[((i * COUT + o) * K + (K - 1 - ky)) * K + (K - 1 - kx)]
                                               : w[((o * CIN + i) * K + ky) * K + kx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(wv, v[j + kx], acc[o][j]);
                    }
                }
            }
        }
        const int oy = ty0 + ty, ox = tx0 + 4 * tx;
        if (oy < H && ox < W) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const int64_t off = (((int64_t)n * COUT + o) * H + oy) * W + ox;
                float r[4] = {acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = fmaxf(r[j], 0.f);
                }
                if (ox + 3 < W && (W & 3) == 0) {
                    if (mask) {
                        const float4 m = *reinterpret_cast<const float4*>(mask + off);
                        r[0] = m.x > 0.f ? r[0] : 0.f; r[1] = m.y > 0.f ? r[1] : 0.f;
                        r[2] = m.z > 0.f ? r[2] : 0.f; r[3] = m.w > 0.f ? r[3] : 0.f;
                    }
                    *reinterpret_cast<float4*>(y + off) = make_float4(r[0], r[1], r[2], r[3]);
                } else {
                    for (int j = 0; j < 4 && ox + j < W; ++j) y[off + j] = (mask && !(mask[off + j] > 0.f)) ? 0.f : r[j];
                }
            }
        }
    }
}

// ---- weight gradient -------------------------------------------------------------------------------------------
// gw[o][i][ky][kx] = sum_{n,y,x} gy[n][o][y][x] * x[n][i][y + ky - PAD][x + kx - PAD];  gb[o] = sum gy[n][o]
// Thread = (input channel i, pixel group): it keeps the whole [COUT][K][K] slice of its channel in registers
// (100 accumulators for 4x5x5) and walks 4-pixel units of the tile: per unit 4 global float4 loads of grad_y
// (shared by the CIN lanes of the unit), K x 2 LDS b128 reads of x, COUT*K*K*4 FMAs.  Channel planes in LDS are
// skewed by 16 floats so the CIN lanes of a unit hit different bank quarters.  Accumulators persist over all
// tiles of the block; one shuffle + LDS reduction per block, block partials summed in fixed order afterwards.
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(NT) void smallconv_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             float* __restrict__ partial, int N, int H, int W) {
    using G = Geo<K>;
    constexpr int GROUPS = NT / CIN;
    constexpr int UNITS = TH * (TW / 4);
    constexpr int PLANE = G::LH * LWS + 16;
    constexpr int NW = COUT * CIN * K * K;
    constexpr int NOUT = NW + COUT;
    constexpr int SLICE = COUT * K * K;
    constexpr int LG = COUT * TH * TW;
    static_assert(CIN * PLANE + LG >= (NT / EAS_WAVE) * CIN * (SLICE + COUT), "reduction scratch does not fit");
    __shared__ __attribute__((aligned(16))) float lx[CIN * PLANE + LG];   // x tile + halo | grad_y tile
    float* lg = lx + CIN * PLANE;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int ci = threadIdx.x % CIN, grp = threadIdx.x / CIN;
    const bool vec_ok = (W & 3) == 0;
    float acc[COUT][K][K];
    float accb[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[o][ky][kx] = 0.f;
    }
    using St = Stager<CIN, K, PLANE>;
    St st;
    st.init(H, W);
    constexpr int GPER = (LG / 4 + NT - 1) / NT;
    float pre[St::PER];
    float4 preg[GPER];
    auto load_gy = [&](int t) {
        const int n = t / (tiles_x * tiles_y);
        const int ty0 = ((t / tiles_x) % tiles_y) * TH, tx0 = (t % tiles_x) * TW;
#pragma unroll
        for (int it = 0; it < GPER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int q = idx % (TW / 4), r = (idx / (TW / 4)) % TH, o = idx / ((TW / 4) * TH);
            const int oy = ty0 + r, ox = tx0 + 4 * q;
            const int64_t off = (((int64_t)n * COUT + o) * H + oy) * W + ox;
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < LG / 4 && oy < H) {
                if (ox + 3 < W && vec_ok) {
                    t4 = *reinterpret_cast<const float4*>(gy + off);
                } else {
                    if (ox < W) t4.x = gy[off];
                    if (ox + 1 < W) t4.y = gy[off + 1];
                    if (ox + 2 < W) t4.z = gy[off + 2];
                    if (ox + 3 < W) t4.w = gy[off + 3];
                }
            }
            preg[it] = t4;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) {
        st.load(x, pre, tile / (tiles_x * tiles_y), H, W, ((tile / tiles_x) % tiles_y) * TH, (tile % tiles_x) * TW);
        load_gy(tile);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                           // previous tile's readers are done
        st.store(lx, pre);
#pragma unroll
        for (int it = 0; it < GPER; ++it) {
            const int idx = threadIdx.x + it * NT;
            if (idx < LG / 4) reinterpret_cast<float4*>(lg)[idx] = preg[it];
        }
        __syncthreads();
        const int nxt = tile + gridDim.x;          // next tile's loads fly during this tile's FMAs
        if (nxt < ntiles) {
            st.load(x, pre, nxt / (tiles_x * tiles_y), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
            load_gy(nxt);
        }
        for (int u = grp; u < UNITS; u += GROUPS) {
            const int r = u / (TW / 4), q = u % (TW / 4);
            float g[COUT][4];
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float4 t = *reinterpret_cast<const float4*>(lg + (o * TH + r) * TW + 4 * q);
                g[o][0] = t.x; g[o][1] = t.y; g[o][2] = t.z; g[o][3] = t.w;
                if (ci == 0) accb[o] += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float v[G::NV4 * 4];
                const float4* row = reinterpret_cast<const float4*>(lx + ci * PLANE + (r + ky) * LWS + 4 * q);
#pragma unroll
                for (int k4 = 0; k4 < G::NV4; ++k4) {
                    const float4 t = row[k4];
                    v[4 * k4] = t.x; v[4 * k4 + 1] = t.y; v[4 * k4 + 2] = t.z; v[4 * k4 + 3] = t.w;
                }
#pragma unroll
                for (int o = 0; o < COUT; ++o)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[o][ky][kx] = fmaf(g[o][j], v[j + kx], acc[o][ky][kx]);
            }
        }
    }
    // lanes l and l + CIN*m hold the same channel: butterfly over the lane bits above log2(CIN)
#pragma unroll
    for (int off = CIN; off < EAS_WAVE; off <<= 1) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            accb[o] += __shfl_xor(accb[o], off, EAS_WAVE);
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[o][ky][kx] += __shfl_xor(acc[o][ky][kx], off, EAS_WAVE);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (EAS_WAVE - 1), wid = threadIdx.x / EAS_WAVE;
    float* red = lx;                                // [wave][ci][SLICE + COUT]
    if (lane < CIN) {
        float* dst = red + (wid * CIN + lane) * (SLICE + COUT);
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) dst[(o * K + ky) * K + kx] = acc[o][ky][kx];
            dst[SLICE + o] = accb[o];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NOUT; e += NT) {
        int c, within;
        if (e < NW) {                               // e indexes [o][ci][ky][kx]
            const int o = e / (CIN * K * K), rem = e % (CIN * K * K);
            c = rem / (K * K);
            within = o * K * K + rem % (K * K);
        } else {
            c = 0;
            within = SLICE + (e - NW);
        }
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / EAS_WAVE; ++wv) s += red[(wv * CIN + c) * (SLICE + COUT) + within];
        partial[(int64_t)blockIdx.x * NOUT + e] = s;
    }
}

// one wave per output element: lanes stride over the block partials, fixed-order butterfly -> deterministic
__global__ __launch_bounds__(EAS_WAVE) void smallconv_wgrad_finalize(const float* __restrict__ partial, int nblocks, int nout,
                                                                     int nw, float* __restrict__ gw, float* __restrict__ gb) {
    const int e = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += EAS_WAVE) s += (double)partial[(int64_t)b * nout + e];
    s = eas_wave_sum(s);
    if (threadIdx.x == 0) {
        if (e < nw) gw[e] = (float)s;
        else if (gb) gb[e - nw] = (float)s;
    }
}

constexpr int kWgradBlocks = 512;
constexpr int kConvBlocks = 256 * 6;

template <int CIN, int COUT, bool DGRAD>
int launch_conv_k(int k, const float* x, const float* w, const float* b, const float* mask, float* y, int N, int H, int W,
                  int relu, hipStream_t st) {
    int tiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (tiles > kConvBlocks) tiles = kConvBlocks;      // persistent blocks loop over tiles with register prefetch
    switch (k) {
        case 3: hipLaunchKernelGGL((smallconv_kernel<CIN, COUT, 3, DGRAD>), dim3(tiles), dim3(NT), 0, st, x, w, b, mask, y, N, H, W, relu); break;
        case 5: hipLaunchKernelGGL((smallconv_kernel<CIN, COUT, 5, DGRAD>), dim3(tiles), dim3(NT), 0, st, x, w, b, mask, y, N, H, W, relu); break;
        case 7: hipLaunchKernelGGL((smallconv_kernel<CIN, COUT, 7, DGRAD>), dim3(tiles), dim3(NT), 0, st, x, w, b, mask, y, N, H, W, relu); break;
        default: return EAS_ERR_UNSUPPORTED;
    }
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

template <int CIN, int COUT>
int launch_wgrad_k(int k, const float* gy, const float* x, float* partial, int nblocks, int N, int H, int W, hipStream_t st) {
    switch (k) {
        case 3: hipLaunchKernelGGL((smallconv_wgrad_kernel<CIN, COUT, 3>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W); break;
        case 5: hipLaunchKernelGGL((smallconv_wgrad_kernel<CIN, COUT, 5>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W); break;
        case 7: hipLaunchKernelGGL((smallconv_wgrad_kernel<CIN, COUT, 7>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W); break;
        default: return EAS_ERR_UNSUPPORTED;
    }
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

bool bad_ptr(const void* p) { return ((uintptr_t)p & 15) != 0; }

// Which weight-gradient kernel eas_smallconv_bwd_weight launches: the matrix-core form for k >= 5 (measured on MI355X, 256 images of
// 256x320: 4->4 k5 324 us against 580 us, k7 79 against 417 us), the vector-ALU form for k = 3 (95 against 165 us).
// EAS_SW_FORM=fma|mfma (development switch) forces one; read per call, so a test can compare the two forms in one process.
bool wgrad_on_mfma(int k) {
    const char* e = getenv("EAS_SW_FORM");
    if (e && e[0]) return e[0] == 'm';
    return k >= 5;
}

// Forward / input gradient at k = 5 also exist in a matrix-core form (smallconv_mfma.hip).  It is exact and correct but NOT faster than
// this file's vector-ALU kernel (64 images of 256x320, 4 -> 4: 98 us against 90 us; see the header there), so it only runs on request:
// EAS_SC_FORM=mfma (read per call so that a test can compare the two forms in one process).
bool conv_on_mfma(int k) {
    const char* e = getenv("EAS_SC_FORM");
    return e && e[0] == 'm' && k == 5;
}

}  // namespace

// smallconv_mfma.hip
int eas_sc5_mfma(const float* x, const float* w, const float* b, const float* mask, float* y, int N, int Cin, int Cout, int H, int W,
                 int relu, int dgrad, int x_terms, hipStream_t st);
// smallconv_wgrad_mfma.hip
int eas_sw_mfma_partials(const float* gy, const float* x, float* partial, int max_blocks, int N, int Cin, int Cout, int H, int W,
                         int k, hipStream_t st);

extern "C" {

int eas_smallconv_fwd(const float* x, const float* w, const float* b, float* y, int N, int Cin, int Cout, int H,
                      int W, int k, int relu, eas_stream_t stream) {
    if (!x || !w || !y || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (bad_ptr(y)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (conv_on_mfma(k)) {
        const int rc = eas_sc5_mfma(x, w, b, nullptr, y, N, Cin, Cout, H, W, relu, 0, 3, st);
        if (rc == EAS_OK) { EAS_CHECK_LAUNCH(); return EAS_OK; }
        if (rc != EAS_ERR_UNSUPPORTED) return rc;
    }
    if (Cin == 2 && Cout == 4) return launch_conv_k<2, 4, false>(k, x, w, b, nullptr, y, N, H, W, relu, st);
    if (Cin == 4 && Cout == 4) return launch_conv_k<4, 4, false>(k, x, w, b, nullptr, y, N, H, W, relu, st);
    if (Cin == 2 && Cout == 2) return launch_conv_k<2, 2, false>(k, x, w, b, nullptr, y, N, H, W, relu, st);
    return EAS_ERR_UNSUPPORTED;
}

int eas_smallconv_bwd_input(const float* grad_y, const float* w, const float* relu_mask, float* grad_x, int N, int Cin,
                            int Cout, int H, int W, int k, eas_stream_t stream) {
    if (!grad_y || !w || !grad_x || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (bad_ptr(grad_x) || bad_ptr(relu_mask)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (conv_on_mfma(k)) {
        const int rc = eas_sc5_mfma(grad_y, w, nullptr, relu_mask, grad_x, N, Cin, Cout, H, W, 0, 1, 3, st);
        if (rc == EAS_OK) { EAS_CHECK_LAUNCH(); return EAS_OK; }
        if (rc != EAS_ERR_UNSUPPORTED) return rc;
    }
    // the kernel's "input" is grad_y (Cout channels), its "output" grad_x (Cin channels)
    if (Cin == 2 && Cout == 4) return launch_conv_k<4, 2, true>(k, grad_y, w, nullptr, relu_mask, grad_x, N, H, W, 0, st);
    if (Cin == 4 && Cout == 4) return launch_conv_k<4, 4, true>(k, grad_y, w, nullptr, relu_mask, grad_x, N, H, W, 0, st);
    if (Cin == 2 && Cout == 2) return launch_conv_k<2, 2, true>(k, grad_y, w, nullptr, relu_mask, grad_x, N, H, W, 0, st);
    return EAS_ERR_UNSUPPORTED;
}

int64_t eas_smallconv_wgrad_workspace_floats(int Cin, int Cout, int k) {
    return (int64_t)kWgradBlocks * ((int64_t)Cout * Cin * k * k + Cout);
}

int eas_smallconv_bwd_weight(const float* grad_y, const float* x, float* grad_w, float* grad_b, float* workspace,
                             int N, int Cin, int Cout, int H, int W, int k, eas_stream_t stream) {
    if (!grad_y || !x || !grad_w || !workspace || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    int nblocks;
    if (wgrad_on_mfma(k)) {
        nblocks = eas_sw_mfma_partials(grad_y, x, workspace, kWgradBlocks, N, Cin, Cout, H, W, k, st);
        if (nblocks < 0) return nblocks;
    } else {
        const int tiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        nblocks = tiles < kWgradBlocks ? tiles : kWgradBlocks;
        int rc;
        if (Cin == 2 && Cout == 4) rc = launch_wgrad_k<2, 4>(k, grad_y, x, workspace, nblocks, N, H, W, st);
        else if (Cin == 4 && Cout == 4) rc = launch_wgrad_k<4, 4>(k, grad_y, x, workspace, nblocks, N, H, W, st);
        else if (Cin == 2 && Cout == 2) rc = launch_wgrad_k<2, 2>(k, grad_y, x, workspace, nblocks, N, H, W, st);
        else return EAS_ERR_UNSUPPORTED;
        if (rc != EAS_OK) return rc;
    }
    const int nw = Cout * Cin * k * k, nout = nw + Cout;
    hipLaunchKernelGGL(smallconv_wgrad_finalize, dim3(nout), dim3(EAS_WAVE), 0, st, workspace, nblocks, nout, nw,
                       grad_w, grad_b);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
