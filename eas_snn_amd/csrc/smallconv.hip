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
#include "smallconv_core.h"

namespace {

// y[n][o] = (b[o]) + sum_{i,ky,kx} wr[i][ky][kx][o] * x[n][i][. + ky - PAD][. + kx - PAD], optional ReLU, optional mask (result zeroed
// where mask <= 0: the ReLU in front of the convolution whose input gradient this is).  DUAL: COUT = 2 * CO outputs of which the
// first CO go to y (masked by mask) and the rest to y2 (masked by mask2): two input gradients that share grad_y in one pass.
template <int CIN, int COUT, int K, bool DUAL, bool VECW>
__global__ __launch_bounds__(NT, 2) void smallconv_kernel(const float* __restrict__ x, const float* __restrict__ wr,
                                                          const float* __restrict__ b, const float* __restrict__ mask,
                                                          float* __restrict__ y, const float* __restrict__ mask2, float* __restrict__ y2, int N,
                                                          int H, int W, int relu, int x_tm) {
    using SG = Stage<CIN, K, VECW>;
    constexpr int LWF = SG::PITCH, PLANE = SG::PLANE;
    constexpr int CO = DUAL ? COUT / 2 : COUT;
    using St = typename SG::St;
    St st;
    st.init(H, W);
    __shared__ __attribute__((aligned(16))) float lds[CIN * PLANE];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    typename SG::Pre pre[St::PER];
    unsigned okm = 0;
    int tile = blockIdx.x;
    const int xS = x_tm ? N / x_tm : 0;                // x given as collated micro-slices (sc_src_image)
    if (tile < ntiles)
        okm = st.load(x, pre, sc_src_image(tile / (tiles_x * tiles_y), xS, x_tm), H, W, ((tile / tiles_x) % tiles_y) * TH, (tile % tiles_x) * TW);
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int ty0 = ((tile / tiles_x) % tiles_y) * TH, tx0 = (tile % tiles_x) * TW;
        __syncthreads();                               // readers of the previous tile are done
        st.store(lds, pre, okm);
        __syncthreads();
        const int nxt = tile + gridDim.x;              // next tile's loads fly during this tile's FMAs
        if (nxt < ntiles)
            okm = st.load(x, pre, sc_src_image(nxt / (tiles_x * tiles_y), xS, x_tm), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
        f2 acc[4][COUT / 2];
#pragma unroll
        for (int op = 0; op < COUT / 2; ++op) {
            const f2 bv = b ? f2{b[2 * op], b[2 * op + 1]} : f2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][op] = bv;
        }
        const int oy = ty0 + ty, ox = tx0 + 4 * tx;
        const bool inside = oy < H && ox < W, vec = ox + 3 < W && (W & 3) == 0;
        // the ReLU masks of the thread's outputs are requested in front of the arithmetic, like the next tile (vector form only)
        float4 mk4[COUT];
        if (mask && inside && vec) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const bool second = DUAL && o >= CO;
                mk4[o] = *reinterpret_cast<const float4*>((second ? mask2 : mask) + ((((int64_t)n * CO + (second ? o - CO : o)) * H + oy) * W + ox));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sc_accumulate<CIN, COUT, K, LWF, SG::NQ, SG::OFS>(lds, PLANE, wr, acc, tx, ty);
        if (inside) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const bool second = DUAL && o >= CO;
                const int oc = second ? o - CO : o;
                const int64_t off = (((int64_t)n * CO + oc) * H + oy) * W + ox;
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = (o & 1) ? acc[j][o >> 1].y : acc[j][o >> 1].x;
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = fmaxf(r[j], 0.f);
                }
                const float* mk = second ? mask2 : mask;
                float* dst = second ? y2 : y;
                if (vec) {
                    if (mk) {
                        const float4 m = mk4[o];
                        r[0] = m.x > 0.f ? r[0] : 0.f; r[1] = m.y > 0.f ? r[1] : 0.f;
                        r[2] = m.z > 0.f ? r[2] : 0.f; r[3] = m.w > 0.f ? r[3] : 0.f;
                    }
                    *reinterpret_cast<float4*>(dst + off) = make_float4(r[0], r[1], r[2], r[3]);
                } else {
                    for (int j = 0; j < 4 && ox + j < W; ++j) dst[off + j] = (mk && !(mk[off + j] > 0.f)) ? 0.f : r[j];
                }
            }
        }
    }
}

// wr[ci][ky][kx][o_total] from w[Cout][Cin][k][k]: mode 0 (forward) wr[i][ky][kx][o_off + o] = w[o][i][ky][kx];
// mode 1 (input gradient: the kernel's input channels are the convolution's OUTPUT channels) wr[co][ky][kx][o_off + ci] =
// w[co][ci][k-1-ky][k-1-kx].  Up to 8 tensors per launch (jobs by value: graph-capturable, no device table).
constexpr int kPackJobs = 8;
struct PackJobs {
    const float* w[kPackJobs];
    float* wr[kPackJobs];
    int Cin[kPackJobs], Cout[kPackJobs], k[kPackJobs], mode[kPackJobs], o_total[kPackJobs], o_off[kPackJobs];
};

__global__ void smallconv_pack_kernel(const PackJobs jobs) {
    const int j = blockIdx.x;
    const float* __restrict__ w = jobs.w[j];
    float* __restrict__ wr = jobs.wr[j];
    const int Cin = jobs.Cin[j], Cout = jobs.Cout[j], k = jobs.k[j], mode = jobs.mode[j], ot = jobs.o_total[j], oo = jobs.o_off[j];
    const int ni = mode ? Cout : Cin, no = mode ? Cin : Cout;
    for (int e = threadIdx.x; e < ni * k * k * no; e += blockDim.x) {
        const int o = e % no;
        int rest = e / no;
        const int kx = rest % k; rest /= k;
        const int ky = rest % k;
        const int i = rest / k;
        const float v = mode ? w[((i * Cin + o) * k + (k - 1 - ky)) * k + (k - 1 - kx)] : w[((o * Cin + i) * k + ky) * k + kx];
        wr[((i * k + ky) * k + kx) * ot + oo + o] = v;
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
                                                             float* __restrict__ partial, int N, int H, int W, int x_tm) {
    using G = Geo<K>;
    constexpr int GROUPS = NT / CIN;
    const int xS = x_tm ? N / x_tm : 0;           // x given as collated micro-slices (sc_src_image)
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
    unsigned okm = 0;
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
        okm = st.load(x, pre, sc_src_image(tile / (tiles_x * tiles_y), xS, x_tm), H, W, ((tile / tiles_x) % tiles_y) * TH, (tile % tiles_x) * TW);
        load_gy(tile);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                           // previous tile's readers are done
        st.store(lx, pre, okm);
#pragma unroll
        for (int it = 0; it < GPER; ++it) {
            const int idx = threadIdx.x + it * NT;
            if (idx < LG / 4) reinterpret_cast<float4*>(lg)[idx] = preg[it];
        }
        __syncthreads();
        const int nxt = tile + gridDim.x;          // next tile's loads fly during this tile's FMAs
        if (nxt < ntiles) {
            okm = st.load(x, pre, sc_src_image(nxt / (tiles_x * tiles_y), xS, x_tm), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
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

template <int CIN, int COUT, bool DUAL>
int launch_conv_k(int k, const float* x, const float* wr, const float* b, const float* mask, float* y, const float* mask2, float* y2, int N,
                  int H, int W, int relu, hipStream_t st, int x_tm = 0) {
    int tiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (tiles > kConvBlocks) tiles = kConvBlocks;      // persistent blocks loop over tiles with register prefetch
    const bool vecw = (W & 3) == 0 && (((uintptr_t)x) & 15) == 0;
#define EAS_SC(K_)                                                                                                                             \
    do {                                                                                                                                       \
        if (vecw) EAS_LAUNCH((smallconv_kernel<CIN, COUT, K_, DUAL, true>), dim3(tiles), dim3(NT), 0, st, x, wr, b, mask, y, mask2, y2, \
                                     N, H, W, relu, x_tm);                                                                                     \
        else EAS_LAUNCH((smallconv_kernel<CIN, COUT, K_, DUAL, false>), dim3(tiles), dim3(NT), 0, st, x, wr, b, mask, y, mask2, y2, N,  \
                                H, W, relu, x_tm);                                                                                             \
    } while (0)
    switch (k) {
        case 3: EAS_SC(3); break;
        case 5: EAS_SC(5); break;
        case 7: EAS_SC(7); break;
        default: return EAS_ERR_UNSUPPORTED;
    }
#undef EAS_SC
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

template <int CIN, int COUT>
int launch_wgrad_k(int k, const float* gy, const float* x, float* partial, int nblocks, int N, int H, int W, int x_tm, hipStream_t st) {
    switch (k) {
        case 3: EAS_LAUNCH((smallconv_wgrad_kernel<CIN, COUT, 3>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W, x_tm); break;
        case 5: EAS_LAUNCH((smallconv_wgrad_kernel<CIN, COUT, 5>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W, x_tm); break;
        case 7: EAS_LAUNCH((smallconv_wgrad_kernel<CIN, COUT, 7>), dim3(nblocks), dim3(NT), 0, st, gy, x, partial, N, H, W, x_tm); break;
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

}  // namespace

// smallconv_wgrad_mfma.hip
int eas_sw_mfma_partials(const float* gy, const float* x, float* partial, int max_blocks, int N, int Cin, int Cout, int H, int W,
                         int k, int x_tm, hipStream_t st);

extern "C" {

int64_t eas_smallconv_packed_floats(int n_in, int k, int o_total) { return (int64_t)n_in * k * k * o_total; }

int eas_smallconv_pack_weights(const EasSmallconvPackJob* jobs, int njobs, eas_stream_t stream) {
    if (!jobs || njobs < 1 || njobs > kPackJobs) return EAS_ERR_INVALID_ARG;
    PackJobs pj{};
    for (int j = 0; j < njobs; ++j) {
        const EasSmallconvPackJob& q = jobs[j];
        const int no = q.mode ? q.Cin : q.Cout;
        if (!q.w || !q.wr || q.Cin < 1 || q.Cout < 1 || (q.k != 3 && q.k != 5 && q.k != 7) || (q.mode != 0 && q.mode != 1) || q.o_off < 0 ||
            q.o_off + no > q.o_total || ((uintptr_t)q.wr & 15))
            return EAS_ERR_INVALID_ARG;
        pj.w[j] = q.w; pj.wr[j] = q.wr; pj.Cin[j] = q.Cin; pj.Cout[j] = q.Cout; pj.k[j] = q.k; pj.mode[j] = q.mode;
        pj.o_total[j] = q.o_total; pj.o_off[j] = q.o_off;
    }
    EAS_CLEAR_ERR();
    EAS_LAUNCH(smallconv_pack_kernel, dim3(njobs), dim3(256), 0, eas_s(stream), pj);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_smallconv_fwd(const float* x, const float* wr, const float* b, float* y, int N, int Cin, int Cout, int H,
                      int W, int k, int relu, int x_tm, eas_stream_t stream) {
    if (!x || !wr || !y || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (bad_ptr(y) || bad_ptr(wr) || x_tm < 0 || (x_tm > 0 && N % x_tm != 0)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (Cin == 2 && Cout == 4) return launch_conv_k<2, 4, false>(k, x, wr, b, nullptr, y, nullptr, nullptr, N, H, W, relu, st, x_tm);
    if (Cin == 4 && Cout == 4) return launch_conv_k<4, 4, false>(k, x, wr, b, nullptr, y, nullptr, nullptr, N, H, W, relu, st, x_tm);
    if (Cin == 2 && Cout == 2) return launch_conv_k<2, 2, false>(k, x, wr, b, nullptr, y, nullptr, nullptr, N, H, W, relu, st, x_tm);
    return EAS_ERR_UNSUPPORTED;
}

int eas_smallconv_bwd_input(const float* grad_y, const float* wr, const float* relu_mask, float* grad_x, int N, int Cin,
                            int Cout, int H, int W, int k, eas_stream_t stream) {
    if (!grad_y || !wr || !grad_x || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (bad_ptr(grad_x) || bad_ptr(relu_mask) || bad_ptr(wr)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    // the kernel's "input" is grad_y (Cout channels), its "output" grad_x (Cin channels)
    if (Cin == 2 && Cout == 4) return launch_conv_k<4, 2, false>(k, grad_y, wr, nullptr, relu_mask, grad_x, nullptr, nullptr, N, H, W, 0, st);
    if (Cin == 4 && Cout == 4) return launch_conv_k<4, 4, false>(k, grad_y, wr, nullptr, relu_mask, grad_x, nullptr, nullptr, N, H, W, 0, st);
    if (Cin == 2 && Cout == 2) return launch_conv_k<2, 2, false>(k, grad_y, wr, nullptr, relu_mask, grad_x, nullptr, nullptr, N, H, W, 0, st);
    return EAS_ERR_UNSUPPORTED;
}

// two input gradients of the same grad_y [N][4][H][W] in one pass: grad_xa = mask_a (.) convT(w_a, grad_y), grad_xb = mask_b (.) convT(w_b,
// grad_y) for two 4 -> 4 convolutions; wr = both weight tensors packed with mode 1, o_total 8, o_off 0 / 4
int eas_smallconv_bwd_input_dual(const float* grad_y, const float* wr, const float* mask_a, const float* mask_b, float* grad_xa,
                                 float* grad_xb, int N, int H, int W, int k, eas_stream_t stream) {
    if (!grad_y || !wr || !grad_xa || !grad_xb || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (bad_ptr(grad_xa) || bad_ptr(grad_xb) || bad_ptr(mask_a) || bad_ptr(mask_b) || bad_ptr(wr)) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    return launch_conv_k<4, 8, true>(k, grad_y, wr, nullptr, mask_a, grad_xa, mask_b, grad_xb, N, H, W, 0, eas_s(stream));
}

int64_t eas_smallconv_wgrad_workspace_floats(int Cin, int Cout, int k) {
    return (int64_t)kWgradBlocks * ((int64_t)Cout * Cin * k * k + Cout);
}

int eas_smallconv_bwd_weight(const float* grad_y, const float* x, float* grad_w, float* grad_b, float* workspace,
                             int N, int Cin, int Cout, int H, int W, int k, int x_tm, eas_stream_t stream) {
    if (!grad_y || !x || !grad_w || !workspace || N < 1 || H < 1 || W < 1) return EAS_ERR_INVALID_ARG;
    if (x_tm < 0 || (x_tm > 0 && N % x_tm != 0)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    int nblocks;
    if (wgrad_on_mfma(k)) {
        nblocks = eas_sw_mfma_partials(grad_y, x, workspace, kWgradBlocks, N, Cin, Cout, H, W, k, x_tm, st);
        if (nblocks < 0) return nblocks;
    } else {
        const int tiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        nblocks = tiles < kWgradBlocks ? tiles : kWgradBlocks;
        int rc;
        if (Cin == 2 && Cout == 4) rc = launch_wgrad_k<2, 4>(k, grad_y, x, workspace, nblocks, N, H, W, x_tm, st);
        else if (Cin == 4 && Cout == 4) rc = launch_wgrad_k<4, 4>(k, grad_y, x, workspace, nblocks, N, H, W, x_tm, st);
        else if (Cin == 2 && Cout == 2) rc = launch_wgrad_k<2, 2>(k, grad_y, x, workspace, nblocks, N, H, W, x_tm, st);
        else return EAS_ERR_UNSUPPORTED;
        if (rc != EAS_OK) return rc;
    }
    const int nw = Cout * Cin * k * k, nout = nw + Cout;
    EAS_LAUNCH(smallconv_wgrad_finalize, dim3(nout), dim3(EAS_WAVE), 0, st, workspace, nblocks, nout, nw,
                       grad_w, grad_b);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
