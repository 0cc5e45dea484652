// Input gradient of a stride-2 3x3 convolution with the four parity classes of the gx pixel in ONE tile (round 5).
//
// eas_conv_dgrad_s2 (conv_mfma.hip) used to run every parity class (ph, pw) of the gx pixel as a stride-1 tap-list convolution of its own:
// four tile families in one launch, each staging grad_y by itself -- the three-term split of a staged element (about ten vector
// instructions) then feeds ONE tap in class (0, 0) against nine in a forward tile -- and each writing every second pixel of every second
// row of gx, so that a 128-byte line of gx is put together by four blocks on four XCDs.  Measured (config 2, scripts/dev_conv_calls_table.py):
// 2.0x the time of a stride-1 input gradient of the same MFMA work.
//
// Here a block stages its grad_y patch once (rows a .. a + 1, columns b .. b + 1 of its positions, 16 channels per chunk, three terms) and
// walks the nine taps of a chunk like a forward 3x3 tile; tap t accumulates into the accumulators of its class (conv_tile_body NC = 4), a
// wave holds 4 classes x WN position tiles (WN = 2: 128 accumulator registers), and the epilogue writes gx[2a + ph][2b .. 2b + 1] as one
// 8-byte store per lane: whole lines from one block.  Per accumulator the products arrive in the order of the class kernels (k-step major,
// the class's taps, smallest term products first), so the result is bit-identical to theirs.  Weights: the mode-2 pack, unchanged.
//
// Tiles are ragged over the rows of an image (conv_mfma_body.h bpi) where whole images do not fill the 32 * WN * WVN positions of a block:
// 3 rows of 40 / 6 rows of 20 positions in a 128-position tile.
#include "conv_mfma_body.h"

namespace {

// VEC = 1: a staging item is ONE position x 8 channels (eight 4-byte loads, coalesced over the lanes' consecutive positions; one 16-byte LDS
// store per term), up to four items per thread: the conversion work of a chunk is spread over every lane of the block and over the last
// steps of the chunk instead of sitting in the first one or two waves behind the last step.
template <int VEC>
constexpr int s2c_nit() { return VEC == 1 ? 4 : 1; }

template <int WN, int WVM, int WVN, int VEC>
__global__ __launch_bounds__(64 * WVM * WVN, 2) void conv_dgrad_s2c_kernel(const float* __restrict__ gy, const bf16x8* __restrict__ wp,
                                                                             float* __restrict__ gx, const ConvGeomCore g) {
    extern __shared__ __align__(16) unsigned char smem[];
    conv_tile_body<9, 1, 3, 1, WN, WVM, WVN, 16, VEC, s2c_nit<VEC>(), false, 0, ConvGeomCore, 4>(gy, wp, nullptr, gx, nullptr, g, 0, smem,
                                                                                                (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

template <int WN, int WVM, int WVN, int VEC>
int launch_s2c(const float* gy, const bf16x8* wp, float* gx, ConvGeomCore g, hipStream_t st) {
    auto kern = conv_dgrad_s2c_kernel<WN, WVM, WVN, VEC>;
    const int nbuf = g.KSTEPS <= 1 ? 1 : 2;
    const size_t lds = (size_t)nbuf * g.Q * 16 * 2 * 3;
    if (lds > 160 * 1024) return EAS_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    if (g.total_rows >= (1 << 20) || g.nseg * g.rows_in * (g.Wst / VEC) >= (1 << 20)) return EAS_ERR_UNSUPPORTED;     // fdiv range
    conv_geom_magics(g, VEC, 16);
    const int ntiles = g.bpi ? g.NI * g.bpi : (g.total_rows + g.RT - 1) / g.RT;
    dim3 grid(ntiles, (g.MT + WVM - 1) / WVM, 1);
    EAS_LAUNCH(kern, grid, dim3(64 * WVM * WVN), lds, st, gy, wp, gx, g);
    return EAS_OK;
}

typedef int (*s2c_fn)(const float*, const bf16x8*, float*, ConvGeomCore, hipStream_t);
struct S2cCand { int wvm, wvn, wn, threads; s2c_fn f4, f2, f1; };
#define EAS_S2C(WVM_, WVN_, WN_) {WVM_, WVN_, WN_, 64 * WVM_ * WVN_, launch_s2c<WN_, WVM_, WVN_, 4>, launch_s2c<WN_, WVM_, WVN_, 2>, launch_s2c<WN_, WVM_, WVN_, 1>}
const S2cCand kS2c[7] = {EAS_S2C(2, 4, 2), EAS_S2C(4, 2, 2), EAS_S2C(2, 2, 2), EAS_S2C(4, 1, 2), EAS_S2C(1, 4, 2), EAS_S2C(4, 1, 1), EAS_S2C(2, 2, 1)};
#undef EAS_S2C

}  // namespace

// grad_y [NI, Cout, Ho, Wo] -> grad_x [NI, Cin, Hi, Wi]; packed_w: mode 2.  EAS_ERR_UNSUPPORTED: no tile (the caller falls back to the class
// kernels).  query: plan only.
int eas_conv_dgrad_s2c_dispatch(const float* gy, const void* packed_w, float* gx, int NI, int Cin, int Cout, int Hi, int Wi, hipStream_t st, bool query) {
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    if (Cout % 8 != 0 || Wo % 2 != 0) return EAS_ERR_UNSUPPORTED;
    static const int force_vec = eas_dev_env("EAS_S2C_VEC") ? atoi(eas_dev_env("EAS_S2C_VEC")) : 0;      // development: 2 = 8-byte staging loads everywhere
    // staging items of 2 positions x 8 channels where a chunk's items fit the block's threads (measured 3-6 % faster than 4 positions: more
    // lanes share the conversion work), else 4 positions, else single positions (four items per thread)
    const int vec_pref[3] = {2, Wo % 4 == 0 ? 4 : 0, 1};
    ConvGeomCore g{};
    g.NI = NI; g.Cin = Cout; g.Cout = Cin; g.Hi = Ho; g.Wi = Wo; g.Ho = Ho; g.Wo = Wo;
    g.RS = Wo + 1;                      // one zero column right of the image: position b + 1 of the last column
    g.pad_t = g.pad_l = 0;
    g.ext_h = 2;                        // rows a and a + 1
    for (int t = 0; t < 9; ++t) {       // class-major taps: class 2 * ph + pw, its taps (ih, iw) with ih <= ph, iw <= pw
        const int cls = kS2Cls[t], pw = cls & 1, lt = kS2Lt[t];
        const int ih = lt / (pw + 1), iw = lt - ih * (pw + 1);
        g.tap_off[t] = ih * g.RS + iw;
    }
    g.oH = Hi; g.oW = Wi; g.os = 2; g.oph = g.opw = 0;
    g.MT = (Cin + 31) / 32; g.KSTEPS = (Cout + 15) / 16;
    g.total_rows = NI * Ho;
    g.Wst = Wo; g.gx0 = 0; g.qshift = 0; g.parts = 1;
    static const int dbg_abl = eas_dev_env("EAS_CONV_DBG") ? atoi(eas_dev_env("EAS_CONV_DBG")) : 0;      // development: ablations (conv_mfma_body.h)
    g.dbg = dbg_abl;
    const int nbuf = g.KSTEPS <= 1 ? 1 : 2;
    int best = -1, best_vec = 0;
    double best_cost = 0.0;
    ConvGeomCore best_g = g;
    static const int force = eas_dev_env("EAS_S2C_TILE") ? atoi(eas_dev_env("EAS_S2C_TILE")) : -1;      // development: force a candidate
    for (int i = 0; i < 7; ++i) {
        const S2cCand& c = kS2c[i];
        if (force >= 0 && i != force) continue;
        if ((c.wvm - 1) * 32 >= Cin) continue;
        ConvGeomCore t = g;
        const int bn = 32 * c.wn * c.wvn;
        if (bn < Wo) continue;
        // whole images per tile where they fit, else RT rows of one image with a short last tile per image
        const int maxrt = bn / Wo;
        if (maxrt >= Ho) t.RT = pick_rows(Ho, Wo, bn);
        else {
            t.RT = maxrt;
            t.bpi = (Ho + t.RT - 1) / t.RT;
            t.RT = (Ho + t.bpi - 1) / t.bpi;       // the same tile count with evener tiles
        }
        if (t.RT == 0) continue;
        t.rows_seg = t.RT < Ho ? t.RT : Ho;
        t.nseg = t.RT / t.rows_seg;
        t.rows_in = (t.rows_seg - 1) + t.ext_h;
        t.Q = t.nseg * t.rows_in * t.RS;
        const size_t lds = (size_t)nbuf * t.Q * 16 * 2 * 3;
        if (lds > 160 * 1024) continue;
        int vec = 0;
        for (int k = 0; k < 3 && !vec; ++k) {
            const int v = vec_pref[k];
            if (v && (force_vec == 0 || force_vec == v) && Wo % v == 0 && t.nseg * t.rows_in * (Wo / v) * 2 <= (v == 1 ? 4 : 1) * c.threads) vec = v;
        }
        if (!vec) continue;
        const long tiles = t.bpi ? (long)NI * t.bpi : (g.total_rows + t.RT - 1) / t.RT;
        const long blocks = tiles * ((g.MT + c.wvm - 1) / c.wvm);
        const int bpc = c.threads == 512 ? 1 : (2 * lds <= 160 * 1024 ? 2 : 1);
        // a round of blocks: the MFMA work of a wave goes with WN whatever part of the tile is valid; 8-wave blocks and pairs of resident
        // 4-wave blocks cost more per round than a lone 4-wave block (the forward tiles' model, conv_mfma_body.h dispatch_tile)
        const double round_cost = (c.threads == 512 ? 1.27 : (bpc == 2 && blocks > 256 ? 1.2 : 1.0)) * (0.3 + 0.7 * c.wn / 2.0);
        const double cost = (double)((blocks + 256 * bpc - 1) / (256 * bpc)) * round_cost;
        if (best < 0 || cost < best_cost - 1e-9) {
            best = i; best_cost = cost; best_g = t; best_vec = vec;
        }
    }
    if (best < 0) return EAS_ERR_UNSUPPORTED;
    if (query) return EAS_OK;
    static const bool dbg = eas_dev_env("EAS_S2C_DBG") != nullptr;
    if (dbg)
        fprintf(stderr, "s2c tile: N %d %dx%d Cout %d -> Cin %d: cand %d (wvm %d wvn %d wn %d) vec %d RT %d bpi %d nseg %d Q %d\n", NI, Ho, Wo, Cout, Cin, best,
                kS2c[best].wvm, kS2c[best].wvn, kS2c[best].wn, best_vec, best_g.RT, best_g.bpi, best_g.nseg, best_g.Q);
    const s2c_fn fn = best_vec == 4 ? kS2c[best].f4 : (best_vec == 2 ? kS2c[best].f2 : kS2c[best].f1);
    return fn(gy, (const bf16x8*)packed_w, gx, best_g, st);
}
