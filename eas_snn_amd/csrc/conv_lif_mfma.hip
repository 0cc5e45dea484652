// The fused eval-mode step  conv -> BatchNorm (running statistics) -> (P)LIF over T  in ONE kernel (eas_conv_bn_lif_eval, include/eas_hip.h):
// BaseConv.forward of a converted block in eval mode, yolox/models/network_blocks.py:52-53 after yolox/utils/utils_snn.py:16-58.
// The 3x3 kernels are conv_mfma_body.h's conv_tile_body with the neuron epilogue of conv_lif_epi.h (template LM); this file holds their
// tile choice and the entry points (the 1x1 layers: conv1x1_mfma.hip).  A translation unit of its own so that it compiles beside conv_mfma.hip.
#include "conv_mfma_body.h"

namespace {

// Tile choice of the fused eval step (LM = 1: time-major, the wave tile is lif.T time steps of 32 pixels; LM = 2: one result for all steps).
// Same cost model as dispatch_tile; the block covers 32 * WVN spatial pixels (time-major) whose T time slices are staged side by side.
template <int TAPS, int S, int XT, int CCH, int VEC, bool PL, int LM>
int dispatch_tile_lif(const float* x, const bf16x8* wp, ConvGeom g, hipStream_t st, bool query) {
    struct Cand { int wvm, wvn, threads, nit, wn; launch_fn fn; };
    constexpr int N8 = XT == 1 ? 2 : 1, N4 = 2;
#define EAS_LC(WVM_, WVN_, WN_, NIT_) {WVM_, WVN_, 64 * WVM_ * WVN_, NIT_, WN_, launch_fwd<TAPS, S, XT, 1, WN_, WVM_, WVN_, CCH, VEC, NIT_, PL, LM>}
    // (only the candidates of this LM are instantiated)
    constexpr int ncand = LM == 1 ? 15 : 4;
    Cand cands[ncand];
    if constexpr (LM == 1) {
        const Cand c[15] = {EAS_LC(2, 4, 6, N8), EAS_LC(4, 2, 6, N8), EAS_LC(1, 8, 6, N8), EAS_LC(1, 4, 6, N4), EAS_LC(2, 2, 6, N4),
                            EAS_LC(2, 4, 3, N8), EAS_LC(4, 2, 3, N8), EAS_LC(1, 8, 3, N8), EAS_LC(1, 4, 3, N4), EAS_LC(2, 2, 3, N4),
                            EAS_LC(2, 4, 5, N8), EAS_LC(4, 2, 5, N8), EAS_LC(1, 8, 5, N8), EAS_LC(1, 4, 5, N4), EAS_LC(2, 2, 5, N4)};
        for (int i = 0; i < ncand; ++i) cands[i] = c[i];
    } else {
        const Cand c[4] = {EAS_LC(2, 4, 5, N8), EAS_LC(4, 2, 5, N8), EAS_LC(1, 4, 5, N4), EAS_LC(2, 2, 5, N4)};
        for (int i = 0; i < ncand; ++i) cands[i] = c[i];
    }
#undef EAS_LC
    const int T = g.lif.T;
    const int nchunks = (g.KSTEPS * 16 + CCH - 1) / CCH;
    int best = -1;
    double best_cost = 0.0;
    int best_rank = 0;
    ConvGeom best_g = g;
    static const int force = eas_dev_env("EAS_LIF_TILE") ? atoi(eas_dev_env("EAS_LIF_TILE")) : -1;      // development: force a candidate
    for (int i = 0; i < ncand; ++i) {
        const Cand& c = cands[i];
        if (force >= 0 && i != force) continue;
        if (LM == 1 && c.wn % T != 0) continue;          // the wave tile holds the T steps of one or two spatial tiles
        const int m2 = LM == 1 ? c.wn / T : 1;
        if ((c.wvm - 1) * 32 >= g.Cout && c.wvm != 1) continue;
        const int nbuf = nchunks == 1 ? 1 : 2;
        ConvGeom t = g;
        t.single = 0;
        const int bn = LM == 1 ? 32 * m2 * c.wvn : 32 * c.wn * c.wvn;  // spatial pixels of the block tile
        bool fits = false;
        for (int cap = bn; cap >= g.Wo && !fits; cap -= 32) {
            // whole images per tile where they fit, else RT rows of one image with a short last tile per image (ragged: any RT)
            const int maxrt = cap / g.Wo;
            t.bpi = 0;
            if (maxrt >= g.Ho) t.RT = pick_rows(g.Ho, g.Wo, cap);
            else {
                t.RT = maxrt;
                t.bpi = (g.Ho + t.RT - 1) / t.RT;
                t.RT = (g.Ho + t.bpi - 1) / t.bpi;       // the same tile count with evener tiles
            }
            if (t.RT == 0) break;
            t.rows_seg = t.RT < g.Ho ? t.RT : g.Ho;
            t.nseg_s = t.RT / t.rows_seg;
            t.nseg = (LM == 1 ? T : 1) * t.nseg_s;
            t.rows_in = (t.rows_seg - 1) * S + g.ext_h;
            t.Q = t.nseg * t.rows_in * t.RS;
            fits = (size_t)nbuf * t.Q * CCH * 2 * XT <= 160 * 1024 && t.nseg * t.rows_in * (g.Wst / VEC) * (CCH / 8) <= c.nit * c.threads;
            cap = t.RT * g.Wo;
        }
        if (!fits) continue;
        const long blocks = (long)(t.bpi ? g.NI * t.bpi : (g.total_rows + t.RT - 1) / t.RT) * ((g.MT + c.wvm - 1) / c.wvm);
        const size_t lds_bytes = (size_t)nbuf * t.Q * CCH * 2 * XT;
        const int bpc = c.threads == 512 ? 1 : (2 * lds_bytes <= 160 * 1024 ? 2 : 1);
        const double per_tile = XT == 1 ? 96.0 : 192.0, lat = 450.0;
        const double mfma_part = (c.wn * per_tile > lat ? c.wn * per_tile : lat) / (5.0 * per_tile);
        const double round_cost = (c.threads == 512 ? 1.27 : (bpc == 2 && blocks > 256 ? 1.2 : 1.0)) * (0.3 + 0.7 * mfma_part);
        // the valid share of the pixel tile matters here (a 32 * WVN pixel block of whole rows): rounds x cost / valid pixels per block
        const int valid = t.RT * g.Wo;
        const double cost = (double)((blocks + 256 * bpc - 1) / (256 * bpc)) * round_cost;
        const int rank = c.wvm * 4096 + (valid < 4096 ? valid : 4095);
        if (best < 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && rank > best_rank)) {
            best = i; best_cost = cost; best_rank = rank; best_g = t;
        }
    }
    if (best < 0) return EAS_ERR_UNSUPPORTED;
    if (query) return EAS_OK;
    static const bool dbg = eas_dev_env("EAS_LIF_DBG") != nullptr;
    if (dbg)
        fprintf(stderr, "lif tile: LM %d T %d N %d %dx%d Cin %d Cout %d S %d -> cand %d (wvm %d wvn %d wn %d) RT %d nseg %d Q %d lds %zu blocks %ld\n", LM, T, g.NI,
                g.Ho, g.Wo, g.Cin, g.Cout, S, best, cands[best].wvm, cands[best].wvn, cands[best].wn, best_g.RT, best_g.nseg, best_g.Q,
                (size_t)(nchunks == 1 ? 1 : 2) * best_g.Q * CCH * 2 * XT,
                (long)(best_g.bpi ? g.NI * best_g.bpi : (g.total_rows + best_g.RT - 1) / best_g.RT) * ((g.MT + cands[best].wvm - 1) / cands[best].wvm));
    return cands[best].fn(x, wp, nullptr, nullptr, nullptr, best_g, st);
}

}  // namespace

int eas_conv1x1_lif_dispatch(const void* x, const void* packed_w, const EasLifEpiDev& lif, int Cin, int x_terms, int x_shared, hipStream_t st, bool query);

// 3x3 convolution whose epilogue is BatchNorm (running statistics) + activation (eas_conv_bn_act_eval, conv_mfma.hip conv_fwd_impl): the
// plain forward's tile choice (dispatch_tile) over kernels instantiated with LM = 3, fp32 three-term input.  geom: the ConvGeom conv_fwd_impl
// prepared (g.bna filled in).
int eas_conv3x3_bna_dispatch(int stride, int v4, const float* x, const void* wp, float* y, const void* geom, hipStream_t st) {
    const ConvGeom& g = *(const ConvGeom*)geom;
    const bf16x8* w = (const bf16x8*)wp;
    if (stride == 1) return v4 ? dispatch_tile<9, 1, 3, 16, 4, false, 3>(x, w, nullptr, y, nullptr, g, st) : dispatch_tile<9, 1, 3, 16, 2, false, 3>(x, w, nullptr, y, nullptr, g, st);
    return v4 ? dispatch_tile<9, 2, 3, 16, 4, false, 3>(x, w, nullptr, y, nullptr, g, st) : dispatch_tile<9, 2, 3, 16, 2, false, 3>(x, w, nullptr, y, nullptr, g, st);
}

extern "C" {

// Fused eval-mode step conv -> BatchNorm (running statistics) -> LIF over T (include/eas_hip.h).  3x3: this file; 1x1: conv1x1_mfma.hip.
static int conv_lif_impl(const EasConvBnLifEval* d, eas_stream_t stream, bool query) {
    if (!d) return EAS_ERR_INVALID_ARG;
    const int T = d->T, N = d->N, Cin = d->Cin, Cout = d->Cout, Hi = d->Hi, Wi = d->Wi, ksize = d->ksize, stride = d->stride;
    if (T < 1 || T > 8 || N < 1 || Cin < 1 || Cout < 1 || Hi < 1 || Wi < 1) return EAS_ERR_INVALID_ARG;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (ksize == 1 && stride != 1) || Cin % 8 != 0 || Cout % 8 != 0 ||
        d->csplit % 8 != 0 || d->csplit < 8 || d->csplit > Cout)
        return EAS_ERR_UNSUPPORTED;
    // inputs: spike planes with distinct frames per step (T = 3 or 5: the wave tile holds the T steps), or one fp32 frame set shared by all steps
    const bool shared = d->x_shared != 0;
    if (shared ? (d->x_terms != 1 && d->x_terms != 3) : (d->x_terms != 2 || (T != 3 && T != 5))) return EAS_ERR_UNSUPPORTED;
    const int pad = ksize / 2;
    const int Ho = (Hi + 2 * pad - ksize) / stride + 1, Wo = (Wi + 2 * pad - ksize) / stride + 1;
    if ((Ho * Wo) % 4 != 0 || (ksize == 3 && Wi % 2 != 0)) return EAS_ERR_UNSUPPORTED;
    EasLifEpiDev lif{};
    lif.T = T; lif.N = N; lif.HW = Ho * Wo; lif.Cout = Cout; lif.csplit = d->csplit;
    lif.r[0] = d->range[0];
    lif.r[1] = d->range[1];
    if (!query) {
        if (!d->x || !d->packed_w) return EAS_ERR_INVALID_ARG;
        if ((uintptr_t)d->x & 15) return EAS_ERR_INVALID_ARG;
        for (int i = 0; i < (d->csplit < Cout ? 2 : 1); ++i) {
            const EasLifRange& r = lif.r[i];
            if ((!r.planes) == (!r.out_f32) || (r.res_planes && r.res_f32) || !r.gamma || !r.beta || !r.mean || !r.invstd) return EAS_ERR_INVALID_ARG;
            if (r.out_ctot % 8 != 0 || r.out_c0 % 8 != 0 || ((r.res_planes || r.res_f32) && r.res_ctot % 8 != 0)) return EAS_ERR_UNSUPPORTED;
            if (((uintptr_t)r.planes | (uintptr_t)r.res_planes) & 15) return EAS_ERR_INVALID_ARG;
        }
    }
    EAS_CLEAR_ERR();
    hipStream_t st = eas_s(stream);
    int rc = EAS_ERR_UNSUPPORTED;
    if (ksize == 1) {
        rc = eas_conv1x1_lif_dispatch(d->x, d->packed_w, lif, Cin, d->x_terms, d->x_shared, st, query);
    } else {
        ConvGeom g{};
        g.NI = N; g.Cin = Cin; g.Cout = Cout; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo;
        g.RS = Wi + 2 * pad;
        g.pad_t = g.pad_l = pad;
        g.ext_h = ksize;
        for (int t = 0; t < 9; ++t) g.tap_off[t] = (t / 3) * g.RS + (t % 3);
        g.oH = Ho; g.oW = Wo; g.os = 1; g.oph = g.opw = 0;
        g.MT = (Cout + 31) / 32;
        g.KSTEPS = (Cin + 15) / 16;
        g.total_rows = N * Ho;
        g.Wst = Wi; g.gx0 = 0; g.qshift = 0; g.parts = 1;
        g.lif = lif;
        static const int dbg = eas_dev_env("EAS_CONV_DBG") ? atoi(eas_dev_env("EAS_CONV_DBG")) : 0;
        g.dbg = dbg;
        const bf16x8* wp = (const bf16x8*)d->packed_w;
        const float* x = (const float*)d->x;
        const bool v4 = Wi % 4 == 0;
#define EAS_LIF_DISPATCH(S_)                                                                                                           \
    rc = !shared ? (v4 ? dispatch_tile_lif<9, S_, 1, 16, 4, true, 1>(x, wp, g, st, query) : dispatch_tile_lif<9, S_, 1, 16, 2, true, 1>(x, wp, g, st, query)) \
       : d->x_terms == 1 ? (v4 ? dispatch_tile_lif<9, S_, 1, 16, 4, false, 2>(x, wp, g, st, query) : dispatch_tile_lif<9, S_, 1, 16, 2, false, 2>(x, wp, g, st, query)) \
                         : (v4 ? dispatch_tile_lif<9, S_, 3, 16, 4, false, 2>(x, wp, g, st, query) : dispatch_tile_lif<9, S_, 3, 16, 2, false, 2>(x, wp, g, st, query))
        if (stride == 1) { EAS_LIF_DISPATCH(1); }
        else { EAS_LIF_DISPATCH(2); }
#undef EAS_LIF_DISPATCH
    }
    if (rc != EAS_OK || query) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_conv_bn_lif_eval(const EasConvBnLifEval* d, eas_stream_t stream) { return conv_lif_impl(d, stream, false); }

int eas_conv_bn_lif_eval_supported(int T, int N, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms, int x_shared) {
    EasConvBnLifEval d{};
    d.T = T; d.N = N; d.Cin = Cin; d.Cout = Cout; d.Hi = Hi; d.Wi = Wi; d.ksize = ksize; d.stride = stride; d.x_terms = x_terms; d.x_shared = x_shared;
    d.csplit = Cout;
    return conv_lif_impl(&d, nullptr, true) == EAS_OK ? 1 : 0;
}

}  // extern "C"
