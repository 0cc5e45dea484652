// Grouped (multi-problem) launches of the dense 3x3 convolution: ONE grid, a descriptor table of {geometry, packed weights, in / out
// pointers} in the kernel arguments, blockIdx.x -> (problem, pixel tile).  The three pyramid levels of the detection head run the same
// layers on 32x40, 16x20 and 8x10 maps and are independent until the loss (yolox/models/yolo_head.py:149-200, the
// `for k, (cls_conv, reg_conv, stride_this_level, x)` loop): as launches of their own the two small levels leave most of the 256 CUs
// without a block (64 images of 8x10 with 128 channels are 64 four-wave blocks) and still pay a kernel's whole prologue / epilogue; in the
// grid of the large level they are a few more blocks of the same kernel.  The tiles are conv_mfma_body.h's (same staging, same MFMA
// order: results bit-identical to eas_conv_fwd per problem); a block first looks its problem up, then runs the body instantiated for the
// problem's staging width (16-byte units where the rows are a multiple of 4 pixels, 8-byte units for the 10-pixel rows).
// Forward and -- with weights packed transposed + flipped (mode 1) and grad_y as x -- the stride-1 input gradient.
#include "conv_mfma_body.h"

namespace {

constexpr int kMaxGroup = 8;

template <typename G>
struct ConvGroupArgsT {
    G g[kMaxGroup];
    const float* x[kMaxGroup];
    const bf16x8* wp[kMaxGroup];
    const float* bias[kMaxGroup];
    float* y[kMaxGroup];
    int first[kMaxGroup + 1];      // problem p owns pixel tiles [first[p], first[p + 1]) of grid.x
    int vec2[kMaxGroup];           // 1: the problem's rows are staged in 2-pixel units (Wi % 4 != 0)
    int n;
};
typedef ConvGroupArgsT<ConvGeomCore> ConvGroupArgs;

// LM = 0: the plain epilogue (y, bias, BatchNorm partial sums); LM = 3: BatchNorm (running statistics) + activation on the accumulators
// (grouped eas_conv_bn_act_eval: the eval-mode head), G = the geometry type that carries what the epilogue reads
template <int XT, int WN, int WVM, int WVN, int NIT, int LM = 0, typename G = ConvGeomCore>
__global__ __launch_bounds__(64 * WVM * WVN, 2) void conv3x3_group_kernel(const ConvGroupArgsT<G> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int bx = (int)blockIdx.x - a.first[p], gx = a.first[p + 1] - a.first[p];
    if ((int)blockIdx.y * WVM >= a.g[p].MT) return;        // grid.y covers the widest problem: no channels left for this block (whole block exits)
    if (a.vec2[p])
        conv_tile_body<9, 1, XT, 1, WN, WVM, WVN, 16, 2, NIT, false, LM, G>(a.x[p], a.wp[p], a.bias[p], a.y[p], nullptr, a.g[p], 0, smem, bx,
                                                                            (int)blockIdx.y, gx);
    else
        conv_tile_body<9, 1, XT, 1, WN, WVM, WVN, 16, 4, NIT, false, LM, G>(a.x[p], a.wp[p], a.bias[p], a.y[p], nullptr, a.g[p], 0, smem, bx,
                                                                            (int)blockIdx.y, gx);
}

template <int XT, int WN, int WVM, int WVN, int NIT, int LM, typename G>
int launch_group3(const ConvGroupArgsT<G>& a, int grid_y, size_t lds, hipStream_t st) {
    auto kern = conv3x3_group_kernel<XT, WN, WVM, WVN, NIT, LM, G>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    EAS_LAUNCH(kern, dim3(a.first[a.n], grid_y), dim3(64 * WVM * WVN), lds, st, a);
    return EAS_OK;
}

struct GCand { int wvm, wvn, wn, nit; };
static const GCand kCands[4] = {{4, 1, 5, 2}, {2, 2, 5, 2}, {4, 1, 3, 2}, {2, 2, 3, 2}};

template <int XT, int LM, typename G>
int launch_cand3(int i, const ConvGroupArgsT<G>& a, int grid_y, size_t lds, hipStream_t st) {
    switch (i) {
        case 0: return launch_group3<XT, 5, 4, 1, 2, LM, G>(a, grid_y, lds, st);
        case 1: return launch_group3<XT, 5, 2, 2, 2, LM, G>(a, grid_y, lds, st);
        case 2: return launch_group3<XT, 3, 4, 1, 2, LM, G>(a, grid_y, lds, st);
        default: return launch_group3<XT, 3, 2, 2, 2, LM, G>(a, grid_y, lds, st);
    }
}

// geometry of one 3x3 stride-1 problem (padding 1), before a tile is chosen: conv_fwd_impl's set-up
void base_geom(ConvGeomCore& g, const EasConvProblem& q) {
    g.NI = q.NI; g.Cin = q.Cin; g.Cout = q.Cout; g.Hi = q.Hi; g.Wi = q.Wi;
    g.Ho = q.Hi; g.Wo = q.Wi;
    g.RS = q.Wi + 2;
    g.pad_t = g.pad_l = 1;
    g.ext_h = 3;
    for (int t = 0; t < 9; ++t) g.tap_off[t] = (t / 3) * g.RS + (t % 3);
    g.oH = g.Ho; g.oW = g.Wo; g.os = 1; g.oph = g.opw = 0;
    g.MT = (q.Cout + 31) / 32;
    g.KSTEPS = (q.Cin + 15) / 16;
    g.total_rows = q.NI * g.Ho;
    g.Wst = q.Wi; g.gx0 = 0; g.qshift = 0; g.parts = 1;
}

// the tile of candidate c for problem geometry g: the largest whole-row tile <= c's pixel count whose patch and staging items fit
template <int XT>
bool fit_tile(ConvGeomCore& t, const GCand& c, int vec) {
    constexpr int CCH = 16;
    const int nchunks = (t.KSTEPS * 16 + CCH - 1) / CCH;
    const int nbuf = nchunks == 1 ? 1 : 2;
    const int threads = 64 * c.wvm * c.wvn;
    for (int cap = 32 * c.wn * c.wvn; cap >= t.Wo; cap -= 32) {
        t.RT = pick_rows(t.Ho, t.Wo, cap);
        if (t.RT == 0) return false;
        t.rows_seg = t.RT < t.Ho ? t.RT : t.Ho;
        t.nseg = t.RT / t.rows_seg;
        t.rows_in = (t.rows_seg - 1) + t.ext_h;
        t.Q = t.nseg * t.rows_in * t.RS;
        if ((size_t)nbuf * t.Q * CCH * 2 * XT <= 160 * 1024 && t.nseg * t.rows_in * (t.Wst / vec) * (CCH / 8) <= c.nit * threads) return true;
        cap = t.RT * t.Wo;
    }
    return false;
}

// One tile shape for all problems of the group: the cost model of dispatch_tile (conv_mfma_body.h) over the SUM of the problems' blocks.
template <int XT, typename G>
int plan_group3(const EasConvProblem* pr, int n, ConvGroupArgsT<G>& a, int& best, int& grid_y, size_t& lds, int* nb_out) {
    const GCand* cands = kCands;
    static const int force = eas_dev_env("EAS_GROUP3_TILE") ? atoi(eas_dev_env("EAS_GROUP3_TILE")) : -1;      // development: force a candidate
    best = -1;
    double best_cost = 0.0;
    ConvGroupArgsT<G> best_a = a;
    for (int i = 0; i < 4; ++i) {
        if (force >= 0 && i != force) continue;
        const GCand& c = cands[i];
        int cout_min = 1 << 30;
        for (int p = 0; p < n; ++p) cout_min = pr[p].Cout < cout_min ? pr[p].Cout : cout_min;
        if ((c.wvm - 1) * 32 >= cout_min) continue;          // every wave row has channels to compute in every problem
        ConvGroupArgsT<G> t = a;
        bool ok = true;
        double load = 0.0;
        size_t lds_max = 0;
        int gy = 1, blocks_x = 0;
        for (int p = 0; p < n && ok; ++p) {
            G& g = t.g[p];
            base_geom(g, pr[p]);
            const int vec = pr[p].Wi % 4 == 0 ? 4 : 2;
            t.vec2[p] = vec == 2;
            ok = fit_tile<XT>(g, c, vec);
            if (!ok) break;
            const int nchunks = g.KSTEPS;
            const size_t l = (size_t)(nchunks == 1 ? 1 : 2) * g.Q * 16 * 2 * XT;
            lds_max = l > lds_max ? l : lds_max;
            const int bxp = (g.total_rows + g.RT - 1) / g.RT;
            const int byp = (g.MT + c.wvm - 1) / c.wvm;
            t.first[p] = blocks_x;
            blocks_x += bxp;
            gy = byp > gy ? byp : gy;
            load += (double)bxp * byp;
        }
        if (!ok) continue;
        t.first[n] = blocks_x;
        const int bpc = 2 * lds_max <= 160 * 1024 ? 2 : 1;
        const double per_tile = XT == 1 ? 96.0 : 192.0;
        const double mfma_part = (c.wn * per_tile > 450.0 ? c.wn * per_tile : 450.0) / (5.0 * per_tile);
        const double round_cost = (bpc == 2 && load > 256.0 ? 1.2 : 1.0) * (0.3 + 0.7 * mfma_part);
        double rounds = load / (256.0 * bpc);
        if (rounds < 1.0) rounds = 1.0;
        const double cost = rounds * round_cost;
        if (best < 0 || cost < best_cost - 1e-9) {
            best = i; best_cost = cost; best_a = t; grid_y = gy; lds = lds_max;
        }
    }
    if (best < 0) return EAS_ERR_UNSUPPORTED;
    a = best_a;
    const GCand& c = cands[best];
    const size_t stats_lds = (size_t)c.wvm * c.wvn * (EAS_STATS_SCRATCH * sizeof(float) + 64 * sizeof(double));
    bool any_stats = false;
    for (int p = 0; p < n; ++p) {
        G& g = a.g[p];
        const int gxp = a.first[p + 1] - a.first[p];
        if (nb_out) nb_out[p] = gxp;
        g.stats = pr[p].stats;
        g.stats_nb = gxp;
        any_stats = any_stats || pr[p].stats;
        if (g.total_rows >= (1 << 20) || g.nseg * g.rows_in * (g.Wst / (a.vec2[p] ? 2 : 4)) >= (1 << 20)) return EAS_ERR_UNSUPPORTED;     // fdiv range
        conv_geom_magics(g, a.vec2[p] ? 2 : 4, 16);
        a.x[p] = (const float*)pr[p].x; a.wp[p] = (const bf16x8*)pr[p].packed_w; a.bias[p] = pr[p].bias; a.y[p] = pr[p].y;
    }
    if (any_stats && lds < stats_lds) lds = stats_lds;
    a.n = n;
    return EAS_OK;
}

template <int XT>
int group3(const EasConvProblem* pr, int n, hipStream_t st, int* nb_out, bool query) {
    ConvGroupArgs a{};
    int best = -1, grid_y = 1;
    size_t lds = 0;
    const int rc = plan_group3<XT, ConvGeomCore>(pr, n, a, best, grid_y, lds, nb_out);
    if (rc != EAS_OK || query) return rc;
    return launch_cand3<XT, 0, ConvGeomCore>(best, a, grid_y, lds, st);
}

// grouped eas_conv_bn_act_eval, 3x3: the same plan, every problem's BatchNorm / activation descriptor in its geometry
int group3_bna(const EasConvProblem* pr, const EasBnActEpiDev* bna, int n, hipStream_t st) {
    ConvGroupArgsT<ConvGeomBna> a{};
    int best = -1, grid_y = 1;
    size_t lds = 0;
    const int rc = plan_group3<3, ConvGeomBna>(pr, n, a, best, grid_y, lds, nullptr);
    if (rc != EAS_OK) return rc;
    for (int p = 0; p < n; ++p) a.g[p].bna = bna[p];
    return launch_cand3<3, 3, ConvGeomBna>(best, a, grid_y, lds, st);
}

}  // namespace

int eas_conv1x1_group(const EasConvProblem* pr, int n, int x_terms, hipStream_t st, int* nb_out, bool query, const void* bna = nullptr);

static int conv_group_impl(const EasConvProblem* pr, int n, int ksize, int x_terms, eas_stream_t stream, int* nb_out, bool query) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (x_terms != 3 || (ksize != 1 && ksize != 3)) return EAS_ERR_UNSUPPORTED;      // real-valued fp32 inputs (the ANN neck / head and every input gradient)
    for (int p = 0; p < n; ++p) {
        const EasConvProblem& q = pr[p];
        if (q.NI <= 0 || q.Cin <= 0 || q.Cout <= 0 || q.Hi <= 0 || q.Wi <= 0) return EAS_ERR_INVALID_ARG;
        if (!query && (!q.x || !q.packed_w || !q.y)) return EAS_ERR_INVALID_ARG;
        if (q.stats && q.bias) return EAS_ERR_INVALID_ARG;
        if (ksize == 3 && (q.Cin % 8 != 0 || q.Wi % 2 != 0 || q.accumulate)) return EAS_ERR_UNSUPPORTED;
    }
    EAS_CLEAR_ERR();
    int rc;
    if (ksize == 1) rc = eas_conv1x1_group(pr, n, x_terms, eas_s(stream), nb_out, query);
    else if (n > kMaxGroup) rc = EAS_ERR_UNSUPPORTED;
    else rc = group3<3>(pr, n, eas_s(stream), nb_out, query);
    if (rc != EAS_OK || query) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

extern "C" {

// include/eas_hip.h: the tile plan of a group without launching anything -- nb_out[p] = statistics partials per channel problem p's
// blocks write (what EasConvProblem.stats must hold: Cout * nb * 2 doubles); EAS_ERR_UNSUPPORTED: run the problems one by one
int eas_conv_fwd_group_plan(const EasConvProblem* problems, int n, int ksize, int x_terms, int* nb_out) {
    return conv_group_impl(problems, n, ksize, x_terms, nullptr, nb_out, true);
}

int eas_conv_fwd_group(const EasConvProblem* problems, int n, int ksize, int x_terms, eas_stream_t stream) {
    return conv_group_impl(problems, n, ksize, x_terms, stream, nullptr, false);
}

// include/eas_hip.h: eas_conv_bn_act_eval for several layers in one launch (the eval-mode head: the three pyramid levels of a stage)
int eas_conv_bn_act_eval_group(const EasConvBnActEval* ds, int n, eas_stream_t stream) {
    if (!ds || n < 1) return EAS_ERR_INVALID_ARG;
    if (n > kMaxGroup) return EAS_ERR_UNSUPPORTED;
    EasConvProblem pr[kMaxGroup] = {};
    EasBnActEpiDev bna[kMaxGroup] = {};
    const int ksize = ds[0].ksize;
    for (int p = 0; p < n; ++p) {
        const EasConvBnActEval* d = &ds[p];
        if (!d->x || !d->packed_w || d->act < 0 || d->act > 1) return EAS_ERR_INVALID_ARG;
        if (d->ksize != ksize || d->stride != 1 || d->x_terms != 3 || (ksize != 1 && ksize != 3)) return EAS_ERR_UNSUPPORTED;
        if (d->Cout % 8 || d->Cin % 8 || d->csplit % 8 || d->csplit < 8 || d->csplit > d->Cout || (ksize == 3 && d->Wi % 2 != 0)) return EAS_ERR_UNSUPPORTED;
        EasBnActEpiDev& b = bna[p];
        b.on = 1; b.act = d->act; b.csplit = d->csplit; b.Cout = d->Cout;
        const int nr = d->csplit < d->Cout ? 2 : 1;
        for (int i = 0; i < nr; ++i) {
            const EasBnActRange& r = d->range[i];
            const int cr = i ? d->Cout - d->csplit : d->csplit;
            if (!r.gamma || !r.beta || !r.mean || !r.invstd || !r.out || r.out_c0 < 0 || r.out_c0 + cr > r.out_ctot) return EAS_ERR_INVALID_ARG;
            if ((((uintptr_t)r.gamma) | ((uintptr_t)r.beta) | ((uintptr_t)r.mean) | ((uintptr_t)r.invstd)) & 15) return EAS_ERR_INVALID_ARG;
            b.r[i] = r;
        }
        if (nr == 1) b.r[1] = b.r[0];
        pr[p].x = d->x; pr[p].packed_w = d->packed_w; pr[p].y = d->range[0].out;
        pr[p].NI = d->NI; pr[p].Cin = d->Cin; pr[p].Cout = d->Cout; pr[p].Hi = d->Hi; pr[p].Wi = d->Wi;
    }
    EAS_CLEAR_ERR();
    const int rc = ksize == 1 ? eas_conv1x1_group(pr, n, 3, eas_s(stream), nullptr, false, bna) : group3_bna(pr, bna, n, eas_s(stream));
    if (rc != EAS_OK) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
