// Shared body of the implicit-GEMM 3x3 convolution kernels (conv_mfma.hip: forward / input gradients; conv_lif_mfma.hip: the fused eval
// step conv -> BatchNorm -> LIF): geometry, LDS staging, the MFMA main loop, the epilogues and the launcher.  See conv_mfma.hip for the
// design notes.  Everything lives in an anonymous namespace: each translation unit instantiates the kernels it dispatches to.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "conv_lif_epi.h"
#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// staging loads of rows outside the image / channels past Cin read these zeros (channel stride 0) instead of being masked
__device__ __attribute__((aligned(16))) float eas_conv_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

// ConvGeomCore: everything the plain tiles read.  ConvGeom adds the descriptors of the fused eval epilogues (624 bytes together); the
// grouped launches (conv_group.hip) carry up to eight ConvGeomCore by value in their kernel arguments.
struct ConvGeomCore {
    int NI, Cin, Cout, Hi, Wi, Ho, Wo;
    int RT;        // output rows (flattened over images) per block tile
    int rows_seg;  // output rows per image segment inside a tile = min(RT, Ho)
    int nseg;      // image segments per tile = RT / rows_seg
    int rows_in;   // input rows staged per segment = (rows_seg-1)*S + ext_h
    int RS;        // staged row stride in pixels = Wi + 2*pad
    int Q;         // staged input pixels per tile = nseg*rows_in*RS
    int MT;        // ceil(Cout/32)
    int KSTEPS;    // ceil(Cin/16)
    int total_rows;
    int pad_t, pad_l;   // rows above / columns left of the image in the staged patch (zero halo)
    int ext_h;          // rows spanned by the taps (3 for a 3x3 kernel)
    int tap_off[9];     // staged-pixel offset (dh*RS + dw) of every tap
    int oH, oW, os, oph, opw;   // output tensor height/width, output stride and phase: y[.., orow*os+oph, ocol*os+opw]
    // Column parts (rows too wide for one LDS patch: 3x3 layers on 320-pixel rows with real-valued inputs): blockIdx.z = part, a
    // tile covers output columns [part*Wo, (part+1)*Wo) of rows that are parts*Wo wide (Wo is the TILE width everywhere in the
    // kernel).  The staged row then holds input columns gx0 + part*Wo*S .. + Wst, units outside [0, Wi) read the zero page, and
    // there is no separately zeroed halo (pad_l = 0, qshift = halo units - pad).  One part: Wst = Wi, gx0 = 0, qshift = 0.
    int Wst, gx0, qshift, parts;
    // ceil(2^40 / d) for the divisors of the per-thread geometry set-up: n / d == (n * m) >> 40 for n, d < 2^20
    unsigned long long m_Wo, m_rows_seg, m_Ho, m_units, m_units_seg, m_units_row, m_hrow, m_rows;
    int dbg;       // development ablation switches (EAS_CONV_DBG): 2 no staging after chunk 0, 4 weights from one address, 16 no barrier
    int single;    // one LDS buffer for the staged patch (the next channel chunk is written after a barrier): chosen by dispatch_tile where
                   // the double-buffered patch leaves one block per CU and the single-buffered one two
    // BatchNorm statistics in the epilogue (eas_conv_fwd_stats): not NULL = every block also writes the sum and the sum of squares of
    // its output tile per channel to stats[(co * stats_nb + pixel block) * 2 + {0, 1}] (doubles); stats_nb = gridDim.x * parts
    double* stats;
    int stats_nb;
    // Fused eval-mode step conv -> BatchNorm (running statistics) -> LIF over T (eas_conv_bn_lif_eval, kernel template LM != 0; see
    // conv_lif_epi.h).  LM = 1, time-major: NI is the number of SAMPLES N, pixel tile j of a wave is time step j of the wave's 32 spatial
    // pixels (WN == lif.T) and the staged patch holds lif.T time slices of the block's rows: nseg = lif.T * nseg_s segments, segment
    // t * nseg_s + s = image t * N + img0 + s.  LM = 2: one convolution result for all T steps (spatial tiles as always).
    int nseg_s;
    // Ragged row tiles (fused eval step): bpi > 0 = every image is cut into bpi tiles of RT rows by itself, the last one short (its rows past
    // the image are masked) -- RT need not divide Ho, so a 32 * 2^k pixel tile can hold 3 or 6 rows of 40 pixels.  grid.x = NI * bpi.
    int bpi;
    int act;       // epilogue activation on (acc + bias): 0 none, 1 SiLU (eas_conv_fwd_act: an eval-mode BaseConv whose BatchNorm is folded into
                   // weights and bias by fuse_model, yolox/utils/model_utils.py:35-80 -- "return self.act(self.conv(x))", network_blocks.py:55-56)
};

// the plain tiles + the BatchNorm / activation epilogue (LM = 3): what a grouped eas_conv_bn_act_eval carries per problem
struct ConvGeomBna : ConvGeomCore {
    EasBnActEpiDev bna;
};

struct ConvGeom : ConvGeomCore {
    EasLifEpiDev lif;
    EasBnActEpiDev bna;   // BatchNorm (running statistics) + activation in the plain epilogue (eas_conv_bn_act_eval); bna.on = 0: the plain store
};

// exact n / d for n, d < 2^20 with m = ceil(2^40 / d): a multiply-shift instead of the ~25-instruction integer division
__device__ __forceinline__ int fdiv(int n, unsigned long long m) { return (int)(((unsigned long long)(unsigned)n * m) >> 40); }
static inline unsigned long long fdiv_magic(int d) { return ((1ULL << 40) + (unsigned long long)d - 1) / (unsigned long long)(d > 0 ? d : 1); }

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    lo = (__bf16)r2;
}

// ---------------------------------------------------------------------------------------------------------------
// 8 consecutive input channels of one staged pixel -> bf16 term(s) -> one 16-byte LDS store per term
template <int XT>
__device__ __forceinline__ void stage_store(unsigned char* dst, int term_stride, const float (&v)[8], unsigned& bad) {
    if constexpr (XT == 1) {
        // the caller promised small integers (spikes, SEW sums): exact in bf16 <=> the low 16 bits of the fp32 pattern are zero.  The
        // patterns are OR-ed into a per-thread register and looked at once, at the end of the kernel (conv_tile_body).
        bf16x8 t0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bad |= __float_as_uint(v[j]);
            t0[j] = (__bf16)v[j];
        }
        *(bf16x8*)dst = t0;
    } else {
        bf16x8 t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, c;
            split3(v[j], a, b, c);
            t0[j] = a;
            t1[j] = b;
            t2[j] = c;
        }
        *(bf16x8*)dst = t0;
        *(bf16x8*)(dst + term_stride) = t1;
        *(bf16x8*)(dst + 2 * term_stride) = t2;
    }
}

// Staging: one item = VEC consecutive pixels of one staged row x 8 consecutive input channels = 8 vector loads (16 B per
// lane for VEC = 4) and VEC 16-byte LDS stores per term.  The next chunk's items are all requested during the first MFMA
// steps of the current chunk and written to the other LDS buffer during its last steps, so each load has several thousand
// cycles to land.  Halo columns are zeroed once; rows outside the image are written as zeros.
// PL: x is a spike-plane tensor (bf16 [NI][Cin/8][Hi*Wi][8], one exact term): the 8 channels of a staged pixel are 16 contiguous bytes
// in HBM and in LDS -- a staging item is VEC 16-byte loads and VEC 16-byte LDS stores, no conversion, half the bytes of fp32.
// bx / by / gx: the block's pixel-tile index, channel-group index and the number of pixel tiles of its problem -- blockIdx.x, blockIdx.y,
// gridDim.x for a launch of one problem; a grouped launch (conv_group.hip) maps its flat grid onto (problem, tile) first.
// NC = 4: the input gradient of a stride-2 3x3 convolution with the four parity classes of the gx pixel in ONE tile (conv_s2d.hip): x is
// grad_y, the nine taps come in class-major order (class 2 * ph + pw has (ph + 1) * (pw + 1) taps, weights packed with mode 2), tap t
// accumulates into the accumulators of its class -- acc[.][class * WN + j] -- and the epilogue writes gx[2a + ph][2b + pw] of the lane's
// grad_y position (a, b): both column parities of a row in one 8-byte store, i.e. whole lines from one block.
constexpr int kS2Cls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3}, kS2Lt[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3}, kS2Ntap[4] = {1, 2, 2, 4}, kS2Cum[4] = {0, 1, 3, 5};

template <int TAPS, int S, int XT, int WM, int WN, int WVM, int WVN, int CCH, int VEC, int NIT, bool PL = false, int LM = 0, typename G = ConvGeom, int NC = 1>
__device__ __forceinline__ void conv_tile_body(const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias,
                                               float* __restrict__ y, int* __restrict__ inexact, const G& g, const int part,
                                               unsigned char* smem, const int bx, const int by, const int gx) {
    static_assert(!PL || XT == 1, "spike planes are one exact bf16 term");
    static_assert(LM == 0 || WM == 1, "the fused neuron epilogue walks one M-tile per wave");
    static_assert(NC == 1 || (NC == 4 && TAPS == 9 && S == 1 && LM == 0 && CCH == 16 && !PL), "NC = 4: the stride-2 input gradient");
    constexpr bool TM = LM == 1;   // time-major pixel tiles (conv_lif_epi.h)
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    constexpr int PIXB = CCH * 2;  // bytes per staged pixel and term
    constexpr int NSTEPS = (CCH / 16) * TAPS;
    constexpr int NT = 64 * WVM * WVN;   // threads; NIT = staging items per thread per chunk (upper bound, surplus skipped block-uniformly)
    constexpr bool SPREAD = NSTEPS >= 2 * NIT;   // commit item i after step NSTEPS-NIT+i, else after the last step
    // B-fragment prefetch one step ahead: spike inputs (one term: WN fragments), and -- EAS_CONV_BPF3, round 5 -- real-valued inputs on the
    // 96-pixel wave tiles (3 x 3 fragments, 36 registers per set): those are the tiles of the small maps, where a SIMD holds one or two waves and
    // the LDS round trip of a step's nine ds_read_b128 sat exposed in front of its 18 MFMAs
#ifndef EAS_CONV_BPF3
#define EAS_CONV_BPF3 1
#endif
#ifndef EAS_S2C_BPF
#define EAS_S2C_BPF 0
#endif
    constexpr bool BPF = NSTEPS > 1 && (XT == 1 || (EAS_CONV_BPF3 && XT == 3 && WN <= 3 && WM == 1 && (NC == 1 || EAS_S2C_BPF)));
    static_assert(WVM * WVN == 4 || WVM * WVN == 8, "4 or 8 waves per block");
    static_assert(CCH % 16 == 0, "ci chunk is a multiple of the MFMA k");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WVN, wn = wave % WVN;
    const int r = lane & 31, h = lane >> 5;
    const int bimg = g.bpi ? bx / g.bpi : 0;
    const int rho0 = g.bpi ? bimg * g.Ho + (bx - bimg * g.bpi) * g.RT : bx * g.RT;
    const int img0 = fdiv(rho0, g.m_Ho), r0 = rho0 - img0 * g.Ho;
    const int mt0 = (by * WVM + wm) * WM;
    const int term_stride = g.Q * PIXB;
    const int grp = g.Q * 16;      // bytes of one 8-channel group plane: LDS layout [term][8-channel group][pixel][8 ch = 16 B]
    const int buf_bytes = term_stride * XT;
    const int npix = g.RT * g.Wo;

    // zero the halo columns of both buffers (never written again): pad_l columns left of the image, the rest right of it
    {
        const int rows = g.nseg * g.rows_in;
        const int hcols = g.RS - g.Wst;
        const int per_row = hcols * (PIXB / 16);
        const int nbuf_k = (g.KSTEPS * 16 <= CCH || g.single) ? 1 : 2;    // a single channel chunk uses (and owns) one buffer only
        for (int i = tid; i < rows * per_row * XT * nbuf_k; i += NT) {
            const int row = fdiv(i, g.m_hrow), k = i - row * per_row;
            const int bt = fdiv(row, g.m_rows), rw = row - bt * rows;   // bt = buffer*XT + term
            const int hc = k / (PIXB / 16), kk = k - hc * (PIXB / 16);
            const int col = hc < g.pad_l ? hc : g.Wst + hc;
            *(uint4*)(smem + (size_t)bt * term_stride + (size_t)kk * grp + ((size_t)rw * g.RS + col) * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    // per-lane geometry of the WN pixel columns this lane owns
    int qoff[WN];
    long ybase[WN];
    constexpr bool NEURON = LM == 1 || LM == 2;      // fused neuron epilogue (LM = 3: BatchNorm + activation on the plain tiles)
    int l_img[NEURON ? WN : 1], l_pix[NEURON ? WN : 1];     // fused neuron epilogue: sample and pixel (inside its image) of the lane's column of tile j
    bool l_ok[NEURON ? WN : 1];
    bool ok_r1[NC == 4 ? WN : 1], ok_c1[NC == 4 ? WN : 1];      // NC = 4: row 2a + 1 / column 2b + 1 of the lane's position exist in gx
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        // time-major: tile j = time step j / M2 of the wave's spatial tile j % M2 (M2 = WN / T spatial tiles per wave)
        int m2 = 1;
        if constexpr (TM) m2 = WN / g.lif.T;
        const int p = TM ? (wn * m2 + j % m2) * 32 + r : (wn * WN + j) * 32 + r;
        const int pc = p < npix ? p : 0;
        const int rl = fdiv(pc, g.m_Wo), c = pc - rl * g.Wo;
        const int seg = fdiv(rl, g.m_rows_seg), rr = rl - seg * g.rows_seg;
        qoff[j] = (((TM ? (j / m2) * g.nseg_s + seg : seg) * g.rows_in + rr * S) * g.RS + c * S + g.qshift) * 16 + h * grp;
        const int rho = rho0 + rl;
        const int img = g.bpi ? img0 : fdiv(rho, g.m_Ho), orow = rho - img * g.Ho;       // (ragged tiles: rows past the image belong to nobody)
        const int yr = orow * g.os + g.oph, yc = (part * g.Wo + c) * g.os + g.opw;
        ybase[j] = (p < npix && orow < g.Ho && rho < g.total_rows && yr < g.oH && yc < g.oW) ? (((long)img * g.Cout * g.oH + yr) * g.oW + yc) : -1;
        if constexpr (NEURON) {
            l_img[j] = img;
            l_pix[j] = yr * g.oW + yc;
            l_ok[j] = ybase[j] >= 0;
        }
        if constexpr (NC == 4) {
            ok_r1[j] = yr + 1 < g.oH;
            ok_c1[j] = yc + 1 < g.oW;
        }
    }

    // per-thread staging items (the same for every channel chunk)
    const size_t plane = (size_t)g.Hi * g.Wi;
    const int units_row = g.Wst / VEC, units_seg = g.rows_in * units_row, units = g.nseg * units_seg;
    const int nitems = units * (CCH / 8);
    const int gcol0 = g.gx0 + part * g.Wo * S;      // input column of the first staged unit (may be < 0 for part 0)
    int gofs[NIT], lofs[NIT], gch[NIT];   // gofs < 0: row (or, with column parts, unit) outside the image (zeros)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int item = it * NT + tid;
        item = item < nitems ? item : nitems - 1;   // surplus threads of the last slice redo its last item (same bytes)
        const int gi = fdiv(item, g.m_units), u = item - gi * units;
        const int seg = fdiv(u, g.m_units_seg), rem = u - seg * units_seg;
        const int rl = fdiv(rem, g.m_units_row), cu = rem - rl * units_row;
        const int ir = r0 * S - g.pad_t + rl;
        // (time-major: segment t * nseg_s + s of the patch is image t * N + img0 + s; g.NI = N samples there)
        const int seg_t = TM ? seg / g.nseg_s : 0, seg_s = TM ? seg - seg_t * g.nseg_s : seg;
        const int img = (TM ? seg_t * g.NI : 0) + img0 + seg_s;
        const int gc = gcol0 + cu * VEC;
        const bool ok = ir >= 0 && ir < g.Hi && img0 + seg_s < g.NI && gc >= 0 && gc < g.Wi;
        gofs[it] = ok ? (PL ? (int)((((size_t)img * (g.Cin / 8)) * g.Hi + ir) * g.Wi + gc)          // planes: in 16-byte units, group 0
                            : (int)((((size_t)img * g.Cin) * g.Hi + ir) * g.Wi + gc)) : -1;
        lofs[it] = ((seg * g.rows_in + rl) * g.RS + g.pad_l + cu * VEC) * 16 + gi * grp;
        gch[it] = gi * 8;
    }

    f32x16 acc[WM][WN * NC];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN * NC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // A fragments: linear in the step index st = kstep*TAPS + tap.  Read with buffer loads: per-lane offset lane*16 in a VGPR,
    // everything else (term, M-tile, step) in a scalar offset -> no vector ALU work per load.  Register sets rotate (the
    // step count of a chunk is a multiple of NSETS), so "current" and "prefetched" fragments never need copying.
    constexpr int NSETS = (NSTEPS % 2 == 0) ? 2 : 3;
    static_assert(NSTEPS % NSETS == 0, "fragment register sets rotate back at the end of a chunk");
    const int steps_total = g.KSTEPS * TAPS;
    const unsigned a_term_bytes = (unsigned)g.MT * steps_total * 1024u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 3 * a_term_bytes, 0x00020000);
    const int a_voff = lane * 16;
    unsigned a_soff[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mt = (mt0 + i) < g.MT ? (mt0 + i) : (g.MT - 1);
        a_soff[i] = __builtin_amdgcn_readfirstlane(NC == 4 ? (unsigned)mt * g.KSTEPS : (unsigned)mt * steps_total * 1024u);
    }
    bf16x8 a[NSETS][WM][3];
    // NC = 4: the mode-2 pack keeps every class by itself -- [class][term][M-tile][k-step][tap of the class][lane]; a_soff = M-tile * KSTEPS
    // there, and a step's fragment sits at 3 * mk * cum(class) + term * mk * ntap(class) + ((M-tile * KSTEPS + k-step) * ntap + tap) KB with
    // mk = MT * KSTEPS KB (all of it scalar arithmetic on compile-time class constants)
    const unsigned a_mk = (unsigned)g.MT * g.KSTEPS * 1024u;
    auto load_a = [&](int set, int step) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                a[set][i][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                              a_rsrc, a_voff, a_soff[i] + t * a_term_bytes + (unsigned)step * 1024u, 0));
    };
    auto load_a4 = [&](int set, int ks, int tap) {      // tap: a constant once the step loop is unrolled
        const int cls = kS2Cls[tap], nt = kS2Ntap[cls];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                a[set][i][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                              a_rsrc, a_voff, a_mk * (unsigned)(3 * kS2Cum[cls] + t * nt) +
                                                                                  ((a_soff[i] + (unsigned)ks) * nt + kS2Lt[tap]) * 1024u, 0));
    };
    // weight fragments in flight ahead of the MFMAs: two steps for spike inputs (a step of a one-term layer is 288-480 MFMA cycles per wave,
    // less than the ~450-cycle latency of the fragment loads it has to cover; the third register set of the 9-step rotation holds them),
    // one step otherwise.  Measured, same box: config 3 45.89 / 45.94 -> 45.76 / 45.83 ms (its convolution family 22.2 -> 22.0 ms),
    // config 2 unchanged.  -DEAS_CONV_APF=1: one step everywhere (development).
#ifndef EAS_CONV_APF
#define EAS_CONV_APF 2
#endif
    constexpr int APF = (EAS_CONV_APF == 2 && XT == 1 && NSETS == 3) ? 2 : 1;
    if constexpr (NC == 4) load_a4(0, 0, 0);
    else load_a(0, 0);
    if constexpr (APF == 2) load_a(1, steps_total > 1 ? 1 : 0);

    // loads are unconditional: a row outside the image or a channel group past Cin (Cin % 8 == 0: a group is valid or not as
    // a whole) reads the zero page with channel stride 0, so nothing branches, waits or needs masking around them
    vecf L[NIT][PL ? 1 : 8];
    u32x4 LP[NIT][PL ? VEC : 1];
    unsigned bad = 0;       // XT == 1 on fp32 input: OR of the staged bit patterns (stage_store)
    auto fetch = [&](int it, int c0) {
        const bool ok = gofs[it] >= 0 && c0 + gch[it] < g.Cin;
        if constexpr (PL) {
            // VEC consecutive pixels of 8-channel group (c0 + gch) / 8: VEC x 16 contiguous bytes
            const u32x4* src = ok ? reinterpret_cast<const u32x4*>(x) + (size_t)gofs[it] + (size_t)((c0 + gch[it]) >> 3) * plane
                                  : reinterpret_cast<const u32x4*>(eas_conv_zero_page);
#pragma unroll
            for (int p = 0; p < VEC; ++p) LP[it][p] = src[ok ? p : 0];
        } else {
            const float* src = ok ? x + (size_t)gofs[it] + (size_t)(c0 + gch[it]) * plane : eas_conv_zero_page;
            const size_t cs = ok ? plane : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) L[it][j] = *(const vecf*)(src + j * cs);
        }
    };
    auto commit = [&](unsigned char* buf, int it, int c0) {
        if constexpr (PL) {
#pragma unroll
            for (int p = 0; p < VEC; ++p) *(u32x4*)(buf + lofs[it] + p * 16) = LP[it][p];
        } else {
#pragma unroll
            for (int p = 0; p < VEC; ++p) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = L[it][j][p];
                stage_store<XT>(buf + lofs[it] + p * 16, term_stride, v, bad);
            }
        }
    };

    // prologue: chunk 0 -> buffer 0
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        if (it * NT < nitems) fetch(it, 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        if (it * NT < nitems) commit(smem, it, 0);
    __syncthreads();

    const int nchunks = (g.KSTEPS * 16 + CCH - 1) / CCH;
    int st = 0;
    bf16x8 bq[NSETS][WN][BPF ? XT : 1];
    for (int c = 0; c < nchunks; ++c) {
        const bool single = g.single != 0;
        const unsigned char* cur = smem + (single ? 0 : (c & 1)) * buf_bytes;
        unsigned char* nxt = smem + (single ? 0 : ((c + 1) & 1)) * buf_bytes;
        const bool more = c + 1 < nchunks && !(g.dbg & 2);
        const int c1 = (c + 1) * CCH;
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const int kk = s / TAPS, tap = s - kk * TAPS;
            ++st;
            const int sta = st + APF - 1;
            const int stn = (g.dbg & 4) ? 0 : (sta < steps_total ? sta : steps_total - 1);   // prefetch the weights APF steps ahead (clamped at the end)
            if constexpr (NC == 4) {
                const int ksn = c * (CCH / 16) + (s + APF) / TAPS;
                load_a4((s + APF) % NSETS, ksn < g.KSTEPS ? ksn : g.KSTEPS - 1, (s + APF) % TAPS);
            } else {
                load_a((s + APF) % NSETS, stn);
            }
            if (more) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if ((SPREAD ? it == s : s == 0) && it * NT < nitems) fetch(it, c1);
            }
            // keep the prefetch loads (next step's weights, the next patch) ahead of this step's MFMAs: without the fence
            // the scheduler sinks them behind the last use of the registers they would share
            __builtin_amdgcn_sched_barrier(0);
            // B fragments of this step; with spike inputs (one term, 20 registers) the NEXT step's fragments are requested
            // before this step's MFMAs so that the LDS latency is off the critical path (first step of a chunk excepted)
            bf16x8 b[WN][XT];
            if constexpr (BPF) {
                if (s == 0) {
                    const int toff0 = g.tap_off[0] * 16;
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int t = 0; t < XT; ++t) bq[0][j][t] = *(const bf16x8*)(cur + t * term_stride + qoff[j] + toff0);
                }
                if (s + 1 < NSTEPS) {
                    const int kk1 = (s + 1) / TAPS, tap1 = (s + 1) - kk1 * TAPS;
                    const int toff1 = g.tap_off[tap1] * 16 + kk1 * 2 * grp;
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int t = 0; t < XT; ++t) bq[(s + 1) % NSETS][j][t] = *(const bf16x8*)(cur + t * term_stride + qoff[j] + toff1);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int t = 0; t < XT; ++t) b[j][t] = bq[s % NSETS][j][t];
            } else {
                const int toff = g.tap_off[tap] * 16 + kk * 2 * grp;
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int t = 0; t < XT; ++t) b[j][t] = *(const bf16x8*)(cur + t * term_stride + qoff[j] + toff);
            }
            bf16x8 (&a_cur)[WM][3] = a[s % NSETS];
            // smallest products first; pairs (weight term, input term) with ta + tb <= 2.  Term-major order: consecutive MFMAs go
            // to different accumulators (a dependent MFMA right behind its producer waits for the result), the order per
            // accumulator is unchanged
            if constexpr (XT == 1) {
#pragma unroll
                for (int ta = 2; ta >= 0; --ta)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[i][ta], b[j][0], acc[i][j], 0, 0, 0);
            } else {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
                const int cj = NC == 4 ? kS2Cls[tap] * WN : 0;      // the accumulators of the tap's parity class
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][cj + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[i][PA[q]], b[j][PB[q]], acc[i][cj + j], 0, 0, 0);
            }
            if (SPREAD && more && !single) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (s == NSTEPS - NIT + it && it * NT < nitems) commit(nxt, it, c1);
            }
        }
        if (single && more) __syncthreads();
        if ((!SPREAD || single) && more) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it * NT < nitems) commit(nxt, it, c1);
        }
        if (!(g.dbg & 16)) __syncthreads();
    }

    if constexpr (XT == 1 && !PL) {
        if (inexact && (bad & 0xffffu)) atomicOr(inexact, 1);      // a tensor tagged "spikes / small integers" was not: say so, loudly
    }

    // ---- BatchNorm statistics of the tile (block-uniform branch): per-wave sums over its valid pixels by DPP, the WVN waves that share
    // the channels added in wave order, one float2 per channel and block.  The staging buffers are free after the last barrier.
    if constexpr (NC == 1) if (g.stats) {
        double* red = reinterpret_cast<double*>(smem + (size_t)WVM * WVN * EAS_STATS_SCRATCH * sizeof(float));
        bool valid[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) valid[j] = ybase[j] >= 0;
        eas_conv_stats_wave<WM, WN>(acc, valid, reinterpret_cast<float*>(smem) + wave * EAS_STATS_SCRATCH, red + wave * (WM * 64), lane);
        __syncthreads();
        if (tid < WVM * WM * 32) {
            const int wmi = tid / (WM * 32), t = tid - wmi * (WM * 32);
            const int co = (by * WVM + wmi) * WM * 32 + t;
            eas_conv_stats_store(red + wmi * WVN * (WM * 64), WM * 64, WVN, t, co, g.Cout, g.stats, g.stats_nb,
                                 bx + gx * part);
        }
    }

    if constexpr (NEURON) {
        // ---- fused BatchNorm (running statistics) + LIF over T: the accumulators are z_t of the lane's neurons, nothing is written but spikes
        if (g.dbg & 32) {        // development ablation: the main loop alone (one store keeps the accumulators alive)
            if (acc[0][0][0] == 12345.678f && g.lif.r[0].out_f32) g.lif.r[0].out_f32[0] = acc[0][WN - 1][3];
            return;
        }
        if (mt0 < g.MT) eas_lif_epilogue<WN, LM>(acc[0], g.lif, mt0, h, l_img, l_pix, l_ok);
        return;
    }

    // ---- epilogue: D tile (col = pixel = lane&31, row = (e&3) + 8*(e>>2) + 4*h) -> y NCHW.  Lean on purpose: one pointer per
    // (M-tile, pixel tile), rows reached by adding multiples of the channel stride; bias values loaded once per M-tile;
    // the per-element channel bound check only for a ragged last M-tile (the epilogue used to be as long as the main loop).
    const long cstride = (long)g.oH * g.oW;
    if constexpr (NC == 4) {
        // gx[2a + ph][2b + pw] of the lane's grad_y position: ybase is the (ph, pw) = (0, 0) pixel; even row width and an 8-byte aligned tensor
        // (block-uniform) -> the two column parities leave as one 8-byte store, 32 lanes = 256 contiguous bytes
        const bool pair = (g.oW & 1) == 0 && ((uintptr_t)y & 7) == 0;
        if (g.dbg & 64) {        // development ablation: the main loop alone (one store keeps the accumulators alive)
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < WN * NC; ++j) sum += acc[0][j][3];
            if (sum == 12345.678f) y[0] = sum;
            return;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (mt0 + i >= g.MT) continue;
            const int co0 = (mt0 + i) * 32 + 4 * h;
            const bool full = (mt0 + i) * 32 + 32 <= g.Cout;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (ybase[j] < 0) continue;
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    if (ph && !ok_r1[j]) continue;
                    float* yp = y + ybase[j] + (long)co0 * cstride + ph * g.oW;
                    const f32x16& v0 = acc[i][(2 * ph) * WN + j];
                    const f32x16& v1 = acc[i][(2 * ph + 1) * WN + j];
                    if (pair) {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout)
                                *reinterpret_cast<float2*>(yp + ((e & 3) + 8 * (e >> 2)) * cstride) = make_float2(v0[e], v1[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout) {
                                yp[((e & 3) + 8 * (e >> 2)) * cstride] = v0[e];
                                if (ok_c1[j]) yp[((e & 3) + 8 * (e >> 2)) * cstride + 1] = v1[e];
                            }
                    }
                }
            }
        }
        return;
    }
    if constexpr (LM == 3) {      // BatchNorm (running statistics) + activation on the accumulators (eas_conv_bn_act_eval)
#pragma unroll
        for (int i = 0; i < WM; ++i)
            if (mt0 + i < g.MT) eas_bnact_epilogue<WN>(acc[i], g.bna, mt0 + i, h, ybase, cstride);
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        if (mt0 + i >= g.MT) continue;
        const int co0 = (mt0 + i) * 32 + 4 * h;
        const bool full = (mt0 + i) * 32 + 32 <= g.Cout;          // block-uniform
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
        for (int j = 0; j < WN; ++j) {
            if (ybase[j] < 0) continue;
            float* yp = y + ybase[j] + (long)co0 * cstride;
            if (g.act) {          // block-uniform: the activation's transcendental never runs for the plain convolution
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (full || co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[((e & 3) + 8 * (e >> 2)) * cstride] = eas_epi_silu(acc[i][j][e] + bv[e]);
            } else if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) yp[((e & 3) + 8 * (e >> 2)) * cstride] = acc[i][j][e] + bv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (co0 + (e & 3) + 8 * (e >> 2) < g.Cout) yp[((e & 3) + 8 * (e >> 2)) * cstride] = acc[i][j][e] + bv[e];
            }
        }
    }
}

template <int TAPS, int S, int XT, int WM, int WN, int WVM, int WVN, int CCH, int VEC, int NIT, bool PL = false, int LM = 0>
__global__ __launch_bounds__(64 * WVM * WVN, 2) void conv_fwd_mfma_kernel(const float* __restrict__ x, const bf16x8* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ y, int* __restrict__ inexact,
                                                            ConvGeom g) {
    extern __shared__ __align__(16) unsigned char smem[];
    conv_tile_body<TAPS, S, XT, WM, WN, WVM, WVN, CCH, VEC, NIT, PL, LM>(x, wp, bias, y, inexact, g, (int)blockIdx.z, smem, (int)blockIdx.x,
                                                                         (int)blockIdx.y, (int)gridDim.x);
}

static void conv_geom_magics(ConvGeomCore& g, int VEC, int CCH) {
    const int units_row = g.Wst / VEC, units_seg = g.rows_in * units_row, units = g.nseg * units_seg;
    const int rows = g.nseg * g.rows_in, per_row = (g.RS - g.Wst) * (CCH * 2 / 16);
    g.m_Wo = fdiv_magic(g.Wo); g.m_rows_seg = fdiv_magic(g.rows_seg); g.m_Ho = fdiv_magic(g.Ho);
    g.m_units = fdiv_magic(units); g.m_units_seg = fdiv_magic(units_seg); g.m_units_row = fdiv_magic(units_row);
    g.m_hrow = fdiv_magic(per_row); g.m_rows = fdiv_magic(rows);
}

template <int TAPS, int S, int XT, int WM, int WN, int WVM, int WVN, int CCH, int VEC, int NIT, bool PL = false, int LM = 0>
int launch_fwd(const float* x, const bf16x8* wp, const float* bias, float* y, int* inexact, ConvGeom g, hipStream_t st) {
    auto kern = conv_fwd_mfma_kernel<TAPS, S, XT, WM, WN, WVM, WVN, CCH, VEC, NIT, PL, LM>;
    const int nbuf = (g.KSTEPS * 16 <= CCH || g.single) ? 1 : 2;      // one channel chunk: the second (prefetch) buffer is never used
    size_t lds = (size_t)nbuf * g.Q * CCH * 2 * XT;
    if (lds > 160 * 1024) return EAS_ERR_UNSUPPORTED;
    if (g.stats && lds < (size_t)WVM * WVN * (EAS_STATS_SCRATCH * sizeof(float) + WM * 64 * sizeof(double)))
        lds = (size_t)WVM * WVN * (EAS_STATS_SCRATCH * sizeof(float) + WM * 64 * sizeof(double));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    if (g.total_rows >= (1 << 20) || g.nseg * g.rows_in * (g.Wst / VEC) >= (1 << 20)) return EAS_ERR_UNSUPPORTED;     // fdiv range
    conv_geom_magics(g, VEC, CCH);
    dim3 grid(g.bpi ? g.NI * g.bpi : (g.total_rows + g.RT - 1) / g.RT, (g.MT + WVM * WM - 1) / (WVM * WM), g.parts);
    if (g.stats && (int)(grid.x * g.parts) != g.stats_nb) return EAS_ERR_INVALID_ARG;
    EAS_LAUNCH(kern, grid, dim3(64 * WVM * WVN), lds, st, x, wp, bias, y, inexact, g);
    return EAS_OK;
}

// rows per tile: the largest RT with RT*Wo <= BN such that tiles never straddle an image boundary mid-image
int pick_rows(int Ho, int Wo, int BN) {
    int best = 0;
    const int maxrt = BN / Wo;
    for (int rt = 1; rt <= maxrt; ++rt)
        if (Ho % rt == 0 || rt % Ho == 0) best = rt;
    return best;
}

typedef int (*launch_fn)(const float*, const bf16x8*, const float*, float*, int*, ConvGeom, hipStream_t);


// pixel blocks per channel (grid.x * parts) of the tile the last dispatch_tile call of this thread chose: the number of statistics
// partials eas_conv_fwd_stats writes per channel
thread_local int tl_pixel_blocks = 0;

template <int TAPS, int S, int XT, int CCH, int VEC, bool PL = false, int LM = 0>
int dispatch_tile(const float* x, const bf16x8* wp, const float* bias, float* y, int* inexact, ConvGeom g, hipStream_t st) {
    // candidate block tiles {waves along M, waves along N}, each wave one 32-channel M-tile x 5 pixel tiles (160 pixels);
    // 8-wave blocks (2 waves per SIMD hide each other's staging, LDS and weight latencies) first, widest pixel tile first
    struct Cand { int wm, wvm, bn, threads, nit, wn; launch_fn fn; };
    constexpr int N8 = XT == 1 ? 2 : 1, N4 = 2;   // staging items per thread (register budget: 2 waves per SIMD either way)
    // wave tile = 32 channels x WN 32-pixel tiles.  WN = 5 (160 pixels) is the efficient shape; WN = 3 (96 pixels: two 40-pixel rows,
    // four 20-pixel rows or one 8x10 image) exists for the small maps of the neck / head and of dark4 / dark5, where 160-pixel wave
    // tiles leave most of the 256 CUs without a block (64 images of 8x10 with 128 channels are 16 eight-wave blocks)
    const Cand cands[14] = {
        {1, 2, 640, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 2, 4, CCH, VEC, N8, PL, LM>}, {1, 4, 320, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 4, 2, CCH, VEC, N8, PL, LM>},
        {1, 8, 160, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 8, 1, CCH, VEC, N8, PL, LM>}, {1, 1, 1280, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 1, 8, CCH, VEC, N8, PL, LM>},
        {1, 1, 640, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 1, 4, CCH, VEC, N4, PL, LM>}, {1, 2, 320, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 2, 2, CCH, VEC, N4, PL, LM>},
        {1, 4, 160, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 4, 1, CCH, VEC, N4, PL, LM>},
        {1, 2, 384, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 2, 4, CCH, VEC, N8, PL, LM>}, {1, 4, 192, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 4, 2, CCH, VEC, N8, PL, LM>},
        {1, 8, 96, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 8, 1, CCH, VEC, N8, PL, LM>}, {1, 1, 768, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 1, 8, CCH, VEC, N8, PL, LM>},
        {1, 1, 384, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 1, 4, CCH, VEC, N4, PL, LM>}, {1, 2, 192, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 2, 2, CCH, VEC, N4, PL, LM>},
        {1, 4, 96, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 4, 1, CCH, VEC, N4, PL, LM>}};
    // cost model (measured on MI355X, scripts/dev_conv.py, scripts/dev_conv_calls.py): one block per CU; a round of 8-wave blocks costs
    // ~1.27x a round of 4-wave blocks; a block's time is a fixed part (prologue, first patch, epilogue) plus its MFMA work, which
    // goes with WN whatever part of the pixel tile is valid
    int best = -1;
    double best_cost = 0.0;
    int best_valid = 0;
    ConvGeom best_g = g;
    static const int force = eas_dev_env("EAS_CONV_TILE") ? atoi(eas_dev_env("EAS_CONV_TILE")) : -1;   // development: force a candidate
    static const int ncand = eas_dev_env("EAS_CONV_NCAND") ? atoi(eas_dev_env("EAS_CONV_NCAND")) : 14;  // development: 7 = the 160-pixel wave tiles only
    const int nchunks = (g.KSTEPS * 16 + CCH - 1) / CCH;
    // Single-buffered patch (mode 1): for layers of few channel chunks whose double-buffered patch fills the LDS (real-valued inputs on wide
    // rows: dark2.0, 135 -> 103 us) -- the second buffer buys one overlapped stage there, a second resident block overlaps everything.
    // With more chunks the exposed write of every chunk costs more than it gains (4 chunks: 100 -> 122 us, 8: 125 -> 154 us); three
    // chunks (the 48-channel layers of SYOLOX-M) still gain: config 4 88.0 -> 87.5 ms.
    static const int single_nch = eas_dev_env("EAS_CONV_SINGLE_NCH") ? atoi(eas_dev_env("EAS_CONV_SINGLE_NCH")) : 3;      // development: 0 = never
    static const double single_pen = eas_dev_env("EAS_CONV_SINGLE_PEN") ? atof(eas_dev_env("EAS_CONV_SINGLE_PEN")) : 1.1;
    for (int i = 0; i < ncand; ++i)
      for (int mode = 0; mode < 2; ++mode) {
        const Cand& c = cands[i];
        if (force >= 0 && i != force) continue;
        if (mode == 1 && (XT != 3 || nchunks < 2 || nchunks > single_nch)) continue;      // spike-input layers: 76 -> 84 us (the patch is a third)
        const int nbuf = (nchunks == 1 || mode == 1) ? 1 : 2;
        if ((c.wvm * c.wm - 1) * 32 >= g.Cout && !(c.wvm == 1 && c.wm == 1)) continue;   // every wave row (and M-tile) has channels to compute
        ConvGeom t = g;
        t.single = mode;
        bool fits = false;
        for (int cap = c.bn; cap >= g.Wo && !fits; cap -= 32) {   // shrink the pixel tile until patch + staging slots fit
            t.RT = pick_rows(g.Ho, g.Wo, cap);
            if (t.RT == 0) break;
            t.rows_seg = t.RT < g.Ho ? t.RT : g.Ho;
            t.nseg = t.RT / t.rows_seg;
            t.rows_in = (t.rows_seg - 1) * S + g.ext_h;
            t.Q = t.nseg * t.rows_in * t.RS;
            fits = (size_t)nbuf * t.Q * CCH * 2 * XT <= 160 * 1024 && t.nseg * t.rows_in * (g.Wst / VEC) * (CCH / 8) <= c.nit * c.threads;
            cap = t.RT * g.Wo;
        }
        if (!fits) continue;
        const long blocks = (long)((g.total_rows + t.RT - 1) / t.RT) * ((g.MT + c.wvm * c.wm - 1) / (c.wvm * c.wm)) * g.parts;
        // resident blocks per CU: one 8-wave block, or two 4-wave blocks when their LDS fits twice (independent barriers
        // overlap one block's prologue/epilogue with the other's MFMA phase: cheaper per round than one 8-wave block)
        const size_t lds_bytes = (size_t)nbuf * t.Q * CCH * 2 * XT;
        int bpc = c.threads == 512 ? 1 : (2 * lds_bytes <= 160 * 1024 ? 2 : 1);
        if (mode == 1) {
            // worth it only where it buys the second resident block (launch bounds keep the registers of two blocks per CU available)
            if (2 * lds_bytes > 160 * 1024 || 2 * (2 * lds_bytes) <= 160 * 1024) continue;
            bpc = 2;
        }
        // (two co-resident 4-wave blocks cost 1.2 rounds of one -- but only when there are more blocks than CUs: a grid of <= 256 blocks
        // puts one block on a CU whatever its LDS size)
        // a step's MFMAs (3 * WN with one-term inputs, 6 * WN with three) hide the ~450-cycle latency of the next step's weight
        // fragments only when they last that long: with spike inputs a 3-tile wave (288 MFMA cycles per step) waits on every step and
        // costs nearly as much as a 5-tile wave (dark5.m.conv2: 71 us against 62 us for the 160-pixel shape the old model ranked behind)
        static const double lat = eas_dev_env("EAS_CONV_STEP_LAT") ? atof(eas_dev_env("EAS_CONV_STEP_LAT")) : 450.0;   // development
        const double per_tile = XT == 1 ? 96.0 : 192.0;
        const double mfma_part = (c.wn * per_tile > lat ? c.wn * per_tile : lat) / (5.0 * per_tile);
        const double round_cost = (c.threads == 512 ? 1.27 : (bpc == 2 && blocks > 256 ? 1.2 : 1.0)) * (0.3 + 0.7 * mfma_part) *
                                  (mode == 1 ? single_pen * (c.threads == 512 && blocks > 256 ? 1.2 : 1.0) : 1.0);
        const double cost = (double)((blocks + 256 * bpc - 1) / (256 * bpc)) * round_cost;
        // ties: the larger valid pixel count; among grids of lone blocks (<= 256) first the shape with more waves along the channels
        // (they share one staged pixel patch: the smaller patch per block is the shorter prologue; 8x10 head layers: 35-38 us against
        // 44-50 us for the four-pixel-group shape)
        const int valid = t.RT * g.Wo;
        static const int tie_wvm = eas_dev_env("EAS_CONV_TIE_WVM") ? atoi(eas_dev_env("EAS_CONV_TIE_WVM")) : 1;        // development: 0 = lone blocks only
        const int rank = ((blocks <= 256 || tie_wvm) ? c.wvm * 4096 : 0) + (valid < 4096 ? valid : 4095);
        if (best < 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && rank > best_valid)) {
            best = i; best_cost = cost; best_valid = rank; best_g = t;
        }
    }
    if (best < 0) return EAS_ERR_UNSUPPORTED;
    tl_pixel_blocks = ((best_g.total_rows + best_g.RT - 1) / best_g.RT) * best_g.parts;
    if (!y) return EAS_OK;                       // geometry query (eas_conv_fwd_supported): a tile exists, nothing is launched
    return cands[best].fn(x, wp, bias, y, inexact, best_g, st);
}

}  // namespace
