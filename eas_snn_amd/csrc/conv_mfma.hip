// K4 (conv half): implicit-GEMM 2-D convolution on the CDNA4 matrix cores for the conv -> BN -> LIF step
// (reference: yolox/models/network_blocks.py:52-53 after yolox/utils/utils_snn.py:16-58).
//
// Tensors stay fp32 NCHW in HBM ([T*N, C, H, W], exactly the SeqToANNContainer view of the [T,N,C,H,W] spike tensors).
// Arithmetic: every fp32 operand is split into bf16 terms (hi + mid + lo = the fp32 value exactly, 3 x 8 mantissa bits);
// bf16 x bf16 products are exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, so the result is an fp32-accumulated
// sum of the SAME products an fp32 conv forms (spike inputs {0,1,2,..} are one exact term; general fp32 inputs use the six
// term pairs whose weight is >= 2^-16 relative, error of the dropped pairs <= 2^-23 relative per product).
//
//   forward :  y[n][co][ho][wo] = sum_{ci,kh,kw} w[co][ci][kh][kw] * x[n][ci][ho*S+kh-P][wo*S+kw-P]
//     GEMM view: D[co][pixel] = A[co][(tap,ci)] * B[(tap,ci)][pixel];  A = weights, packed once per call into MFMA
//     fragment order (conv_pack_weights_kernel), read straight from L2 into registers; B = the input patch, staged
//     through LDS as [term][input pixel incl. zero halo][ci chunk] bf16, so each lane's 8 consecutive ci are one 16-byte
//     ds_read_b128 and the KS*KS taps are plain address offsets into the same image; the two 8-channel halves of a k-step
//     live in separate planes ([term][8-channel group][pixel][16 B]) so that the 32 lanes of a half-wave read 512 contiguous
//     bytes (pixel-major 32-byte rows cost a 2-way bank conflict on every fragment read).
//   input gradient of a stride-1 conv is the same kernel on grad_y with the weights packed transposed + flipped.
//
//   input gradient of a stride-2 3x3 conv: per parity class of the input pixel a stride-1 tap-list convolution (eas_conv_dgrad_s2).
//
// Wave tile = one 32-channel M-tile x five 32-pixel N-tiles (80 accumulator registers); block = 4 or 8 waves arranged
// (waves along M) x (waves along N), block tile = RT whole output rows (possibly several whole images); two 4-wave blocks or
// one 8-wave block per CU (two waves per SIMD hide each other's staging and LDS latency).  The kernel is written for a
// low vector-ALU count per MFMA (PMC: a wave can hide about five VALU instructions per MFMA slot): fragment register sets
// rotate instead of being copied, weights come by buffer loads with scalar offsets, staging loads that fall outside the
// image read a zero page with stride 0 instead of being masked, and the epilogue walks one pointer per tile.
#include <stdlib.h>

#include "eas_common.h"
#include "conv_lif_epi.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// staging loads of rows outside the image / channels past Cin read these zeros (channel stride 0) instead of being masked
__device__ __attribute__((aligned(16))) float eas_conv_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct ConvGeom {
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
    EasLifEpiDev lif;
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
// weights [Cout][Cin][KS][KS] fp32 -> A fragments: wp[term][mt][kstep][tap][lane] (8 bf16 each)
//   mode 0 (forward):        M = Cout, K = Cin :  A[m][k][tap] = w[m][k][tap]
//   mode 1 (input gradient): M = Cin,  K = Cout:  A[m][k][tap] = w[k][m][TAPS-1-tap]      (transposed + flipped)
// wb / ca: the [Cout][Cin][KS][KS] tensor may be the concatenation along Cout of two tensors, w (output channels [0, ca)) and wb
// (the rest) -- the one convolution that computes two layers reading the same input (conv1 | conv2 of a CSPLayer, the first cls / reg
// tower convolutions of the head) is packed straight from its two parameters, without a torch.cat.  wb == nullptr: one tensor.
__device__ __forceinline__ void pack_fragment(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin, int TAPS, int mode,
                                              int idx, int total, const float* __restrict__ wb = nullptr, int ca = 0) {
    const int M = mode ? Cin : Cout, K = mode ? Cout : Cin;
    auto wat = [&](int co, int ci, int tap) {
        return (wb && co >= ca) ? wb[((size_t)(co - ca) * Cin + ci) * TAPS + tap] : w[((size_t)co * Cin + ci) * TAPS + tap];
    };
    const int KSTEPS = (K + 15) / 16;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int tap = rest % TAPS;
    rest /= TAPS;
    const int ks = rest % KSTEPS;
    const int mt = rest / KSTEPS;
    const int m = mt * 32 + (lane & 31);
    bf16x8 t0, t1, t2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * 16 + 8 * (lane >> 5) + j;
        float v = 0.0f;
        if (m < M && k < K) v = mode ? wat(k, m, TAPS - 1 - tap) : wat(m, k, tap);
        __bf16 a, b, c;
        split3(v, a, b, c);
        t0[j] = a;
        t1[j] = b;
        t2[j] = c;
    }
    wp[idx] = t0;
    wp[(size_t)total + idx] = t1;
    wp[(size_t)2 * total + idx] = t2;
}

// Input gradient of a stride-2 3x3 convolution (padding 1), by output-pixel parity class (ph, pw): grad_x[2a+ph][2b+pw]
// only receives taps with kh = ph+1 (mod 2): a stride-1 "convolution" of grad_y with 1, 2, 2 or 4 taps per class
//   ph = 0: kh = 1 reads grad_y row a ;  ph = 1: kh = 2 reads row a (dh = 0), kh = 0 reads row a+1 (dh = 1); same for columns.
// Packed: class c = 2*ph+pw at bf16x8 offset 3*MT*KSTEPS*64*{0,1,3,5}[c], inside [term][mt][kstep][local tap = ih*(pw+1)+iw][lane];
// A[m = ci][k = co][tap] = w[co][ci][kh][kw].
__device__ __forceinline__ void pack_fragment_s2dgrad(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin, int idx) {
    const int M = Cin, K = Cout;
    const int MT = (M + 31) / 32, KSTEPS = (K + 15) / 16;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int tap9 = rest % 9;
    rest /= 9;
    const int ks = rest % KSTEPS;
    const int mt = rest / KSTEPS;
    const int kh = tap9 / 3, kw = tap9 - kh * 3;
    const int ph = kh == 1 ? 0 : 1, pw = kw == 1 ? 0 : 1;
    const int ih = kh == 0 ? 1 : 0, iw = kw == 0 ? 1 : 0;
    const int cls = 2 * ph + pw, ntap = (ph + 1) * (pw + 1), lt = ih * (pw + 1) + iw;
    const int cum = cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5));
    const size_t base = (size_t)3 * MT * KSTEPS * 64 * cum;
    const size_t term = (size_t)MT * KSTEPS * ntap * 64;
    const size_t pos = (((size_t)mt * KSTEPS + ks) * ntap + lt) * 64 + lane;
    const int m = mt * 32 + (lane & 31);
    bf16x8 t0, t1, t2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * 16 + 8 * (lane >> 5) + j;
        const float v = (m < M && k < K) ? w[((size_t)k * Cin + m) * 9 + tap9] : 0.0f;
        __bf16 a, b, c;
        split3(v, a, b, c);
        t0[j] = a;
        t1[j] = b;
        t2[j] = c;
    }
    wp[base + pos] = t0;
    wp[base + term + pos] = t1;
    wp[base + 2 * term + pos] = t2;
}

__device__ __forceinline__ int pack_total(int Cout, int Cin, int ksize, int mode) {
    const int M = mode ? Cin : Cout, K = mode ? Cout : Cin;
    return ((M + 31) / 32) * ((K + 15) / 16) * ksize * ksize * 64;
}

__global__ void conv_pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin, int ksize, int mode) {
    const int total = pack_total(Cout, Cin, ksize, mode);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        if (mode == 2) pack_fragment_s2dgrad(w, wp, Cout, Cin, idx);
        else pack_fragment(w, wp, Cout, Cin, ksize * ksize, mode, idx, total);
    }
}

// every weight tensor of a model in ONE launch (blockIdx.y = job): a job is eight int64 {w, packed, Cout, Cin, ksize, mode, wb, ca}
// (wb != 0: the weight is the concatenation [w (ca output channels) ; wb] along Cout, Cout = the total; modes 0 and 1 only)
__global__ void conv_pack_weights_many_kernel(const long long* __restrict__ jobs) {
    const long long* j = jobs + (size_t)blockIdx.y * 8;
    const float* w = (const float*)j[0];
    bf16x8* wp = (bf16x8*)j[1];
    const int Cout = (int)j[2], Cin = (int)j[3], ksize = (int)j[4], mode = (int)j[5];
    const float* wb = (const float*)j[6];
    const int ca = (int)j[7];
    const int total = pack_total(Cout, Cin, ksize, mode);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        if (mode == 2) pack_fragment_s2dgrad(w, wp, Cout, Cin, idx);
        else pack_fragment(w, wp, Cout, Cin, ksize * ksize, mode, idx, total, wb, ca);
    }
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
template <int TAPS, int S, int XT, int WM, int WN, int WVM, int WVN, int CCH, int VEC, int NIT, bool PL = false, int LM = 0>
__device__ __forceinline__ void conv_tile_body(const float* __restrict__ x, const bf16x8* __restrict__ wp, const float* __restrict__ bias,
                                               float* __restrict__ y, int* __restrict__ inexact, const ConvGeom& g, const int part,
                                               unsigned char* smem) {
    static_assert(!PL || XT == 1, "spike planes are one exact bf16 term");
    static_assert(LM == 0 || WM == 1, "the fused neuron epilogue walks one M-tile per wave");
    constexpr bool TM = LM == 1;   // time-major pixel tiles (conv_lif_epi.h)
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    constexpr int PIXB = CCH * 2;  // bytes per staged pixel and term
    constexpr int NSTEPS = (CCH / 16) * TAPS;
    constexpr int NT = 64 * WVM * WVN;   // threads; NIT = staging items per thread per chunk (upper bound, surplus skipped block-uniformly)
    constexpr bool SPREAD = NSTEPS >= 2 * NIT;   // commit item i after step NSTEPS-NIT+i, else after the last step
    constexpr bool BPF = XT == 1 && NSTEPS > 1;   // B-fragment prefetch one step ahead
    static_assert(WVM * WVN == 4 || WVM * WVN == 8, "4 or 8 waves per block");
    static_assert(CCH % 16 == 0, "ci chunk is a multiple of the MFMA k");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WVN, wn = wave % WVN;
    const int r = lane & 31, h = lane >> 5;
    const int rho0 = blockIdx.x * g.RT;
    const int img0 = fdiv(rho0, g.m_Ho), r0 = rho0 - img0 * g.Ho;
    const int mt0 = (blockIdx.y * WVM + wm) * WM;
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
    int l_img[LM ? WN : 1], l_pix[LM ? WN : 1];     // fused neuron epilogue: sample and pixel (inside its image) of the lane's column of tile j
    bool l_ok[LM ? WN : 1];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int p = TM ? wn * 32 + r : (wn * WN + j) * 32 + r;       // time-major: tile j = time step j of the SAME 32 pixels
        const int pc = p < npix ? p : 0;
        const int rl = fdiv(pc, g.m_Wo), c = pc - rl * g.Wo;
        const int seg = fdiv(rl, g.m_rows_seg), rr = rl - seg * g.rows_seg;
        qoff[j] = (((TM ? j * g.nseg_s + seg : seg) * g.rows_in + rr * S) * g.RS + c * S + g.qshift) * 16 + h * grp;
        const int rho = rho0 + rl;
        const int img = fdiv(rho, g.m_Ho), orow = rho - img * g.Ho;
        const int yr = orow * g.os + g.oph, yc = (part * g.Wo + c) * g.os + g.opw;
        ybase[j] = (p < npix && rho < g.total_rows && yr < g.oH && yc < g.oW) ? (((long)img * g.Cout * g.oH + yr) * g.oW + yc) : -1;
        if constexpr (LM != 0) {
            l_img[j] = img;
            l_pix[j] = yr * g.oW + yc;
            l_ok[j] = ybase[j] >= 0;
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

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
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
        a_soff[i] = __builtin_amdgcn_readfirstlane((unsigned)mt * steps_total * 1024u);
    }
    bf16x8 a[NSETS][WM][3];
    auto load_a = [&](int set, int step) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                a[set][i][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                              a_rsrc, a_voff, a_soff[i] + t * a_term_bytes + (unsigned)step * 1024u, 0));
    };
    load_a(0, 0);

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
    bf16x8 bq[NSETS][WN];
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
            const int stn = (g.dbg & 4) ? 0 : (st < steps_total ? st : steps_total - 1);   // prefetch next step's weights (clamped at the end)
            load_a((s + 1) % NSETS, stn);
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
                    for (int j = 0; j < WN; ++j) bq[0][j] = *(const bf16x8*)(cur + qoff[j] + toff0);
                }
                if (s + 1 < NSTEPS) {
                    const int kk1 = (s + 1) / TAPS, tap1 = (s + 1) - kk1 * TAPS;
                    const int toff1 = g.tap_off[tap1] * 16 + kk1 * 2 * grp;
#pragma unroll
                    for (int j = 0; j < WN; ++j) bq[(s + 1) % NSETS][j] = *(const bf16x8*)(cur + qoff[j] + toff1);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j][0] = bq[s % NSETS][j];
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
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
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
    if (g.stats) {
        double* red = reinterpret_cast<double*>(smem + (size_t)WVM * WVN * EAS_STATS_SCRATCH * sizeof(float));
        bool valid[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) valid[j] = ybase[j] >= 0;
        eas_conv_stats_wave<WM, WN>(acc, valid, reinterpret_cast<float*>(smem) + wave * EAS_STATS_SCRATCH, red + wave * (WM * 64), lane);
        __syncthreads();
        if (tid < WVM * WM * 32) {
            const int wmi = tid / (WM * 32), t = tid - wmi * (WM * 32);
            const int co = (blockIdx.y * WVM + wmi) * WM * 32 + t;
            eas_conv_stats_store(red + wmi * WVN * (WM * 64), WM * 64, WVN, t, co, g.Cout, g.stats, g.stats_nb,
                                 (int)blockIdx.x + (int)gridDim.x * part);
        }
    }

    if constexpr (LM != 0) {
        // ---- fused BatchNorm (running statistics) + LIF over T: the accumulators are z_t of the lane's neurons, nothing is written but spikes
        if (mt0 < g.MT) eas_lif_epilogue<WN, LM>(acc[0], g.lif, mt0, h, l_img, l_pix, l_ok);
        return;
    }

    // ---- epilogue: D tile (col = pixel = lane&31, row = (e&3) + 8*(e>>2) + 4*h) -> y NCHW.  Lean on purpose: one pointer per
    // (M-tile, pixel tile), rows reached by adding multiples of the channel stride; bias values loaded once per M-tile;
    // the per-element channel bound check only for a ragged last M-tile (the epilogue used to be as long as the main loop).
    const long cstride = (long)g.oH * g.oW;
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
            if (full) {
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
    conv_tile_body<TAPS, S, XT, WM, WN, WVM, WVN, CCH, VEC, NIT, PL, LM>(x, wp, bias, y, inexact, g, (int)blockIdx.z, smem);
}

// Input gradient of a stride-2 3x3 convolution in ONE launch: blockIdx.z = parity class of the input pixel, every class a stride-1
// tap-list convolution over grad_y with its own tap count (1, 2, 2, 4), channel chunk (64, 32, 32, 16: four MFMA steps per chunk
// each), weights and tile geometry -- the four class kernels used to be four launches, each leaving most CUs idle on the small
// maps and each staging grad_y by itself.
struct S2Geoms {
    ConvGeom g[4];
    const bf16x8* wp[4];
};

template <int WM, int WN, int WVM, int WVN, int VEC, int NIT>
__global__ __launch_bounds__(64 * WVM * WVN, 2) void conv_dgrad_s2_kernel(const float* __restrict__ gy, float* __restrict__ gx, const S2Geoms sg) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int cls = blockIdx.z;
    if ((int)blockIdx.x * sg.g[cls].RT >= sg.g[cls].total_rows) return;
    switch (cls) {
        case 0: conv_tile_body<1, 1, 3, WM, WN, WVM, WVN, 64, VEC, NIT>(gy, sg.wp[0], nullptr, gx, nullptr, sg.g[0], 0, smem); break;
        case 1: conv_tile_body<2, 1, 3, WM, WN, WVM, WVN, 32, VEC, NIT>(gy, sg.wp[1], nullptr, gx, nullptr, sg.g[1], 0, smem); break;
        case 2: conv_tile_body<2, 1, 3, WM, WN, WVM, WVN, 32, VEC, NIT>(gy, sg.wp[2], nullptr, gx, nullptr, sg.g[2], 0, smem); break;
        default: conv_tile_body<4, 1, 3, WM, WN, WVM, WVN, 16, VEC, NIT>(gy, sg.wp[3], nullptr, gx, nullptr, sg.g[3], 0, smem); break;
    }
}

static void conv_geom_magics(ConvGeom& g, int VEC, int CCH) {
    const int units_row = g.Wst / VEC, units_seg = g.rows_in * units_row, units = g.nseg * units_seg;
    const int rows = g.nseg * g.rows_in, per_row = (g.RS - g.Wst) * (CCH * 2 / 16);
    g.m_Wo = fdiv_magic(g.Wo); g.m_rows_seg = fdiv_magic(g.rows_seg); g.m_Ho = fdiv_magic(g.Ho);
    g.m_units = fdiv_magic(units); g.m_units_seg = fdiv_magic(units_seg); g.m_units_row = fdiv_magic(units_row);
    g.m_hrow = fdiv_magic(per_row); g.m_rows = fdiv_magic(rows);
}

template <int WM, int WN, int WVM, int WVN, int VEC, int NIT>
int launch_s2(const float* gy, float* gx, S2Geoms sg, hipStream_t st) {
    auto kern = conv_dgrad_s2_kernel<WM, WN, WVM, WVN, VEC, NIT>;
    static const int cch[4] = {64, 32, 32, 16};
    size_t lds = 0;
    int gx_blocks = 0;
    for (int c = 0; c < 4; ++c) {
        ConvGeom& g = sg.g[c];
        const int nbuf = g.KSTEPS * 16 <= cch[c] ? 1 : 2;
        const size_t l = (size_t)nbuf * g.Q * cch[c] * 2 * 3;
        lds = l > lds ? l : lds;
        const int units = g.nseg * g.rows_in * (g.Wst / VEC);
        if (g.total_rows >= (1 << 20) || units >= (1 << 20)) return EAS_ERR_UNSUPPORTED;
        conv_geom_magics(g, VEC, cch[c]);
        const int b = (g.total_rows + g.RT - 1) / g.RT;
        gx_blocks = b > gx_blocks ? b : gx_blocks;
    }
    if (lds > 160 * 1024) return EAS_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(gx_blocks, (sg.g[0].MT + WVM * WM - 1) / (WVM * WM), 4);
    EAS_LAUNCH(kern, grid, dim3(64 * WVM * WVN), lds, st, gy, gx, sg);
    return EAS_OK;
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
    dim3 grid((g.total_rows + g.RT - 1) / g.RT, (g.MT + WVM * WM - 1) / (WVM * WM), g.parts);
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

template <int TAPS, int S, int XT, int CCH, int VEC, bool PL = false>
int dispatch_tile(const float* x, const bf16x8* wp, const float* bias, float* y, int* inexact, ConvGeom g, hipStream_t st) {
    // candidate block tiles {waves along M, waves along N}, each wave one 32-channel M-tile x 5 pixel tiles (160 pixels);
    // 8-wave blocks (2 waves per SIMD hide each other's staging, LDS and weight latencies) first, widest pixel tile first
    struct Cand { int wm, wvm, bn, threads, nit, wn; launch_fn fn; };
    constexpr int N8 = XT == 1 ? 2 : 1, N4 = 2;   // staging items per thread (register budget: 2 waves per SIMD either way)
    // wave tile = 32 channels x WN 32-pixel tiles.  WN = 5 (160 pixels) is the efficient shape; WN = 3 (96 pixels: two 40-pixel rows,
    // four 20-pixel rows or one 8x10 image) exists for the small maps of the neck / head and of dark4 / dark5, where 160-pixel wave
    // tiles leave most of the 256 CUs without a block (64 images of 8x10 with 128 channels are 16 eight-wave blocks)
    const Cand cands[14] = {
        {1, 2, 640, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 2, 4, CCH, VEC, N8, PL>}, {1, 4, 320, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 4, 2, CCH, VEC, N8, PL>},
        {1, 8, 160, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 8, 1, CCH, VEC, N8, PL>}, {1, 1, 1280, 512, N8, 5, launch_fwd<TAPS, S, XT, 1, 5, 1, 8, CCH, VEC, N8, PL>},
        {1, 1, 640, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 1, 4, CCH, VEC, N4, PL>}, {1, 2, 320, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 2, 2, CCH, VEC, N4, PL>},
        {1, 4, 160, 256, N4, 5, launch_fwd<TAPS, S, XT, 1, 5, 4, 1, CCH, VEC, N4, PL>},
        {1, 2, 384, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 2, 4, CCH, VEC, N8, PL>}, {1, 4, 192, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 4, 2, CCH, VEC, N8, PL>},
        {1, 8, 96, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 8, 1, CCH, VEC, N8, PL>}, {1, 1, 768, 512, N8, 3, launch_fwd<TAPS, S, XT, 1, 3, 1, 8, CCH, VEC, N8, PL>},
        {1, 1, 384, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 1, 4, CCH, VEC, N4, PL>}, {1, 2, 192, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 2, 2, CCH, VEC, N4, PL>},
        {1, 4, 96, 256, N4, 3, launch_fwd<TAPS, S, XT, 1, 3, 4, 1, CCH, VEC, N4, PL>}};
    // cost model (measured on MI355X, scripts/dev_conv.py, scripts/dev_conv_calls.py): one block per CU; a round of 8-wave blocks costs
    // ~1.27x a round of 4-wave blocks; a block's time is a fixed part (prologue, first patch, epilogue) plus its MFMA work, which
    // goes with WN whatever part of the pixel tile is valid
    int best = -1;
    double best_cost = 0.0;
    int best_valid = 0;
    ConvGeom best_g = g;
    static const int force = getenv("EAS_CONV_TILE") ? atoi(getenv("EAS_CONV_TILE")) : -1;   // development: force a candidate
    static const int ncand = getenv("EAS_CONV_NCAND") ? atoi(getenv("EAS_CONV_NCAND")) : 14;  // development: 7 = the 160-pixel wave tiles only
    const int nchunks = (g.KSTEPS * 16 + CCH - 1) / CCH;
    // Single-buffered patch (mode 1): for layers of few channel chunks whose double-buffered patch fills the LDS (real-valued inputs on wide
    // rows: dark2.0, 135 -> 103 us) -- the second buffer buys one overlapped stage there, a second resident block overlaps everything.
    // With more chunks the exposed write of every chunk costs more than it gains (4 chunks: 100 -> 122 us, 8: 125 -> 154 us); three
    // chunks (the 48-channel layers of SYOLOX-M) still gain: config 4 88.0 -> 87.5 ms.
    static const int single_nch = getenv("EAS_CONV_SINGLE_NCH") ? atoi(getenv("EAS_CONV_SINGLE_NCH")) : 3;      // development: 0 = never
    static const double single_pen = getenv("EAS_CONV_SINGLE_PEN") ? atof(getenv("EAS_CONV_SINGLE_PEN")) : 1.1;
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
        static const double lat = getenv("EAS_CONV_STEP_LAT") ? atof(getenv("EAS_CONV_STEP_LAT")) : 450.0;   // development
        const double per_tile = XT == 1 ? 96.0 : 192.0;
        const double mfma_part = (c.wn * per_tile > lat ? c.wn * per_tile : lat) / (5.0 * per_tile);
        const double round_cost = (c.threads == 512 ? 1.27 : (bpc == 2 && blocks > 256 ? 1.2 : 1.0)) * (0.3 + 0.7 * mfma_part) *
                                  (mode == 1 ? single_pen * (c.threads == 512 && blocks > 256 ? 1.2 : 1.0) : 1.0);
        const double cost = (double)((blocks + 256 * bpc - 1) / (256 * bpc)) * round_cost;
        // ties: the larger valid pixel count; among grids of lone blocks (<= 256) first the shape with more waves along the channels
        // (they share one staged pixel patch: the smaller patch per block is the shorter prologue; 8x10 head layers: 35-38 us against
        // 44-50 us for the four-pixel-group shape)
        const int valid = t.RT * g.Wo;
        static const int tie_wvm = getenv("EAS_CONV_TIE_WVM") ? atoi(getenv("EAS_CONV_TIE_WVM")) : 1;        // development: 0 = lone blocks only
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

// Tile choice of the fused eval step (LM = 1: time-major, the wave tile is lif.T time steps of 32 pixels; LM = 2: one result for all steps).
// Same cost model as dispatch_tile; the block covers 32 * WVN spatial pixels (time-major) whose T time slices are staged side by side.
template <int TAPS, int S, int XT, int CCH, int VEC, bool PL, int LM>
int dispatch_tile_lif(const float* x, const bf16x8* wp, ConvGeom g, hipStream_t st, bool query) {
    struct Cand { int wvm, wvn, threads, nit, wn; launch_fn fn; };
    constexpr int N8 = XT == 1 ? 2 : 1, N4 = 2;
#define EAS_LC(WVM_, WVN_, WN_, NIT_) {WVM_, WVN_, 64 * WVM_ * WVN_, NIT_, WN_, launch_fwd<TAPS, S, XT, 1, WN_, WVM_, WVN_, CCH, VEC, NIT_, PL, LM>}
    // (only the candidates of this LM are instantiated)
    constexpr int ncand = LM == 1 ? 10 : 4;
    Cand cands[ncand];
    if constexpr (LM == 1) {
        const Cand c[10] = {EAS_LC(2, 4, 3, N8), EAS_LC(4, 2, 3, N8), EAS_LC(1, 8, 3, N8), EAS_LC(1, 4, 3, N4), EAS_LC(2, 2, 3, N4),
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
    for (int i = 0; i < ncand; ++i) {
        const Cand& c = cands[i];
        if (LM == 1 && c.wn != T) continue;
        if ((c.wvm - 1) * 32 >= g.Cout && c.wvm != 1) continue;
        const int nbuf = nchunks == 1 ? 1 : 2;
        ConvGeom t = g;
        t.single = 0;
        const int bn = LM == 1 ? 32 * c.wvn : 32 * c.wn * c.wvn;       // spatial pixels of the block tile
        bool fits = false;
        for (int cap = bn; cap >= g.Wo && !fits; cap -= 32) {
            t.RT = pick_rows(g.Ho, g.Wo, cap);
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
        const long blocks = (long)((g.total_rows + t.RT - 1) / t.RT) * ((g.MT + c.wvm - 1) / c.wvm);
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
    return cands[best].fn(x, wp, nullptr, nullptr, nullptr, best_g, st);
}

}  // namespace

int eas_conv1x1_dispatch(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                         hipStream_t st, double* stats, int stats_nb, int* nb_out, int planes, int* inexact);
int eas_conv1x1_lif_dispatch(const void* x, const void* packed_w, const EasLifEpiDev& lif, int Cin, int x_terms, int x_shared, hipStream_t st, bool query);

extern "C" {

int64_t eas_conv_packed_weight_bytes(int Cout, int Cin, int ksize, int mode) {
    const int M = mode ? Cin : Cout, K = mode ? Cout : Cin;
    return (int64_t)3 * ((M + 31) / 32) * ((K + 15) / 16) * ksize * ksize * 64 * 16;
}

int eas_conv_pack_weights(const float* w, void* packed, int Cout, int Cin, int ksize, int mode, eas_stream_t stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || mode < 0 || mode > 2 || (mode == 2 && ksize != 3))
        return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    const int M = mode ? Cin : Cout, K = mode ? Cout : Cin;
    const int total = ((M + 31) / 32) * ((K + 15) / 16) * ksize * ksize * 64;
    EAS_LAUNCH(conv_pack_weights_kernel, dim3(eas_grid_1d(total)), dim3(EAS_BLOCK), 0, eas_s(stream), w, (bf16x8*)packed, Cout, Cin, ksize,
                       mode);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// jobs: device array of njobs x 8 int64 {weight pointer, packed pointer, Cout, Cin, ksize, mode, second weight pointer or 0, output
// channels of the first}; one launch packs them all (the weights of a model only change at the optimizer step, so a training step needs
// this once).  A second pointer packs the concatenation of two weight tensors along Cout (modes 0 and 1).
int eas_conv_pack_weights_many(const void* jobs, int njobs, eas_stream_t stream) {
    if (!jobs || njobs < 1) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(conv_pack_weights_many_kernel, dim3(24, njobs), dim3(EAS_BLOCK), 0, eas_s(stream), (const long long*)jobs);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// y[NI,Cout,Ho,Wo] = conv2d(x[NI,Cin,Hi,Wi], weights packed with mode 0 (or mode 1 for the stride-1 input gradient, called
// with grad_y as x and Cin/Cout swapped), padding ksize/2.  x_terms = 1: x holds small integers (spikes and their SEW sums,
// exact in bf16; `inexact_flag`, if given, is set to 1 should any element not be); x_terms = 3: general fp32 input.
static int conv_fwd_impl(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                         int stride, int x_terms, int* inexact_flag, eas_stream_t stream, bool query, double* stats = nullptr, int stats_nb = 0,
                         int* nb_out = nullptr) {
    if (!query && (!x || !packed_w || !y)) return EAS_ERR_INVALID_ARG;
    if (stats && (bias || stats_nb < 1)) return EAS_ERR_INVALID_ARG;
    if (NI <= 0 || Cin <= 0 || Cout <= 0 || Hi <= 0 || Wi <= 0) return EAS_ERR_INVALID_ARG;
    // x_terms: 1 = fp32 tensor holding small integers (one bf16 term), 3 = general fp32, 2 = SPIKE PLANES (bf16 [NI][Cin/8][Hi*Wi][8], one term)
    const bool planes = x_terms == 2;
    if (planes) x_terms = 1;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (x_terms != 1 && x_terms != 3) || (ksize == 3 && (Cin % 8 != 0 || Wi % 2 != 0)) ||
        (planes && Cin % 8 != 0))
        return EAS_ERR_UNSUPPORTED;
    if (planes && !query && ((uintptr_t)x & 15)) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    ConvGeom g{};
    const int pad = ksize / 2;
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.Hi = Hi; g.Wi = Wi;
    g.Ho = (Hi + 2 * pad - ksize) / stride + 1;
    g.Wo = (Wi + 2 * pad - ksize) / stride + 1;
    g.RS = Wi + 2 * pad;
    g.pad_t = g.pad_l = pad;
    g.ext_h = ksize;
    for (int t = 0; t < ksize * ksize; ++t) g.tap_off[t] = (t / ksize) * g.RS + (t % ksize);
    g.oH = g.Ho; g.oW = g.Wo; g.os = 1; g.oph = g.opw = 0;
    g.MT = (Cout + 31) / 32;
    g.KSTEPS = (Cin + 15) / 16;
    g.total_rows = NI * g.Ho;
    g.Wst = Wi; g.gx0 = 0; g.qshift = 0; g.parts = 1;
    static const int dbg = getenv("EAS_CONV_DBG") ? atoi(getenv("EAS_CONV_DBG")) : 0;
    g.dbg = dbg;
    g.stats = stats; g.stats_nb = stats_nb;
    const bf16x8* wp = (const bf16x8*)packed_w;
    hipStream_t st = eas_s(stream);
    int rc = EAS_ERR_UNSUPPORTED;
    const bool v4 = Wi % 4 == 0;
    tl_pixel_blocks = 0;
#define EAS_CONV_DISPATCH(TAPS_, S_, CCH_)                                                                                        \
    rc = planes ? (v4 ? dispatch_tile<TAPS_, S_, 1, CCH_, 4, true>(x, wp, bias, y, inexact_flag, g, st)                           \
                      : dispatch_tile<TAPS_, S_, 1, CCH_, 2, true>(x, wp, bias, y, inexact_flag, g, st))                          \
       : x_terms == 1 ? (v4 ? dispatch_tile<TAPS_, S_, 1, CCH_, 4>(x, wp, bias, y, inexact_flag, g, st)                           \
                            : dispatch_tile<TAPS_, S_, 1, CCH_, 2>(x, wp, bias, y, inexact_flag, g, st))                          \
                      : (v4 ? dispatch_tile<TAPS_, S_, 3, CCH_, 4>(x, wp, bias, y, inexact_flag, g, st)                           \
                            : dispatch_tile<TAPS_, S_, 3, CCH_, 2>(x, wp, bias, y, inexact_flag, g, st))
    const ConvGeom g_full = g;
    static const int force_parts = getenv("EAS_CONV_PARTS") ? atoi(getenv("EAS_CONV_PARTS")) : 0;   // development: force column parts
    for (int parts = force_parts > 0 ? force_parts : 1; parts <= 8 && rc == EAS_ERR_UNSUPPORTED && ksize == 3; parts *= 2) {
        g = g_full;
        if (parts > 1) {
            // column parts: the tile is Wo/parts output columns wide; the staged row holds their input columns plus one staging
            // unit (VEC columns, aligned) of halo on either side, all fetched like interior units (zero page outside the image)
            const int hv = v4 ? 4 : 2;
            if (g_full.Wo % parts != 0 || ((g_full.Wo / parts) * stride) % hv != 0) continue;
            g.parts = parts;
            g.Wo = g_full.Wo / parts;
            g.Wst = g.Wo * stride + 2 * hv;
            g.RS = g.Wst;
            g.pad_l = 0;
            g.gx0 = -hv;
            g.qshift = hv - pad;
            for (int t = 0; t < ksize * ksize; ++t) g.tap_off[t] = (t / ksize) * g.RS + (t % ksize);
        }
        if (stride == 1) { EAS_CONV_DISPATCH(9, 1, 16); }
        else { EAS_CONV_DISPATCH(9, 2, 16); }
        if (force_parts > 0) break;
    }
    if (ksize == 1 && stride == 1) {
        int nb1 = 0;
        rc = eas_conv1x1_dispatch(x, packed_w, bias, query ? nullptr : y, NI, Cin, Cout, Hi * Wi, x_terms, st, stats, stats_nb, &nb1, planes ? 1 : 0, inexact_flag);
        tl_pixel_blocks = nb1;
    }
#undef EAS_CONV_DISPATCH
    if (nb_out) *nb_out = rc == EAS_OK ? tl_pixel_blocks : 0;
    if (rc != EAS_OK || query) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

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

int eas_conv_fwd(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                 int stride, int x_terms, int* inexact_flag, eas_stream_t stream) {
    return conv_fwd_impl(x, packed_w, bias, y, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, inexact_flag, stream, false);
}

// eas_conv_fwd (no bias) that also leaves the per-channel sums of its output for the BatchNorm behind it: stats[Cout][nb][2] doubles,
// (sum, sum of squares) of every pixel block's tile, nb = eas_conv_fwd_stats_blocks(same geometry).  The consumer (eas_bn_lif_fwd_ex /
// eas_bn_silu_fwd_ex with EasBnPending.partial = stats, chunks = pitch = nb) adds them in a fixed order: no separate statistics pass over y.
int eas_conv_fwd_stats(const float* x, const void* packed_w, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride,
                       int x_terms, int* inexact_flag, double* stats, int nb, eas_stream_t stream) {
    if (!stats || nb < 1) return EAS_ERR_INVALID_ARG;
    return conv_fwd_impl(x, packed_w, nullptr, y, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, inexact_flag, stream, false, stats, nb);
}

// number of statistics partials per channel eas_conv_fwd_stats writes for this geometry (the pixel blocks of the tile it picks);
// 0 = no tile (eas_conv_fwd_supported is 0)
int eas_conv_fwd_stats_blocks(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms) {
    int nb = 0;
    const int rc = conv_fwd_impl(nullptr, nullptr, nullptr, nullptr, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, nullptr, nullptr, true, nullptr, 0,
                                 &nb);
    return rc == EAS_OK ? nb : 0;
}

// eas_conv_fwd / eas_conv_fwd_stats reading x as SPIKE PLANES (eas_hip.h): bf16 [NI][Cin/8][Hi*Wi][8] as eas_bn_lif_fwd_ex writes them.
// Same products summed in the same order as eas_conv_fwd with x_terms = 1 on the fp32 values: bit-identical y.  stats / nb: nullable /
// 0, else as eas_conv_fwd_stats (geometry queries: x_terms = 2).
int eas_conv_fwd_planes(const void* x_planes, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                        int stride, double* stats, int nb, eas_stream_t stream) {
    if (stats && (nb < 1 || bias)) return EAS_ERR_INVALID_ARG;
    return conv_fwd_impl(reinterpret_cast<const float*>(x_planes), packed_w, bias, y, NI, Cin, Cout, Hi, Wi, ksize, stride, 2, nullptr, stream, false,
                         stats, stats ? nb : 0);
}

// 1 when eas_conv_fwd has a tile for this geometry.  Rows whose staged patch does not fit LDS in one piece (3x3 layers with
// real-valued inputs wider than ~280 pixels: stem / dark2.0 of the 384x640 canvas) run in 2, 4 or 8 column parts.
int eas_conv_fwd_supported(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms) {
    return conv_fwd_impl(nullptr, nullptr, nullptr, nullptr, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, nullptr, nullptr, true) == EAS_OK ? 1 : 0;
}

// grad_x[NI,Cin,Hi,Wi] of a stride-2 3x3 convolution (padding 1) from grad_y[NI,Cout,Ho,Wo] and the weights packed with
// mode 2: four launches, one per parity class of the input pixel (1, 2, 2 and 4 taps), each a stride-1 tap-list
// convolution over grad_y (general fp32, three bf16 terms) that writes every second row/column of grad_x.
int eas_conv_dgrad_s2(const float* grad_y, const void* packed_w, float* grad_x, int NI, int Cin, int Cout, int Hi, int Wi, eas_stream_t stream) {
    if (!grad_y || !packed_w || !grad_x || NI <= 0 || Cin <= 0 || Cout <= 0 || Hi <= 0 || Wi <= 0) return EAS_ERR_INVALID_ARG;
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    if (Cout % 8 != 0 || Wo % 2 != 0) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    hipStream_t st = eas_s(stream);
    static const int dbg = getenv("EAS_CONV_DBG") ? atoi(getenv("EAS_CONV_DBG")) : 0;
    const int MT = (Cin + 31) / 32, KSTEPS = (Cout + 15) / 16;
    const int cum[4] = {0, 1, 3, 5};
    const bool v4 = Wo % 4 == 0;
    S2Geoms sg{};
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        ConvGeom& g = sg.g[cls];
        g.NI = NI; g.Cin = Cout; g.Cout = Cin; g.Hi = Ho; g.Wi = Wo; g.Ho = Ho; g.Wo = Wo;
        g.RS = Wo + pw;
        g.pad_t = g.pad_l = 0;
        g.ext_h = ph + 1;
        for (int ih = 0; ih <= ph; ++ih)
            for (int iw = 0; iw <= pw; ++iw) g.tap_off[ih * (pw + 1) + iw] = ih * g.RS + iw;
        g.oH = Hi; g.oW = Wi; g.os = 2; g.oph = ph; g.opw = pw;
        g.MT = MT; g.KSTEPS = KSTEPS;
        g.total_rows = NI * Ho;
        g.Wst = Wo; g.gx0 = 0; g.qshift = 0; g.parts = 1;
        g.dbg = dbg;
        sg.wp[cls] = (const bf16x8*)packed_w + (size_t)3 * MT * KSTEPS * 64 * cum[cls];
    }
    static const int one_launch = getenv("EAS_S2_FORM") ? (getenv("EAS_S2_FORM")[0] == '1') : 1;   // development: 0 = four launches
    if (one_launch) {
        // one block shape for the four classes, a tile geometry per class (their staged patches differ: taps, channel chunk)
        typedef int (*s2_fn)(const float*, float*, S2Geoms, hipStream_t);
        struct Shape { int wvm, bn, threads, nit, wn; s2_fn f4, f2; };
        const Shape shapes[14] = {
            {2, 640, 512, 1, 5, launch_s2<1, 5, 2, 4, 4, 1>, launch_s2<1, 5, 2, 4, 2, 1>}, {4, 320, 512, 1, 5, launch_s2<1, 5, 4, 2, 4, 1>, launch_s2<1, 5, 4, 2, 2, 1>},
            {8, 160, 512, 1, 5, launch_s2<1, 5, 8, 1, 4, 1>, launch_s2<1, 5, 8, 1, 2, 1>}, {1, 1280, 512, 1, 5, launch_s2<1, 5, 1, 8, 4, 1>, launch_s2<1, 5, 1, 8, 2, 1>},
            {1, 640, 256, 2, 5, launch_s2<1, 5, 1, 4, 4, 2>, launch_s2<1, 5, 1, 4, 2, 2>}, {2, 320, 256, 2, 5, launch_s2<1, 5, 2, 2, 4, 2>, launch_s2<1, 5, 2, 2, 2, 2>},
            {4, 160, 256, 2, 5, launch_s2<1, 5, 4, 1, 4, 2>, launch_s2<1, 5, 4, 1, 2, 2>},
            {2, 384, 512, 1, 3, launch_s2<1, 3, 2, 4, 4, 1>, launch_s2<1, 3, 2, 4, 2, 1>}, {4, 192, 512, 1, 3, launch_s2<1, 3, 4, 2, 4, 1>, launch_s2<1, 3, 4, 2, 2, 1>},
            {8, 96, 512, 1, 3, launch_s2<1, 3, 8, 1, 4, 1>, launch_s2<1, 3, 8, 1, 2, 1>}, {1, 768, 512, 1, 3, launch_s2<1, 3, 1, 8, 4, 1>, launch_s2<1, 3, 1, 8, 2, 1>},
            {1, 384, 256, 2, 3, launch_s2<1, 3, 1, 4, 4, 2>, launch_s2<1, 3, 1, 4, 2, 2>}, {2, 192, 256, 2, 3, launch_s2<1, 3, 2, 2, 4, 2>, launch_s2<1, 3, 2, 2, 2, 2>},
            {4, 96, 256, 2, 3, launch_s2<1, 3, 4, 1, 4, 2>, launch_s2<1, 3, 4, 1, 2, 2>}};
        static const int cch[4] = {64, 32, 32, 16};
        static const double work[4] = {0.45, 0.65, 0.65, 1.0};      // relative block time of the classes (1, 2, 2, 4 taps)
        const int vec = v4 ? 4 : 2;
        int best = -1;
        double best_cost = 0.0;
        S2Geoms best_sg = sg;
        double four[4] = {1e30, 1e30, 1e30, 1e30};      // the same model's cost of each class as a launch of its own (its best shape)
        for (int i = 0; i < 14; ++i) {
            const Shape& c = shapes[i];
            if ((c.wvm - 1) * 32 >= Cin && c.wvm != 1) continue;
            S2Geoms t = sg;
            bool ok = true;
            double load = 0.0;
            size_t lds_max = 0;
            for (int cls = 0; cls < 4 && ok; ++cls) {
                ConvGeom& g = t.g[cls];
                const int nbuf = g.KSTEPS * 16 <= cch[cls] ? 1 : 2;
                bool fits = false;
                for (int cap = c.bn; cap >= g.Wo && !fits; cap -= 32) {
                    g.RT = pick_rows(g.Ho, g.Wo, cap);
                    if (g.RT == 0) break;
                    g.rows_seg = g.RT < g.Ho ? g.RT : g.Ho;
                    g.nseg = g.RT / g.rows_seg;
                    g.rows_in = (g.rows_seg - 1) + g.ext_h;
                    g.Q = g.nseg * g.rows_in * g.RS;
                    fits = (size_t)nbuf * g.Q * cch[cls] * 2 * 3 <= 160 * 1024 && g.nseg * g.rows_in * (g.Wst / vec) * (cch[cls] / 8) <= c.nit * c.threads;
                    cap = g.RT * g.Wo;
                }
                ok = fits;
                if (!ok) break;
                const size_t l = (size_t)nbuf * g.Q * cch[cls] * 2 * 3;
                lds_max = l > lds_max ? l : lds_max;
                const long blocks = (long)((g.total_rows + g.RT - 1) / g.RT) * ((MT + c.wvm - 1) / c.wvm);
                load += work[cls] * (double)blocks;
                const int bpc1 = c.threads == 512 ? 1 : (2 * l <= 160 * 1024 ? 2 : 1);
                const double alone = (double)((blocks + 256 * bpc1 - 1) / (256 * bpc1)) * (c.threads == 512 ? 1.27 : (bpc1 == 2 ? 1.2 : 1.0)) *
                                     (0.3 + 0.7 * c.wn / 5.0) * work[cls];
                if (alone < four[cls]) four[cls] = alone;
            }
            if (!ok) continue;
            const int bpc = c.threads == 512 ? 1 : (2 * lds_max <= 160 * 1024 ? 2 : 1);
            const double round_cost = (c.threads == 512 ? 1.27 : (bpc == 2 ? 1.2 : 1.0)) * (0.3 + 0.7 * c.wn / 5.0);
            double rounds = load / (256.0 * bpc);
            if (rounds < 1.0) rounds = 1.0;
            const double cost = rounds * round_cost;
            if (best < 0 || cost < best_cost) {
                best = i; best_cost = cost; best_sg = t;
            }
        }
        // large maps: every class fills the chip by itself and does better with a tile of its own -> four launches
        if (best >= 0 && four[0] + four[1] + four[2] + four[3] < 0.97 * best_cost) best = -1;
        if (best >= 0) {
            const int rc = v4 ? shapes[best].f4(grad_y, grad_x, best_sg, st) : shapes[best].f2(grad_y, grad_x, best_sg, st);
            if (rc != EAS_OK) return rc;
            EAS_CHECK_LAUNCH();
            return EAS_OK;
        }
    }
    for (int cls = 0; cls < 4; ++cls) {
        const ConvGeom& g = sg.g[cls];
        const bf16x8* wp = sg.wp[cls];
        int rc;
        if (cls == 0) rc = v4 ? dispatch_tile<1, 1, 3, 64, 4>(grad_y, wp, nullptr, grad_x, nullptr, g, st) : dispatch_tile<1, 1, 3, 64, 2>(grad_y, wp, nullptr, grad_x, nullptr, g, st);
        else if (cls == 3) rc = v4 ? dispatch_tile<4, 1, 3, 16, 4>(grad_y, wp, nullptr, grad_x, nullptr, g, st) : dispatch_tile<4, 1, 3, 16, 2>(grad_y, wp, nullptr, grad_x, nullptr, g, st);
        else rc = v4 ? dispatch_tile<2, 1, 3, 32, 4>(grad_y, wp, nullptr, grad_x, nullptr, g, st) : dispatch_tile<2, 1, 3, 32, 2>(grad_y, wp, nullptr, grad_x, nullptr, g, st);
        if (rc != EAS_OK) return rc;
        EAS_CHECK_LAUNCH();
    }
    return EAS_OK;
}

}  // extern "C"
