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
//     bytes (pixel-major 32-byte rows cost a 2-way bank conflict on every fragment read).  Conflict-free reads hold while a
//     32-pixel fragment stays inside one staged row; a tile that wraps rows jumps by RS - Wo + 1 pixels (2-way and worse on
//     the 40- / 20- / 10-pixel rows; padding the staged rows to (RS - Wo) % 16 == 0 removes it -- EAS_CONV_RS_PAD, measured
//     without effect on config 2 and slower on config 3, off by default).  The stride-2 forward reads every second
//     16-byte slot (2-way) and the staging stores of a thread's VEC consecutive pixels are VEC-way conflicted
//     (scripts/lds_conflict_model.py reproduces the measured SQ_LDS_BANK_CONFLICT share); the LDS pipe is active for 3-10 %
//     of these kernels' wave cycles, so neither is what they wait for.
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
#include "conv_mfma_body.h"

namespace {

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

// The same with the work spread evenly: the launch above gives every job the same 24 blocks, so the largest tensors (dark5: 2.4 M weights,
// two orders) set its duration while the blocks of the ~200 small ones have long finished (103 us for SYOLOX-S against ~25 us of traffic).
// Here a block takes 256-lane units from ONE list over all jobs: every block builds the jobs' unit prefix in LDS (<= 1024 jobs) and finds
// the job of a unit by binary search.
constexpr int kPackJobsMax = 1024;
__global__ __launch_bounds__(256) void conv_pack_weights_flat_kernel(const long long* __restrict__ jobs, int njobs) {
    __shared__ int pref[kPackJobsMax + 1];
    for (int j = threadIdx.x; j < njobs; j += blockDim.x) {
        const long long* q = jobs + (size_t)j * 8;
        pref[j + 1] = (pack_total((int)q[2], (int)q[3], (int)q[4], (int)q[5]) + 255) / 256;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pref[0] = 0;
        for (int j = 0; j < njobs; ++j) pref[j + 1] += pref[j];
    }
    __syncthreads();
    const int units = pref[njobs];
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        int lo = 0, hi = njobs;                       // the job with pref[j] <= u < pref[j + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pref[mid] <= u) lo = mid; else hi = mid;
        }
        const long long* q = jobs + (size_t)lo * 8;
        const float* w = (const float*)q[0];
        bf16x8* wp = (bf16x8*)q[1];
        const int Cout = (int)q[2], Cin = (int)q[3], ksize = (int)q[4], mode = (int)q[5];
        const float* wb = (const float*)q[6];
        const int ca = (int)q[7];
        const int total = pack_total(Cout, Cin, ksize, mode);
        const int idx = (u - pref[lo]) * 256 + (int)threadIdx.x;
        if (idx < total) {
            if (mode == 2) pack_fragment_s2dgrad(w, wp, Cout, Cin, idx);
            else pack_fragment(w, wp, Cout, Cin, ksize * ksize, mode, idx, total, wb, ca);
        }
    }
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
        case 0: conv_tile_body<1, 1, 3, WM, WN, WVM, WVN, 64, VEC, NIT>(gy, sg.wp[0], nullptr, gx, nullptr, sg.g[0], 0, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x); break;
        case 1: conv_tile_body<2, 1, 3, WM, WN, WVM, WVN, 32, VEC, NIT>(gy, sg.wp[1], nullptr, gx, nullptr, sg.g[1], 0, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x); break;
        case 2: conv_tile_body<2, 1, 3, WM, WN, WVM, WVN, 32, VEC, NIT>(gy, sg.wp[2], nullptr, gx, nullptr, sg.g[2], 0, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x); break;
        default: conv_tile_body<4, 1, 3, WM, WN, WVM, WVN, 16, VEC, NIT>(gy, sg.wp[3], nullptr, gx, nullptr, sg.g[3], 0, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x); break;
    }
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

}  // namespace

int eas_conv1x1_dispatch(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int HW, int x_terms,
                         hipStream_t st, double* stats, int stats_nb, int* nb_out, int planes, int* inexact, int act, const void* bna);
// conv_s2d.hip: the stride-2 input gradient with the four parity classes in one tile
int eas_conv_dgrad_s2c_dispatch(const float* gy, const void* packed_w, float* gx, int NI, int Cin, int Cout, int Hi, int Wi, hipStream_t st, bool query);
// conv_lif_mfma.hip: the 3x3 tiles with the BatchNorm + activation epilogue (kernel template LM = 3), fp32 three-term input
int eas_conv3x3_bna_dispatch(int stride, int v4, const float* x, const void* wp, float* y, const void* geom, hipStream_t st);

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
    static const bool flat = !(eas_dev_env("EAS_PACK_FORM") && eas_dev_env("EAS_PACK_FORM")[0] == 'j');      // EAS_PACK_FORM=jobs: 24 blocks per job (development)
    if (flat && njobs <= kPackJobsMax)
        EAS_LAUNCH(conv_pack_weights_flat_kernel, dim3(2048), dim3(256), 0, eas_s(stream), (const long long*)jobs, njobs);
    else
        EAS_LAUNCH(conv_pack_weights_many_kernel, dim3(24, njobs), dim3(EAS_BLOCK), 0, eas_s(stream), (const long long*)jobs);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// y[NI,Cout,Ho,Wo] = conv2d(x[NI,Cin,Hi,Wi], weights packed with mode 0 (or mode 1 for the stride-1 input gradient, called
// with grad_y as x and Cin/Cout swapped), padding ksize/2.  x_terms = 1: x holds small integers (spikes and their SEW sums,
// exact in bf16; `inexact_flag`, if given, is set to 1 should any element not be); x_terms = 3: general fp32 input.
static int conv_fwd_impl(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                         int stride, int x_terms, int* inexact_flag, eas_stream_t stream, bool query, double* stats = nullptr, int stats_nb = 0,
                         int* nb_out = nullptr, int act = 0, const EasBnActEpiDev* bna = nullptr) {
    if (!query && (!x || !packed_w || !y)) return EAS_ERR_INVALID_ARG;
    if (bna && (act || stats || bias)) return EAS_ERR_INVALID_ARG;
    if (bna && x_terms != 3) return EAS_ERR_UNSUPPORTED;      // the epilogue exists for real-valued fp32 inputs (the ANN neck / head)
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
    {
        // Rows narrower than / not a multiple of the 32-pixel fragment: a fragment's lanes then cross a staged-row boundary, where the pixel
        // index jumps by RS - Wo + 1.  With (RS - Wo) a multiple of 16 the lanes' 16-byte slots stay distinct modulo 16 across the jump, i.e.
        // the ds_read_b128 stays conflict-free (scripts/lds_conflict_model.py: 32x40 maps 3.2 -> 2.0 LDS cycles per half wave, 16x20 maps
        // 4.0 -> 2.0); costs 14 more staged (zero) columns per row.  Measured (same box): config 2 unchanged (21.23 / 21.24 ms without,
        // 21.24 / 21.20 ms with), config 3 SLOWER (45.19 -> 46.7 ms: the wider patches of the 96 / 192-channel layers cost a resident block) --
        // the LDS pipe is not what these kernels wait for.  Off by default; EAS_CONV_RS_PAD=1: the padded rows.
        static const int rs_pad = eas_dev_env("EAS_CONV_RS_PAD") ? atoi(eas_dev_env("EAS_CONV_RS_PAD")) : 0;
        if (rs_pad && ksize == 3 && stride == 1 && g.Wo % 32 != 0) g.RS += (16 - ((g.RS - g.Wo) % 16)) % 16;
    }
    g.pad_t = g.pad_l = pad;
    g.ext_h = ksize;
    for (int t = 0; t < ksize * ksize; ++t) g.tap_off[t] = (t / ksize) * g.RS + (t % ksize);
    g.oH = g.Ho; g.oW = g.Wo; g.os = 1; g.oph = g.opw = 0;
    g.MT = (Cout + 31) / 32;
    g.KSTEPS = (Cin + 15) / 16;
    g.total_rows = NI * g.Ho;
    g.Wst = Wi; g.gx0 = 0; g.qshift = 0; g.parts = 1;
    static const int dbg = eas_dev_env("EAS_CONV_DBG") ? atoi(eas_dev_env("EAS_CONV_DBG")) : 0;
    g.dbg = dbg;
    g.stats = stats; g.stats_nb = stats_nb;
    g.act = act;
    if (bna) g.bna = *bna;
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
    static const int force_parts = eas_dev_env("EAS_CONV_PARTS") ? atoi(eas_dev_env("EAS_CONV_PARTS")) : 0;   // development: force column parts
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
        if (bna) rc = eas_conv3x3_bna_dispatch(stride, v4 ? 1 : 0, x, wp, query ? nullptr : y, &g, st);
        else if (stride == 1) { EAS_CONV_DISPATCH(9, 1, 16); }
        else { EAS_CONV_DISPATCH(9, 2, 16); }
        if (force_parts > 0) break;
    }
    if (ksize == 1 && stride == 1) {
        int nb1 = 0;
        rc = eas_conv1x1_dispatch(x, packed_w, bias, query ? nullptr : y, NI, Cin, Cout, Hi * Wi, x_terms, st, stats, stats_nb, &nb1, planes ? 1 : 0, inexact_flag, act, bna);
        tl_pixel_blocks = nb1;
    }
#undef EAS_CONV_DISPATCH
    if (nb_out) *nb_out = rc == EAS_OK ? tl_pixel_blocks : 0;
    if (rc != EAS_OK || query) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_conv_fwd(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                 int stride, int x_terms, int* inexact_flag, eas_stream_t stream) {
    return conv_fwd_impl(x, packed_w, bias, y, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, inexact_flag, stream, false);
}

// eas_conv_fwd with an activation in the epilogue: y = act(conv(x) + bias), act 1 = SiLU.  The eval-mode form of a real-valued BaseConv
// after fuse_model folded its BatchNorm into weights and bias (yolox/utils/model_utils.py:35-80; network_blocks.py:55-56 fuseforward):
// ONE kernel instead of convolution + BatchNorm + SiLU passes.
int eas_conv_fwd_act(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                     int stride, int x_terms, int act, int* inexact_flag, eas_stream_t stream) {
    if (act < 0 || act > 1) return EAS_ERR_INVALID_ARG;
    return conv_fwd_impl(x, packed_w, bias, y, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, inexact_flag, stream, false, nullptr, 0, nullptr, act);
}

// conv -> BatchNorm (running statistics) -> activation in one kernel (include/eas_hip.h): eas_conv_fwd's tiles, the epilogue normalises and
// activates the accumulators (conv_lif_epi.h eas_bnact_epilogue) and writes them where the caller's concatenation wants them.
int eas_conv_bn_act_eval(const EasConvBnActEval* d, int* inexact_flag, eas_stream_t stream) {
    if (!d || !d->x || !d->packed_w || d->act < 0 || d->act > 1) return EAS_ERR_INVALID_ARG;
    if (d->Cout % 8 || d->csplit % 8 || d->csplit < 8 || d->csplit > d->Cout) return EAS_ERR_UNSUPPORTED;
    EasBnActEpiDev b{};
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
    return conv_fwd_impl((const float*)d->x, d->packed_w, nullptr, d->range[0].out, d->NI, d->Cin, d->Cout, d->Hi, d->Wi, d->ksize, d->stride,
                         d->x_terms, inexact_flag, stream, false, nullptr, 0, nullptr, 0, &b);
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
    static const int dbg = eas_dev_env("EAS_CONV_DBG") ? atoi(eas_dev_env("EAS_CONV_DBG")) : 0;
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
    // development: EAS_S2_FORM=1 the class kernels in one launch (round 4), 0 = in four launches; default: the four classes in one tile
    static const char form = eas_dev_env("EAS_S2_FORM") ? eas_dev_env("EAS_S2_FORM")[0] : 'c';
    if (form == 'c') {
        const int rc = eas_conv_dgrad_s2c_dispatch(grad_y, packed_w, grad_x, NI, Cin, Cout, Hi, Wi, st, false);
        if (rc == EAS_OK) {
            EAS_CHECK_LAUNCH();
            return EAS_OK;
        }
        if (rc != EAS_ERR_UNSUPPORTED) return rc;
    }
    const int one_launch = form != '0';
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
