// K4 (conv half): weight gradient of the 3x3 convolutions of the conv -> BN -> LIF step on the CDNA4 matrix cores
// (replaces ATen/MIOpen convolution_backward's grad_weight for nn.Conv2d inside BaseConv,
// yolox/models/network_blocks.py:31-56 after yolox/utils/utils_snn.py:25-27).
//
//   dW[co][ci][kh][kw] = sum_{n,ho,wo} gy[n][co][ho][wo] * x[n][ci][ho*S+kh-1][wo*S+kw-1]
//
// GEMM view, one per tap: D_tap[co][ci] = sum_p A[co][p] * B_tap[p][ci] with the reduction index p = output pixel.
// Both operands are staged pixel-major in LDS as exact bf16 terms ([term][32-channel plane][pixel][32 ch], 64-byte rows:
// conflict-free for the transposing read), grad_y always as three terms (hi+mid+lo = the fp32 value), x as one term when
// it holds spikes / small integers or three for general fp32.  Each lane's 8 consecutive reduction indices of one channel
// come from two ds_read_b64_tr_b16 (4 pixel rows x 16 channel columns per 16-lane group, delivered column-major), and the
// nine taps are plain address offsets into the same x image (zero halo staged with it).
//
// Block = 3*PM*PN waves: PM x PN (32 co x 32 ci) output tiles, three waves per tile (one kernel row kh each, 3 taps =
// 48 accumulator registers).  A block owns one (co block, ci block) and every KSL-th 80-pixel tile of output rows; its
// partial sums go to a slab that eas_conv_wgrad's second kernel reduces in fixed order (deterministic, no float atomics).
#include <stdio.h>
#include <stdlib.h>

#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TP = 80;       // output pixels per tile (5 MFMA k-steps of 16)
constexpr int KST = TP / 16;
#ifndef EAS_WG_COMMIT_KS
#define EAS_WG_COMMIT_KS (KST - 2)
#endif
constexpr int COMMIT_KS = EAS_WG_COMMIT_KS;      // k-step after which the next tile's registers are converted and written to LDS
constexpr int ROWB = 64;     // bytes per staged pixel row of one 32-channel plane
constexpr int A_PLANE = TP * ROWB;

// staging loads of rows outside the image / channels past the tensor read these zeros (channel stride 0)
__device__ __attribute__((aligned(16))) float eas_wg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct WgGeom {
    int NI, Cin, Cout, Hi, Wi, Ho, Wo;
    int RT, rows_seg, nseg, rows_in, RS, Q;
    int total_rows, ntiles, kslices;
    int ci_blocks;
    int pitchX, pitchY;      // row strides of x / grad_y in floats = the full image widths
    // Column parts (output rows wider than the 80-pixel reduction tile, or input rows whose staged patch does not fit LDS: stem and
    // dark2.0 at 128x160 and wider): a tile is `Wo` output columns of the RT rows, Wo = full width / parts, and consecutive tile
    // indices walk the parts of one row block (kslices is a multiple of parts, so a block always works on the same part and its
    // per-thread geometry stays tile-independent).  The staged x rows then hold input columns part*Wo*S - hv .. (Wst of them,
    // hv = one staging unit of halo on either side, units outside the image read the zero page) and there is no separately
    // zeroed halo: lpad = 0, qshift = hv - 1.  One part: Wst = Wi, lpad = 1, qshift = 0.
    int parts, Wst, lpad, qshift, hv;
    int pn;                  // ci tiles per block chosen by wg_geom (1 or 2)
    int single;              // one LDS buffer (the next tile is written after a barrier): where that lets a CU hold two blocks instead of one
    int nit;                 // staging items per thread the tile needs (wg_geom): 3 = the NITX = 3 instances (stride 2, spike inputs, 6-wave blocks)
};

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    lo = (__bf16)r2;
}

// two transposing reads -> one MFMA operand fragment (8 reduction indices of this lane's channel)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NTERMS>
__device__ __forceinline__ void stage_terms(unsigned char* dst, int term_stride, const float (&v)[8]) {
    if constexpr (NTERMS == 1) {
        bf16x8 t0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t0[j] = (__bf16)v[j];
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

// XPL: x is a spike-plane tensor (bf16 [NI][Cin/8][Hi*Wi][8], XT = 1): an x staging item is VEC 16-byte loads copied as they are into
// the pixel-major LDS image (the 8 channels of a pixel are 16 contiguous bytes in both).
// bx / by: the block's pixel slice (its first tile and its slab) and (co, ci) block index -- blockIdx.x / .y of a launch of one layer; a
// grouped launch (conv_wgrad_group_kernel) maps its flat grid onto (layer, slice, channel block) first
// NITX: staging items per thread, 0 = the rule below (1 for the 12-wave blocks, else 2); 3 = the stride-2 spike-input tiles whose x patch
// (four input pixels per output pixel) does not fit two items per thread at 80 output pixels (wg_geom, WgGeom.nit)
template <int S, int XT, int PM, int PN, int VEC, bool XPL, int NITX = 0>
__device__ __forceinline__ void conv_wgrad_body(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ slabs,
                                                const WgGeom& g, unsigned char* smem, const int bx, const int by) {
    static_assert(!XPL || XT == 1, "spike planes are one exact bf16 term");
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NW = 3 * PM * PN, NT = 64 * NW, NIT = NITX ? NITX : (PM * PN == 4 ? 1 : 2);   // staging items per thread (register budget of 12-wave blocks)
    constexpr int A_TERM = PM * A_PLANE, A_BYTES = 3 * A_TERM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave % 3, pair = wave / 3, pm = pair / PN, pn = pair % PN;
    const int cob = by / g.ci_blocks, cib = by - cob * g.ci_blocks;
    const int co0 = cob * 32 * PM, ci0 = cib * 32 * PN;
    const int b_plane = g.Q * ROWB, b_term = PN * b_plane;
    const int buf_bytes = A_BYTES + XT * b_term;
    const int npix = g.RT * g.Wo;

    // zero the halo columns of the x image in both buffers (never written again; column parts stage their halo like any unit)
    if (g.parts == 1) {
        const int rows = g.nseg * g.rows_in;
        for (int i = tid; i < rows * 2 * 4 * PN * XT * (g.single ? 1 : 2); i += NT) {   // 2 sides x 4 16-byte pieces x planes x terms x buffers
            int rest = i;
            const int piece = rest & 3; rest >>= 2;
            const int side = rest & 1; rest >>= 1;
            const int row = rest % rows; rest /= rows;
            const int pl = rest % (PN * XT), buf = rest / (PN * XT);
            const int col = side ? g.RS - 1 : 0;
            *(uint4*)(smem + (size_t)buf * buf_bytes + A_BYTES + (size_t)pl * b_plane + ((size_t)row * g.RS + col) * ROWB + piece * 16) =
                make_uint4(0, 0, 0, 0);
        }
    }

    if (PN == 1 && g.Cin - ci0 < 32) {
        const int ng = (g.Cin - ci0 + 7) / 8;                  // groups [ng, 4) of every staged pixel row stay zero
        for (int i = tid; i < g.Q * (4 - ng) * XT * (g.single ? 1 : 2); i += NT) {
            int rest = i;
            const int grp = ng + rest % (4 - ng); rest /= (4 - ng);
            const int q = rest % g.Q; rest /= g.Q;
            const int t = rest % XT, buf = rest / XT;
            *(uint4*)(smem + (size_t)buf * buf_bytes + A_BYTES + (size_t)t * b_term + (size_t)q * ROWB + grp * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    // ---- per-lane transposing-read geometry: 16-lane group gq -> channel column block (gq&1)*16, reduction rows (gq>>1)*8 ..
    const int gq = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
    const int lane_col_bytes = ((gq & 1) * 16 + 4 * pcol) * 2;
    const int kb = (gq >> 1) * 8 + qrow;                       // first reduction row (pixel) this lane addresses
    const int a_lane = pm * A_PLANE + kb * ROWB + lane_col_bytes;   // + ks*16*ROWB + rd*4*ROWB + term*A_TERM
    int b_lane[KST][2];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int p = ks * 16 + kb + 4 * rd;
            int q = 0;
            if (p < npix) {
                const int rl = p / g.Wo, c = p - rl * g.Wo;
                const int seg = rl / g.rows_seg, rr = rl - seg * g.rows_seg;
                q = (seg * g.rows_in + rr * S) * g.RS + c * S + g.qshift;
            }
            b_lane[ks][rd] = A_BYTES + pn * b_plane + q * ROWB + lane_col_bytes;   // + tap offset + term*b_term
        }

    // ---- staging items: item = VEC pixels x 8 channels; ids [0,nA) grad_y, [nA,nA+nB) x
    const int unitsA = TP / VEC, nA = unitsA * 4 * PM;
    // 8-channel groups of x that exist (a layer with fewer than 32 input channels -- the stem has 8 -- stages only those; the
    // other groups of its single plane are zeroed once, below)
    const int nbg = PN == 1 ? (g.Cin - ci0 >= 32 ? 4 : (g.Cin - ci0 + 7) / 8) : 4 * PN;
    const int units_row = g.Wst / VEC, units_seg = g.rows_in * units_row, unitsB = g.nseg * units_seg, nB = unitsB * nbg;
    const int part = g.parts > 1 ? bx % g.parts : 0;
    const int ycol0 = part * g.Wo;                              // first grad_y column of this block's tiles
    const int xcol0 = g.parts > 1 ? ycol0 * S - g.hv : 0;       // input column of the first staged unit (may be < 0)
    const int nitems = nA + nB;
    const size_t planeY = (size_t)g.Ho * g.pitchY, planeX = (size_t)g.Hi * g.pitchX;
    // Addresses: a staged element sits at  tile base (scalar: image img0, row r0 of the tile)  +  a per-item constant, so the constant is
    // worked out ONCE (it_off, bytes) and a tile costs an item a bounds test instead of the whole 64-bit address arithmetic (the per-tile
    // address code was 340 vector-ALU instructions per wave and tile against 45 MFMAs: the kernel was bound by the vector ALU).  Loads are
    // raw buffer loads: the descriptor's base is the tile base, the 8 channel rows of an fp32 item differ by a SCALAR offset, and an item
    // that is out of the image (halo rows, images past NI, channels past the tensor, padding) uses an offset past num_records, for which
    // the hardware returns zeros without touching memory (measured on gfx950: the range check covers voffset + soffset, scripts/micro/buffer_oob.hip;
    // masked items do not rely on it -- their voffset alone is past num_records; launch_wgrad refuses tiles whose span would reach 2^31).
    // Spike-input instances only (BUF): with three-term x the pointer form below measured 5-13 % faster (those instances spend twice the
    // time in MFMAs per tile and lost more to the changed instruction order than they gained).
    constexpr bool BUF = XT == 1;
    constexpr unsigned OOB = 0x80000000u;     // = num_records of the descriptors
    int it_kind[NIT], it_seg[NIT], it_row[NIT], it_lofs[NIT];   // kind: 0 grad_y, 1 x, 2 padding (zeros)
    unsigned it_off[BUF ? NIT : 1];
    int it_col[BUF ? 1 : NIT], it_ch[BUF ? 1 : NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int id = it * NT + tid;
        id = id < nitems ? id : nitems - 1;
        if (id < nA) {
            // channel group fastest: the lanes of one ds_write_b128 then spread over the four 16-byte pieces of a staged pixel row and
            // over both channel planes (with the pixel unit fastest every lane of a wave wrote the SAME four banks -- rows 4 apart are
            // 256 bytes apart; 5-8 % of the kernel time on the SYOLOX-M layers)
            const int u = id / (4 * PM), gi = id - u * (4 * PM);
            const int p0 = u * VEC;
            const int rl = p0 / g.Wo, c = p0 - rl * g.Wo;
            it_kind[it] = p0 < npix ? 0 : 2;
            it_seg[it] = rl / g.rows_seg;
            it_row[it] = rl - it_seg[it] * g.rows_seg;
            const int ch = co0 + gi * 8;
            if constexpr (BUF) {
                it_off[it] = (p0 < npix && ch < g.Cout)
                                 ? (unsigned)((((size_t)it_seg[it] * g.Cout + ch) * g.Ho + it_row[it]) * g.pitchY + ycol0 + c) * 4u : OOB;
            } else {
                it_col[it] = c;
                it_ch[it] = gi * 8;
            }
            it_lofs[it] = (gi >> 2) * A_PLANE + p0 * ROWB + (gi & 3) * 16;
        } else {
            const int idb = id - nA;
            const int u = idb / nbg, gi = idb - u * nbg;
            const int seg = u / units_seg, rem = u - seg * units_seg;
            const int rl = rem / units_row, cu = rem - rl * units_row;
            it_kind[it] = 1;
            it_seg[it] = seg;
            it_row[it] = rl;
            const int ch = ci0 + gi * 8, gc = xcol0 + cu * VEC;
            if constexpr (BUF) {
                const bool sok = ch < g.Cin && gc >= 0 && gc < g.pitchX;
                // relative to input row r0 * S - 1 of image img0 (the tile base): planes in 16-byte pixels of a group, fp32 in floats
                it_off[it] = !sok ? OOB
                             : XPL ? (unsigned)((((size_t)seg * (g.Cin / 8) + (ch >> 3)) * g.Hi + rl) * g.pitchX + gc) * 16u
                                   : (unsigned)((((size_t)seg * g.Cin + ch) * g.Hi + rl) * g.pitchX + gc) * 4u;
            } else {
                it_col[it] = cu * VEC;
                it_ch[it] = gi * 8;
            }
            it_lofs[it] = A_BYTES + (gi >> 2) * b_plane + ((seg * g.rows_in + rl) * g.RS + g.lpad + cu * VEC) * ROWB + (gi & 3) * 16;
        }
    }

    vecf L[NIT][8];
    // 12-wave instances (one item per thread, 168 registers): a spike-plane item, VEC x 16 bytes, lives in the first registers of the same
    // array as raw bits -- an item is one or the other, and a separate array cost 16 registers they do not have (spills in the tile
    // loop).  With two items per thread the compiler keeps the shared array in scratch instead, so those keep two arrays.
    constexpr bool UNI = NIT == 1;
    u32x4 LP[UNI ? 1 : NIT][(XPL && !UNI) ? VEC : 1];
    const unsigned planeYb = (unsigned)planeY * 4u, planeXb = (unsigned)planeX * 4u;
    auto fetch = [&](int it, int tile) {
        const int rho0 = (g.parts > 1 ? tile / g.parts : tile) * g.RT;
        const int img0 = rho0 / g.Ho, r0 = rho0 - img0 * g.Ho;
        if constexpr (!BUF) {
            // pointer form: addresses from scratch per tile, lanes outside the image read a zero page with channel stride 0
            const int img = img0 + it_seg[it];
            const float* src;
            size_t plane;
            bool ok;
            if (it_kind[it] == 1) {
                const int ir = r0 * S - 1 + it_row[it];
                const int gc = xcol0 + it_col[it];
                ok = ir >= 0 && ir < g.Hi && img < g.NI && ci0 + it_ch[it] < g.Cin && gc >= 0 && gc < g.pitchX;
                const int ch = ci0 + it_ch[it] < g.Cin ? ci0 + it_ch[it] : g.Cin - 8;
                src = x + (((size_t)(ok ? img : 0) * g.Cin + ch) * g.Hi + (ok ? ir : 0)) * g.pitchX + (ok ? gc : 0);
                plane = planeX;
            } else {
                ok = it_kind[it] == 0 && img < g.NI && co0 + it_ch[it] < g.Cout;
                const int ch = co0 + it_ch[it] < g.Cout ? co0 + it_ch[it] : g.Cout - 8;
                src = gy + (((size_t)(ok ? img : 0) * g.Cout + ch) * g.Ho + (ok ? r0 + it_row[it] : 0)) * g.pitchY + (ok ? ycol0 + it_col[it] : 0);
                plane = planeY;
            }
            const float* sp = ok ? src : eas_wg_zero_page;
            const size_t cs = ok ? plane : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) L[it][j] = *(const vecf*)(sp + j * cs);
        } else {
        const bool img_ok = img0 + it_seg[it] < g.NI;
        if (it_kind[it] == 1) {
            const int r0s = r0 * S - 1;
            const bool ok = img_ok && (unsigned)(r0s + it_row[it]) < (unsigned)g.Hi;
            const unsigned voff = ok ? it_off[it] : OOB;
            if constexpr (XPL) {
                const char* base = reinterpret_cast<const char*>(x) + (((long)img0 * (g.Cin / 8)) * g.Hi + r0s) * (long)g.pitchX * 16;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, OOB, 0x00020000);
#pragma unroll
                for (int p = 0; p < VEC; ++p) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * p, 0, 0);
                    if constexpr (!UNI) {
                        LP[it][p] = v;
                    } else if constexpr (VEC == 4) {
                        L[it][p] = __builtin_bit_cast(vecf, v);
                    } else {
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        L[it][2 * p] = __builtin_bit_cast(vecf, (u32x2)__builtin_shufflevector(v, v, 0, 1));
                        L[it][2 * p + 1] = __builtin_bit_cast(vecf, (u32x2)__builtin_shufflevector(v, v, 2, 3));
                    }
                }
            } else {
                const char* base = reinterpret_cast<const char*>(x) + (((long)img0 * g.Cin) * g.Hi + r0s) * (long)g.pitchX * 4;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, OOB, 0x00020000);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (VEC == 4) L[it][j] = __builtin_bit_cast(vecf, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * planeXb, 0));
                    else L[it][j] = __builtin_bit_cast(vecf, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, j * planeXb, 0));
                }
            }
        } else {
            const unsigned voff = img_ok ? it_off[it] : OOB;
            const char* base = reinterpret_cast<const char*>(gy) + (((long)img0 * g.Cout) * g.Ho + r0) * (long)g.pitchY * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, OOB, 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (VEC == 4) L[it][j] = __builtin_bit_cast(vecf, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * planeYb, 0));
                else L[it][j] = __builtin_bit_cast(vecf, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, j * planeYb, 0));
            }
        }
        }
    };
    auto commit = [&](unsigned char* buf, int it) {
        if (XPL && it_kind[it] == 1) {
#pragma unroll
            for (int p = 0; p < VEC; ++p) {
                if constexpr (!UNI) {
                    *(u32x4*)(buf + it_lofs[it] + p * ROWB) = LP[it][p];
                } else if constexpr (VEC == 4) {
                    *(vecf*)(buf + it_lofs[it] + p * ROWB) = L[it][p];
                } else {
                    *(vecf*)(buf + it_lofs[it] + p * ROWB) = L[it][2 * p];
                    *(vecf*)(buf + it_lofs[it] + p * ROWB + 8) = L[it][2 * p + 1];
                }
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < VEC; ++p) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = L[it][j][p];
            unsigned char* dst = buf + it_lofs[it] + p * ROWB;
            if (it_kind[it] == 1) stage_terms<XT>(dst, b_term, v);
            else stage_terms<3>(dst, A_TERM, v);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    int tile = bx;
    if (tile < g.ntiles) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (it * NT < nitems) fetch(it, tile);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (it * NT < nitems) commit(smem, it);
    }
    __syncthreads();

    const bool single = g.single != 0;
    int par = 0;
    for (; tile < g.ntiles; tile += g.kslices, par ^= (single ? 0 : 1)) {
        const unsigned char* cur = smem + par * buf_bytes;
        unsigned char* nxt = smem + (single ? 0 : (par ^ 1)) * buf_bytes;
        const bool more = tile + g.kslices < g.ntiles;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            if (ks == 0 && more) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (it * NT < nitems) fetch(it, tile + g.kslices);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const unsigned char* pa = cur + a_lane + t * A_TERM + ks * 16 * ROWB;
                a[t] = tr_frag(pa, pa + 4 * ROWB);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int toff = (kh * g.RS + kw) * ROWB;
                bf16x8 b[XT];
#pragma unroll
                for (int t = 0; t < XT; ++t) b[t] = tr_frag(cur + b_lane[ks][0] + toff + t * b_term, cur + b_lane[ks][1] + toff + t * b_term);
                if constexpr (XT == 1) {
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[kw], 0, 0, 0);
                } else {
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[kw], 0, 0, 0);
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[kw], 0, 0, 0);
                }
            }
            if (ks == COMMIT_KS && more && !single) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (it * NT < nitems) commit(nxt, it);
            }
        }
        if (single && more) {
            __syncthreads();              // every wave has read its fragments of this tile: the buffer may be overwritten
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it * NT < nitems) commit(nxt, it);
        }
        __syncthreads();
    }

    // ---- partial sums of this block -> slab [kslice][co][ci][kh][kw]; D: col = ci = lane&31, row = co = (e&3)+8*(e>>2)+4*(lane>>5)
    const int ci = ci0 + pn * 32 + (lane & 31);
    float* slab = slabs + (size_t)bx * g.Cout * g.Cin * 9;
    if (ci < g.Cin) {
        const int cob = co0 + pm * 32 + 4 * (lane >> 5);
        float* d0 = slab + ((size_t)cob * g.Cin + ci) * 9 + kh * 3;
        const size_t rs = (size_t)g.Cin * 9;
        const bool full = co0 + pm * 32 + 32 <= g.Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (full || cob + (e & 3) + 8 * (e >> 2) < g.Cout) {
                float* d = d0 + ((e & 3) + 8 * (e >> 2)) * rs;
                d[0] = acc[0][e];
                d[1] = acc[1][e];
                d[2] = acc[2][e];
            }
        }
    }
}

template <int S, int XT, int PM, int PN, int VEC, bool XPL = false, int NITX = 0>
__global__ __launch_bounds__(192 * PM * PN) void conv_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                        float* __restrict__ slabs, WgGeom g) {
    extern __shared__ __align__(16) unsigned char smem[];
    conv_wgrad_body<S, XT, PM, PN, VEC, XPL, NITX>(x, gy, slabs, g, smem, (int)blockIdx.x, (int)blockIdx.y);
}

// Grouped launch (include/eas_hip.h eas_conv_wgrad_group_partial): layer p owns blocks [first[p], first[p + 1]) of the flat grid,
// kslices x (co, ci) blocks of them, slice fastest.  Stride 1, fp32 three-term x, whole-row tiles; a layer's rows are staged in 16-byte
// units or (10-pixel rows) 8-byte units.
constexpr int kMaxWgGroup = 8;
struct WgGroupArgs {
    WgGeom g[kMaxWgGroup];
    const float* x[kMaxWgGroup];
    const float* gy[kMaxWgGroup];
    float* slabs[kMaxWgGroup];
    int first[kMaxWgGroup + 1];
    int vec2[kMaxWgGroup];
    int n;
};

template <int XT, int PM, int PN>
__global__ __launch_bounds__(192 * PM * PN) void conv_wgrad_group_kernel(const WgGroupArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    int p = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const int lb = (int)blockIdx.x - a.first[p];
    const int ks = a.g[p].kslices, by = lb / ks, bx = lb - by * ks;
    if (a.vec2[p]) conv_wgrad_body<1, XT, PM, PN, 2, false>(a.x[p], a.gy[p], a.slabs[p], a.g[p], smem, bx, by);
    else conv_wgrad_body<1, XT, PM, PN, 4, false>(a.x[p], a.gy[p], a.slabs[p], a.g[p], smem, bx, by);
}

// dW[i] = sum over slabs in a fixed order (deterministic): slice lane kl = 0..3 adds slabs kl, kl + 4, kl + 8, ... in double, in that order,
// then the four partial sums are added in lane order.  A block = 4 slice lanes x 64 threads, a thread = 4 consecutive outputs (one 16-byte
// load per slab, eight slabs in flight: the first form read one float per thread and slab with four in flight and ran at 2 TB/s on the
// ~1 GB of slabs a step reduces); n not a multiple of 4 or unaligned slabs: one output per thread.  Both forms add the same values in
// the same order per output.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slabs, float* __restrict__ gw, int n, int kslices, int block) {
    __shared__ double part[4][64][4];
    const int o = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const bool v4 = (n & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)gw) & 15) == 0;
    const int per = v4 ? 4 : 1;
    const int i = (block * 64 + o) * per;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < n) {
        int k = kl;
        if (v4) {
            for (; k + 28 < kslices; k += 32) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(slabs + (size_t)(k + 4 * u) * n + i);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w;
                }
            }
            for (; k < kslices; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)k * n + i);
                s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            }
        } else {
            for (; k + 12 < kslices; k += 16) {
                const float a = slabs[(size_t)k * n + i], b = slabs[(size_t)(k + 4) * n + i], c = slabs[(size_t)(k + 8) * n + i],
                            d = slabs[(size_t)(k + 12) * n + i];
                s[0] += (double)a;
                s[0] += (double)b;
                s[0] += (double)c;
                s[0] += (double)d;
            }
            for (; k < kslices; k += 4) s[0] += (double)slabs[(size_t)k * n + i];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[kl][o][j] = s[j];
    __syncthreads();
    if (kl == 0 && i < n) {
        if (v4) {
            float4 r;
            r.x = (float)(((part[0][o][0] + part[1][o][0]) + part[2][o][0]) + part[3][o][0]);
            r.y = (float)(((part[0][o][1] + part[1][o][1]) + part[2][o][1]) + part[3][o][1]);
            r.z = (float)(((part[0][o][2] + part[1][o][2]) + part[2][o][2]) + part[3][o][2]);
            r.w = (float)(((part[0][o][3] + part[1][o][3]) + part[2][o][3]) + part[3][o][3]);
            *reinterpret_cast<float4*>(gw + i) = r;
        } else {
            gw[i] = (float)(((part[0][o][0] + part[1][o][0]) + part[2][o][0]) + part[3][o][0]);
        }
    }
}

// outputs a block covers (host side: blocks of a job)
static inline int wgrad_reduce_blocks(const float* slabs, const float* gw, int n) {
    const bool v4 = (n & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)gw) & 15) == 0;
    return (n + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ gw, int n, int kslices) {
    wgrad_reduce_block(slabs, gw, n, kslices, (int)blockIdx.x);
}

// The same reduction for MANY weight gradients in one launch (eas_conv_wgrad_reduce_many): a training step has ~80 weight
// gradients whose slab reductions are each a few microseconds of work behind a launch; their results are not needed before the
// optimizer step, so the host collects them and reduces them together.  Jobs travel BY VALUE in the kernel arguments (no device
// table to fill, graph-capturable); block -> job by the prefix of 64-output blocks.
constexpr int kReduceJobsPerLaunch = 96;
struct ReduceJobs {
    const float* slabs[kReduceJobsPerLaunch];
    float* gw[kReduceJobsPerLaunch];
    int n[kReduceJobsPerLaunch];
    int kslices[kReduceJobsPerLaunch];
    int first_block[kReduceJobsPerLaunch + 1];
    int njobs;
};

__global__ __launch_bounds__(256) void conv_wgrad_reduce_many_kernel(const ReduceJobs jobs) {
    int lo = 0, hi = jobs.njobs;                 // job j owns blocks [first_block[j], first_block[j+1])
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (jobs.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    // identical summation order to conv_wgrad_reduce_kernel: bit-identical results
    wgrad_reduce_block(jobs.slabs[lo], jobs.gw[lo], jobs.n[lo], jobs.kslices[lo], (int)blockIdx.x - jobs.first_block[lo]);
}

int pick_rows(int Ho, int Wo, int cap) {
    int best = 0;
    for (int rt = 1; rt * Wo <= cap; ++rt)
        if (Ho % rt == 0 || rt % Ho == 0) best = rt;
    return best;
}

template <int S, int XT, int PM, int PN, int VEC, bool XPL = false, int NITX = 0>
int launch_wgrad(const float* x, const float* gy, float* slabs, WgGeom g, hipStream_t st) {
    auto kern = conv_wgrad_mfma_kernel<S, XT, PM, PN, VEC, XPL, NITX>;
    constexpr int NT = 192 * PM * PN;
    const size_t lds = (size_t)(g.single ? 1 : 2) * (3 * PM * A_PLANE + (size_t)XT * PN * g.Q * ROWB);
    const int nbg = PN == 1 && g.Cin < 32 ? (g.Cin + 7) / 8 : 4 * PN;
    const int nitems = (TP / VEC) * 4 * PM + g.nseg * g.rows_in * (g.Wst / VEC) * nbg;
    if (lds > 160 * 1024 || nitems > (NITX ? NITX : (PM * PN == 4 ? 1 : 2)) * NT) {
        if (eas_dev_env("EAS_CONV_DBG")) fprintf(stderr, "wgrad launch: lds %zu nitems %d NT %d PM %d PN %d VEC %d RT %d Q %d\n", lds, nitems, NT, PM, PN, VEC, g.RT, g.Q);
        return EAS_ERR_UNSUPPORTED;
    }
    if constexpr (XT == 1) {
        // the spike-input instances address a tile's items as 32-bit byte offsets from the tile base in raw buffer loads whose descriptors say
        // num_records = 2^31 (offsets from there on mean "outside: zeros"): the images of a tile plus the 8 channel rows of an item must stay
        // below that (far above every EAS-SNN shape: a 64-channel 128x160 image is 5 MB)
        const size_t img_y = (size_t)g.Cout * g.Ho * g.pitchY * 4, img_x = (size_t)g.Cin * g.Hi * g.pitchX * (XPL ? 2 : 4);
        const size_t span = (size_t)(g.nseg + 1) * (img_y > img_x ? img_y : img_x) + (size_t)8 * g.Hi * g.pitchX * 4;
        if (span >= ((size_t)1 << 31)) return EAS_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    g.ci_blocks = (g.Cin + 32 * PN - 1) / (32 * PN);
    const int co_blocks = (g.Cout + 32 * PM - 1) / (32 * PM);
    dim3 grid(g.kslices, co_blocks * g.ci_blocks);
    EAS_LAUNCH(kern, grid, dim3(NT), lds, st, x, gy, slabs, g);
    return EAS_OK;
}

// blocks of this kernel instance a CU holds at once with `lds` bytes of dynamic LDS (registers, waves and LDS all count), cached
template <int S, int XT, int PM, int PN, int VEC, bool XPL = false, int NITX = 0>
int resident_wgrad(size_t lds) {
    static size_t seen[16];
    static int val[16], n = 0;
    for (int i = 0; i < n; ++i)
        if (seen[i] == lds) return val[i];
    auto kern = conv_wgrad_mfma_kernel<S, XT, PM, PN, VEC, XPL, NITX>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 192 * PM * PN, lds) != hipSuccess || nb < 1) nb = 1;
    if (n < 16) { seen[n] = lds; val[n] = nb; ++n; }
    return nb;
}

struct WgPlan { int pm, pn, kslices; };

// block shape and number of pixel slices for a layer (shared by the workspace query and the launch)
// pn = 2 (a block owns 64 input channels: grad_y is staged by half as many blocks) for spike inputs with >= 64 channels whenever the two
// x planes fit LDS next to the grad_y tile (always at stride 1; at stride 2 on the small maps, where the layers with many channels
// are bound by re-staging the same pixels from L2 in every (co, ci) block)
WgPlan wg_plan(int Cin, int Cout, int stride, int x_terms, int ntiles, int parts, int pn, int resident = 0) {
    WgPlan p;
    p.pm = Cout >= 64 ? 2 : 1;
    p.pn = pn;
    const int yz = ((Cout + 32 * p.pm - 1) / (32 * p.pm)) * ((Cin + 32 * p.pn - 1) / (32 * p.pn));
    // Pixel slices: every block walks ntiles / ks tiles and the launch takes ceil(ks * yz / slots) rounds of the chip's block slots
    // (slots = 256 CUs x the blocks of THIS kernel instance a CU really holds: registers, waves and LDS -- 108 KB of double-buffered LDS
    // leave one, where the old rule of thumb "4 / (pm pn)" assumed two and sized the grid for 512 slots: 2.25 rounds).  The slice count
    // with the fewest tile periods wins; ties go to fewer slabs.  resident = 0: geometry search only (pm / pn are all it reads).
    static const int slot_pct = eas_dev_env("EAS_WG_SLOT_PCT") ? atoi(eas_dev_env("EAS_WG_SLOT_PCT")) : 100;      // development: share of the chip a launch is sized for
    const int slots = 256 * (resident > 0 ? resident : 1) * slot_pct / 100;
    const int step = parts > 1 ? parts : 1;
    int best = step;
    static const double pen = eas_dev_env("EAS_WG_SHARE_PEN") ? atof(eas_dev_env("EAS_WG_SHARE_PEN")) : 0.0;      // development
    const int res1 = resident > 0 ? resident : 1;
    double best_cost = -1.0;
    for (int ks = step; ks <= ntiles && (long)ks * yz <= 4L * slots; ks += step) {
        const long blocks = (long)ks * yz;
        const long rounds = (blocks + slots - 1) / slots;
        // blocks that share a CU share its matrix cores: a tile period stretches by `pen` per co-resident block
        long share = (blocks + 255) / 256;
        if (share > res1) share = res1;
        const double cost = (double)rounds * (double)((ntiles + ks - 1) / ks) * (1.0 + pen * (double)(share - 1));
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    p.kslices = best;
    return p;
}

// geometry of a layer cut into `parts` column parts per row (1 = whole rows), tile <= cap output pixels
bool wg_geom_cap(WgGeom& g, int NI, int Cin, int Cout, int Hi, int Wi, int stride, int cap, int parts) {
    g.NI = NI; g.Cin = Cin; g.Cout = Cout; g.Hi = Hi; g.Wi = Wi;
    g.Ho = (Hi + 2 - 3) / stride + 1;
    const int WoF = (Wi + 2 - 3) / stride + 1;
    g.parts = parts;
    g.pitchX = Wi; g.pitchY = WoF;
    if (parts == 1) {
        g.Wo = WoF; g.Wst = Wi; g.lpad = 1; g.qshift = 0; g.hv = 0;
        g.RS = Wi + 2;
    } else {
        if (WoF % parts != 0 || Wi % 4 != 0 || ((WoF / parts) * stride) % 4 != 0 || (WoF / parts) % 4 != 0 || WoF * stride != Wi) return false;
        g.Wo = WoF / parts; g.hv = 4; g.Wst = g.Wo * stride + 2 * g.hv; g.lpad = 0; g.qshift = g.hv - 1;
        g.RS = g.Wst;
    }
    g.RT = pick_rows(g.Ho, g.Wo, cap);
    if (g.RT == 0) return false;
    g.rows_seg = g.RT < g.Ho ? g.RT : g.Ho;
    g.nseg = g.RT / g.rows_seg;
    g.rows_in = (g.rows_seg - 1) * stride + 3;
    g.Q = g.nseg * g.rows_in * g.RS;
    g.total_rows = NI * g.Ho;
    g.ntiles = ((g.total_rows + g.RT - 1) / g.RT) * parts;
    return true;
}

// largest tile (<= 80 output pixels) whose double-buffered images and staging items fit the block; whole rows first, then 2, 4
// and 8 column parts per row
bool wg_geom(WgGeom& g, int NI, int Cin, int Cout, int Hi, int Wi, int stride, int x_terms) {
    static const int force_parts = eas_dev_env("EAS_WG_PARTS") ? atoi(eas_dev_env("EAS_WG_PARTS")) : 0;      // development switch
    static const int pn2_s2 = eas_dev_env("EAS_WG_PN2_S2") ? atoi(eas_dev_env("EAS_WG_PN2_S2")) : 1;       // development switch
    for (int parts = force_parts > 0 ? force_parts : 1; parts <= 8; parts *= 2) {
        for (int cap = TP; cap >= 8; cap /= 2)
          for (int pn = (Cin >= 64 && x_terms == 1 && (stride == 1 || (pn2_s2 && cap == TP))) ? 2 : 1; pn >= 1; --pn) {
            if (!wg_geom_cap(g, NI, Cin, Cout, Hi, Wi, stride, cap, parts)) continue;
            g.pn = pn;
            const WgPlan p = wg_plan(Cin, Cout, stride, x_terms, g.ntiles, parts, pn);
            const int vec = parts > 1 ? 4 : ((Wi % 4 == 0 && g.Wo % 4 == 0) ? 4 : 2);
            const size_t lds = (size_t)2 * (3 * p.pm * A_PLANE + (size_t)x_terms * p.pn * g.Q * ROWB);
            const int nbg = p.pn == 1 && Cin < 32 ? (Cin + 7) / 8 : 4 * p.pn;
            const int nitems = (TP / vec) * 4 * p.pm + g.nseg * g.rows_in * (g.Wst / vec) * nbg;
            const int nt = 192 * p.pm * p.pn;
            g.nit = p.pm * p.pn == 4 ? 1 : 2;
            if (lds <= 160 * 1024 && nitems <= g.nit * nt) return true;
            // stride 2, spike inputs, 64-output-channel blocks: a third item per thread keeps the 80-pixel tile (dark5.0 at 16x20 -> 8x10: 1000
            // items for 384 threads; with two the tile shrank to 40 pixels -- half of every MFMA step on zeros, twice the barriers: 296 -> 199 us)
            if (stride == 2 && x_terms == 1 && p.pm == 2 && p.pn == 1 && cap == TP && lds <= 160 * 1024 && nitems <= 3 * nt) {
                g.nit = 3;
                return true;
            }
        }
        if (force_parts > 0) break;
    }
    return false;
}

// the plan of a layer whose geometry is fixed: block shape, then the residency of the kernel instance the launch will use, then slices
WgPlan wg_plan_final(const WgGeom& g, int Cin, int Cout, int stride, int x_terms, bool planes, int& single) {
    const WgPlan p0 = wg_plan(Cin, Cout, stride, x_terms, g.ntiles, g.parts, g.pn);
    const size_t lds = (size_t)2 * (3 * p0.pm * A_PLANE + (size_t)x_terms * p0.pn * g.Q * ROWB);
    const bool v4 = g.parts > 1 || (g.Wi % 4 == 0 && g.Wo % 4 == 0);
    int res = 1;
    if (g.nit == 3) {        // (wg_geom: stride 2, one term, pm 2, pn 1; never single-buffered)
        res = planes ? (v4 ? resident_wgrad<2, 1, 2, 1, 4, true, 3>(lds) : resident_wgrad<2, 1, 2, 1, 2, true, 3>(lds))
                     : (v4 ? resident_wgrad<2, 1, 2, 1, 4, false, 3>(lds) : resident_wgrad<2, 1, 2, 1, 2, false, 3>(lds));
        single = 0;
        return wg_plan(Cin, Cout, stride, x_terms, g.ntiles, g.parts, g.pn, res);
    }
#define EAS_RS(S_, XT_, PM_, PN_, PL_) (v4 ? resident_wgrad<S_, XT_, PM_, PN_, 4, PL_>(lds) : resident_wgrad<S_, XT_, PM_, PN_, 2, PL_>(lds))
#define EAS_RS_SHAPE(S_, XT_, PL_)                                              \
    do {                                                                        \
        if (p0.pm == 2 && p0.pn == 2) res = EAS_RS(S_, XT_, 2, 2, PL_);         \
        else if (p0.pm == 2) res = EAS_RS(S_, XT_, 2, 1, PL_);                  \
        else if (p0.pn == 2) res = EAS_RS(S_, XT_, 1, 2, PL_);                  \
        else res = EAS_RS(S_, XT_, 1, 1, PL_);                                  \
    } while (0)
    if (planes && stride == 1) EAS_RS_SHAPE(1, 1, true);
    else if (planes) EAS_RS_SHAPE(2, 1, true);
    else if (stride == 1 && x_terms == 1) EAS_RS_SHAPE(1, 1, false);
    else if (stride == 1) EAS_RS_SHAPE(1, 3, false);
    else if (x_terms == 1) EAS_RS_SHAPE(2, 1, false);
    else EAS_RS_SHAPE(2, 3, false);
#undef EAS_RS_SHAPE
#undef EAS_RS
    // single-buffered LDS where the double-buffered tile leaves ONE block per CU and half of it two: the second block overlaps everything
    // the first one waits for, which a second buffer inside one 6- or 12-wave block cannot (a tile's write is exposed instead: +15 %)
    static const int single_env = eas_dev_env("EAS_WG_SINGLE") ? atoi(eas_dev_env("EAS_WG_SINGLE")) : -1;      // development: 0 never, 1 whenever it fits twice
    single = 0;
    static const int single_cap = eas_dev_env("EAS_WG_SINGLE_CAP") ? atoi(eas_dev_env("EAS_WG_SINGLE_CAP")) : 2;       // development
    // (measured: 3-wave blocks -- layers of fewer than 64 output channels on the big maps -- gain 20-30 %: real-input 48 -> 48 channels at
    // 64x80 699 -> 496 us; 6- and 12-wave blocks lose 5-8 %: they already keep the SIMDs' issue slots busy and the exposed write costs more)
    if (single_env != 0 && res == 1 && (p0.pm * p0.pn == 1 || single_env == 1)) {
        const size_t lds1 = lds / 2;
        int res1 = 1;
#define EAS_RS(S_, XT_, PM_, PN_, PL_) (v4 ? resident_wgrad<S_, XT_, PM_, PN_, 4, PL_>(lds1) : resident_wgrad<S_, XT_, PM_, PN_, 2, PL_>(lds1))
#define EAS_RS_SHAPE(S_, XT_, PL_)                                              \
    do {                                                                        \
        if (p0.pm == 2 && p0.pn == 2) res1 = EAS_RS(S_, XT_, 2, 2, PL_);        \
        else if (p0.pm == 2) res1 = EAS_RS(S_, XT_, 2, 1, PL_);                 \
        else if (p0.pn == 2) res1 = EAS_RS(S_, XT_, 1, 2, PL_);                 \
        else res1 = EAS_RS(S_, XT_, 1, 1, PL_);                                 \
    } while (0)
        if (planes && stride == 1) EAS_RS_SHAPE(1, 1, true);
        else if (planes) EAS_RS_SHAPE(2, 1, true);
        else if (stride == 1 && x_terms == 1) EAS_RS_SHAPE(1, 1, false);
        else if (stride == 1) EAS_RS_SHAPE(1, 3, false);
        else if (x_terms == 1) EAS_RS_SHAPE(2, 1, false);
        else EAS_RS_SHAPE(2, 3, false);
#undef EAS_RS_SHAPE
#undef EAS_RS
        if (res1 >= 2) {
            single = 1;
            res = res1 > single_cap ? single_cap : res1;
        }
    }
    return wg_plan(Cin, Cout, stride, x_terms, g.ntiles, g.parts, g.pn, res);
}

// ---- grouped launch: plan.  Every layer keeps its own (co, ci) blocks; the pixel slices of ALL layers are sized together: a common
// number of tiles per block tau, slices_p = ceil(ntiles_p / tau), chosen like wg_plan chooses a layer's -- the fewest tile periods
// (rounds of the chip's block slots x tau), ties to the larger tau (fewer slabs).  A layer alone needs its slices to fill 256 CUs by
// itself (512 slabs for a 128 -> 128 layer on 32x40 maps); in a group the other layers' blocks fill them.
template <int XT, int PM, int PN>
int resident_wgrad_group(size_t lds) {
    static size_t seen[8];
    static int val[8], n = 0;
    for (int i = 0; i < n; ++i)
        if (seen[i] == lds) return val[i];
    auto kern = conv_wgrad_group_kernel<XT, PM, PN>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 192 * PM * PN, lds) != hipSuccess || nb < 1) nb = 1;
    if (n < 8) { seen[n] = lds; val[n] = nb; ++n; }
    return nb;
}

// tau for block counts yz[p] and tile counts nt[p]
int group_tau(const int* yz, const int* nt, int n, long slots) {
    int max_nt = 0;
    for (int p = 0; p < n; ++p) max_nt = nt[p] > max_nt ? nt[p] : max_nt;
    int best = max_nt;
    long best_cost = -1;
    for (int tau = max_nt; tau >= 1; --tau) {
        long blocks = 0;
        for (int p = 0; p < n; ++p) blocks += (long)((nt[p] + tau - 1) / tau) * yz[p];
        if (blocks > 4 * slots && best_cost >= 0) break;
        const long cost = ((blocks + slots - 1) / slots) * tau;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = tau; }
    }
    return best;
}

int wgrad_group3(const EasWgradProblem* pr, int n, int x_terms, hipStream_t st, int* slabs_out, bool query) {
    if (n > kMaxWgGroup || x_terms != 3) return EAS_ERR_UNSUPPORTED;
    WgGroupArgs a{};
    int yz[kMaxWgGroup], nt[kMaxWgGroup];
    size_t lds = 0;
    const int pm = pr[0].Cout >= 64 ? 2 : 1;
    for (int p = 0; p < n; ++p) {
        const EasWgradProblem& q = pr[p];
        if (q.NI <= 0 || q.Cin <= 0 || q.Cout <= 0 || q.Hi <= 0 || q.Wi <= 0) return EAS_ERR_INVALID_ARG;
        if (!query && (!q.x || !q.grad_y || !q.workspace)) return EAS_ERR_INVALID_ARG;
        if (q.Cin % 8 != 0 || q.Cout % 8 != 0 || q.Wi % 2 != 0 || (q.Cout >= 64 ? 2 : 1) != pm) return EAS_ERR_UNSUPPORTED;
        WgGeom& g = a.g[p];
        if (!wg_geom(g, q.NI, q.Cin, q.Cout, q.Hi, q.Wi, 1, x_terms) || g.parts != 1 || g.pn != 1) return EAS_ERR_UNSUPPORTED;
        if (g.Wo % 2 != 0 || (g.Ho * g.pitchY) % 4 != 0) return EAS_ERR_UNSUPPORTED;
        a.vec2[p] = (g.Wi % 4 == 0 && g.Wo % 4 == 0) ? 0 : 1;
        const size_t l = (size_t)2 * (3 * pm * A_PLANE + (size_t)x_terms * g.Q * ROWB);
        lds = l > lds ? l : lds;
        g.ci_blocks = (g.Cin + 31) / 32;
        yz[p] = ((g.Cout + 32 * pm - 1) / (32 * pm)) * g.ci_blocks;
        nt[p] = g.ntiles;
        a.x[p] = (const float*)q.x; a.gy[p] = q.grad_y; a.slabs[p] = q.workspace;
    }
    if (lds > 160 * 1024) return EAS_ERR_UNSUPPORTED;
    const int res = pm == 2 ? resident_wgrad_group<3, 2, 1>(lds) : resident_wgrad_group<3, 1, 1>(lds);
    static const int slot_pct_g = eas_dev_env("EAS_WG_SLOT_PCT") ? atoi(eas_dev_env("EAS_WG_SLOT_PCT")) : 100;      // development (wg_plan)
    const int tau = group_tau(yz, nt, n, 256L * res * slot_pct_g / 100);
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        a.g[p].kslices = (nt[p] + tau - 1) / tau;
        a.g[p].single = 0;
        if (slabs_out) slabs_out[p] = a.g[p].kslices;
        a.first[p] = blocks;
        blocks += a.g[p].kslices * yz[p];
    }
    a.first[n] = blocks;
    a.n = n;
    if (query) return EAS_OK;
    if (pm == 2) EAS_LAUNCH((conv_wgrad_group_kernel<3, 2, 1>), dim3(blocks), dim3(384), lds, st, a);
    else EAS_LAUNCH((conv_wgrad_group_kernel<3, 1, 1>), dim3(blocks), dim3(192), lds, st, a);
    return EAS_OK;
}

}  // namespace

int eas_conv1x1_wgrad_group(const EasWgradProblem* pr, int n, int x_terms, hipStream_t st, int* slabs_out, bool query);
int eas_conv1x1_wgrad_slices(int NI, int Cin, int Cout, int HW, int x_terms, int planes);
int eas_conv1x1_wgrad_dispatch(const void* x, const float* gy, float* slabs, int NI, int Cin, int Cout, int HW, int x_terms, hipStream_t st,
                               int planes);

extern "C" {

int64_t eas_conv_wgrad_workspace_floats(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms) {
    const bool planes = x_terms == 2;
    if (planes) x_terms = 1;                // spike planes: the geometry of one-term inputs
    if (ksize == 1) return stride == 1 ? (int64_t)eas_conv1x1_wgrad_slices(NI, Cin, Cout, Hi * Wi, x_terms, planes ? 1 : 0) * Cout * Cin : 0;
    if (ksize != 3) return 0;
    WgGeom g{};
    if (!wg_geom(g, NI, Cin, Cout, Hi, Wi, stride, x_terms)) return 0;
    int single = 0;
    return (int64_t)wg_plan_final(g, Cin, Cout, stride, x_terms, planes, single).kslices * Cout * Cin * 9;
}

// number of column parts per row eas_conv_wgrad uses for a 3x3 layer: 1 = whole rows fit one reduction tile, 2..8 = column parts
// (same kernel, one launch), 0 = unsupported
int eas_conv_wgrad_parts(int NI, int Cin, int Cout, int Hi, int Wi, int stride, int x_terms) {
    if (x_terms == 2) x_terms = 1;
    WgGeom g{};
    return wg_geom(g, NI, Cin, Cout, Hi, Wi, stride, x_terms) ? g.parts : 0;
}

// grad_w[Cout][Cin][3][3] of a 3x3 convolution (padding 1, stride 1 or 2) from x[NI][Cin][Hi][Wi] and grad_y[NI][Cout][Ho][Wo].
// x_terms as in eas_conv_fwd.  workspace: eas_conv_wgrad_workspace_floats(...) floats.
// slab kernel only; returns the number of slabs written (> 0) or a negative status
static int wgrad_partial(const float* x, const float* grad_y, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                         int stride, int x_terms, eas_stream_t stream) {
    if (!x || !grad_y || !workspace || NI <= 0 || Cin <= 0 || Cout <= 0 || Hi <= 0 || Wi <= 0) return EAS_ERR_INVALID_ARG;
    const bool planes = x_terms == 2;       // x is a spike-plane tensor (bf16 [NI][Cin/8][Hi*Wi][8], one exact term)
    if (planes) x_terms = 1;
    if ((x_terms != 1 && x_terms != 3) || (planes && (Cin % 8 != 0 || ((uintptr_t)x & 15)))) return EAS_ERR_UNSUPPORTED;
    if (ksize == 1) {
        if (stride != 1) return EAS_ERR_UNSUPPORTED;
        EAS_CLEAR_ERR();
        const int slices = eas_conv1x1_wgrad_slices(NI, Cin, Cout, Hi * Wi, x_terms, planes ? 1 : 0);
        if (slices <= 0) return EAS_ERR_UNSUPPORTED;
        const int rc1 = eas_conv1x1_wgrad_dispatch(x, grad_y, workspace, NI, Cin, Cout, Hi * Wi, x_terms, eas_s(stream), planes ? 1 : 0);
        if (rc1 != EAS_OK) return rc1;
        EAS_CHECK_LAUNCH();
        return slices;
    }
    if (ksize != 3 || (stride != 1 && stride != 2) || Cin % 8 != 0 || Cout % 8 != 0 || Wi % 2 != 0) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    WgGeom g{};
    if (!wg_geom(g, NI, Cin, Cout, Hi, Wi, stride, x_terms)) return EAS_ERR_UNSUPPORTED;
    if (g.Wo % 2 != 0 || (g.Ho * g.pitchY) % 4 != 0) return EAS_ERR_UNSUPPORTED;
    int single = 0;
    const WgPlan p = wg_plan_final(g, Cin, Cout, stride, x_terms, planes, single);
    g.kslices = p.kslices;
    g.single = single;
    const int slabs = p.kslices;
    hipStream_t st = eas_s(stream);
    const int n = Cout * Cin * 9;
    int rc = EAS_ERR_UNSUPPORTED;
    const bool v4 = g.parts > 1 || (g.Wi % 4 == 0 && g.Wo % 4 == 0);
#define EAS_WG(S_, XT_, PM_, PN_, PL_) (v4 ? launch_wgrad<S_, XT_, PM_, PN_, 4, PL_>(x, grad_y, workspace, g, st) : launch_wgrad<S_, XT_, PM_, PN_, 2, PL_>(x, grad_y, workspace, g, st))
#define EAS_WG_SHAPE(S_, XT_, PL_)                                             \
    do {                                                                       \
        if (p.pm == 2 && p.pn == 2) rc = EAS_WG(S_, XT_, 2, 2, PL_);           \
        else if (p.pm == 2) rc = EAS_WG(S_, XT_, 2, 1, PL_);                   \
        else if (p.pn == 2) rc = EAS_WG(S_, XT_, 1, 2, PL_);                   \
        else rc = EAS_WG(S_, XT_, 1, 1, PL_);                                  \
    } while (0)
    if (g.nit == 3)
        rc = planes ? (v4 ? launch_wgrad<2, 1, 2, 1, 4, true, 3>(x, grad_y, workspace, g, st) : launch_wgrad<2, 1, 2, 1, 2, true, 3>(x, grad_y, workspace, g, st))
                    : (v4 ? launch_wgrad<2, 1, 2, 1, 4, false, 3>(x, grad_y, workspace, g, st) : launch_wgrad<2, 1, 2, 1, 2, false, 3>(x, grad_y, workspace, g, st));
    else if (planes && stride == 1) EAS_WG_SHAPE(1, 1, true);
    else if (planes) EAS_WG_SHAPE(2, 1, true);
    else if (stride == 1 && x_terms == 1) EAS_WG_SHAPE(1, 1, false);
    else if (stride == 1) EAS_WG_SHAPE(1, 3, false);
    else if (x_terms == 1) EAS_WG_SHAPE(2, 1, false);
    else EAS_WG_SHAPE(2, 3, false);
#undef EAS_WG_SHAPE
#undef EAS_WG
    if (rc != EAS_OK) return rc;
    EAS_CHECK_LAUNCH();
    (void)n;
    return slabs;
}

int eas_conv_wgrad(const float* x, const float* grad_y, float* grad_w, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                   int stride, int x_terms, eas_stream_t stream) {
    if (!grad_w) return EAS_ERR_INVALID_ARG;
    const int slabs = wgrad_partial(x, grad_y, workspace, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, stream);
    if (slabs < 0) return slabs;
    const int n = Cout * Cin * ksize * ksize;
    EAS_LAUNCH(conv_wgrad_reduce_kernel, dim3(wgrad_reduce_blocks(workspace, grad_w, n)), dim3(256), 0, eas_s(stream), workspace, grad_w, n, slabs);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_conv_wgrad_partial(const float* x, const float* grad_y, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                           int stride, int x_terms, eas_stream_t stream) {
    if (x_terms == 2) return EAS_ERR_INVALID_ARG;      // spike planes go through eas_conv_wgrad_planes_partial
    return wgrad_partial(x, grad_y, workspace, NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms, stream);
}

// eas_conv_wgrad_partial with x given as SPIKE PLANES (bf16 [NI][Cin/8][Hi*Wi][8]): the same products in the same order as x_terms = 1 on
// the fp32 values, bit-identical slabs; workspace: eas_conv_wgrad_workspace_floats(..., x_terms = 2)
int eas_conv_wgrad_planes_partial(const void* x_planes, const float* grad_y, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                                  int stride, eas_stream_t stream) {
    return wgrad_partial(reinterpret_cast<const float*>(x_planes), grad_y, workspace, NI, Cin, Cout, Hi, Wi, ksize, stride, 2, stream);
}

// include/eas_hip.h: several layers' slab kernels as one grid
static int wgrad_group_impl(const EasWgradProblem* pr, int n, int ksize, int x_terms, eas_stream_t stream, int* slabs_out, bool query) {
    if (!pr || n < 1) return EAS_ERR_INVALID_ARG;
    if (ksize != 1 && ksize != 3) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    const int rc = ksize == 1 ? eas_conv1x1_wgrad_group(pr, n, x_terms, eas_s(stream), slabs_out, query)
                              : wgrad_group3(pr, n, x_terms, eas_s(stream), slabs_out, query);
    if (rc != EAS_OK || query) return rc;
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_conv_wgrad_group_plan(const EasWgradProblem* problems, int n, int ksize, int x_terms, int* slabs_out) {
    return wgrad_group_impl(problems, n, ksize, x_terms, nullptr, slabs_out, true);
}

int eas_conv_wgrad_group_partial(const EasWgradProblem* problems, int n, int ksize, int x_terms, eas_stream_t stream) {
    return wgrad_group_impl(problems, n, ksize, x_terms, stream, nullptr, false);
}

int eas_conv_wgrad_reduce_many(const EasWgradReduceJob* jobs, int njobs, eas_stream_t stream) {
    if (!jobs || njobs < 1) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    for (int j0 = 0; j0 < njobs; j0 += kReduceJobsPerLaunch) {
        ReduceJobs a;
        a.njobs = njobs - j0 < kReduceJobsPerLaunch ? njobs - j0 : kReduceJobsPerLaunch;
        int blocks = 0;
        for (int j = 0; j < a.njobs; ++j) {
            const EasWgradReduceJob& q = jobs[j0 + j];
            if (!q.slabs || !q.grad_w || q.n < 1 || q.slabs_count < 1) return EAS_ERR_INVALID_ARG;
            a.slabs[j] = q.slabs; a.gw[j] = q.grad_w; a.n[j] = q.n; a.kslices[j] = q.slabs_count;
            a.first_block[j] = blocks;
            blocks += wgrad_reduce_blocks(q.slabs, q.grad_w, q.n);
        }
        a.first_block[a.njobs] = blocks;
        EAS_LAUNCH(conv_wgrad_reduce_many_kernel, dim3(blocks), dim3(256), 0, eas_s(stream), a);
        EAS_CHECK_LAUNCH();
    }
    return EAS_OK;
}

}  // extern "C"
