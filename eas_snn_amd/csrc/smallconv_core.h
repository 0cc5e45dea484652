// Shared pieces of the sampler's vector-ALU convolution kernels (smallconv.hip, arsnn.hip): tile geometry, register-prefetch
// staging of the input window into LDS, and the accumulation core.
#pragma once
#include <type_traits>

#include "eas_common.h"

namespace {

constexpr int TH = 16, TW = 64, LWS = 80, NT = 256;

// Image n of a convolution whose input is the COLLATED micro-slice tensor [S][Tm][C][H][W] (trainer.py:99; embedding.py:147-156 flips the
// time axis and makes it the leading one) read time-major, newest slice first, without that flipped copy: image n = t * S + s is micro-slice
// Tm - 1 - t of sample s.  Tm = 0: the input is an ordinary [N][C][H][W] tensor.
__device__ __forceinline__ int sc_src_image(int n, int S, int Tm) {
    if (Tm == 0) return n;
    const int t = n / S;
    return (n - t * S) * Tm + (Tm - 1 - t);
}

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
// global loads back to back and does NOT touch the loaded values (out-of-range lanes read element 0; which elements were in range
// goes into a bit mask), so nothing waits for them until `store`, after the compute phase of the previous tile, writes them to LDS
// (zeros where the mask says so; channel planes PLANE floats apart, rows PITCH floats apart).
// (The first form zeroed the out-of-range values with a select right behind the loads: the compiler then placed the wait for ALL of a
// tile's loads in front of the FMA phase -- the prefetch never overlapped the arithmetic and the kernels ran at 30 % of the FMA rate.)
template <int CIN, int K, int PLANE, int PITCH = LWS>
struct Stager {
    using G = Geo<K>;
    static constexpr int NE = CIN * G::LH * G::LW;
    static constexpr int PER = (NE + NT - 1) / NT;
    static_assert(PER <= 32, "in-range mask is one 32-bit word");
    // which window element a thread stages does not depend on the tile: (channel, row, column) packed once per thread
    int rc[PER];         // r | col << 8 | c << 16, or -1 for the surplus elements of the last slice

    __device__ __forceinline__ void init(int H, int W) {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int col = idx % G::LW;
            const int r = (idx / G::LW) % G::LH;
            const int c = idx / (G::LW * G::LH);
            rc[it] = idx < NE ? (r | (col << 8) | (c << 16)) : -1;
        }
    }

    // returns the mask of the elements that lie inside the image (bit `it`)
    __device__ __forceinline__ unsigned load(const float* __restrict__ x, float (&pre)[PER], int n, int H, int W, int y0, int x0) const {
        const int oy = y0 - G::PAD, ox = x0 - G::PAD;
        const float* win = x + ((int64_t)n * CIN * H + oy) * W + ox;       // window origin (may lie outside the image)
        unsigned okm = 0;
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int r = rc[it] & 255, col = (rc[it] >> 8) & 255, c = rc[it] >> 16;
            const unsigned gy = (unsigned)(oy + r), gx = (unsigned)(ox + col);
            const bool ok = rc[it] >= 0 && gy < (unsigned)H && gx < (unsigned)W;
            pre[it] = *(ok ? win + ((c * H + r) * W + col) : x);
            okm |= ok ? (1u << it) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);        // the loads are issued HERE, in front of the arithmetic that follows the call
        return okm;
    }

    // unconditional stores (no control flow for the scheduler to rotate the loop around): the surplus elements of the last slice
    // write their zero into a pad column of row 0 (columns >= LW of a row are never read; PITCH > LW)
    __device__ __forceinline__ void store(float* __restrict__ lds, const float (&pre)[PER], unsigned okm) const {
        static_assert(PITCH > G::LW, "pad column for the surplus elements");
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int r = rc[it] & 255, col = (rc[it] >> 8) & 255, c = rc[it] >> 16;
            const int ofs = rc[it] >= 0 ? c * PLANE + r * PITCH + col : G::LW;
            lds[ofs] = ((okm >> it) & 1u) ? pre[it] : 0.f;
        }
    }
};

// The same for image widths that are multiples of 4 (every model canvas): the window is staged as aligned 16-byte quads.  The LDS frame
// starts 4 columns left of the tile (x0 - 4, a multiple of 4) whatever the kernel size, so a quad lies entirely inside or outside the
// image: one float4 load + one ds_write_b128 per quad (6 per thread and tile for four channels at k = 5 instead of 22 dword loads and
// stores with their index arithmetic).  Row pitch 80 floats (18 quads + 2 pad quads that take the surplus threads' zeros).
template <int CIN, int K>
struct StagerV {
    using G = Geo<K>;
    static constexpr int QW = (TW + 8) / 4;              // quads per staged row: columns x0-4 .. x0+TW+3
    static constexpr int PITCH = 80, PLANE = G::LH * PITCH;
    static constexpr int NE = CIN * G::LH * QW;
    static constexpr int PER = (NE + NT - 1) / NT;
    static constexpr int OFS = 4 - G::PAD;               // index of the first needed value in a thread's 12 loaded ones
    static_assert(G::PAD <= 4 && OFS + 3 + K - 1 < 12, "three quads per row cover the taps");
    int rc[PER];         // r | q << 8 | c << 16, or -1

    __device__ __forceinline__ void init(int, int) {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int q = idx % QW;
            const int r = (idx / QW) % G::LH;
            const int c = idx / (QW * G::LH);
            rc[it] = idx < NE ? (r | (q << 8) | (c << 16)) : -1;
        }
    }

    __device__ __forceinline__ unsigned load(const float* __restrict__ x, float4 (&pre)[PER], int n, int H, int W, int y0, int x0) const {
        const int oy = y0 - G::PAD, ox = x0 - 4;
        const float* win = x + ((int64_t)n * CIN * H + oy) * W + ox;
        unsigned okm = 0;
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int r = rc[it] & 255, q = (rc[it] >> 8) & 255, c = rc[it] >> 16;
            const unsigned gy = (unsigned)(oy + r), gx = (unsigned)(ox + 4 * q);
            const bool ok = rc[it] >= 0 && gy < (unsigned)H && gx < (unsigned)W;
            pre[it] = *reinterpret_cast<const float4*>(ok ? win + ((c * H + r) * W + 4 * q) : x);
            okm |= ok ? (1u << it) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);        // the loads are issued HERE, in front of the arithmetic that follows the call
        return okm;
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, const float4 (&pre)[PER], unsigned okm) const {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int r = rc[it] & 255, q = (rc[it] >> 8) & 255, c = rc[it] >> 16;
            const int ofs = rc[it] >= 0 ? c * PLANE + r * PITCH + 4 * q : 4 * QW;       // surplus: pad quad of row 0
            *reinterpret_cast<float4*>(lds + ofs) = ((okm >> it) & 1u) ? pre[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};

// ---- convolution core -------------------------------------------------------------------------------------------
// The weights are read through the scalar cache in the order wr[ci][ky][kx][o] (output channel fastest; eas_smallconv_pack_weights
// arranges them, also flipped + channel-transposed for the input gradient): the two output channels of an (o, o+1) pair sit in an
// aligned scalar-register pair, so one v_pk_fma_f32 multiplies ONE input value (a lane's pixel, broadcast to both halves by op_sel) with
// the pair -- two FMAs per lane and instruction, no register moves to line operands up.  (The earlier form packed two neighbouring PIXELS
// against a broadcast weight: every odd kernel column needed shifted copies of the input registers, 25 moves per 200 packed FMAs, and
// kept 100 weights of an input channel in scalar registers at once, which spilled.)  Per output the order of the additions is
// (bias), ci, ky, kx -- the order of the kernels this replaces: results are bit-identical.
// NQ quads are read per window row at column 4 * tx; the thread's first needed value is element OFS of them.
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CIN, int COUT, int K, int PITCH, int NQ, int OFS>
__device__ __forceinline__ void sc_accumulate(const float* __restrict__ lds, int plane, const float* __restrict__ wr, f2 (&acc)[4][COUT / 2],
                                              int tx, int ty) {
#pragma unroll 1
    for (int i = 0; i < CIN; ++i) {                // not unrolled: one input channel's weights in scalar registers at a time
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            float v[NQ * 4];
            const float4* row = reinterpret_cast<const float4*>(lds + i * plane + (ty + ky) * PITCH + 4 * tx);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 t = row[q];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
                for (int op = 0; op < COUT / 2; ++op) {
                    const f2 w2 = *reinterpret_cast<const f2*>(wr + ((i * K + ky) * K + kx) * COUT + 2 * op);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][op] = __builtin_elementwise_fma(w2, f2{v[OFS + j + kx], v[OFS + j + kx]}, acc[j][op]);
                }
            }
        }
    }
}

// staging form of a kernel instantiation: VECW (image width a multiple of 4) = aligned quads, else single elements
template <int CIN, int K, bool VECW>
struct Stage {
    using G = Geo<K>;
    static constexpr int PITCH = VECW ? StagerV<CIN, K>::PITCH : 128;      // element form: a multiple of the 64-bank row
    static constexpr int PLANE = G::LH * PITCH;
    static constexpr int NQ = VECW ? 3 : G::NV4;
    static constexpr int OFS = VECW ? StagerV<CIN, K>::OFS : 0;
    using St = typename std::conditional<VECW, StagerV<CIN, K>, Stager<CIN, K, PLANE, PITCH>>::type;
    using Pre = typename std::conditional<VECW, float4, float>::type;
};

}  // namespace
