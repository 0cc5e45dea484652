// Weight gradient of the adaptive sampler's tiny-channel convolutions (Conv2d(2->4 / 4->4 / 2->2, k, pad k/2), stride 1;
// yolox/models/embedding.py:106-111) on the CDNA4 matrix cores.
//
//   dW[co][ci][kh][kw] = sum_{n,ho,wo} gy[n][co][ho][wo] * x[n][ci][ho+kh-P][wo+kw-P],  db[co] = sum gy[n][co]
//
// Two to four channels cannot fill an MFMA tile by themselves, but (channel, kernel row) x (channel, kernel column) can: for one
// row h of x and 16 consecutive columns w (the reduction index),
//
//   A[m = kh*COUT + co][w] = gy[co][h - kh + P][w]      (<= 28 rows of 32)
//   B[w][n = ci*K + kw] = x[ci][h][w + kw - P]          (<= 28 columns of 32; column CIN*K holds ones -> row sums = db)
//   D[m][n] += A * B                                    = dW[co][ci][kh][kw]
//
// Both operands are staged in LDS as exact bf16 terms (hi + mid + lo = the fp32 value; the six products that matter are
// accumulated in fp32, like the other MFMA convolutions of this library).  A fragments are aligned 16-byte reads.  B fragments
// start at column w + kw - P: x is staged twice, as is ("even") and shifted by one column ("odd"), so that every fragment
// starts on a 4-byte boundary of one of the two copies and is read as four dwords.
//
// Persistent blocks over (image, 8-row band, 64-column chunk) tiles, one block per CU: 8 staging waves convert tile t+1 to bf16
// terms (global loads issued two tiles ahead) into one LDS buffer while 8 multiplying waves, one x row each, run the MFMAs of tile t
// from the other; one barrier per tile.  The 32x32 accumulator tiles stay in registers over all tiles; per block one fixed-order
// reduction over the waves, block partials in the layout of smallconv.hip's VALU kernel (its finalize kernel sums them in fixed
// order: deterministic, no float atomics).  Measured (MI355X, 256 images of 256x320, 4->4 k5, inside the training step): 232 us
// against 495 us for the VALU kernel; LDS images free of bank conflicts (SQ_LDS_BANK_CONFLICT ~ 0), LDS array 40 % busy; the
// barrier per tile between the two kinds of waves is what remains.
#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RB = 8;              // x rows per tile
constexpr int WC = 64;             // reduction columns per tile (4 k-steps of 16)
constexpr int HALO = 8;            // staged x columns left / right of the chunk (>= PAD + 1; 8 keeps chunks 16-byte aligned)
constexpr int XC = WC + 2 * HALO;  // staged x columns
constexpr int NTH = 256;
constexpr int NWV = NTH / 64;

template <int CIN, int COUT, int K>
struct Sw {
    static constexpr int PAD = K / 2;
    static constexpr int GR = RB + K - 1;                  // staged grad_y rows
    static constexpr int GP = WC + 8;                      // grad_y row pitch (elements): 144 bytes
    static constexpr int G_ROWS = GR * COUT;
    static constexpr int G_TERM = G_ROWS * GP * 2;         // bytes per term
    // x rows: one per (row, channel), pitch 40 dwords; the odd copy starts 4 banks after the even one.  ds_read_b32 / ds_read2_b32
    // bank 32 lanes over 32 banks: channel c's <= 5 distinct dwords sit on banks 3..8 + 8c (+4 for the second k-half, which is the
    // other lane group), the bias column's constant row on bank 2: conflict-free for k <= 5 (k = 7 with 4 channels: 2-way on half)
    static constexpr int XP = XC;                          // x row pitch (elements): 160 bytes
    static constexpr int X_ROWS = RB * CIN;                // + one constant row: ones (bias column)
    static constexpr int X_RAW = (X_ROWS + 1) * XP * 2;
    static constexpr int X_COPY = X_RAW + ((4 + 32 - (X_RAW / 4) % 32) % 32) * 4;   // bytes per (term, copy)
    static constexpr int X_ONES = X_ROWS * XP * 2 + 8;     // constant row inside the even copy, from bank 2
    static constexpr int X_TERM = 2 * X_COPY;
    static constexpr int G_BYTES = 3 * G_TERM;
    static constexpr int X_BYTES = 3 * X_TERM;
    static constexpr int STAGE_BYTES = G_BYTES + X_BYTES;
    static constexpr int RED_BYTES = 2 * NWV * 32 * 33 * 4;
    static constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
    static constexpr int NWT = COUT * CIN * K * K;
    static constexpr int NOUT = NWT + COUT;
    static constexpr int G_ITEMS = GR * COUT * (WC / 8);
    static constexpr int X_ITEMS = RB * CIN * (XC / 8);
    static_assert(COUT * K <= 32 && CIN * K + 1 <= 32, "does not fit one 32x32 tile");
    static_assert(PAD + 1 <= HALO && XC <= XP && (X_COPY / 4) % 32 == 4 && X_COPY % 16 == 0, "x staging geometry");
    static_assert(2 * STAGE_BYTES <= 160 * 1024 && STAGE_BYTES % 16 == 0, "LDS");
    static_assert(RB == 2 * NWV, "two rows per wave (4 waves) / one row per wave (8 waves)");
};

__device__ __forceinline__ void sw_split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// 8 consecutive values v[o..o+7] as three bf16 term vectors at dst, dst + term_stride, dst + 2 term_stride
template <int O, int NV>
__device__ __forceinline__ void sw_store_terms(unsigned char* dst, int term_stride, const float (&v)[NV]) {
    bf16x8 t0, t1, t2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        sw_split3(v[O + j], a, b, c);
        t0[j] = a;
        t1[j] = b;
        t2[j] = c;
    }
    *(bf16x8*)dst = t0;
    *(bf16x8*)(dst + term_stride) = t1;
    *(bf16x8*)(dst + 2 * term_stride) = t2;
}

// NV values of one image row starting at column c0 (may be negative / past the row): zeros outside [0, W) and for a missing row
// (generic path: any W, any alignment)
template <int NV>
__device__ __forceinline__ void sw_load_row(const float* __restrict__ row, bool row_ok, int c0, int W, float (&v)[NV]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = c0 + j;
        v[j] = (row_ok && c >= 0 && c < W) ? row[c] : 0.f;
    }
}

// VEC (W % 4 == 0, aligned tensors): 16 waves, two LDS buffers -- waves 0-7 convert and stage tile t+1 (vector ALU, LDS stores) while
// waves 8-15 multiply tile t, one row of the band each (matrix cores, LDS reads); each SIMD hosts two waves of either kind, one
// barrier per tile.
// !VEC (any W / alignment): 4 waves do both in turn on one buffer.
template <int CIN, int COUT, int K, bool VEC>
__global__ __launch_bounds__(VEC ? 4 * NTH : NTH) void smallconv_wgrad_mfma_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                                   float* __restrict__ partial, int N, int H, int W, int x_tm) {
    using S = Sw<CIN, COUT, K>;
    const int xS = x_tm ? N / x_tm : 0;                                          // x given as collated micro-slices: image n of the
    auto x_img = [&](int n) {                                                     // convolution = micro-slice x_tm - 1 - n / xS of sample n % xS
        if (x_tm == 0) return n;                                                  // (smallconv_core.h sc_src_image)
        const int t = n / xS;
        return (n - t * xS) * x_tm + (x_tm - 1 - t);
    };
    constexpr int PAD = S::PAD;
    constexpr int NBUF = VEC ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // NBUF x STAGE_BYTES
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NTS = VEC ? 2 * NTH : NTH;                                    // staging threads
    constexpr int NCW = VEC ? 2 * NWV : NWV;                                    // multiplying waves
    constexpr int RPW = RB / NCW;                                               // rows of the band per multiplying wave
    const int wave = VEC ? (tid >> 6) - NTS / 64 : (tid >> 6);                  // index among the multiplying waves (< 0: staging wave)
    const int r = lane & 31, hh = lane >> 5;

    // this lane's A row m = kh*COUT + co and B column n = ci*K + kw.  Rows / columns of the 32x32 tile past those ranges are never read
    // back, so their lanes repeat the address of a used lane of the same LDS lane group (a broadcast, no extra bank cycle); the bias
    // column reads a constant row of ones.  The kernel-row-major order of m puts the <= 16 rows one ds_read_b128 lane group touches
    // on distinct 16-byte bank groups (row pitch 9 x 16 bytes): no branches, no selects and no bank conflicts in the compute phase.
    const bool g0 = (r & 3) == r || (r >= 12 && r < 16) || (r >= 20 && r < 28);   // ds_read_b128 lane groups {0-3,12-15,20-27} / the rest
    const int ra = r < COUT * K ? r : (g0 ? 0 : 4);
    const int a_kh = ra / COUT, a_co = ra % COUT;
    const int a_off = (((2 * PAD - a_kh) * COUT + a_co) * S::GP + 8 * hh) * 2;
    const int a_rs = COUT * S::GP * 2, a_kss = 32;                               // + row*a_rs + ks*a_kss + term*G_TERM
    const bool b_ones = r == CIN * K;                                            // bias column
    const int rb = r < CIN * K ? r : 0;
    const int b_ci = rb / K, b_kw = rb % K;
    const int sft = b_kw - PAD;                                                   // column shift of this lane's fragment
    const int odd = sft & 1;
    const int b_off = b_ones ? S::X_ONES : odd * S::X_COPY + (b_ci * S::XP + HALO + 8 * hh + (sft - odd)) * 2;
    const int b_rs = b_ones ? 0 : CIN * S::XP * 2, b_kss = b_ones ? 0 : 32;       // + row*b_rs + ks*b_kss + term*X_TERM

    // constant row of ones (first term; zeros in the other two), never overwritten by the staging
    for (int i = tid; i < NBUF * 3 * 4; i += blockDim.x) {
        const int buf = i / 12, tm = (i % 12) / 4, d = i & 3;
        *(unsigned int*)(smem + buf * S::STAGE_BYTES + S::G_BYTES + tm * S::X_TERM + S::X_ONES + 4 * d) = tm == 0 ? 0x3f803f80u : 0u;
    }

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc0[e] = 0.0f;
        acc1[e] = 0.0f;
    }

    const int bands = (H + RB - 1) / RB, chunks = (W + WC - 1) / WC;
    const int ntiles = N * bands * chunks;
    const size_t plane = (size_t)H * W;

    auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
        n = t / (bands * chunks);
        const int rest = t - n * bands * chunks;
        const int band = rest / chunks;
        h0 = band * RB;
        w0 = (rest - band * chunks) * WC;
    };

    // ---- this wave's RPW rows of the staged band: RPW * 4 (k-step, row) steps, the fragments of step s+1 read while step s
    //      multiplies; two accumulators used in turn (independent MFMA chains)
    struct Frag { bf16x8 a[3], b[3]; };
    const unsigned char* pa0 = smem + a_off + wave * a_rs;
    const unsigned char* pb0 = smem + S::G_BYTES + b_off + wave * b_rs;
    auto read_frag = [&](Frag& f, int step) {
        const int ks = step / RPW, i = step % RPW;
        const unsigned char* pa = pa0 + i * (NCW * a_rs) + ks * a_kss;
        const unsigned char* pb = pb0 + i * (NCW * b_rs) + ks * b_kss;
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
            f.a[tm] = *(const bf16x8*)(pa + tm * S::G_TERM);
            const unsigned int* q = (const unsigned int*)(pb + tm * S::X_TERM);
            u32x4 d;
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
            f.b[tm] = __builtin_bit_cast(bf16x8, d);
        }
    };
    auto mul6 = [&](f32x16& acc, const Frag& f) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2], f.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[0], acc, 0, 0, 0);
    };
    auto compute = [&](int buf) {
        constexpr int STEPS = RPW * (WC / 16);
        pa0 = smem + buf * S::STAGE_BYTES + a_off + wave * a_rs;
        pb0 = smem + buf * S::STAGE_BYTES + S::G_BYTES + b_off + wave * b_rs;
        Frag f0, f1;
        read_frag(f0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; st += 2) {
            read_frag(f1, st + 1);
            mul6(acc0, f0);
            if (st + 2 < STEPS) read_frag(f0, st + 2);
            mul6(acc1, f1);
        }
    };

    // accumulators of a multiplying wave -> LDS (after the last barrier of the tile loop: every buffer has been consumed)
    auto write_red = [&]() {
        float* red = (float*)smem;                       // [wave][m][33]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 4 * hh + (e & 3) + 8 * (e >> 2);
            red[(wave * 32 + m) * 33 + r] = acc0[e] + acc1[e];
        }
    };

    if constexpr (VEC) {
        // W % 4 == 0 and 16-byte aligned tensors: every float4 of a staged row lies entirely inside or outside the image.  Each thread
        // of the staging waves owns NG + NX fixed items (8 columns of one staged row); its global loads are issued unconditionally (from
        // a clamped address, masked when they are converted) two tiles before the tile is written to LDS.
        constexpr int NG = (S::G_ITEMS + NTS - 1) / NTS, NX = (S::X_ITEMS + NTS - 1) / NTS;
        int g_dr[NG], g_dc[NG], g_ch[NG], g_lds[NG];
        int x_dr[NX], x_dc[NX], x_ch[NX], x_lds[NX];
        bool g_valid[NG], x_valid[NX];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int id = j * NTS + tid;
            g_valid[j] = tid < NTS && id < S::G_ITEMS;
            const int idc = g_valid[j] ? id : 0;
            const int c8 = idc % (WC / 8), rc = idc / (WC / 8);
            const int co = rc % COUT, jrow = rc / COUT;
            g_dr[j] = jrow - PAD; g_dc[j] = 8 * c8; g_ch[j] = co;
            g_lds[j] = ((jrow * COUT + co) * S::GP + 8 * c8) * 2;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int id = j * NTS + (((j + NG) & 1) ? NTS - 1 - tid : tid);   // alternate rounds from the top: spreads the ragged rounds over the waves
            x_valid[j] = tid < NTS && id >= 0 && id < S::X_ITEMS;
            const int idc = x_valid[j] ? id : 0;
            const int c8 = idc % (XC / 8), rc = idc / (XC / 8);
            const int ci = rc % CIN, row_l = rc / CIN;
            x_dr[j] = row_l; x_dc[j] = 8 * c8 - HALO; x_ch[j] = ci;
            x_lds[j] = S::G_BYTES + ((row_l * CIN + ci) * S::XP + 8 * c8) * 2;
        }
        // two register sets: the loads of tile t+2 are issued right after tile t has been written to LDS, so one tile of loads is
        // always in flight -- also while the vector ALU converts (the memory pipe idled during the conversion with one set)
        struct Pre {
            float4 ga[NG], gb[NG], xa[NX], xb[NX];
            float xc[NX];
            int gm[NG], xm[NX];
        };
        auto issue = [&](Pre& p, int t) {
            int n, h0, w0;
            tile_origin(t, n, h0, w0);
            const int nx = x_img(n);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int row = h0 + g_dr[j], col = w0 + g_dc[j];
                const bool row_ok = g_valid[j] && (unsigned)row < (unsigned)H;
                const bool ok0 = row_ok && col + 4 <= W, ok1 = row_ok && col + 8 <= W;
                const float* prow = gy + ((size_t)n * COUT + g_ch[j]) * plane + (size_t)(row_ok ? row : 0) * W + col;
                p.ga[j] = *(const float4*)(ok0 ? prow : gy);
                p.gb[j] = *(const float4*)(ok1 ? prow + 4 : gy);
                p.gm[j] = (ok0 ? 1 : 0) | (ok1 ? 2 : 0);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int row = h0 + x_dr[j], col = w0 + x_dc[j];
                const bool row_ok = x_valid[j] && row < H;
                const bool ok0 = row_ok && col >= 0 && col + 4 <= W, ok1 = row_ok && col + 4 >= 0 && col + 8 <= W;
                const bool ok2 = row_ok && col + 8 < W;
                const float* prow = x + ((size_t)nx * CIN + x_ch[j]) * plane + (size_t)(row_ok ? row : 0) * W + col;
                p.xa[j] = *(const float4*)(ok0 ? prow : x);
                p.xb[j] = *(const float4*)(ok1 ? prow + 4 : x);
                p.xc[j] = *(ok2 ? prow + 8 : x);
                p.xm[j] = (ok0 ? 1 : 0) | (ok1 ? 2 : 0) | (ok2 ? 4 : 0);
            }
        };
        auto commit = [&](const Pre& p, int buf) {
            unsigned char* const base = smem + buf * S::STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if (!g_valid[j]) continue;
                const bool ok0 = p.gm[j] & 1, ok1 = p.gm[j] & 2;
                float v[8];
                v[0] = ok0 ? p.ga[j].x : 0.f; v[1] = ok0 ? p.ga[j].y : 0.f; v[2] = ok0 ? p.ga[j].z : 0.f; v[3] = ok0 ? p.ga[j].w : 0.f;
                v[4] = ok1 ? p.gb[j].x : 0.f; v[5] = ok1 ? p.gb[j].y : 0.f; v[6] = ok1 ? p.gb[j].z : 0.f; v[7] = ok1 ? p.gb[j].w : 0.f;
                sw_store_terms<0, 8>(base + g_lds[j], S::G_TERM, v);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if (!x_valid[j]) continue;
                const bool ok0 = p.xm[j] & 1, ok1 = p.xm[j] & 2, ok2 = p.xm[j] & 4;
                float v[9];
                v[0] = ok0 ? p.xa[j].x : 0.f; v[1] = ok0 ? p.xa[j].y : 0.f; v[2] = ok0 ? p.xa[j].z : 0.f; v[3] = ok0 ? p.xa[j].w : 0.f;
                v[4] = ok1 ? p.xb[j].x : 0.f; v[5] = ok1 ? p.xb[j].y : 0.f; v[6] = ok1 ? p.xb[j].z : 0.f; v[7] = ok1 ? p.xb[j].w : 0.f;
                v[8] = ok2 ? p.xc[j] : 0.f;
                unsigned char* dst = base + x_lds[j];
                sw_store_terms<0, 9>(dst, S::X_TERM, v);
                sw_store_terms<1, 9>(dst + S::X_COPY, S::X_TERM, v);
            }
        };
        // The two kinds of waves run separate loops (wave-uniform branch on a scalar wave index: the register allocator sees the
        // staging registers and the accumulators / fragments on different paths) with the same number of barriers.
        const int G = gridDim.x;
        const bool stager = __builtin_amdgcn_readfirstlane(tid >> 6) < NTS / 64;
        int t = blockIdx.x;
        if (stager) {
            Pre pa, pb;                                // tiles t, t+2G, .. in pa; t+G, t+3G, .. in pb
            issue(pa, t);
            if (t + G < ntiles) issue(pb, t + G);
            commit(pa, 0);
            if (t + 2 * G < ntiles) issue(pa, t + 2 * G);
            __syncthreads();
            for (;;) {
                if (t + G < ntiles) {
                    commit(pb, 1);
                    if (t + 3 * G < ntiles) issue(pb, t + 3 * G);
                }
                __syncthreads();
                t += G;
                if (t >= ntiles) break;
                if (t + G < ntiles) {
                    commit(pa, 0);
                    if (t + 3 * G < ntiles) issue(pa, t + 3 * G);
                }
                __syncthreads();
                t += G;
                if (t >= ntiles) break;
            }
        } else {
            __syncthreads();
            for (;;) {
                compute(0);
                __syncthreads();
                t += G;
                if (t >= ntiles) break;
                compute(1);
                __syncthreads();
                t += G;
                if (t >= ntiles) break;
            }
            write_red();
        }
    } else {
        __syncthreads();
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            int n, h0, w0;
            tile_origin(t, n, h0, w0);
            // ---- stage grad_y rows h0-PAD .. h0+RB-1+PAD, columns w0 .. w0+WC-1
            for (int id = tid; id < S::G_ITEMS; id += NTH) {
                const int c8 = id % (WC / 8), rc = id / (WC / 8);
                const int co = rc % COUT, jrow = rc / COUT;
                const int ho = h0 - PAD + jrow;
                const bool row_ok = ho >= 0 && ho < H;
                const float* row = gy + ((size_t)n * COUT + co) * plane + (size_t)(row_ok ? ho : 0) * W;
                float v[8];
                sw_load_row<8>(row, row_ok, w0 + 8 * c8, W, v);
                sw_store_terms<0, 8>(smem + ((jrow * COUT + co) * S::GP + 8 * c8) * 2, S::G_TERM, v);
            }
            // ---- stage x rows h0 .. h0+RB-1, columns w0-HALO .. w0+WC+HALO-1, as is and shifted by one column
            for (int id = tid; id < S::X_ITEMS; id += NTH) {
                const int c8 = id % (XC / 8), rc = id / (XC / 8);
                const int ci = rc % CIN, row_l = rc / CIN;
                const int h = h0 + row_l;
                const bool row_ok = h < H;
                const float* row = x + ((size_t)x_img(n) * CIN + ci) * plane + (size_t)(row_ok ? h : 0) * W;
                float v[9];
                sw_load_row<9>(row, row_ok, w0 - HALO + 8 * c8, W, v);
                unsigned char* dst = smem + S::G_BYTES + ((row_l * CIN + ci) * S::XP + 8 * c8) * 2;
                sw_store_terms<0, 9>(dst, S::X_TERM, v);
                sw_store_terms<1, 9>(dst + S::X_COPY, S::X_TERM, v);
            }
            __syncthreads();
            compute(0);
            __syncthreads();
        }
        write_red();
    }

    // ---- block partial: waves summed in fixed order
    const float* red = (const float*)smem;
    __syncthreads();
    for (int e = tid; e < S::NOUT; e += blockDim.x) {
        int m, nn;
        if (e < S::NWT) {                            // e indexes [co][ci][kh][kw]
            const int co = e / (CIN * K * K), rem = e % (CIN * K * K);
            const int ci = rem / (K * K), kk = rem % (K * K);
            m = (kk / K) * COUT + co;
            nn = ci * K + kk % K;
        } else {
            m = PAD * COUT + (e - S::NWT);
            nn = CIN * K;
        }
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < NCW; ++wv) s += red[(wv * 32 + m) * 33 + nn];
        partial[(int64_t)blockIdx.x * S::NOUT + e] = s;
    }
}

template <int CIN, int COUT, int K>
int sw_launch_k(bool vec, const float* gy, const float* x, float* partial, int nblocks, int N, int H, int W, int x_tm, hipStream_t st) {
    using S = Sw<CIN, COUT, K>;
    if (vec) {
        auto kern = smallconv_wgrad_mfma_kernel<CIN, COUT, K, true>;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
            attr_set = true;
        }
        EAS_LAUNCH(kern, dim3(nblocks), dim3(4 * NTH), 2 * S::STAGE_BYTES, st, gy, x, partial, N, H, W, x_tm);
    } else {
        EAS_LAUNCH((smallconv_wgrad_mfma_kernel<CIN, COUT, K, false>), dim3(nblocks), dim3(NTH), S::LDS_BYTES, st, gy, x, partial, N, H, W, x_tm);
    }
    return EAS_OK;
}

template <int CIN, int COUT>
int sw_launch(int k, const float* gy, const float* x, float* partial, int nblocks, int N, int H, int W, int x_tm, hipStream_t st) {
    const bool vec = (W & 3) == 0 && ((((uintptr_t)gy) | ((uintptr_t)x)) & 15) == 0;   // float4 staging loads with register prefetch
    int rc;
    if (vec && nblocks > 256) nblocks = 256;             // 8-wave blocks with two LDS buffers: one per CU
    switch (k) {
        case 3: rc = sw_launch_k<CIN, COUT, 3>(vec, gy, x, partial, nblocks, N, H, W, x_tm, st); break;
        case 5: rc = sw_launch_k<CIN, COUT, 5>(vec, gy, x, partial, nblocks, N, H, W, x_tm, st); break;
        case 7: rc = sw_launch_k<CIN, COUT, 7>(vec, gy, x, partial, nblocks, N, H, W, x_tm, st); break;
        default: return EAS_ERR_UNSUPPORTED;
    }
    if (rc != EAS_OK) return rc;
    EAS_CHECK_LAUNCH();
    return nblocks;
}

}  // namespace

// Internal to the library (smallconv.hip's eas_smallconv_bwd_weight): block partials [nblocks][Cout*Cin*k*k + Cout] of the MFMA
// form; returns the number of blocks launched (<= max_blocks) or a negative status.
int eas_sw_mfma_partials(const float* gy, const float* x, float* partial, int max_blocks, int N, int Cin, int Cout, int H, int W,
                         int k, int x_tm, hipStream_t st) {
    const int64_t tiles = (int64_t)N * ((H + RB - 1) / RB) * ((W + WC - 1) / WC);
    if (tiles > 0x7fffffff) return EAS_ERR_UNSUPPORTED;
    const int nblocks = tiles < max_blocks ? (int)tiles : max_blocks;
    int rc;
    if (Cin == 2 && Cout == 4) rc = sw_launch<2, 4>(k, gy, x, partial, nblocks, N, H, W, x_tm, st);
    else if (Cin == 4 && Cout == 4) rc = sw_launch<4, 4>(k, gy, x, partial, nblocks, N, H, W, x_tm, st);
    else if (Cin == 2 && Cout == 2) rc = sw_launch<2, 2>(k, gy, x, partial, nblocks, N, H, W, x_tm, st);
    else return EAS_ERR_UNSUPPORTED;
    return rc;
}
