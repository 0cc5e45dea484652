// The sampler's 5x5 tiny-channel convolutions (Conv2d(2->4,5,p=2) + ReLU + Conv2d(4->4,5,p=2), yolox/models/embedding.py:106-111)
// forward and input gradient on the CDNA4 matrix cores.
//
// The vector-ALU kernel of smallconv.hip is compute-bound: 100 FMAs per output value at ~30 % of the packed-FMA peak.  With 2..4
// channels a convolution does not fill an MFMA tile the usual way (M = output channels = 4 of 32), so the GEMM is cut differently,
// per OUTPUT ROW y of a 32-column strip:
//
//     D[(kx, co)][x'] = sum over (ky, ci) of  Wt[(kx, co)][(ky, ci)] * in[ci][y + ky - 2][x']          M = 5*4 = 20 of 32, K = 5*4 = 20 of 32
//     out[co][y][x]   = sum over kx of        D[(kx, co)][x + kx - 2]                                   (shift-add over the kernel columns)
//
// i.e. the kernel columns move from the reduction index into the M index and come back as five shifted adds of the MFMA result
// (through a wave-private LDS patch).  One v_mfma_f32_32x32x16_bf16 pair (K = 32) per bf16 term pair yields 28 x 4 outputs: 12 MFMAs
// with exact three-term operands (fp32 = hi + mid + lo, products with ta + tb <= 2), 6 when the input holds spikes / small integers
// (one term) -- 3.4 / 1.7 MFMA cycles per output value against ~10 vector-ALU cycles.  A, the weights, never changes: each lane
// builds its 2 x 3 fragments once.  B comes from the input tile staged in LDS as [term][row][column][4 channels] bf16 (8 bytes per
// pixel): a lane's 8 reduction values of a k-step are two rows' 4 channels = two ds_read_b64 at consecutive columns across the
// lanes (conflict-free).  Rows ky = 5..7 of the second k-step meet zero weights (their row index is clamped into the tile).
// The input gradient is the same kernel with the filter flipped and channel-transposed (+ the ReLU mask of the tensor it flows into).
//
// MEASURED (MI355X, 64 images of 256x320, 4 -> 4 channels, three-term input): 98 us against 90 us for the vector-ALU kernel, so this
// form is an opt-in (EAS_SC_FORM=mfma), kept with its parity test.  Ablations (EAS_SC_DBG): tile loads + staging alone 34 us, the MFMAs
// add 42 us (their own bound: 2.4 M MFMAs x 32 cycles over 1024 SIMDs = 31-35 us), the shift-add epilogue + stores 25 us -- and the
// three do not overlap: the four waves of a block share every barrier, so only the two co-resident blocks of a CU (LDS: 79 KB each)
// can hide each other's phases (one block per CU: 140 us).  The utilisation of the MFMA tile (20/32 x 20/32, 28/32 columns) makes the
// matrix-core time alone a third of the vector kernel's total; a version that wins needs the phases in different waves (loader /
// MFMA / epilogue specialisation) rather than a faster phase.
#include <stdlib.h>

#include "eas_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TR = 16;            // output rows per block tile
constexpr int SWV = 28;           // output columns per wave strip (32 staged columns minus the 4 halo columns)
constexpr int NWV = 4;            // waves (column strips) per block
constexpr int TC = SWV * NWV;     // output columns per block tile
constexpr int SC = TC + 4;        // staged columns
constexpr int SR = TR + 4;        // staged rows
constexpr int PP = 36;            // floats per (kx, co) row of the shift-add patch (32 columns + reach of kx)

__device__ __attribute__((aligned(16))) float eas_sc_zero[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    lo = (__bf16)r2;
}

__device__ __forceinline__ void wave_lds_sync() {
    // a wave's LDS operations execute in order: only the compiler has to keep writes before the reads of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XT: bf16 terms of the input (1: spikes / small integers, exact in bf16; 3: general fp32).  CI / CO: input / output channels of THIS
// kernel (for the input gradient: the layer's Cout / Cin).  DGRAD: w is the layer's [Cout][Cin][5][5] filter read flipped and
// channel-transposed; mask (nullable): output zeroed where mask <= 0.
template <int XT, int CI, int CO, bool DGRAD>
__global__ __launch_bounds__(256, 2) void smallconv5_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, const float* __restrict__ mask,
                                                                 float* __restrict__ y, int N, int H, int W, int relu, int tiles_x,
                                                                 int tiles_y, int ntiles, int dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;

    // ---- A fragments: lane = row m = (kx, co) of Wt, 8 reduction indices k = (ky, ci) per k-step and half wave
    bf16x8 a[2][3];
    {
        const int kx = n >> 2, co = n & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * h + j;
                const int ky = k >> 2, ci = k & 3;
                float wv = 0.0f;
                if (ky < 5 && kx < 5 && ci < CI && co < CO)
                    wv = DGRAD ? w[((ci * CO + co) * 5 + (4 - ky)) * 5 + (4 - kx)] : w[((co * CI + ci) * 5 + ky) * 5 + kx];
                __bf16 t0, t1, t2;
                split3(wv, t0, t1, t2);
                a[s][0][j] = t0;
                a[s][1][j] = t1;
                a[s][2][j] = t2;
            }
    }

    // ---- staging: input tile (+ 2-pixel halo, zeros outside the image) -> LDS [term][row][column][4 channels] bf16.  The blocks are
    // persistent; the global loads of the NEXT tile are issued before the current tile's rows are computed and land in registers
    // meanwhile (a tile's loads alone are ~30 us of a 64-image call: HBM time that has to hide behind the MFMAs).
    constexpr int TERM_BYTES = SR * SC * 8;
    constexpr int NPT = (SR * SC + 255) / 256;
    const size_t plane = (size_t)H * W;
    int prc[NPT];                                           // staged (row << 16 | column) of this thread's pixels, -1 past the tile
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
        const int p = tid + it * 256;
        const int r = p / SC, c = p - r * SC;
        prc[it] = p < SR * SC ? (r << 16 | c) : -1;
    }
    float v[NPT][4];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int img = t / tiles_y;
        const float* xi = x + (size_t)img * CI * plane;
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int gy = ty * TR - 2 + (prc[it] >> 16), gx = tx * TC - 2 + (prc[it] & 0xffff);
            const bool ok = prc[it] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            // pixels outside the image read a zero word with channel stride 0: no select behind the load (a select would make the
            // kernel wait for the prefetch right here instead of at the commit)
            const float* src = ok ? xi + (size_t)gy * W + gx : eas_sc_zero;
            const size_t cs = ok ? plane : 0;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) v[it][ci] = ci < CI ? src[ci * cs] : 0.0f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            if (prc[it] >= 0 && !(dbg & 4)) {
                unsigned char* dst = smem + (size_t)(tid + it * 256) * 8;
                if constexpr (XT == 1) {
                    bf16x4 t0;
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) t0[ci] = (__bf16)v[it][ci];
                    *(bf16x4*)dst = t0;
                } else {
                    bf16x4 t0, t1, t2;
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) {
                        __bf16 p0, p1, p2;
                        split3(v[it][ci], p0, p1, p2);
                        t0[ci] = p0; t1[ci] = p1; t2[ci] = p2;
                    }
                    *(bf16x4*)dst = t0;
                    *(bf16x4*)(dst + TERM_BYTES) = t1;
                    *(bf16x4*)(dst + 2 * TERM_BYTES) = t2;
                }
            }
        }
    };

    float* patch = reinterpret_cast<float*>(smem + XT * TERM_BYTES) + wave * (2 * 20 * PP);
    const int cbase = SWV * wave + n;                    // staged column of this lane's x'
    // output channels of this lane in the shift-add: CO = 4 -> {2h, 2h + 1}; CO = 2 -> {h}
    constexpr int NCO = CO == 4 ? 2 : 1;
    const int oc0 = CO == 4 ? 2 * h : h;
    float bv[NCO];
#pragma unroll
    for (int q = 0; q < NCO; ++q) bv[q] = bias ? bias[oc0 + q] : 0.0f;
    const size_t oplane = plane;

    int tile = blockIdx.x;                              // < ntiles (the launch has at most ntiles blocks)
    fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        commit();
        __syncthreads();
        // unconditional (the last round re-reads its own tile): loads behind a branch come with register copies at the join, and
        // the copies wait for the loads -- the prefetch would be awaited right here
        fetch(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch loads here: the scheduler otherwise sinks them to their first use
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int img = t / tiles_y;
        const int y0 = ty * TR, x0 = tx * TC;
        const int xo = x0 + SWV * wave + n;              // output column of this lane (lanes n < SWV)
        float* yi = y + (size_t)img * CO * oplane;
        const float* mi = mask ? mask + (size_t)img * CO * oplane : nullptr;

        // two output rows per iteration: independent accumulator chains for the matrix pipe, one patch round trip for both
        for (int rl = 0; rl < TR; rl += 2) {
            if (y0 + rl >= H) break;
            f32x16 acc0, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (dbg & 1) break;
                const int ky0 = 4 * s + 2 * h;
                const int r0 = rl + ky0 < SR ? rl + ky0 : SR - 1;          // rows past the tile only meet zero weights (ky >= 5)
                const int r1 = rl + ky0 + 1 < SR ? rl + ky0 + 1 : SR - 1;
                const int r2 = rl + ky0 + 2 < SR ? rl + ky0 + 2 : SR - 1;
                bf16x8 b0[XT], b1[XT];
#pragma unroll
                for (int tt = 0; tt < XT; ++tt) {
                    const bf16x4 q0 = *(const bf16x4*)(smem + tt * TERM_BYTES + ((size_t)r0 * SC + cbase) * 8);
                    const bf16x4 q1 = *(const bf16x4*)(smem + tt * TERM_BYTES + ((size_t)r1 * SC + cbase) * 8);
                    const bf16x4 q2 = *(const bf16x4*)(smem + tt * TERM_BYTES + ((size_t)r2 * SC + cbase) * 8);
                    b0[tt] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
                    b1[tt] = __builtin_shufflevector(q1, q2, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if constexpr (XT == 1) {
#pragma unroll
                    for (int ta = 2; ta >= 0; --ta) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][ta], b0[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][ta], b1[0], acc1, 0, 0, 0);
                    }
                } else {
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};       // smallest products first
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][PA[q]], b0[PB[q]], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][PA[q]], b1[PB[q]], acc1, 0, 0, 0);
                    }
                }
            }
            if (dbg & 2) {
                if (acc0[0] == 123.456f) yi[0] = acc1[1];
                continue;
            }
            // D[(kx, co)][x']: this lane holds column x' = n, rows m = 8q + 4h + co -> kx = 2q + h
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int kx = 2 * q + h;
                if (kx < 5) {
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        patch[(kx * 4 + co) * PP + n] = acc0[4 * q + co];
                        patch[(20 + kx * 4 + co) * PP + n] = acc1[4 * q + co];
                    }
                }
            }
            wave_lds_sync();
            if (n < SWV && xo < W) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (y0 + rl + rr < H && rl + rr < TR) {
                        const size_t o = (size_t)(y0 + rl + rr) * W + xo;
#pragma unroll
                        for (int q = 0; q < NCO; ++q) {
                            const int co = oc0 + q;
                            float u = patch[(rr * 20 + co) * PP + n];
#pragma unroll
                            for (int kx = 1; kx < 5; ++kx) u += patch[(rr * 20 + kx * 4 + co) * PP + n + kx];
                            u += bv[q];
                            if (relu) u = fmaxf(u, 0.0f);
                            if (mi) u = mi[co * oplane + o] > 0.0f ? u : 0.0f;
                            yi[co * oplane + o] = u;
                        }
                    }
                }
            }
            wave_lds_sync();
        }
        __syncthreads();          // every wave is done with the staged tile before the next one is committed
    }
}

template <int XT, int CI, int CO, bool DGRAD>
int launch_sc5(const float* x, const float* w, const float* b, const float* mask, float* y, int N, int H, int W, int relu, hipStream_t st) {
    auto kern = smallconv5_mfma_kernel<XT, CI, CO, DGRAD>;
    const size_t lds = (size_t)XT * SR * SC * 8 + (size_t)NWV * 2 * 20 * PP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
    const long ntiles = (long)N * tiles_x * tiles_y;
    if (ntiles > 0x7fffffffL) return EAS_ERR_UNSUPPORTED;
    static const int dbg = getenv("EAS_SC_DBG") ? atoi(getenv("EAS_SC_DBG")) : 0;      // development ablations: 1 no MFMA, 2 no epilogue, 4 no staging stores
    static const int maxb = getenv("EAS_SC_BLOCKS") ? atoi(getenv("EAS_SC_BLOCKS")) : 512;   // persistent blocks: two per CU
    const int blocks = ntiles < maxb ? (int)ntiles : maxb;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, x, w, b, mask, y, N, H, W, relu, tiles_x, tiles_y, (int)ntiles, dbg);
    return EAS_OK;
}

}  // namespace

// 5x5 forward (dgrad = 0: x [N][Cin][H][W] -> y [N][Cout][H][W]) or input gradient (dgrad = 1: x = grad_y [N][Cout][H][W] ->
// y = grad_x [N][Cin][H][W]) of a layer with weights w [Cout][Cin][5][5]; x_terms = 1 when x holds spikes / small integers.
// EAS_ERR_UNSUPPORTED for channel counts other than (2|4) -> (2|4): the caller falls back to the vector-ALU kernel.
int eas_sc5_mfma(const float* x, const float* w, const float* b, const float* mask, float* y, int N, int Cin, int Cout, int H, int W,
                 int relu, int dgrad, int x_terms, hipStream_t st) {
#define EAS_SC5(XT_, CI_, CO_, DG_) return launch_sc5<XT_, CI_, CO_, DG_>(x, w, b, mask, y, N, H, W, relu, st)
    if (!dgrad) {
        if (Cin == 2 && Cout == 4) { if (x_terms == 1) EAS_SC5(1, 2, 4, false); EAS_SC5(3, 2, 4, false); }
        if (Cin == 4 && Cout == 4) { if (x_terms == 1) EAS_SC5(1, 4, 4, false); EAS_SC5(3, 4, 4, false); }
        if (Cin == 2 && Cout == 2) { if (x_terms == 1) EAS_SC5(1, 2, 2, false); EAS_SC5(3, 2, 2, false); }
    } else {            // kernel input = grad_y (Cout channels), output = grad_x (Cin channels)
        if (Cin == 2 && Cout == 4) EAS_SC5(3, 4, 2, true);
        if (Cin == 4 && Cout == 4) EAS_SC5(3, 4, 4, true);
        if (Cin == 2 && Cout == 2) EAS_SC5(3, 2, 2, true);
    }
#undef EAS_SC5
    return EAS_ERR_UNSUPPORTED;
}
