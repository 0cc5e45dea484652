// Shared device helpers for libeas_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/eas_hip.h"

#define EAS_WAVE 64
#define EAS_BLOCK 256

// hipGetLastError() is sticky per thread and PyTorch leaves benign codes behind (hipErrorNotReady from
// event/stream queries): EAS_CLEAR_ERR() at entry, EAS_CHECK_LAUNCH() right after each launch.
#define EAS_CLEAR_ERR() (void)hipGetLastError()
#define EAS_CHECK_LAUNCH()                                          \
    do {                                                            \
        if (hipGetLastError() != hipSuccess) return EAS_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t eas_s(eas_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel-instance trace (eas_kernel_trace_begin / _dump, capi.hip): while it is on, every launch of the library records the symbol of
// the device kernel it starts -- the names rocprofv3 reports.  tests/test_gpu_bench_shapes.py uses it to prove that every kernel
// instance a bench step launches is also launched by a test that compares with fp64 / the oracle.  One predictable branch when off.
// Both library-internal (hidden visibility: not part of the ABI; eas_launch_counter() reads the count).
__attribute__((visibility("hidden"))) extern int eas_trace_on;
__attribute__((visibility("hidden"))) extern long long eas_launch_count;      // every kernel launch of the library since it was loaded
__attribute__((visibility("hidden"))) void eas_trace_kernel(const void* host_function);
#define EAS_LAUNCH(kern, ...)                                                \
    do {                                                                     \
        auto eas_kern_ = (kern);                                             \
        ++eas_launch_count;                                                  \
        if (eas_trace_on) eas_trace_kernel((const void*)eas_kern_);          \
        hipLaunchKernelGGL(eas_kern_, __VA_ARGS__);                          \
    } while (0)

// Development switches (tile / plan overrides, ablations: EAS_CONV_TILE, EAS_WG_SINGLE, EAS_CONV_DBG, ... -- INTEGRATION.md lists them) are read
// from the environment only in a development build (make DEV=1 -> -DEAS_DEV); the default library has no hidden state of that kind: the
// expression folds to "not set".  Three switches select between kernel FORMS that are all product code and stay readable by the tests in
// every build: EAS_HIST_FORM, EAS_BNLIF_BWD, EAS_SW_FORM.
#ifdef EAS_DEV
#define eas_dev_env(name) getenv(name)
#else
#define eas_dev_env(name) ((const char*)nullptr)
#endif

// memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int eas_grid_1d(int64_t work_items, int block = EAS_BLOCK, int max_blocks = 2048 * 4) {
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

__device__ __forceinline__ float eas_sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// surrogate gradient g'(u), u = h - v_th
__device__ __forceinline__ float eas_surrogate_grad(int id, float alpha, float u) {
    if (id == EAS_SG_ATAN || id == EAS_SG_PATAN) {   // PATAN: the same slope with alpha = |learnable alpha| (resolved by the caller)
        // alpha/2 / (1 + q^2) with the hardware reciprocal (1 ulp) instead of the IEEE division sequence (~10 instructions): the BN+LIF
        // backward is vector-ALU bound (two recomputing passes), and 1 ulp on a surrogate slope is far inside the 1e-4 parity bar
        const float q = 1.57079632679489661923f * alpha * u;
        return alpha * 0.5f * __builtin_amdgcn_rcpf(1.0f + q * q);
    } else if (id == EAS_SG_SIGMOID) {
        const float sg = eas_sigmoidf(alpha * u);
        return (1.0f - sg) * sg * alpha;
    } else {  // rectangle: [|u| < 0.5/alpha] * alpha
        return (fabsf(u) < 0.5f / alpha) ? alpha : 0.0f;
    }
}

// ---- wave / block reductions (wave64) ----------------------------------------------------
template <typename T>
__device__ __forceinline__ T eas_wave_sum(T v) {
#pragma unroll
    for (int off = EAS_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, EAS_WAVE);
    return v;  // valid in lane 0
}

// sum over a block of NW waves; result valid in thread 0. `smem` must hold NW values.
template <typename T, int NW>
__device__ __forceinline__ T eas_block_sum(T v, T* smem) {
    const int lane = threadIdx.x & (EAS_WAVE - 1);
    const int wid = threadIdx.x / EAS_WAVE;
    v = eas_wave_sum(v);
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r += smem[i];
    }
    __syncthreads();
    return r;
}

// ---- sums over the 32 lanes that share lane>>5 (the pixel columns of a 32x32 MFMA accumulator), DPP only (no LDS):
// butterfly inside the quads, mirror inside the half rows and rows of 16, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3.
// The total of lanes 0..31 is valid in lanes 16..31, the total of lanes 32..63 in lanes 48..63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float eas_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float eas_half_wave_sum(float v) {
    v += eas_dpp<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v += eas_dpp<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v += eas_dpp<0x141, 0xF>(v);     // row_half_mirror
    v += eas_dpp<0x140, 0xF>(v);     // row_mirror
    v += eas_dpp<0x142, 0xA>(v);     // row_bcast15 into rows 1 and 3 (other rows add 0)
    return v;
}

// Per-channel sums of a convolution's output tile for the BatchNorm that follows (eas_conv_fwd_stats): the wave's accumulators
// acc[WM][WN] (32 channels x 32 pixels each: pixel = lane & 31, channel = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)), `valid[n]` = this
// lane's pixel of tile n exists.  Writes sum and sum of squares of the wave's WM * 32 channels to red[(m * 32 + channel) * 2 + {0, 1}].
// Vector-ALU budget matters here (the waves of a SIMD share it and the MFMA kernels' epilogues are short): the sums over the wave's pixel
// tiles are packed-fp32 operations, and the sum over the 32 pixel lanes goes through a wave-private LDS patch (`scratch`,
// EAS_STATS_SCRATCH floats: 16 ds_write_b32, then lane c < 32 reads the 32 lane values of channel c with 8 conflict-free ds_read_b128 and
// adds them in lane order) instead of 5 DPP steps per value.
#define EAS_STATS_PITCH 36                          // floats per channel row of the patch (32 lanes + 4: b128 rows land on distinct banks)
#define EAS_STATS_SCRATCH (32 * EAS_STATS_PITCH)    // floats per wave
template <int WM, int WN, typename ACC>
__device__ __forceinline__ void eas_conv_stats_wave(const ACC (&acc)[WM][WN], const bool (&valid)[WN], float* scratch, double* red, int lane) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int h = lane >> 5, r = lane & 31;
    bool all = true;
#pragma unroll
    for (int n = 0; n < WN; ++n) all = all && valid[n];
    const bool uniform_all = __all(all);
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        f2 s[8], q[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { s[p] = f2{0.0f, 0.0f}; q[p] = f2{0.0f, 0.0f}; }
        if (uniform_all) {
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const f2 a = f2{acc[m][n][2 * p], acc[m][n][2 * p + 1]};
                    s[p] += a;
                    q[p] = __builtin_elementwise_fma(a, a, q[p]);
                }
        } else {
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const f2 a = f2{valid[n] ? acc[m][n][2 * p] : 0.0f, valid[n] ? acc[m][n][2 * p + 1] : 0.0f};
                    s[p] += a;
                    q[p] = __builtin_elementwise_fma(a, a, q[p]);
                }
        }
#pragma unroll
        for (int stat = 0; stat < 2; ++stat) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const f2 v = stat ? q[e >> 1] : s[e >> 1];
                scratch[(4 * h + (e & 3) + 8 * (e >> 2)) * EAS_STATS_PITCH + r] = (e & 1) ? v.y : v.x;
            }
            // the wave's own LDS writes are visible to its own later reads (LDS executes a wave's operations in order): only the
            // compiler must keep the order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 32) {
                const float4* row = reinterpret_cast<const float4*>(scratch + lane * EAS_STATS_PITCH);
                double t = 0.0;        // from here on in double (full-rate v_add_f64): sum of squares minus squared sum cancels downstream
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 v = row[k];
                    t += (double)v.x; t += (double)v.y; t += (double)v.z; t += (double)v.w;
                }
                red[(m * 32 + lane) * 2 + stat] = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// The waves of a block that computed the same channels (NWV waves, `red` = their eas_conv_stats_wave outputs, wave w at
// red + w * wave_stride) added in wave order; thread t < nch handles the block's channel t: co = first channel + t.
// stats[(co * nb + block) * 2 + {0, 1}].  Call after a __syncthreads().
__device__ __forceinline__ void eas_conv_stats_store(const double* red, int wave_stride, int nwv, int t, int co, int Cout, double* stats, int nb,
                                                     int block) {
    if (co >= Cout) return;
    double s = 0.0, q = 0.0;
    for (int w = 0; w < nwv; ++w) {
        s += red[w * wave_stride + t * 2 + 0];
        q += red[w * wave_stride + t * 2 + 1];
    }
    *reinterpret_cast<double2*>(stats + ((int64_t)co * nb + block) * 2) = make_double2(s, q);
}

#define EAS_BN_MAX_CHUNKS 64   // chunk partials per channel in the BN workspaces (kMaxChunks of bn_lif.hip / bn_act.hip)

// ------------------------------------------------------------------------------------------------ fused finalize
// Statistics finalize folded into the consumer: every block of channel c reduces that channel's chunk partials in the
// same fixed order (identical result in every block), block 0 of the channel publishes mean / invstd for the backward
// and updates the running statistics.  Saves one launch per layer (74 per SYOLOX-S step).
struct BnFin {
    const double* part;     // NULL: mean / invstd are inputs; else [C][pitch][2] partial sums, nchunks <= pitch of them used per channel
                            // (eas_bn_stats_partial: pitch 64; the producing convolution's epilogue, eas_conv_fwd_stats: pitch = nchunks
                            // = its pixel blocks, possibly thousands)
    int nchunks, pitch, replicas;
    double count;
    float eps, momentum;
    float* mean_out;
    float* invstd_out;
    float* rmean;           // nullable
    float* rvar;
};

// mean / invstd of channel c from its totals (sum, sum of squares); `publish` (one block per channel): write them for the backward
// and update the running statistics.  Every thread may call it: same double arithmetic, same result.
__device__ __forceinline__ void bn_from_totals(const BnFin& f, int c, double s, double ss, bool publish, float& mu, float& istd) {
    const double m = s / f.count;
    double var = ss / f.count - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    istd = (float)(1.0 / sqrt(var + (double)f.eps));
    if (publish) {
        f.mean_out[c] = mu;
        f.invstd_out[c] = istd;
        if (f.rmean) {
            const double full = f.count * f.replicas;
            const double unbiased = full > 1.0 ? var * full / (full - 1.0) : var;
            f.rmean[c] = (float)((1.0 - f.momentum) * f.rmean[c] + f.momentum * m);
            f.rvar[c] = (float)((1.0 - f.momentum) * f.rvar[c] + f.momentum * unbiased);
        }
    }
}

__device__ __forceinline__ void bn_finalize_in_block(const BnFin& f, int c, float* st, float& mu, float& istd, bool publish) {
    if (f.nchunks > EAS_WAVE) {
        // many partials (a convolution epilogue's), added in a fixed order (the same in every block of the channel and in every run):
        // thread t takes partials t, t + blockDim, ...; four loads in flight per thread (the finalize is pure load latency in front of
        // the block's work: a one-wave loop over 384 partials cost +12 us on a 38 us layer), then the block tree.
        __shared__ double fred[2][EAS_BLOCK / EAS_WAVE];
        const double2* p = reinterpret_cast<const double2*>(f.part) + (int64_t)c * f.pitch;
        const int B = blockDim.x, n = f.nchunks;
        double s = 0.0, ss = 0.0;
        for (int i = threadIdx.x; i < n; i += 4 * B) {
            const double2 z = make_double2(0.0, 0.0);
            const double2 v0 = p[i];
            const double2 v1 = i + B < n ? p[i + B] : z;
            const double2 v2 = i + 2 * B < n ? p[i + 2 * B] : z;
            const double2 v3 = i + 3 * B < n ? p[i + 3 * B] : z;
            s += v0.x; ss += v0.y;
            s += v1.x; ss += v1.y;
            s += v2.x; ss += v2.y;
            s += v3.x; ss += v3.y;
        }
        s = eas_wave_sum(s);
        ss = eas_wave_sum(ss);
        if ((threadIdx.x & (EAS_WAVE - 1)) == 0) {
            fred[0][threadIdx.x / EAS_WAVE] = s;
            fred[1][threadIdx.x / EAS_WAVE] = ss;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s = 0.0; ss = 0.0;
            for (int w = 0; w < (int)(blockDim.x / EAS_WAVE); ++w) { s += fred[0][w]; ss += fred[1][w]; }
            float m_, i_;
            bn_from_totals(f, c, s, ss, publish, m_, i_);
            st[0] = m_;
            st[1] = i_;
        }
        __syncthreads();
        mu = st[0];
        istd = st[1];
        return;
    }
    if (threadIdx.x < EAS_WAVE) {
        double s = 0.0, ss = 0.0;
        if ((int)threadIdx.x < f.nchunks) {
            s = f.part[((int64_t)c * f.pitch + threadIdx.x) * 2 + 0];
            ss = f.part[((int64_t)c * f.pitch + threadIdx.x) * 2 + 1];
        }
        s = eas_wave_sum(s);
        ss = eas_wave_sum(ss);
        if (threadIdx.x == 0) {
            const double m = s / f.count;
            double var = ss / f.count - m * m;
            if (var < 0.0) var = 0.0;
            st[0] = (float)m;
            st[1] = (float)(1.0 / sqrt(var + (double)f.eps));
            if (publish) {                  // one block per channel
                f.mean_out[c] = st[0];
                f.invstd_out[c] = st[1];
                if (f.rmean) {
                    const double full = f.count * f.replicas;
                    const double unbiased = full > 1.0 ? var * full / (full - 1.0) : var;
                    f.rmean[c] = (float)((1.0 - f.momentum) * f.rmean[c] + f.momentum * m);
                    f.rvar[c] = (float)((1.0 - f.momentum) * f.rvar[c] + f.momentum * unbiased);
                }
            }
        }
    }
    __syncthreads();
    mu = st[0];
    istd = st[1];
}

// ------------------------------------------------------------------------------------------------ channel grids
// The BN kernels decompose as (channel, chunk of the channel's pixel groups).  Grid = (8, chunks, ceil(C / 8)), channel = blockIdx.z * 8 +
// blockIdx.x: work-groups are dealt round-robin over the 8 XCDs in linear order (observed; a matter of speed only), so all chunks of a
// channel run on ONE XCD and share its L2.
#define EAS_CHAN_GRID(chunks, C) dim3(8, (chunks), ((C) + 7) / 8)

// (image, float4-group-in-plane) of group index g0 and its advance by `stride` groups, without a 64-bit division per group
struct GroupWalk {
    int n, q, dn, dq, hw4;
    __device__ __forceinline__ GroupWalk(int64_t g0, int64_t stride, int hw4_) : hw4(hw4_) {
        n = (int)(g0 / hw4_);
        q = (int)(g0 - (int64_t)n * hw4_);
        dn = (int)(stride / hw4_);
        dq = (int)(stride - (int64_t)dn * hw4_);
    }
    __device__ __forceinline__ void next() {
        n += dn;
        q += dq;
        if (q >= hw4) {
            q -= hw4;
            ++n;
        }
    }
};

struct EasLifParams {
    const float* w_logit;  // PLIF: k = sigmoid(*w_logit); else k_const
    float k_const;
    float v_th;
    float v_reset;
    int flags;
};

__device__ __forceinline__ float eas_lif_k(const EasLifParams& p) {
    return p.w_logit ? eas_sigmoidf(*p.w_logit) : p.k_const;
}

// One LIF step, arithmetic ordered like the PyTorch expression it replaces
// (-ffp-contract=off: every * and + rounds separately, as ATen elementwise kernels do).
template <bool HARD, bool DI, bool STRICT>
__device__ __forceinline__ void eas_lif_step(float& v, float x, float k, float omk, float v_th, float v_reset,
                                             float& h, float& s) {
    if (DI) {
        h = HARD ? v + (x - (v - v_reset)) * k : v + (x - v) * k;
    } else {
        h = (HARD && v_reset != 0.0f) ? (v - (v - v_reset) * k) + x : v * omk + x;
    }
    const float u = h - v_th;
    s = STRICT ? (u > 0.0f ? 1.0f : 0.0f) : (u >= 0.0f ? 1.0f : 0.0f);
    v = HARD ? (1.0f - s) * h + s * v_reset : h - s * v_th;
}

// dL/dh of one step and the pieces the caller needs.
//   gs: dL/ds_t from the output, gv: dL/dv_t from the future (updated to dL/dv_{t-1} on return)
//   returns dh = dL/dh_t;  dk_term = dh * dh/dk (for grad of the decay)
//   da (accumulated, EAS_SG_PATAN only): dL/d|alpha| of the learnable arctan surrogate sigma(u) = atan(pi/2 |alpha| u)/pi + 1/2
//   (yolox/models/activation.py:121-130): every use of s_t -- the output and, unless detached, the reset term -- carries
//   d sigma/d|alpha| = u/2 / (1 + (pi/2 |alpha| u)^2); the sign of alpha is applied by the finalize.
template <bool HARD, bool DI, bool STRICT>
__device__ __forceinline__ float eas_lif_step_bwd(float gs, float& gv, float h, float v_prev, float x, float k,
                                                  float omk, float v_th, float v_reset, bool detach, int sg_id,
                                                  float alpha, float& dk_term, float& gx, float& da) {
    const float u = h - v_th;
    const float s = STRICT ? (u > 0.0f ? 1.0f : 0.0f) : (u >= 0.0f ? 1.0f : 0.0f);
    const float sg = eas_surrogate_grad(sg_id, alpha, u);
    if (sg_id == EAS_SG_PATAN) {     // block-uniform branch
        const float q = 1.57079632679489661923f * alpha * u;
        const float gs_all = detach ? gs : (HARD ? gs + gv * (v_reset - h) : gs - gv * v_th);
        da += gs_all * (0.5f * u * __builtin_amdgcn_rcpf(1.0f + q * q));
    }
    float dvdh;
    if (HARD) {
        dvdh = detach ? (1.0f - s) : (1.0f - s) + (v_reset - h) * sg;
    } else {
        dvdh = detach ? 1.0f : 1.0f - v_th * sg;
    }
    const float dh = gs * sg + gv * dvdh;
    if (DI) {
        gx = dh * k;
        dk_term = dh * (HARD ? (x - (v_prev - v_reset)) : (x - v_prev));
    } else {
        gx = dh;
        dk_term = dh * ((HARD && v_reset != 0.0f) ? -(v_prev - v_reset) : -v_prev);
    }
    gv = dh * omk;
    return dh;
}
