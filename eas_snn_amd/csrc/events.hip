// K1: raw (t, x, y, p) event streams -> per-polarity count frames / voxel grids.
// Integer scatter: HBM/atomic bound.  Algorithmic bytes per sample: 9 B per event read
// (u32 t + u16 x + u16 y + u8 p) + 4*Tm*2*H*W B for the frames (zero-fill + result).
// One thread handles 4 consecutive events (16-B / 8-B / 8-B / 4-B vector loads); the atomics are
// int32 `global_atomic_add` without return, so results are bit-exact and order independent.
#include <stdlib.h>

#include <string.h>

#include "eas_common.h"

namespace {

// sample index of event i: largest b with offsets[b] <= i  (offsets ascending, offsets[B] = nev)
__device__ __forceinline__ int find_sample(const int64_t* __restrict__ offsets, int B, int64_t i) {
    int lo = 0, hi = B;  // invariant: offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

struct SampleWin {
    uint32_t t0;
    uint32_t win;  // 0 -> nothing is binned
};

__device__ __forceinline__ SampleWin sample_window(const uint32_t* __restrict__ t, const int64_t* __restrict__ offsets,
                                                   int b, int Tm) {
    const int64_t a = offsets[b], e = offsets[b + 1];
    SampleWin w;
    w.t0 = t[a];
    w.win = (t[e - 1] - w.t0) / (uint32_t)Tm;
    return w;
}

__device__ __forceinline__ void bin_one(uint32_t tt, uint32_t xx, uint32_t yy, uint32_t pp, const SampleWin& w, int b,
                                        int Tm, int H, int W, int32_t* __restrict__ out, uint32_t* __restrict__ oob) {
    if (w.win == 0) return;
    const uint32_t k = (tt - w.t0) / w.win;
    if (k >= (uint32_t)Tm) return;  // tail beyond t0 + Tm*win is dropped (gen1.py:321-326)
    if (xx >= (uint32_t)W || yy >= (uint32_t)H) {
        if (oob) atomicAdd(oob, 1u);
        return;
    }
    const int c = pp != 0 ? 1 : 0;
    const int64_t idx = ((((int64_t)b * Tm + k) * 2 + c) * H + yy) * W + xx;
    atomicAdd(out + idx, 1);
}

template <bool VEC4>
__global__ __launch_bounds__(EAS_BLOCK) void event_hist_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                               int64_t nev, const int64_t* __restrict__ offsets, int B,
                                                               int Tm, int H, int W, int32_t* __restrict__ out,
                                                               uint32_t* __restrict__ oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC4) {
        const int64_t ngroups = nev / 4;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
            const uint4 tv = reinterpret_cast<const uint4*>(t)[g];
            const ushort4 xv = reinterpret_cast<const ushort4*>(x)[g];
            const ushort4 yv = reinterpret_cast<const ushort4*>(y)[g];
            const uchar4 pv = reinterpret_cast<const uchar4*>(p)[g];
            const int64_t i0 = g * 4;
            int b = find_sample(offsets, B, i0);
            SampleWin w = sample_window(t, offsets, b, Tm);
            int64_t end = offsets[b + 1];
            const uint32_t ts[4] = {tv.x, tv.y, tv.z, tv.w};
            const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
            const uint32_t ys[4] = {yv.x, yv.y, yv.z, yv.w};
            const uint32_t ps[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                while (i0 + e >= end) {  // group straddles a sample boundary (also skips empty samples)
                    ++b;
                    end = offsets[b + 1];
                    if (i0 + e < end) w = sample_window(t, offsets, b, Tm);
                }
                bin_one(ts[e], xs[e], ys[e], ps[e], w, b, Tm, H, W, out, oob);
            }
        }
        // the last nev % 4 events
        if (blockIdx.x == 0 && threadIdx.x < (nev & 3)) {
            const int64_t i = (nev & ~(int64_t)3) + threadIdx.x;
            const int b = find_sample(offsets, B, i);
            const SampleWin w = sample_window(t, offsets, b, Tm);
            bin_one(t[i], x[i], y[i], p[i], w, b, Tm, H, W, out, oob);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += stride) {
            const int b = find_sample(offsets, B, i);
            const SampleWin w = sample_window(t, offsets, b, Tm);
            bin_one(t[i], x[i], y[i], p[i], w, b, Tm, H, W, out, oob);
        }
    }
}

// Same binning straight from Prophesee .dat records (8 bytes: u32 t, u32 packed with x = bits 0..13, y = bits 14..27,
// p = bit 28; yolox/utils/psee_loader/io/dat_events_tools.py:24-54): decode + window + histogram in one pass, so the raw
// file bytes are the only thing that crosses PCIe.  Two records per thread (one 16-byte load) when the buffer is aligned.
__device__ __forceinline__ SampleWin sample_window_dat(const uint2* __restrict__ rec, const int64_t* __restrict__ offsets, int b, int Tm) {
    const int64_t a = offsets[b], e = offsets[b + 1];
    SampleWin w;
    w.t0 = rec[a].x;
    w.win = (rec[e - 1].x - w.t0) / (uint32_t)Tm;
    return w;
}

__device__ __forceinline__ void bin_dat(uint32_t tt, uint32_t packed, const SampleWin& w, int b, int Tm, int H, int W,
                                        int32_t* __restrict__ out, uint32_t* __restrict__ oob) {
    bin_one(tt, packed & 16383u, (packed >> 14) & 16383u, (packed >> 28) & 1u, w, b, Tm, H, W, out, oob);
}

template <bool VEC2>
__global__ __launch_bounds__(EAS_BLOCK) void event_hist_dat_kernel(const uint2* __restrict__ rec, int64_t nev,
                                                                   const int64_t* __restrict__ offsets, int B, int Tm, int H, int W,
                                                                   int32_t* __restrict__ out, uint32_t* __restrict__ oob) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC2) {
        const int64_t npairs = nev / 2;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += stride) {
            const uint4 v = reinterpret_cast<const uint4*>(rec)[g];
            const int64_t i0 = g * 2;
            int b = find_sample(offsets, B, i0);
            SampleWin w = sample_window_dat(rec, offsets, b, Tm);
            bin_dat(v.x, v.y, w, b, Tm, H, W, out, oob);
            if (i0 + 1 >= offsets[b + 1]) {           // the pair straddles a sample boundary (also skips empty samples)
                b = find_sample(offsets, B, i0 + 1);
                w = sample_window_dat(rec, offsets, b, Tm);
            }
            bin_dat(v.z, v.w, w, b, Tm, H, W, out, oob);
        }
        if ((nev & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t i = nev - 1;
            const int b = find_sample(offsets, B, i);
            bin_dat(rec[i].x, rec[i].y, sample_window_dat(rec, offsets, b, Tm), b, Tm, H, W, out, oob);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += stride) {
            const int b = find_sample(offsets, B, i);
            bin_dat(rec[i].x, rec[i].y, sample_window_dat(rec, offsets, b, Tm), b, Tm, H, W, out, oob);
        }
    }
}

// ---- timestamp-window search on the device (second half) ------------------------------------------
// GEN1Dataset.search_events (yolox/data/datasets/gen1.py:217-232) over PSEELoader.seek_time / load_delta_t
// (yolox/utils/psee_loader/io/psee_loader.py:128-238), per label: the events of [ts + w0, ts + w0 + (w1 - w0)); while that is
// empty the window steps back by its own length, num_slice + 2 attempts at most.  The reader's quirks are kept, because they
// decide which events a sample is made of:
//   seek_time(T): T past the last timestamp -> end of file; T <= 0 -> first event AND current time 0 (not T); otherwise a
//   bisection that probes t[middle] while more than 100 000 events remain -- a probe that equals T exactly leaves the reader
//   one event AFTER the probed one -- then the first event with t >= T;  load_delta_t(d): events up to the first with
//   t >= current time + d.
// One thread per label; a recording is records [file_offsets[f], file_offsets[f + 1]) of the .dat image in HBM.
constexpr int64_t kSeekTermCriterion = 100000;

__device__ __forceinline__ int64_t lower_bound_rec(const uint2* __restrict__ rec, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rec[mid].x < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(EAS_WAVE) void event_window_search_kernel(const uint2* __restrict__ rec, const int64_t* __restrict__ file_offsets,
                                                                       const int32_t* __restrict__ file_id,
                                                                       const int64_t* __restrict__ label_t, int B, int64_t w0, int64_t w1,
                                                                       int num_slice, int64_t* __restrict__ ranges) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int f = file_id ? file_id[b] : 0;
    const int64_t base = file_offsets[f], n = file_offsets[f + 1] - base;
    const uint2* r = rec + base;
    const int64_t delta = w1 - w0;
    const int64_t total = n > 0 ? (int64_t)r[n - 1].x : 0;
    int64_t cur = label_t[b] + w0;
    int64_t a = 0, e = 0;
    for (int attempt = 0; attempt <= num_slice + 1; ++attempt, cur -= delta) {
        int64_t pos, now;
        if (cur > total) { pos = n; now = total + 1; }
        else if (cur <= 0) { pos = 0; now = 0; }
        else {
            int64_t low = 0, high = n;
            pos = -1;
            while (high - low > kSeekTermCriterion) {
                const int64_t middle = (low + high) / 2;
                const int64_t mid = (int64_t)r[middle].x;
                if (mid > cur) high = middle;
                else if (mid < cur) low = middle + 1;
                else { pos = middle + 1; break; }
            }
            if (pos < 0) pos = lower_bound_rec(r, low, high, cur);
            now = cur;
        }
        a = pos;
        e = pos >= n ? pos : lower_bound_rec(r, pos, n, now + delta);
        if (e > a) break;
    }
    ranges[2 * b] = base + a;
    ranges[2 * b + 1] = base + e;
}

// micro-slice count frames of arbitrary (possibly overlapping) record ranges: blockIdx.y = sample, blocks stride over its range
__global__ __launch_bounds__(EAS_BLOCK) void event_hist_dat_ranges_kernel(const uint2* __restrict__ rec, const int64_t* __restrict__ ranges,
                                                                          int Tm, int H, int W, int32_t* __restrict__ out,
                                                                          uint32_t* __restrict__ oob) {
    const int b = blockIdx.y;
    const int64_t a = ranges[2 * b], e = ranges[2 * b + 1];
    if (e <= a) return;
    SampleWin w;
    w.t0 = rec[a].x;
    w.win = (rec[e - 1].x - w.t0) / (uint32_t)Tm;
    for (int64_t i = a + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 v = rec[i];
        bin_dat(v.x, v.y, w, b, Tm, H, W, out, oob);
    }
}

// LDS-privatised form of the histogram for streams with many events per frame: one block owns one (sample, micro-slice,
// band of rows) and keeps that band's two polarity planes in LDS (<= 150 KB).  The events of a micro-slice are a
// contiguous run of the time-sorted stream (two binary searches = the reference's np.searchsorted, gen1.py:324-325), so the
// block streams through that run with coalesced loads, counts the events that fall into its rows with LDS atomics, and
// writes its band once with coalesced stores: no global atomics, no zero-fill pass, every output element written
// exactly once.  Each event is read by every band of its slice (L2-resident re-reads); bit-exact like the scatter form.
constexpr int kBandThreads = 1024;
constexpr int kBandLdsBytes = 150 * 1024;
constexpr int kMaxBands = 8;

__device__ __forceinline__ int64_t lower_bound_t(const uint32_t* __restrict__ t, int64_t lo, int64_t hi, uint32_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (t[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The same bound found by the 64 lanes of one wave: a round probes the last elements of 64 equal sub-ranges with ONE load per lane and
// keeps the sub-range that holds the answer -- 3 dependent rounds for a 200 k-event sample instead of the 18 dependent loads of the binary
// search (each a full HBM round trip issued by a single thread: the two searches of a block were ~25 us of its ~45 us).  All lanes of
// the wave call it with the same arguments and get the same result.
__device__ __forceinline__ int64_t lower_bound_wave(const uint32_t* __restrict__ t, int64_t lo, int64_t hi, uint32_t key) {
    const int lane = threadIdx.x & (EAS_WAVE - 1);
    while (hi - lo > EAS_WAVE) {
        const int64_t n = hi - lo, step = (n + EAS_WAVE - 1) / EAS_WAVE;
        const int64_t end = (lane + 1) * step < n ? (lane + 1) * step : n;          // sub-range `lane` = [lane * step, end) (empty ones probe the last element)
        const bool below = t[lo + end - 1] < key;                                    // sorted: the whole sub-range is below the key
        const int c = __popcll(__ballot(below));                                     // the lanes form 1..10..0: c sub-ranges lie below
        if (c == EAS_WAVE) return hi;
        const int64_t nlo = lo + c * step, nhi = lo + ((c + 1) * step < n ? (c + 1) * step : n);
        lo = nlo;
        hi = nhi;
    }
    const int64_t i = lo + lane;
    const bool below = i < hi && t[i] < key;
    return lo + __popcll(__ballot(below));
}

__global__ __launch_bounds__(kBandThreads) void event_hist_banded_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                                         const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                                         const int64_t* __restrict__ offsets, int B, int Tm, int H, int W,
                                                                         int rows, int nbands, int32_t* __restrict__ out,
                                                                         uint32_t* __restrict__ oob, float* __restrict__ canvas, int Hc,
                                                                         int Wc) {
    extern __shared__ int cnt[];            // [2][rows][W]
    __shared__ int64_t range[2];
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so the bands of one (sample, slice) get ids
    // 8 apart -- same XCD, one L2: the slice's events come from HBM once and from that L2 for the other bands
    const int L = blockIdx.x, tid = threadIdx.x;
    const int sl = (L & 7) + 8 * (L / (8 * nbands)), band = (L >> 3) % nbands;
    if (sl >= B * Tm) return;
    const int b = sl / Tm, k = sl - b * Tm;
    const int y0 = band * rows, y1 = y0 + rows < H ? y0 + rows : H;
    const int plane = rows * W;
    if ((plane & 3) == 0) {
        for (int i = tid; i < plane / 2; i += kBandThreads) reinterpret_cast<int4*>(cnt)[i] = make_int4(0, 0, 0, 0);
    } else {
        for (int i = tid; i < 2 * plane; i += kBandThreads) cnt[i] = 0;
    }
    if (tid < EAS_WAVE) {                   // the slice's event range [lo, hi): searched by the first wave (lower_bound_wave)
        const int64_t a = offsets[b], e = offsets[b + 1];
        int64_t lo = 0, hi = 0;
        if (e > a) {
            const uint32_t t0 = t[a];
            const uint32_t win = (t[e - 1] - t0) / (uint32_t)Tm;
            if (win != 0) {
                lo = lower_bound_wave(t, a, e, t0 + (uint32_t)k * win);
                hi = lower_bound_wave(t, lo, e, t0 + (uint32_t)(k + 1) * win);
            }
        }
        if (tid == 0) {
            range[0] = lo;
            range[1] = hi;
        }
    }
    __syncthreads();
    const int64_t lo = range[0], hi = range[1];
    unsigned bad = 0;
    constexpr int UN = 4;                   // events in flight per thread (all loads issued before the first LDS atomic)
    for (int64_t i0 = lo + tid; i0 < hi; i0 += (int64_t)UN * kBandThreads) {
        unsigned xs[UN], ys[UN], ps[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + (int64_t)u * kBandThreads;
            const bool in = i < hi;
            xs[u] = in ? x[i] : 0u;
            ys[u] = in ? y[i] : 0xffffffffu;          // sentinel: neither counted nor out-of-range
            ps[u] = in ? p[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ys[u] == 0xffffffffu) continue;
            if (xs[u] >= (unsigned)W || ys[u] >= (unsigned)H) {
                ++bad;
                continue;
            }
            if (ys[u] < (unsigned)y0 || ys[u] >= (unsigned)y1) continue;
            atomicAdd(&cnt[(ps[u] != 0 ? plane : 0) + (int)(ys[u] - y0) * W + (int)xs[u]], 1);
        }
    }
    if (band == 0 && oob && bad) atomicAdd(oob, bad);
    __syncthreads();
    const int n = (y1 - y0) * W;
    if (canvas) {
        // fp32 frames straight into the zero-padded model canvas [B][Tm][2][Hc][Wc] (gen1.py:447-455 + trainer.py:99 cast):
        // this band's rows with their right padding; the last band also writes the bottom padding rows
        const int yend = band == nbands - 1 ? Hc : y1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* dst = canvas + ((((int64_t)b * Tm + k) * 2 + c) * Hc + y0) * Wc;
            const int total = (yend - y0) * Wc;
            for (int i = tid; i < total; i += kBandThreads) {
                const int r = i / Wc, col = i - r * Wc;
                dst[i] = (r < y1 - y0 && col < W) ? (float)cnt[c * plane + r * W + col] : 0.0f;
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int32_t* dst = out + ((((int64_t)b * Tm + k) * 2 + c) * H + y0) * W;
        if ((n & 3) == 0 && (plane & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
            for (int i = tid; i < n / 4; i += kBandThreads) reinterpret_cast<int4*>(dst)[i] = reinterpret_cast<const int4*>(cnt + c * plane)[i];
        } else {
            for (int i = tid; i < n; i += kBandThreads) dst[i] = cnt[c * plane + i];
        }
    }
}

// The same band form with 16-bit counters, two per LDS word: a band is twice as many rows (2 bands instead of 4 for the 240-row sensor:
// every event is read by half as many blocks and the grid is two rounds of blocks instead of four), and the events come in aligned
// groups of four (one 8-byte load each for x and y, one 4-byte load for p, instead of twelve 1- and 2-byte loads).  A counter cannot
// overflow while the slice holds < 65536 events; a block whose slice holds more (known only on the device) counts its band in two halves
// with 32-bit counters, one after the other -- the same LDS, the events read twice by that block only.  Bit-exact like the other forms.
__global__ __launch_bounds__(kBandThreads) void event_hist_banded16_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                                           const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                                           const int64_t* __restrict__ offsets, int64_t nev, int B, int Tm, int H,
                                                                           int W, int rows, int nbands, int32_t* __restrict__ out,
                                                                           uint32_t* __restrict__ oob, float* __restrict__ canvas, int Hc, int Wc) {
    extern __shared__ int cnt[];            // packed: [2][rows][W] 16-bit counters; wide (fallback): [2][rows / 2 (rounded up)][W] 32-bit
    __shared__ int64_t range[2];
    const int L = blockIdx.x, tid = threadIdx.x;
    const int sl = (L & 7) + 8 * (L / (8 * nbands)), band = (L >> 3) % nbands;      // XCD-aware order, see event_hist_banded_kernel
    if (sl >= B * Tm) return;
    const int b = sl / Tm, k = sl - b * Tm;
    const int y0 = band * rows, y1 = y0 + rows < H ? y0 + rows : H;
    if (tid < EAS_WAVE) {                   // the slice's event range [lo, hi): searched by the first wave (lower_bound_wave)
        const int64_t a = offsets[b], e = offsets[b + 1];
        int64_t lo = 0, hi = 0;
        if (e > a) {
            const uint32_t t0 = t[a];
            const uint32_t win = (t[e - 1] - t0) / (uint32_t)Tm;
            if (win != 0) {
                lo = lower_bound_wave(t, a, e, t0 + (uint32_t)k * win);
                hi = lower_bound_wave(t, lo, e, t0 + (uint32_t)(k + 1) * win);
            }
        }
        if (tid == 0) {
            range[0] = lo;
            range[1] = hi;
        }
    }
    __syncthreads();
    const int64_t lo = range[0], hi = range[1];
    const bool packed = hi - lo < 65536;
    const int hrows = (rows + 1) / 2;                         // rows per pass of the wide fallback
    const int npass = packed ? 1 : 2;
    unsigned bad = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int ya = packed ? y0 : y0 + pass * hrows;
        const int yb = packed ? y1 : (ya + hrows < y1 ? ya + hrows : y1);
        const int prow = packed ? rows : hrows;               // rows per polarity plane in this pass
        const int plane = prow * W;
        const int words = packed ? (2 * plane + 1) / 2 : 2 * plane;
        __syncthreads();                                      // pass 1: pass 0's counters have been written out
        for (int i = tid; i < words; i += kBandThreads) cnt[i] = 0;
        __syncthreads();
        constexpr int UN = 4;                                 // aligned groups of four events in flight per thread
        const int64_t g0 = lo & ~(int64_t)3;
        for (int64_t i0 = g0 + 4 * (int64_t)tid; i0 < hi; i0 += (int64_t)UN * 4 * kBandThreads) {
            uint2 xs[UN], ys[UN];
            unsigned ps[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = i0 + (int64_t)u * 4 * kBandThreads;
                if (i < hi && i + 3 < nev) {
                    xs[u] = *reinterpret_cast<const uint2*>(x + i);
                    ys[u] = *reinterpret_cast<const uint2*>(y + i);
                    ps[u] = *reinterpret_cast<const unsigned*>(p + i);
                } else {                                      // past the slice, or the last (ragged) group of the whole stream
                    unsigned xv[4] = {0, 0, 0, 0}, yv[4] = {0, 0, 0, 0}, pv = 0;
                    for (int j = 0; j < 4; ++j)
                        if (i + j < hi) { xv[j] = x[i + j]; yv[j] = y[i + j]; pv |= (unsigned)p[i + j] << (8 * j); }
                    xs[u] = make_uint2(xv[0] | (xv[1] << 16), xv[2] | (xv[3] << 16));
                    ys[u] = make_uint2(yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16));
                    ps[u] = pv;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = i0 + (int64_t)u * 4 * kBandThreads;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t e = i + j;
                    if (e < lo || e >= hi) continue;
                    const unsigned xv = ((j < 2 ? xs[u].x : xs[u].y) >> (16 * (j & 1))) & 0xffffu;
                    const unsigned yv = ((j < 2 ? ys[u].x : ys[u].y) >> (16 * (j & 1))) & 0xffffu;
                    const unsigned pv = (ps[u] >> (8 * j)) & 0xffu;
                    if (xv >= (unsigned)W || yv >= (unsigned)H) {
                        if (pass == 0) ++bad;
                        continue;
                    }
                    if (yv < (unsigned)ya || yv >= (unsigned)yb) continue;
                    const int idx = (pv != 0 ? plane : 0) + (int)(yv - ya) * W + (int)xv;
                    if (packed) atomicAdd(&cnt[idx >> 1], (idx & 1) ? 65536 : 1);
                    else atomicAdd(&cnt[idx], 1);
                }
            }
        }
        __syncthreads();
        auto count_at = [&](int c, int r, int col) -> int {
            const int idx = c * plane + r * W + col;
            return packed ? (int)(((unsigned)cnt[idx >> 1] >> (16 * (idx & 1))) & 0xffffu) : cnt[idx];
        };
        if (canvas) {
            // this pass's rows with their right padding; the last pass of the last band also writes the bottom padding rows
            const bool last = band == nbands - 1 && pass == npass - 1;
            const int yend = last ? Hc : yb;
            const bool quads = packed && (W & 3) == 0 && (Wc & 3) == 0 && (plane & 3) == 0 && ((uintptr_t)canvas & 15) == 0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float* dst = canvas + ((((int64_t)b * Tm + k) * 2 + c) * Hc + ya) * Wc;
                if (quads) {
                    // four columns per thread: one 8-byte LDS read of four packed counters, one 16-byte store
                    const int q4 = Wc / 4, total4 = (yend - ya) * q4;
                    const unsigned short* c16 = reinterpret_cast<const unsigned short*>(cnt) + c * plane;
                    for (int i = tid; i < total4; i += kBandThreads) {
                        const int r = i / q4, col = (i - r * q4) * 4;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r < yb - ya && col < W) {
                            const uint2 w = *reinterpret_cast<const uint2*>(c16 + r * W + col);
                            v = make_float4((float)(w.x & 0xffffu), (float)(w.x >> 16), (float)(w.y & 0xffffu), (float)(w.y >> 16));
                        }
                        *reinterpret_cast<float4*>(dst + (int64_t)r * Wc + col) = v;
                    }
                    continue;
                }
                const int total = (yend - ya) * Wc;
                for (int i = tid; i < total; i += kBandThreads) {
                    const int r = i / Wc, col = i - r * Wc;
                    dst[i] = (r < yb - ya && col < W) ? (float)count_at(c, r, col) : 0.0f;
                }
            }
        } else {
            const int n = (yb - ya) * W;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int32_t* dst = out + ((((int64_t)b * Tm + k) * 2 + c) * H + ya) * W;
                if (packed && (n & 3) == 0 && (plane & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
                    // the band's rows are contiguous in the output as in LDS: four packed counters per 8-byte read, one 16-byte store
                    const unsigned short* c16 = reinterpret_cast<const unsigned short*>(cnt) + c * plane;
                    for (int i = tid; i < n / 4; i += kBandThreads) {
                        const uint2 w = *reinterpret_cast<const uint2*>(c16 + 4 * i);
                        reinterpret_cast<int4*>(dst)[i] = make_int4((int)(w.x & 0xffffu), (int)(w.x >> 16), (int)(w.y & 0xffffu), (int)(w.y >> 16));
                    }
                    continue;
                }
                for (int i = tid; i < n; i += kBandThreads) {
                    const int r = i / W, col = i - r * W;
                    dst[i] = count_at(c, r, col);
                }
            }
        }
    }
    if (band == 0 && oob && bad) atomicAdd(oob, bad);
}

__global__ __launch_bounds__(EAS_BLOCK) void counts_to_canvas_kernel(const int32_t* __restrict__ counts, int64_t F, int H,
                                                                     int W, int Hc, int Wc, float* __restrict__ out) {
    const int64_t total = F * Hc * Wc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wc);
        const int64_t r = i / Wc;
        const int yy = (int)(r % Hc);
        const int64_t f = r / Hc;
        out[i] = (xx < W && yy < H) ? (float)counts[(f * H + yy) * W + xx] : 0.0f;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void voxel_grid_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                               int64_t nev, const int64_t* __restrict__ offsets, int B,
                                                               int nb, int H, int W, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = find_sample(offsets, B, i);
        const int64_t a = offsets[b], e = offsets[b + 1];
        const double t0 = (double)t[a];
        const double span = (double)t[e - 1] - t0;
        if (span == 0.0) continue;  // single-timestamp stream: declared invalid, left zero
        const uint32_t xx = x[i], yy = y[i];
        if (xx >= (uint32_t)W || yy >= (uint32_t)H) continue;
        // same operation order as event_reps.py:53-57 in float64
        const double ts = (double)nb * ((double)t[i] - t0) / span;
        const int64_t ti = (int64_t)ts;
        const double dt = ts - (double)ti;
        const double pol = p[i] != 0 ? 1.0 : -1.0;
        double* base = out + (int64_t)b * nb * H * W + (int64_t)yy * W + xx;
        if (ti < nb) atomicAdd(base + ti * H * W, pol * (1.0 - dt));
        if (ti + 1 < nb) atomicAdd(base + (ti + 1) * H * W, pol * dt);
    }
}

// Voxel cube (to_voxel_cube_numpy, yolox/utils/event_reps.py:92-138): counts per (slice, channel, y, x) with
// slice = floor(float32(t_rel) / float32(window)) and channel = (p + 1) * (tbin + 1) - 1, tbin = floor((t_rel % window) /
// (window / tbins)) in float64 -- the reference's own arithmetic, quirks included (channel collisions for tbins >= 2).
__global__ __launch_bounds__(EAS_BLOCK) void voxel_cube_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                               int64_t nev, const int64_t* __restrict__ offsets, int B, int ns,
                                                               int tbins, int H, int W, int32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = find_sample(offsets, B, i);
        const int64_t a = offsets[b], e = offsets[b + 1];
        const uint32_t t0 = t[a];
        const uint32_t win = (t[e - 1] - t0) / (uint32_t)ns;
        if (win == 0) continue;
        const uint32_t tr = t[i] - t0;
        if ((uint64_t)tr >= (uint64_t)win * (uint64_t)ns) continue;        // tail beyond ns whole windows is dropped
        const uint32_t xx = x[i], yy = y[i];
        if (xx >= (uint32_t)W || yy >= (uint32_t)H) continue;
        const int sl = (int)floorf((float)(int32_t)tr / (float)win);
        const int tb = (int)floor((double)(tr % win) / ((double)win / (double)tbins));
        const int ch = ((int)p[i] + 1) * (tb + 1) - 1;
        if (sl >= ns || ch >= 2 * tbins) continue;                         // the reference raises on these; never for p in {0,1}
        atomicAdd(out + ((((int64_t)b * ns + sl) * (2 * tbins) + ch) * H + yy) * W + xx, 1);
    }
}

// Time surface (GEN1Dataset.agrregate 'timesurface', gen1.py:362-369 + to_timesurface_numpy, event_reps.py:141-160).
// Pass 1: latest timestamp per (sample, slice, polarity, pixel) by integer atomic max (timestamps ascend, so the
// maximum is the reference's "last write wins").  Pass 2: running maximum over the slices, then
// exp(-((i + 1) * window + t0 - latest) / tau) in float64; untouched pixels keep latest = 0 exactly like the reference.
__global__ __launch_bounds__(EAS_BLOCK) void time_surface_scatter_kernel(const uint32_t* __restrict__ t, const uint16_t* __restrict__ x,
                                                                         const uint16_t* __restrict__ y, const uint8_t* __restrict__ p,
                                                                         int64_t nev, const int64_t* __restrict__ offsets, int B,
                                                                         int ns, int H, int W, uint32_t* __restrict__ latest) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nev; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = find_sample(offsets, B, i);
        const SampleWin w = sample_window(t, offsets, b, ns);
        if (w.win == 0) continue;
        const uint32_t k = (t[i] - w.t0) / w.win;
        if (k >= (uint32_t)ns) continue;
        const uint32_t xx = x[i], yy = y[i], pp = p[i];
        if (xx >= (uint32_t)W || yy >= (uint32_t)H || pp > 1u) continue;
        atomicMax(latest + ((((int64_t)b * ns + k) * 2 + pp) * H + yy) * W + xx, t[i]);
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void time_surface_exp_kernel(const uint32_t* __restrict__ t, const int64_t* __restrict__ offsets,
                                                                     int B, int ns, int H, int W, double tau,
                                                                     const uint32_t* __restrict__ latest, double* __restrict__ out) {
    const int64_t plane = (int64_t)2 * H * W, total = (int64_t)B * plane;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / plane);
        const int64_t q = i - (int64_t)b * plane;
        const int64_t a = offsets[b], e = offsets[b + 1];
        SampleWin w{0u, 0u};
        if (e > a) w = sample_window(t, offsets, b, ns);
        uint32_t m = 0;
        for (int k = 0; k < ns; ++k) {
            const int64_t idx = ((int64_t)b * ns + k) * plane + q;
            if (w.win == 0) {                       // no events, or a window of zero length (invalid input): zeros
                out[idx] = 0.0;
                continue;
            }
            const uint32_t l = latest[idx];
            m = l > m ? l : m;
            const uint32_t end = (uint32_t)(k + 1) * w.win + w.t0;       // uint32 arithmetic as numpy's
            const int64_t diff = -((int64_t)end - (int64_t)m);
            out[idx] = exp((double)diff / tau);
        }
    }
}

// Letterbox / jitter augmentation of count frames on the device (GEN1Dataset.get_random_data, gen1.py:433-521): every
// frame of sample b is resized to (nw, nh) with cv2.resize(INTER_LINEAR) semantics, pasted at (dx, dy) into a zero canvas,
// optionally mirrored left-right, and cast to fp32 (trainer.py:99).  cv2's linear resize for float64 images, restated from
// OpenCV's resize.cpp (not in the reference tree, opencv-python pinned by pip-requirements.txt; no cv2 in this image:
// parity unpinned): fx = float((j + 0.5) * (iw / nw) - 0.5), sx = floor(fx), fx -= sx, clamped at both borders with
// fx = 0; float32 weights, float64 arithmetic, horizontal pass first: out = (S[sy][sx]*a0 + S[sy][sx+1]*a1)*b0 + (...)*b1.
struct AxisTap { int s0, s1; float w0, w1; };

__device__ __forceinline__ AxisTap linear_tap(int j, int n_src, int n_dst) {
    const double scale = (double)n_src / (double)n_dst;
    float f = (float)(((double)j + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    AxisTap t;
    t.s0 = s;
    t.s1 = s + 1 < n_src ? s + 1 : n_src - 1;
    t.w0 = 1.f - f;
    t.w1 = f;
    return t;
}

__global__ __launch_bounds__(EAS_BLOCK) void counts_letterbox_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ params,
                                                                     int B, int F, int H, int W, int Hc, int Wc, float* __restrict__ out) {
    const int64_t total = (int64_t)B * F * Hc * Wc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wc);
        const int64_t r = i / Wc;
        const int yy = (int)(r % Hc);
        const int64_t f = r / Hc;
        const int b = (int)(f / F);
        const int32_t* pr = params + 5 * b;
        const int nw = pr[0], nh = pr[1], dx = pr[2], dy = pr[3], flip = pr[4];
        const int xs = (flip ? Wc - 1 - xx : xx) - dx, ys = yy - dy;
        float v = 0.f;
        if (xs >= 0 && xs < nw && ys >= 0 && ys < nh) {
            const int32_t* src = counts + f * H * W;
            if (nw == W && nh == H) {
                v = (float)src[ys * W + xs];                  // cv2.resize with dsize == size is a copy
            } else {
                const AxisTap tx = linear_tap(xs, W, nw), ty = linear_tap(ys, H, nh);
                const double r0 = (double)src[ty.s0 * W + tx.s0] * (double)tx.w0 + (double)src[ty.s0 * W + tx.s1] * (double)tx.w1;
                const double r1 = (double)src[ty.s1 * W + tx.s0] * (double)tx.w0 + (double)src[ty.s1 * W + tx.s1] * (double)tx.w1;
                v = (float)(r0 * (double)ty.w0 + r1 * (double)ty.w1);
            }
        }
        out[i] = v;
    }
}

}  // namespace

extern "C" {

// counts into ``out``; or, when ``canvas`` is given and the banded form applies, fp32 frames straight into the canvas
// (*wrote_canvas = 1) without touching ``out``
static int histogram_impl(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                          const int64_t* sample_offsets, int B, int Tm, int H, int W, int32_t* out,
                          uint32_t* oob_count, float* canvas, int Hc, int Wc, int* wrote_canvas, eas_stream_t stream) {
    if (wrote_canvas) *wrote_canvas = 0;
    if (!out || !sample_offsets || B < 1 || Tm < 1 || H < 1 || W < 1 || nev < 0) return EAS_ERR_INVALID_ARG;
    if (nev > 0 && (!t || !x || !y || !p)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const size_t bytes = (size_t)B * Tm * 2 * H * W * sizeof(int32_t);
    if (oob_count && hipMemsetAsync(oob_count, 0, sizeof(uint32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    // dense streams (>= 2048 events per frame on average) whose frame splits into at most 8 LDS-sized row bands: LDS form
    int rows = kBandLdsBytes / (8 * W);
    if (rows > H) rows = H;
    const int nbands = rows > 0 ? (H + rows - 1) / rows : kMaxBands + 1;
    const char* force = getenv("EAS_HIST_FORM");      // development switch: "scatter" / "banded"
    const bool dense = nev >= (int64_t)B * Tm * 2048;
    // 16-bit counters + vector loads when the event arrays allow 8- / 4-byte loads (EAS_HIST_FORM=banded32: the 32-bit form)
    const bool al16 = ((((uintptr_t)x | (uintptr_t)y) & 7) | ((uintptr_t)p & 3)) == 0 && !(force && strcmp(force, "banded32") == 0);
    int rows16 = kBandLdsBytes / (4 * W);
    if (rows16 > H) rows16 = H;
    const int nbands16 = rows16 > 0 ? (H + rows16 - 1) / rows16 : kMaxBands + 1;
    if (al16 && nbands16 <= kMaxBands && (int64_t)B * Tm < (1 << 24) && (force ? force[0] == 'b' : dense)) {
        const int r16 = (H + nbands16 - 1) / nbands16;
        static bool attr16 = false;
        if (!attr16) {
            if (hipFuncSetAttribute((const void*)event_hist_banded16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBandLdsBytes) != hipSuccess)
                return EAS_ERR_LAUNCH;
            attr16 = true;
        }
        const int nb = (H + r16 - 1) / r16;
        const int64_t groups = ((int64_t)B * Tm + 7) / 8;
        // LDS: packed [2][r16][W] halves, or the fallback's [2][ceil(r16 / 2)][W] words -- the larger of the two
        const size_t lds16 = (size_t)4 * (((size_t)2 * r16 * W + 1) / 2 > (size_t)2 * ((r16 + 1) / 2) * W ? ((size_t)2 * r16 * W + 1) / 2
                                                                                                        : (size_t)2 * ((r16 + 1) / 2) * W);
        EAS_LAUNCH(event_hist_banded16_kernel, dim3((unsigned)(groups * 8 * nb)), dim3(kBandThreads), lds16, st, t, x, y, p,
                           sample_offsets, nev, B, Tm, H, W, r16, nb, out, oob_count, canvas, Hc, Wc);
        EAS_CHECK_LAUNCH();
        if (canvas && wrote_canvas) *wrote_canvas = 1;
        return EAS_OK;
    }
    if (nbands <= kMaxBands && (int64_t)B * Tm < (1 << 24) && (force ? force[0] == 'b' : dense)) {
        rows = (H + nbands - 1) / nbands;              // even bands
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)event_hist_banded_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBandLdsBytes) != hipSuccess)
                return EAS_ERR_LAUNCH;
            attr_set = true;
        }
        const int nb = (H + rows - 1) / rows;
        const int64_t groups = ((int64_t)B * Tm + 7) / 8;
        EAS_LAUNCH(event_hist_banded_kernel, dim3((unsigned)(groups * 8 * nb)), dim3(kBandThreads), (size_t)2 * rows * W * 4, st, t, x, y,
                           p, sample_offsets, B, Tm, H, W, rows, nb, out, oob_count, canvas, Hc, Wc);
        EAS_CHECK_LAUNCH();
        if (canvas && wrote_canvas) *wrote_canvas = 1;
        return EAS_OK;
    }
    if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (nev == 0) return EAS_OK;
    const bool vec = (((uintptr_t)t & 15) | ((uintptr_t)x & 7) | ((uintptr_t)y & 7) | ((uintptr_t)p & 3)) == 0;
    if (vec) {
        EAS_LAUNCH(event_hist_kernel<true>, dim3(eas_grid_1d((nev + 3) / 4)), dim3(EAS_BLOCK), 0, st, t, x, y, p,
                           nev, sample_offsets, B, Tm, H, W, out, oob_count);
    } else {
        EAS_LAUNCH(event_hist_kernel<false>, dim3(eas_grid_1d(nev)), dim3(EAS_BLOCK), 0, st, t, x, y, p, nev,
                           sample_offsets, B, Tm, H, W, out, oob_count);
    }
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_event_histogram(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                        const int64_t* sample_offsets, int B, int Tm, int H, int W, int32_t* out,
                        uint32_t* oob_count, eas_stream_t stream) {
    return histogram_impl(t, x, y, p, nev, sample_offsets, B, Tm, H, W, out, oob_count, nullptr, 0, 0, nullptr, stream);
}

int eas_event_window_search(const void* records, const int64_t* file_offsets, int F, const int32_t* file_id, const int64_t* label_t, int B,
                            int64_t window_lo, int64_t window_hi, int num_slice, int64_t* ranges, eas_stream_t stream) {
    if (!records || !file_offsets || !label_t || !ranges || F < 1 || B < 1 || window_hi - window_lo < 1 || num_slice < 0) return EAS_ERR_INVALID_ARG;
    if ((uintptr_t)records & 7) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(event_window_search_kernel, dim3((B + EAS_WAVE - 1) / EAS_WAVE), dim3(EAS_WAVE), 0, eas_s(stream), (const uint2*)records,
                       file_offsets, file_id, label_t, B, window_lo, window_hi, num_slice, ranges);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_event_histogram_dat_ranges(const void* records, const int64_t* ranges, int B, int Tm, int H, int W, int32_t* out, uint32_t* oob_count,
                                   eas_stream_t stream) {
    if (!records || !ranges || !out || B < 1 || Tm < 1 || H < 1 || W < 1 || ((uintptr_t)records & 7)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (hipMemsetAsync(out, 0, (size_t)B * Tm * 2 * H * W * sizeof(int32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (oob_count && hipMemsetAsync(oob_count, 0, sizeof(uint32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    // the ranges live on the device (no host read): a fixed number of blocks per sample strides over whatever it holds
    EAS_LAUNCH(event_hist_dat_ranges_kernel, dim3(64, B), dim3(EAS_BLOCK), 0, st, (const uint2*)records, ranges, Tm, H, W, out, oob_count);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_counts_to_canvas(const int32_t* counts, int64_t F, int H, int W, int Hc, int Wc, float* out, eas_stream_t stream);

int eas_event_frames(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                     const int64_t* sample_offsets, int B, int Tm, int H, int W, int Hc, int Wc, float* frames,
                     int32_t* scratch_counts, uint32_t* oob_count, eas_stream_t stream) {
    if (!frames || !scratch_counts || Hc < H || Wc < W) return EAS_ERR_INVALID_ARG;
    int wrote = 0;
    const int rc = histogram_impl(t, x, y, p, nev, sample_offsets, B, Tm, H, W, scratch_counts, oob_count, frames, Hc, Wc, &wrote, stream);
    if (rc != EAS_OK || wrote) return rc;
    return eas_counts_to_canvas(scratch_counts, (int64_t)B * Tm * 2, H, W, Hc, Wc, frames, stream);
}

int eas_event_histogram_dat(const void* records, int64_t nev, const int64_t* sample_offsets, int B, int Tm, int H, int W,
                            int32_t* out, uint32_t* oob_count, eas_stream_t stream) {
    if (!out || !sample_offsets || B < 1 || Tm < 1 || H < 1 || W < 1 || nev < 0) return EAS_ERR_INVALID_ARG;
    if (nev > 0 && (!records || ((uintptr_t)records & 7))) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const size_t bytes = (size_t)B * Tm * 2 * H * W * sizeof(int32_t);
    if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (oob_count && hipMemsetAsync(oob_count, 0, sizeof(uint32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (nev == 0) return EAS_OK;
    const uint2* rec = (const uint2*)records;
    if (((uintptr_t)records & 15) == 0)
        EAS_LAUNCH(event_hist_dat_kernel<true>, dim3(eas_grid_1d((nev + 1) / 2)), dim3(EAS_BLOCK), 0, st, rec, nev, sample_offsets, B,
                           Tm, H, W, out, oob_count);
    else
        EAS_LAUNCH(event_hist_dat_kernel<false>, dim3(eas_grid_1d(nev)), dim3(EAS_BLOCK), 0, st, rec, nev, sample_offsets, B, Tm, H,
                           W, out, oob_count);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_counts_to_canvas(const int32_t* counts, int64_t F, int H, int W, int Hc, int Wc, float* out,
                         eas_stream_t stream) {
    if (!counts || !out || F < 0 || H < 1 || W < 1 || Hc < H || Wc < W) return EAS_ERR_INVALID_ARG;
    if (F == 0) return EAS_OK;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(counts_to_canvas_kernel, dim3(eas_grid_1d(F * Hc * Wc)), dim3(EAS_BLOCK), 0, eas_s(stream), counts,
                       F, H, W, Hc, Wc, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_event_voxel_grid(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                         const int64_t* sample_offsets, int B, int n_bins, int H, int W, double* out,
                         eas_stream_t stream) {
    if (!out || !sample_offsets || B < 1 || n_bins < 1 || H < 1 || W < 1 || nev < 0) return EAS_ERR_INVALID_ARG;
    if (nev > 0 && (!t || !x || !y || !p)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (hipMemsetAsync(out, 0, (size_t)B * n_bins * H * W * sizeof(double), st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (nev == 0) return EAS_OK;
    EAS_LAUNCH(voxel_grid_kernel, dim3(eas_grid_1d(nev)), dim3(EAS_BLOCK), 0, st, t, x, y, p, nev, sample_offsets,
                       B, n_bins, H, W, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_event_voxel_cube(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                         const int64_t* sample_offsets, int B, int num_slices, int tbins, int H, int W, int32_t* out,
                         eas_stream_t stream) {
    if (!out || !sample_offsets || B < 1 || num_slices < 1 || tbins < 1 || H < 1 || W < 1 || nev < 0) return EAS_ERR_INVALID_ARG;
    if (nev > 0 && (!t || !x || !y || !p)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    if (hipMemsetAsync(out, 0, (size_t)B * num_slices * 2 * tbins * H * W * sizeof(int32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (nev == 0) return EAS_OK;
    EAS_LAUNCH(voxel_cube_kernel, dim3(eas_grid_1d(nev)), dim3(EAS_BLOCK), 0, st, t, x, y, p, nev, sample_offsets, B, num_slices,
                       tbins, H, W, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_event_time_surface(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                           const int64_t* sample_offsets, int B, int num_slices, int H, int W, double tau, uint32_t* workspace,
                           double* out, eas_stream_t stream) {
    if (!out || !workspace || !sample_offsets || B < 1 || num_slices < 1 || H < 1 || W < 1 || nev < 0 || !(tau > 0.0)) return EAS_ERR_INVALID_ARG;
    if (nev > 0 && (!t || !x || !y || !p)) return EAS_ERR_INVALID_ARG;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int64_t total = (int64_t)B * num_slices * 2 * H * W;
    if (hipMemsetAsync(workspace, 0, (size_t)total * sizeof(uint32_t), st) != hipSuccess) return EAS_ERR_LAUNCH;
    if (nev > 0) {
        EAS_LAUNCH(time_surface_scatter_kernel, dim3(eas_grid_1d(nev)), dim3(EAS_BLOCK), 0, st, t, x, y, p, nev, sample_offsets, B,
                           num_slices, H, W, workspace);
        EAS_CHECK_LAUNCH();
    }
    EAS_LAUNCH(time_surface_exp_kernel, dim3(eas_grid_1d((int64_t)B * 2 * H * W)), dim3(EAS_BLOCK), 0, st, t, sample_offsets, B,
                       num_slices, H, W, tau, workspace, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_counts_letterbox(const int32_t* counts, const int32_t* params, int B, int F, int H, int W, int Hc, int Wc, float* out,
                         eas_stream_t stream) {
    if (!counts || !params || !out || B < 1 || F < 1 || H < 1 || W < 1 || Hc < 1 || Wc < 1) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(counts_letterbox_kernel, dim3(eas_grid_1d((int64_t)B * F * Hc * Wc)), dim3(EAS_BLOCK), 0, eas_s(stream), counts,
                       params, B, F, H, W, Hc, Wc, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
