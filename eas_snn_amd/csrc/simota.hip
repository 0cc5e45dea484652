// SimOTA label assignment for the whole batch in one launch (the `get_assignments` + `dynamic_k_matching` step of the
// reference's loss, yolox/models/spiking_yolo_head.py:401-560 via yolo_head.get_losses; no gradient flows through it).
// The PyTorch formulation materialises [B, G, A] cost / IoU tensors for all G = 50 padded label rows (5.4 M elements at
// B = 64, A = 1680, ~100 small kernels, two topk calls); here one block owns one image, keeps per-anchor state in LDS and
// loops over the image's VALID ground-truth rows only:
//   cand[a]   = anchor centre inside the 1.5-stride box around some valid gt centre
//   per gt g: iou[a] (0 unless cand), cost[a] = cls_cost + 3 * (-log(iou + 1e-8)) + 1e6 * [centre outside g's box]
//             (1e12 unless cand); dyn_k = max(1, int(sum of the 10 largest iou)); the dyn_k cheapest anchors match g
//   an anchor matched by several gts keeps only the gt of least cost; fg = matched by at least one.
// Every float operation is written in the order of the tensor expression it replaces (no fma contraction); selection
// rounds break ties towards the lower anchor / gt index.
#include "eas_common.h"

namespace {

constexpr int SB = 256;
constexpr int NWV = SB / EAS_WAVE;

struct SimGeom { int B, G, A, nc, kk; int sb, so, sc; };   // sb / so / sc: floats per anchor in the bbox / obj / cls arrays

__device__ __forceinline__ unsigned ord_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// block-wide minimum of a 64-bit key (every thread gets the result)
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, unsigned long long* sm) {
#pragma unroll
    for (int off = EAS_WAVE / 2; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), off, EAS_WAVE), hi = __shfl_xor((unsigned)(v >> 32), off, EAS_WAVE);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    if ((threadIdx.x & (EAS_WAVE - 1)) == 0) sm[threadIdx.x / EAS_WAVE] = v;
    __syncthreads();
    unsigned long long r = sm[0];
#pragma unroll
    for (int i = 1; i < NWV; ++i) r = sm[i] < r ? sm[i] : r;
    __syncthreads();
    return r;
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// binary_cross_entropy(p, t) for t in {0, 1} with ATen's clamp of the logarithms at -100
__device__ __forceinline__ float bce01(float p, bool t) {
    const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
    return t ? -lp : -lq;
}

// BIG (A > 4096: the 640x640 canvas has 8400 anchors, 25 B of state each would not fit LDS): the three per-anchor floats that are only
// touched in the per-gt fold (least cost so far, its IoU, the first match's IoU) live in the OUTPUT arrays until the last pass overwrites
// them with the results -- (best_cost, iou_first) as the two halves of the anchor's int64 `matched` slot, iou_best in `matched_iou`.  A
// thread always walks the same anchors (a = tid, tid + SB, ...), so it only ever reads back what it wrote itself.  LDS: 13 B per anchor.
template <bool BIG>
__global__ __launch_bounds__(SB) void simota_kernel(const float* __restrict__ grids, const float* __restrict__ strides,
                                                    const float* __restrict__ gt_boxes, const float* __restrict__ gt_cls,
                                                    const unsigned char* __restrict__ gt_valid, const float* __restrict__ bbox,
                                                    const float* __restrict__ obj, const float* __restrict__ cls,
                                                    unsigned char* __restrict__ fg, long long* __restrict__ matched,
                                                    float* __restrict__ matched_iou, SimGeom g) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* iou = (float*)smem;                         // [A] of the current gt
    float* cost = iou + g.A;                           // [A]
    const int b_ = blockIdx.x;
    float* pair = reinterpret_cast<float*>(matched + (size_t)b_ * g.A);
    // [A] least cost over the gts so far / iou of that gt / iou of the first matching gt: LDS, or (BIG) parked in the output arrays
    float* best_cost = BIG ? pair : cost + g.A;
    float* iou_best = BIG ? matched_iou + (size_t)b_ * g.A : best_cost + g.A;
    float* iou_first = BIG ? pair + 1 : iou_best + g.A;
    constexpr int PS = BIG ? 2 : 1;                    // element stride of best_cost / iou_first
    unsigned char* cand = BIG ? (unsigned char*)(cost + g.A) : (unsigned char*)(iou_first + g.A);   // [A]
    unsigned char* cnt = cand + g.A;                   // [A] number of matching gts (saturating)
    unsigned char* first_g = cnt + g.A;                // [A]
    unsigned char* best_g = first_g + g.A;             // [A]
    unsigned char* removed = best_g + g.A;             // [A] selection marker of the current round series
    __shared__ unsigned long long red[NWV];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* GB = gt_boxes + (size_t)b * g.G * 4;
    const float* GC = gt_cls + (size_t)b * g.G;
    const unsigned char* GV = gt_valid + (size_t)b * g.G;
    const float* BX = bbox + (size_t)b * g.A * g.sb;
    const float* OB = obj + (size_t)b * g.A * g.so;
    const float* CL = cls + (size_t)b * g.A * g.sc;

    for (int a = tid; a < g.A; a += SB) {
        const float s = strides[a];
        const float xc = (grids[2 * a] + 0.5f) * s, yc = (grids[2 * a + 1] + 0.5f) * s, dist = s * 1.5f;
        bool any = false;
        for (int q = 0; q < g.G; ++q) {
            if (!GV[q]) continue;
            const float gx = GB[4 * q], gy = GB[4 * q + 1];
            const float d0 = xc - (gx - dist), d1 = yc - (gy - dist), d2 = (gx + dist) - xc, d3 = (gy + dist) - yc;
            any |= fminf(fminf(d0, d1), fminf(d2, d3)) > 0.0f;
        }
        cand[a] = any;
        cnt[a] = 0;
        first_g[a] = 0;
        best_g[a] = 0;
        best_cost[PS * a] = 1e12f;
        iou_best[a] = 0.0f;
        iou_first[PS * a] = 0.0f;
    }
    __syncthreads();

    for (int q = 0; q < g.G; ++q) {
        if (!GV[q]) continue;                                        // block-uniform
        const float gx = GB[4 * q], gy = GB[4 * q + 1], gw = GB[4 * q + 2], gh = GB[4 * q + 3];
        int tc = (int)(long long)GC[q];
        tc = tc < 0 ? 0 : (tc > g.nc - 1 ? g.nc - 1 : tc);
        for (int a = tid; a < g.A; a += SB) {
            const float s = strides[a];
            const float xc = (grids[2 * a] + 0.5f) * s, yc = (grids[2 * a + 1] + 0.5f) * s, dist = s * 1.5f;
            const float d0 = xc - (gx - dist), d1 = yc - (gy - dist), d2 = (gx + dist) - xc, d3 = (gy + dist) - yc;
            const bool inc = fminf(fminf(d0, d1), fminf(d2, d3)) > 0.0f;
            float io = 0.0f, co = 1e12f;
            if (cand[a]) {
                const float* bp = BX + (size_t)a * g.sb;
                const float px = bp[0], py = bp[1], pw = bp[2], ph = bp[3];
                const float tlx = fmaxf(gx - gw / 2, px - pw / 2), tly = fmaxf(gy - gh / 2, py - ph / 2);
                const float brx = fminf(gx + gw / 2, px + pw / 2), bry = fminf(gy + gh / 2, py + ph / 2);
                const float en = ((tlx < brx) ? 1.0f : 0.0f) * ((tly < bry) ? 1.0f : 0.0f);
                const float area_i = ((brx - tlx) * (bry - tly)) * en;
                io = area_i / (gw * gh + pw * ph - area_i);
                const float iou_cost = -logf(io + 1e-8f);
                const float so = sigm(OB[(size_t)a * g.so]);
                float cc = 0.0f;
                for (int c = 0; c < g.nc; ++c) cc += bce01(sqrtf(sigm(CL[(size_t)a * g.sc + c]) * so), c == tc);
                co = cc + 3.0f * iou_cost + 1e6f * (inc ? 0.0f : 1.0f);
            }
            iou[a] = io;
            cost[a] = co;
            removed[a] = 0;
        }
        __syncthreads();
        // dyn_k = clamp(int(sum of the kk largest iou), min 1): kk rounds of block arg-max (ties -> lower anchor)
        float ksum = 0.0f;
        for (int rnd = 0; rnd < g.kk; ++rnd) {
            unsigned long long key = ~0ull;
            for (int a = tid; a < g.A; a += SB)
                if (!removed[a]) {
                    const unsigned long long k2 = ((unsigned long long)(~ord_bits(iou[a])) << 32) | (unsigned)a;
                    key = k2 < key ? k2 : key;
                }
            key = block_min_u64(key, red);
            const int a = (int)(key & 0xffffffffull);
            ksum += iou[a];
            if (tid == 0) removed[a] = 1;
            __syncthreads();
        }
        int dyn_k = (int)ksum;
        dyn_k = dyn_k < 1 ? 1 : dyn_k;
        for (int a = tid; a < g.A; a += SB) removed[a] = 0;
        __syncthreads();
        // the dyn_k cheapest anchors (ties -> lower anchor)
        for (int rnd = 0; rnd < g.kk && rnd < dyn_k; ++rnd) {
            unsigned long long key = ~0ull;
            for (int a = tid; a < g.A; a += SB)
                if (!removed[a]) {
                    const unsigned long long k2 = ((unsigned long long)ord_bits(cost[a]) << 32) | (unsigned)a;
                    key = k2 < key ? k2 : key;
                }
            key = block_min_u64(key, red);
            if (tid == 0) removed[(int)(key & 0xffffffffull)] = 1;
            __syncthreads();
        }
        // fold this gt into the per-anchor state
        for (int a = tid; a < g.A; a += SB) {
            if (removed[a] && cand[a]) {
                if (cnt[a] == 0) { first_g[a] = (unsigned char)q; iou_first[PS * a] = iou[a]; }
                if (cnt[a] < 255) cnt[a] = cnt[a] + 1;
            }
            // argmin over ALL label rows: rows before the first valid one hold 1e12 and win ties by their lower index
            if (cost[a] < best_cost[PS * a] && !(cost[a] >= 1e12f)) { best_cost[PS * a] = cost[a]; best_g[a] = (unsigned char)q; iou_best[a] = iou[a]; }
        }
        __syncthreads();
    }
    for (int a = tid; a < g.A; a += SB) {
        const int c = cnt[a];
        int m = 0;
        float mi = 0.0f;
        if (c == 1) { m = first_g[a]; mi = iou_first[PS * a]; }
        else if (c > 1) { m = best_g[a]; mi = iou_best[a]; }
        fg[(size_t)b * g.A + a] = c > 0;
        if constexpr (BIG) {
            const long long m64 = m;                   // the slot held (best_cost, iou_first) as floats until now: a byte copy, so that the
            __builtin_memcpy(pair + 2 * a, &m64, 8);   // compiler orders it behind the float reads above whatever the types say
        } else {
            matched[(size_t)b * g.A + a] = m;
        }
        matched_iou[(size_t)b * g.A + a] = mi;
    }
}

}  // namespace

extern "C" {

// grids [A][2], strides [A], gt_boxes [B][G][4] (cx, cy, w, h), gt_cls [B][G] (float class ids), gt_valid [B][G] (0/1),
// bbox [B][A][4] decoded (cx, cy, w, h), obj [B][A], cls [B][A][nc] raw logits.
// Out: fg [B][A] (0/1), matched [B][A] int64 (gt row, 0 where none), matched_iou [B][A].  G <= 255, A <= 12288 (above 4096 anchors part of the per-anchor state is parked in the output arrays).
static int simota_launch(const float* grids, const float* strides, const float* gt_boxes, const float* gt_cls, const unsigned char* gt_valid,
                         const float* bbox, const float* obj, const float* cls, int sb, int so, int sc, int B, int G, int A, int nc,
                         unsigned char* fg, long long* matched, float* matched_iou, eas_stream_t stream) {
    if (!grids || !strides || !gt_boxes || !gt_cls || !gt_valid || !bbox || !obj || !cls || !fg || !matched || !matched_iou) return EAS_ERR_INVALID_ARG;
    if (B < 1 || G < 1 || A < 1 || nc < 1) return EAS_ERR_INVALID_ARG;
    if (G > 255 || A > 12288) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    SimGeom g{B, G, A, nc, A < 10 ? A : 10, sb, so, sc};
    const bool big = A > 4096;                          // (the 640x640 canvas: 8400 anchors)
    const size_t lds = (size_t)A * (big ? 2 * 4 + 5 : 5 * 4 + 5);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)simota_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 25) != hipSuccess) return EAS_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)simota_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 13) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    if (big) EAS_LAUNCH(simota_kernel<true>, dim3(B), dim3(SB), lds, eas_s(stream), grids, strides, gt_boxes, gt_cls, gt_valid, bbox, obj, cls, fg,
                        matched, matched_iou, g);
    else EAS_LAUNCH(simota_kernel<false>, dim3(B), dim3(SB), lds, eas_s(stream), grids, strides, gt_boxes, gt_cls, gt_valid, bbox, obj, cls, fg,
                    matched, matched_iou, g);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_simota_assign(const float* grids, const float* strides, const float* gt_boxes, const float* gt_cls,
                      const unsigned char* gt_valid, const float* bbox, const float* obj, const float* cls, int B, int G, int A,
                      int nc, unsigned char* fg, long long* matched, float* matched_iou, eas_stream_t stream) {
    return simota_launch(grids, strides, gt_boxes, gt_cls, gt_valid, bbox, obj, cls, 4, 1, nc, B, G, A, nc, fg, matched, matched_iou, stream);
}

// the same on decoded rows [B][A][5+nc] (cx, cy, w, h, obj logit, class logits) as eas_det_decode writes them
int eas_simota_assign_rows(const float* grids, const float* strides, const float* gt_boxes, const float* gt_cls,
                           const unsigned char* gt_valid, const float* dec, int B, int G, int A, int nc, unsigned char* fg,
                           long long* matched, float* matched_iou, eas_stream_t stream) {
    if (!dec) return EAS_ERR_INVALID_ARG;
    const int row = 5 + nc;
    return simota_launch(grids, strides, gt_boxes, gt_cls, gt_valid, dec, dec + 4, dec + 5, row, row, row, B, G, A, nc, fg, matched, matched_iou,
                         stream);
}

}  // extern "C"
