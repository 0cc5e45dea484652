// Detection post-processing on the device: the reference's ``postprocess``
// (yolox/utils/boxes.py:33-77) = centre/size -> corners, best class, confidence mask, class-aware NMS through
// torchvision.ops.batched_nms.  torchvision (0.16.1 pinned by pip-requirements.txt:147) is not part of the reference tree
// and not installed here, so its published algorithm is restated (parity unpinned):
//   nms: candidates in stable descending-score order; candidate i is kept unless an earlier kept candidate j has
//        inter / (area_i + area_j - inter) > threshold (float32, nms_kernel.cu devIoU);
//   batched_nms on a CUDA device: up to 5000 candidates (numel <= 20000) boxes are shifted by class * (max coordinate + 1)
//        and one NMS runs over all of them ("coordinate trick"); above that NMS runs per class on the unshifted boxes.
// Three kernels per call, all images at once, no host synchronisation:
//   1. one block per image: scores / classes / corners, 64-bit keys (score descending, anchor ascending = the stable order),
//      bitonic sort in LDS, sorted candidate rows + NMS boxes to the workspace;
//   2. suppression bit matrix: 64 x 64 candidate tiles, one 64-bit word per (row, column block);
//   3. one wavefront per image walks the sorted candidates, OR-ing the rows of kept candidates into a removed-set held in
//      registers, and writes the kept rows in order.
#include "eas_common.h"

namespace {

constexpr int kSortThreads = 1024;
constexpr int kMaxCand = 16384;
constexpr int kTrickMax = 5000;    // 4 * K <= 20000: torchvision's CUDA threshold for the coordinate trick

struct PpWs {
    float* det;        // [B][A][7] sorted candidate rows
    float* nbox;       // [B][A][4] boxes the NMS sees
    int* count;        // [B] candidates
    unsigned long long* mask;   // [B][A][words]
};

__device__ __forceinline__ unsigned order_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kSortThreads) void pp_sort_kernel(const float* __restrict__ pred, int A, int ncls, float conf_thre, int NP,
                                                               int class_agnostic, PpWs ws) {
    extern __shared__ unsigned long long keys[];
    __shared__ float red[kSortThreads / EAS_WAVE];
    __shared__ int cnt_sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int row = 5 + ncls;
    const float* P = pred + (size_t)b * A * row;
    float mx = -INFINITY;
    int local = 0;
    for (int a = tid; a < NP; a += blockDim.x) {
        unsigned long long key = ~0ull;
        if (a < A) {
            const float* r = P + (size_t)a * row;
            float best = r[5];
            for (int c = 1; c < ncls; ++c) best = r[5 + c] > best ? r[5 + c] : best;     // first maximum wins
            const float score = r[4] * best;
            if (score >= conf_thre) {
                key = ((unsigned long long)(~order_bits(score)) << 32) | (unsigned)a;
                const float x1 = r[0] - r[2] / 2, y1 = r[1] - r[3] / 2, x2 = r[0] + r[2] / 2, y2 = r[1] + r[3] / 2;
                mx = fmaxf(mx, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
                ++local;
            }
        }
        keys[a] = key;
    }
    // block maximum of the candidate coordinates and candidate count
    for (int off = EAS_WAVE / 2; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, EAS_WAVE));
    if ((tid & (EAS_WAVE - 1)) == 0) red[tid / EAS_WAVE] = mx;
    if (tid == 0) cnt_sh = 0;
    __syncthreads();
    if (local) atomicAdd(&cnt_sh, local);
    if (tid == 0) {
        float m = red[0];
        for (int i = 1; i < (int)(blockDim.x / EAS_WAVE); ++i) m = fmaxf(m, red[i]);
        red[0] = m;
    }
    __syncthreads();
    const float max_coord = red[0];
    const int K = cnt_sh;
    // bitonic sort, ascending keys = descending score, ascending anchor among equal scores
    for (int k = 2; k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = keys[i], y = keys[p];
                    if ((x > y) == ((i & k) == 0)) { keys[i] = y; keys[p] = x; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) ws.count[b] = K;
    const bool trick = class_agnostic || K <= kTrickMax;
    for (int r = tid; r < K; r += blockDim.x) {
        const int a = (int)(keys[r] & 0xffffffffull);
        const float* s = P + (size_t)a * row;
        float best = s[5];
        int arg = 0;
        for (int c = 1; c < ncls; ++c)
            if (s[5 + c] > best) { best = s[5 + c]; arg = c; }
        const float x1 = s[0] - s[2] / 2, y1 = s[1] - s[3] / 2, x2 = s[0] + s[2] / 2, y2 = s[1] + s[3] / 2;
        float* d = ws.det + ((size_t)b * A + r) * 7;
        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = s[4]; d[5] = best; d[6] = (float)arg;
        const float off = (trick && !class_agnostic) ? (float)arg * (max_coord + 1.0f) : 0.0f;
        float* nb = ws.nbox + ((size_t)b * A + r) * 4;
        nb[0] = x1 + off; nb[1] = y1 + off; nb[2] = x2 + off; nb[3] = y2 + off;
    }
}

__device__ __forceinline__ bool iou_over(const float* a, const float* b, float thr) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return (inter / (Sa + Sb - inter)) > thr;
}

__global__ __launch_bounds__(EAS_WAVE) void pp_mask_kernel(int A, int words, float nms_thre, int class_agnostic, PpWs ws) {
    const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
    const int K = ws.count[b];
    if (rb * 64 >= K || cb * 64 >= K || cb < rb) return;
    __shared__ float cbox[64][4];
    __shared__ float ccls[64];
    const int tid = threadIdx.x;
    const int j0 = cb * 64;
    if (j0 + tid < K) {
        const float* nb = ws.nbox + ((size_t)b * A + j0 + tid) * 4;
        cbox[tid][0] = nb[0]; cbox[tid][1] = nb[1]; cbox[tid][2] = nb[2]; cbox[tid][3] = nb[3];
        ccls[tid] = ws.det[((size_t)b * A + j0 + tid) * 7 + 6];
    }
    __syncthreads();
    const int i = rb * 64 + tid;
    if (i >= K) return;
    const bool per_class = !class_agnostic && K > kTrickMax;        // unshifted boxes: only the same class suppresses
    float mine[4];
    const float* nb = ws.nbox + ((size_t)b * A + i) * 4;
    mine[0] = nb[0]; mine[1] = nb[1]; mine[2] = nb[2]; mine[3] = nb[3];
    const float mycls = ws.det[((size_t)b * A + i) * 7 + 6];
    unsigned long long bits = 0;
    const int n = K - j0 < 64 ? K - j0 : 64;
    for (int j = (rb == cb ? tid + 1 : 0); j < n; ++j)
        if ((!per_class || ccls[j] == mycls) && iou_over(mine, cbox[j], nms_thre)) bits |= 1ull << j;
    ws.mask[((size_t)b * A + i) * words + cb] = bits;
}

__global__ __launch_bounds__(EAS_WAVE) void pp_scan_kernel(int A, int words, PpWs ws, float* __restrict__ out, int* __restrict__ out_count) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = ws.count[b];
    constexpr int SL = kMaxCand / 64 / 64;     // removed-set words per lane
    unsigned long long remv[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) remv[s] = 0;
    int kept = 0;
    for (int i = 0; i < K; ++i) {
        const int w = i >> 6, owner = w & 63, slot = w >> 6;
        unsigned long long word = 0;
#pragma unroll
        for (int s = 0; s < SL; ++s) word = s == slot ? remv[s] : word;
        const unsigned lo = __shfl((unsigned)(word & 0xffffffffull), owner, EAS_WAVE), hi = __shfl((unsigned)(word >> 32), owner, EAS_WAVE);
        const unsigned long long ow = ((unsigned long long)hi << 32) | lo;
        if ((ow >> (i & 63)) & 1ull) continue;
        if (lane < 7) out[((size_t)b * A + kept) * 7 + lane] = ws.det[((size_t)b * A + i) * 7 + lane];
        ++kept;
        const unsigned long long* mrow = ws.mask + ((size_t)b * A + i) * words;
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int ww = lane + 64 * s;
            if (ww >= w && ww < words && ww * 64 < K) remv[s] |= mrow[ww];
        }
    }
    if (lane == 0) out_count[b] = kept;
}

int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int64_t eas_postprocess_workspace_bytes(int B, int A) {
    if (B < 1 || A < 1) return 0;
    const size_t words = (A + 63) / 64;
    return (int64_t)(align256((size_t)B * A * 7 * 4) + align256((size_t)B * A * 4 * 4) + align256((size_t)B * 4) + align256((size_t)B * A * words * 8));
}

// pred [B][A][5+ncls] decoded head output (cx, cy, w, h, obj, class scores); out [B][A][7] rows
// (x1, y1, x2, y2, obj_conf, class_conf, class_pred) of the kept detections in descending score order, out_count [B].
int eas_postprocess(const float* pred, int B, int A, int ncls, float conf_thre, float nms_thre, int class_agnostic, float* out,
                    int* out_count, void* workspace, eas_stream_t stream) {
    if (!pred || !out || !out_count || !workspace || B < 1 || A < 1 || ncls < 1) return EAS_ERR_INVALID_ARG;
    if (A > kMaxCand) return EAS_ERR_UNSUPPORTED;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int words = (A + 63) / 64;
    PpWs ws;
    char* p = (char*)workspace;
    ws.det = (float*)p; p += align256((size_t)B * A * 7 * 4);
    ws.nbox = (float*)p; p += align256((size_t)B * A * 4 * 4);
    ws.count = (int*)p; p += align256((size_t)B * 4);
    ws.mask = (unsigned long long*)p;
    const int NP = next_pow2(A);
    const size_t lds = (size_t)NP * 8;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)pp_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxCand * 8) != hipSuccess) return EAS_ERR_LAUNCH;
        attr_set = true;
    }
    EAS_LAUNCH(pp_sort_kernel, dim3(B), dim3(kSortThreads), lds, st, pred, A, ncls, conf_thre, NP, class_agnostic, ws);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(pp_mask_kernel, dim3(words, words, B), dim3(EAS_WAVE), 0, st, A, words, nms_thre, class_agnostic, ws);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(pp_scan_kernel, dim3(B), dim3(EAS_WAVE), 0, st, A, words, ws, out, out_count);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
