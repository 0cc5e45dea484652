// Detection loss of (S)YOLOX after the label assignment, forward and gradient in one pass (the loss terms of
// YOLOXHead.get_losses, yolox/models/yolo_head.py:296-420 as inherited by spiking_yolo_head.py): IoU loss (1 - iou^2) and
// L1 loss on the boxes, BCE-with-logits on objectness and classes, all divided by the number of foreground anchors.
// PyTorch runs ~35 small kernels forward and ~45 backward on [B*A, .] tensors for this; here
//   eas_det_decode : raw head maps [B][C][H][W] per level -> decoded rows [B][A][5+nc] (cx, cy, w, h, obj, cls logits)
//                    (the cat / permute / (o + grid) * stride / exp(o) * stride chain, :160-175)
//   eas_det_loss   : one thread per anchor computes its loss terms and d(loss * num_fg)/d(raw outputs), written straight
//                    into gradient maps in the raw [B][C][H][W] layout; block partial sums -> fixed-order finalize
//                    (deterministic) -> the six numbers get_losses returns.  The caller scales the gradient maps by
//                    grad_total / num_fg in backward (num_fg is only known after the pass).
// Float expressions follow the tensor code term by term (-ffp-contract=off); torch.max / torch.min ties split the
// gradient in halves like ATen's maximum / minimum backward.
#include "eas_common.h"

namespace {

constexpr int kMaxLevels = 4;
constexpr int NWB = EAS_BLOCK / EAS_WAVE;

struct DetLevels {
    const float* reg[kMaxLevels];
    const float* obj[kMaxLevels];
    const float* cls[kMaxLevels];
    float* g_reg[kMaxLevels];
    float* g_obj[kMaxLevels];
    float* g_cls[kMaxLevels];
    int H[kMaxLevels], W[kMaxLevels], a0[kMaxLevels + 1];
    float stride[kMaxLevels];
    int L, A, B, nc;
};

__device__ __forceinline__ int level_of(const DetLevels& d, int a) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < d.L && a >= d.a0[i]) l = i;
    return l;
}

__device__ __forceinline__ float sigmoidf_(float x);

// SIG: objectness and class columns as probabilities (the inference output, eas_det_decode_eval); else the logits (the loss reads those)
template <bool SIG>
__global__ __launch_bounds__(EAS_BLOCK) void det_decode_kernel(DetLevels d, float* __restrict__ dec) {
    const int64_t total = (int64_t)d.B * d.A;
    const int row = 5 + d.nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / d.A), a = (int)(i - (int64_t)b * d.A);
        const int l = level_of(d, a), pix = a - d.a0[l], HW = d.H[l] * d.W[l];
        const float s = d.stride[l];
        const float gx = (float)(pix % d.W[l]), gy = (float)(pix / d.W[l]);
        const float* r = d.reg[l] + (size_t)b * 4 * HW + pix;
        float* o = dec + (size_t)i * row;
        o[0] = (r[0] + gx) * s;
        o[1] = (r[HW] + gy) * s;
        o[2] = expf(r[2 * (size_t)HW]) * s;
        o[3] = expf(r[3 * (size_t)HW]) * s;
        const float ob = d.obj[l][(size_t)b * HW + pix];
        o[4] = SIG ? sigmoidf_(ob) : ob;
        const float* c = d.cls[l] + (size_t)b * d.nc * HW + pix;
        for (int k = 0; k < d.nc; ++k) o[5 + k] = SIG ? sigmoidf_(c[(size_t)k * HW]) : c[(size_t)k * HW];
    }
}

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// d max(p, t)/dp and d min(p, t)/dp as ATen's maximum / minimum backward (ties: one half)
__device__ __forceinline__ float dmax(float p, float t) { return p > t ? 1.0f : (p == t ? 0.5f : 0.0f); }
__device__ __forceinline__ float dmin(float p, float t) { return p < t ? 1.0f : (p == t ? 0.5f : 0.0f); }

__global__ __launch_bounds__(EAS_BLOCK) void det_loss_kernel(DetLevels d, const float* __restrict__ dec, const float* __restrict__ gt_boxes,
                                                             const float* __restrict__ gt_cls, int G, const unsigned char* __restrict__ fg,
                                                             const long long* __restrict__ matched, const float* __restrict__ miou,
                                                             int use_l1, double* __restrict__ part) {
    __shared__ double red[NWB];
    const int64_t total = (int64_t)d.B * d.A;
    const int row = 5 + d.nc;
    float s_iou = 0.f, s_obj = 0.f, s_cls = 0.f, s_l1 = 0.f, s_fg = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / d.A), a = (int)(i - (int64_t)b * d.A);
        const int l = level_of(d, a), pix = a - d.a0[l], HW = d.H[l] * d.W[l];
        const float s = d.stride[l];
        const float* o = dec + (size_t)i * row;
        const bool f = fg[i] != 0;
        const float ff = f ? 1.0f : 0.0f;
        float* gr = d.g_reg[l] + (size_t)b * 4 * HW + pix;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        // objectness: BCE with logits against the foreground flag, every anchor
        {
            const float x = o[4];
            s_obj += (1.0f - ff) * x - log_sigmoid(x);
            d.g_obj[l][(size_t)b * HW + pix] = sigmoidf_(x) - ff;
        }
        float* gc = d.g_cls[l] + (size_t)b * d.nc * HW + pix;
        if (f) {
            const int m = (int)matched[i];
            const float* gb = gt_boxes + ((size_t)b * G + m) * 4;
            const float tx = gb[0], ty = gb[1], tw = gb[2], th = gb[3];
            const float px = o[0], py = o[1], pw = o[2], ph = o[3];
            // IoU loss 1 - iou^2 (losses.py:19-33)
            const float plx = px - pw / 2, ply = py - ph / 2, phx = px + pw / 2, phy = py + ph / 2;
            const float tlx_ = tx - tw / 2, tly_ = ty - th / 2, thx = tx + tw / 2, thy = ty + th / 2;
            const float tlx = fmaxf(plx, tlx_), tly = fmaxf(ply, tly_), brx = fminf(phx, thx), bry = fminf(phy, thy);
            const float area_p = pw * ph, area_g = tw * th;
            const float en = ((tlx < brx) ? 1.0f : 0.0f) * ((tly < bry) ? 1.0f : 0.0f);
            const float dx = brx - tlx, dy = bry - tly;
            const float area_i = (dx * dy) * en;
            const float area_u = area_p + area_g - area_i;
            const float den = area_u + 1e-16f;
            const float iou = area_i / den;
            s_iou += 1.0f - iou * iou;
            // d(1 - iou^2): through iou = I / den, den = P + Gt - I + eps
            const float dL_diou = -2.0f * iou;
            const float diou_dI = 1.0f / den + area_i / (den * den);       // dI enters the numerator and (with a minus sign) den
            const float diou_dP = -area_i / (den * den);
            const float dI_dtlx = -dy * en, dI_dbrx = dy * en, dI_dtly = -dx * en, dI_dbry = dx * en;
            const float dI_dplx = dI_dtlx * dmax(plx, tlx_), dI_dply = dI_dtly * dmax(ply, tly_);
            const float dI_dphx = dI_dbrx * dmin(phx, thx), dI_dphy = dI_dbry * dmin(phy, thy);
            const float dI_dpx = dI_dplx + dI_dphx, dI_dpy = dI_dply + dI_dphy;
            const float dI_dpw = (dI_dphx - dI_dplx) * 0.5f, dI_dph = (dI_dphy - dI_dply) * 0.5f;
            const float w5 = 5.0f * dL_diou;                                // reg_weight = 5
            const float d_px = w5 * (diou_dI * dI_dpx), d_py = w5 * (diou_dI * dI_dpy);
            const float d_pw = w5 * (diou_dI * dI_dpw + diou_dP * ph), d_ph = w5 * (diou_dI * dI_dph + diou_dP * pw);
            // decode: px = (o0 + gx) * s, pw = exp(o2) * s
            g0 = d_px * s; g1 = d_py * s; g2 = d_pw * pw; g3 = d_ph * ph;
            // classes: BCE with logits against onehot(class) * matched_iou
            int tc = (int)(long long)gt_cls[(size_t)b * G + m];
            tc = tc < 0 ? 0 : (tc > d.nc - 1 ? d.nc - 1 : tc);
            const float mi = miou[i];
            for (int k = 0; k < d.nc; ++k) {
                const float x = o[5 + k], t = (k == tc ? 1.0f : 0.0f) * mi;
                s_cls += (1.0f - t) * x - log_sigmoid(x);
                gc[(size_t)k * HW] = sigmoidf_(x) - t;
            }
            if (use_l1) {
                const float gx = (float)(pix % d.W[l]), gy = (float)(pix / d.W[l]);
                const float t0 = tx / s - gx, t1 = ty / s - gy, t2 = logf(tw / s + 1e-8f), t3 = logf(th / s + 1e-8f);
                const float* r = d.reg[l] + (size_t)b * 4 * HW + pix;
                const float e0 = r[0] - t0, e1 = r[HW] - t1, e2 = r[2 * (size_t)HW] - t2, e3 = r[3 * (size_t)HW] - t3;
                s_l1 += ((fabsf(e0) + fabsf(e1)) + fabsf(e2)) + fabsf(e3);
                g0 += (e0 > 0.f) - (e0 < 0.f); g1 += (e1 > 0.f) - (e1 < 0.f); g2 += (e2 > 0.f) - (e2 < 0.f); g3 += (e3 > 0.f) - (e3 < 0.f);
            }
            s_fg += 1.0f;
        } else {
            for (int k = 0; k < d.nc; ++k) gc[(size_t)k * HW] = 0.0f;
        }
        gr[0] = g0; gr[HW] = g1; gr[2 * (size_t)HW] = g2; gr[3 * (size_t)HW] = g3;
    }
    const double v[5] = {(double)s_iou, (double)s_obj, (double)s_cls, (double)s_l1, (double)s_fg};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double t = eas_block_sum<double, NWB>(v[k], red);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * 5 + k] = t;
    }
}

// out[0..5] = total, 5 * iou, obj, cls, l1, num_fg / num_gts ; out[6] = 1 / num_fg (the backward scale)
__global__ __launch_bounds__(EAS_BLOCK) void det_loss_finalize(const double* __restrict__ part, int nblocks, const float* __restrict__ num_gts,
                                                               int use_l1, float* __restrict__ out) {
    __shared__ double red[NWB];
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += part[(size_t)i * 5 + k];
    double t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = eas_block_sum<double, NWB>(v[k], red);
    if (threadIdx.x == 0) {
        const float num_fg = fmaxf((float)t[4], 1.0f);
        const float l_iou = (float)t[0] / num_fg, l_obj = (float)t[1] / num_fg, l_cls = (float)t[2] / num_fg;
        const float l_l1 = use_l1 ? (float)t[3] / num_fg : 0.0f;
        out[0] = ((5.0f * l_iou + l_obj) + l_cls) + l_l1;
        out[1] = 5.0f * l_iou;
        out[2] = l_obj;
        out[3] = l_cls;
        out[4] = l_l1;
        out[5] = num_fg / fmaxf(*num_gts, 1.0f);
        out[6] = 1.0f / num_fg;
    }
}

// labels [B][G][5] (class, cx, cy, w, h; zero rows = padding) -> what the assignment and the loss read: gt_valid [B][G] (row g < number of
// rows whose five values sum to > 0, the reference's `nlabel = (labels.sum(dim=2) > 0).sum(dim=1)`, yolo_head.py:277-278), gt_cls [B][G],
// gt_boxes [B][G][4], num_gts = sum of nlabel.  One block: ten tiny tensor operators of the step's dependency chain as one launch.
__global__ __launch_bounds__(EAS_BLOCK) void det_labels_kernel(const float* __restrict__ labels, int B, int G, unsigned char* __restrict__ gt_valid,
                                                               float* __restrict__ gt_cls, float* __restrict__ gt_boxes, float* __restrict__ num_gts) {
    __shared__ int nl[1024];
    __shared__ int total;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        int n = 0;
        for (int q = 0; q < G; ++q) {
            const float* r = labels + ((size_t)b * G + q) * 5;
            n += ((((r[0] + r[1]) + r[2]) + r[3]) + r[4]) > 0.0f ? 1 : 0;       // (label values are non-negative: the sign test does not depend on the summation order)
        }
        nl[b] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int b = 0; b < B; ++b) t += nl[b];
        total = t;
    }
    for (int i = threadIdx.x; i < B * G; i += blockDim.x) {
        const int b = i / G, q = i - b * G;
        const float* r = labels + (size_t)i * 5;
        gt_valid[i] = q < nl[b];
        gt_cls[i] = r[0];
        gt_boxes[(size_t)i * 4 + 0] = r[1]; gt_boxes[(size_t)i * 4 + 1] = r[2];
        gt_boxes[(size_t)i * 4 + 2] = r[3]; gt_boxes[(size_t)i * 4 + 3] = r[4];
    }
    __syncthreads();
    if (threadIdx.x == 0) *num_gts = (float)total;
}

int fill_levels(DetLevels& d, int L, const float* const* reg, const float* const* obj, const float* const* cls, float* const* g_reg,
                float* const* g_obj, float* const* g_cls, const int* hw, const float* strides, int B, int nc) {
    if (L < 1 || L > kMaxLevels || B < 1 || nc < 1 || !reg || !obj || !cls || !hw || !strides) return EAS_ERR_INVALID_ARG;
    d = DetLevels{};
    d.L = L; d.B = B; d.nc = nc;
    int a = 0;
    for (int l = 0; l < L; ++l) {
        if (!reg[l] || !obj[l] || !cls[l] || hw[2 * l] < 1 || hw[2 * l + 1] < 1) return EAS_ERR_INVALID_ARG;
        d.reg[l] = reg[l]; d.obj[l] = obj[l]; d.cls[l] = cls[l];
        d.g_reg[l] = g_reg ? g_reg[l] : nullptr; d.g_obj[l] = g_obj ? g_obj[l] : nullptr; d.g_cls[l] = g_cls ? g_cls[l] : nullptr;
        d.H[l] = hw[2 * l]; d.W[l] = hw[2 * l + 1]; d.stride[l] = strides[l];
        d.a0[l] = a;
        a += d.H[l] * d.W[l];
    }
    for (int l = L; l <= kMaxLevels; ++l) d.a0[l] = a;
    d.A = a;
    return EAS_OK;
}

}  // namespace

extern "C" {

// reg / obj / cls: host arrays of L device pointers to the raw head maps [B][4|1|nc][H_l][W_l]; hw: host array [L][2];
// strides: host array [L].  dec: [B][A][5+nc], A = sum H_l * W_l.
int eas_det_decode(int L, const float* const* reg, const float* const* obj, const float* const* cls, const int* hw, const float* strides, int B,
                   int nc, float* dec, eas_stream_t stream) {
    DetLevels d;
    if (int rc = fill_levels(d, L, reg, obj, cls, nullptr, nullptr, nullptr, hw, strides, B, nc)) return rc;
    if (!dec) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(det_decode_kernel<false>, dim3(eas_grid_1d((int64_t)B * d.A)), dim3(EAS_BLOCK), 0, eas_s(stream), d, dec);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// The inference output of YOLOXHead (yolo_head.py:187-199 + decode_outputs :201-216 of the reference): per level cat[reg, sigmoid(obj),
// sigmoid(cls)], the levels concatenated along the anchors, boxes decoded with grid and stride -- ~25 tensor operators as ONE launch.
int eas_det_decode_eval(int L, const float* const* reg, const float* const* obj, const float* const* cls, const int* hw, const float* strides,
                        int B, int nc, float* dec, eas_stream_t stream) {
    DetLevels d;
    if (int rc = fill_levels(d, L, reg, obj, cls, nullptr, nullptr, nullptr, hw, strides, B, nc)) return rc;
    if (!dec) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(det_decode_kernel<true>, dim3(eas_grid_1d((int64_t)B * d.A)), dim3(EAS_BLOCK), 0, eas_s(stream), d, dec);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int64_t eas_det_loss_workspace_doubles(void) { return 5 * 1024; }

// fg / matched / matched_iou: the assignment (eas_simota_assign); g_reg / g_obj / g_cls: host arrays of L device pointers to the
// gradient maps (same shapes as the raw maps; written for every element); num_gts: device scalar (float).
// out: 7 floats = total, 5 * iou, obj, cls, l1, num_fg / num_gts, 1 / num_fg.  Gradient maps hold d(total * num_fg)/d(raw).
int eas_det_loss(int L, const float* const* reg, const float* const* obj, const float* const* cls, float* const* g_reg, float* const* g_obj,
                 float* const* g_cls, const int* hw, const float* strides, int B, int nc, const float* dec, const float* gt_boxes,
                 const float* gt_cls, int G, const unsigned char* fg, const long long* matched, const float* matched_iou,
                 const float* num_gts, int use_l1, float* out, double* workspace, eas_stream_t stream) {
    DetLevels d;
    if (int rc = fill_levels(d, L, reg, obj, cls, g_reg, g_obj, g_cls, hw, strides, B, nc)) return rc;
    if (!g_reg || !g_obj || !g_cls || !dec || !gt_boxes || !gt_cls || !fg || !matched || !matched_iou || !num_gts || !out || !workspace || G < 1)
        return EAS_ERR_INVALID_ARG;
    for (int l = 0; l < L; ++l)
        if (!d.g_reg[l] || !d.g_obj[l] || !d.g_cls[l]) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    const int blocks = eas_grid_1d((int64_t)B * d.A, EAS_BLOCK, 1024);
    EAS_LAUNCH(det_loss_kernel, dim3(blocks), dim3(EAS_BLOCK), 0, eas_s(stream), d, dec, gt_boxes, gt_cls, G, fg, matched, matched_iou,
                       use_l1, workspace);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(det_loss_finalize, dim3(1), dim3(EAS_BLOCK), 0, eas_s(stream), workspace, blocks, num_gts, use_l1, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// labels [B][G][5] fp32 -> gt_valid [B][G] u8, gt_cls [B][G], gt_boxes [B][G][4], num_gts (device float): the label preparation of
// YOLOXHead.get_losses (yolo_head.py:272-277 of the mirror; reference yolo_head.py:277-300) as one launch.  B <= 1024.
int eas_det_labels(const float* labels, int B, int G, unsigned char* gt_valid, float* gt_cls, float* gt_boxes, float* num_gts, eas_stream_t stream) {
    if (!labels || !gt_valid || !gt_cls || !gt_boxes || !num_gts || B < 1 || G < 1) return EAS_ERR_INVALID_ARG;
    if (B > 1024) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(det_labels_kernel, dim3(1), dim3(EAS_BLOCK), 0, eas_s(stream), labels, B, G, gt_valid, gt_cls, gt_boxes, num_gts);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
