// Spike-count / synaptic-operation statistics of one convolution's input (replaces the
// RecordHook -> CPU clone -> "convolution with all-ones weights" of energy_estimation, yolox/evaluators/event_evaluator.py:
// 473-487 calc_layer_sop and :519-541).  For x [NI][Cin][H][W] and a k x k convolution (padding (k-1)/2, stride s, Cout
// output channels, groups = 1):
//     out[0] = sum x                                   (spike count; firing rate = out[0] / numel)
//     out[1] = sum_{n,ci,h,w} x * Cout * fan(h, w)     (= conv(x, ones).sum(): accumulate operations, "SOP")
// fan(h, w) = fan_h(h) * fan_w(w) = number of output positions that read input pixel (h, w).  One pass over x at the HBM
// roofline (4 B per element), wavefront + block reductions, per-block partials reduced in fixed order by a second kernel
// (deterministic; exact for spike data since every term is a small integer).
#include "eas_common.h"

namespace {

constexpr int NW = EAS_BLOCK / EAS_WAVE;
constexpr int kMaxDim = 2048;
constexpr int kMaxBlocks = 2048;

__device__ __forceinline__ int fan_1d(int i, int n_in, int n_out, int k, int s, int pad) {
    int c = 0;
    for (int t = 0; t < k; ++t) {
        const int num = i + pad - t;
        if (num >= 0 && num % s == 0 && num / s < n_out) ++c;
    }
    return c;
}

__global__ __launch_bounds__(EAS_BLOCK) void spike_sop_partial(const float* __restrict__ x, int64_t planes, int H, int W, int k, int s,
                                                               double* __restrict__ part) {
    __shared__ float fh[kMaxDim], fw[kMaxDim];
    __shared__ double red[NW];
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    for (int i = threadIdx.x; i < H; i += blockDim.x) fh[i] = (float)fan_1d(i, H, Ho, k, s, pad);
    for (int i = threadIdx.x; i < W; i += blockDim.x) fw[i] = (float)fan_1d(i, W, Wo, k, s, pad);
    __syncthreads();
    const int64_t HW = (int64_t)H * W, total = planes * HW;
    double cnt = 0.0, sop = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((W & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int w4 = W / 4;
        const int64_t groups = total / 4;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
            const float4 v = reinterpret_cast<const float4*>(x)[g];
            const int64_t row = g / w4;
            const int wq = (int)(g - row * w4) * 4;
            const float f = fh[(int)(row % H)];
            cnt += (double)((v.x + v.y) + (v.z + v.w));
            sop += (double)(f * ((v.x * fw[wq] + v.y * fw[wq + 1]) + (v.z * fw[wq + 2] + v.w * fw[wq + 3])));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const float v = x[i];
            const int64_t row = i / W;
            cnt += (double)v;
            sop += (double)(v * fh[(int)(row % H)] * fw[(int)(i - row * W)]);
        }
    }
    const double c = eas_block_sum<double, NW>(cnt, red);
    const double o = eas_block_sum<double, NW>(sop, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = c;
        part[2 * blockIdx.x + 1] = o;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void spike_sop_finalize(const double* __restrict__ part, int nblocks, double cout, double* __restrict__ out) {
    __shared__ double red[NW];
    double c = 0.0, o = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        c += part[2 * i];
        o += part[2 * i + 1];
    }
    const double tc = eas_block_sum<double, NW>(c, red);
    const double to = eas_block_sum<double, NW>(o, red);
    if (threadIdx.x == 0) {
        out[0] = tc;
        out[1] = to * cout;
    }
}

}  // namespace

extern "C" {

int64_t eas_spike_sop_workspace_doubles(void) { return 2 * kMaxBlocks; }

int eas_spike_sop(const float* x, int64_t NI, int Cin, int H, int W, int ksize, int stride, int Cout, double* out, double* workspace,
                  eas_stream_t stream) {
    if (!x || !out || !workspace || NI < 1 || Cin < 1 || H < 1 || W < 1 || ksize < 1 || stride < 1 || Cout < 1) return EAS_ERR_INVALID_ARG;
    if (!(ksize & 1) || H > kMaxDim || W > kMaxDim) return EAS_ERR_UNSUPPORTED;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
    const int64_t planes = NI * Cin, total = planes * H * W;
    const int blocks = eas_grid_1d((total + 3) / 4, EAS_BLOCK, kMaxBlocks);
    EAS_LAUNCH(spike_sop_partial, dim3(blocks), dim3(EAS_BLOCK), 0, st, x, planes, H, W, ksize, stride, workspace);
    EAS_CHECK_LAUNCH();
    EAS_LAUNCH(spike_sop_finalize, dim3(1), dim3(EAS_BLOCK), 0, st, workspace, blocks, (double)Cout, out);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
