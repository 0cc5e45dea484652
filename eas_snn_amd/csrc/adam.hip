// Adam step of ALL parameter groups as one launch (eas_adam_step) + one launch that advances the step counters.
//
// The reference trains with torch.optim.Adam (event_yolox_base.py:352-414: five parameter groups).  torch's fused implementation is one
// multi-tensor launch per <= ~30 tensors and group with 64 K-element chunks: eleven kernels per step for SYOLOX-S whose grids (a few dozen blocks
// each) leave most of the chip idle -- 185 us per step for 250 MB of traffic -- plus six tiny launches that add 1 to the step counters.  Here a
// table in device memory describes every tensor (parameter, gradient, both moments, its step counter, its group's learning rate and weight
// decay); a block owns 4096 consecutive elements of one tensor.
//
// The arithmetic is torch's (ATen/native/cuda/fused_adam_utils.cuh adam_math, ADAM_MODE::ORIGINAL, no amsgrad, no grad scaler), with its
// float / double promotions: the hyper-parameters are doubles there, so grad + param * weight_decay, the lerp of the first moment, the second
// moment and lr / bias_correction1 are formed in double and rounded to float; bias corrections come from pow(beta, step) in double.  The step
// counters are read here (value + 1 is the step being taken) and advanced by eas_adam_advance_steps AFTER this kernel, so no block races a write.
#include "eas_common.h"

namespace {

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* step;            // fp32 scalar on the device (torch: state['step'] of a fused / capturable Adam)
    const float* lr_ptr;    // the group's learning rate as a device scalar (a captured trainer), or NULL: lr below
    double lr;
    double wd;
    long long n;
    long long first_block;  // blocks [first_block, first_block of the next entry) belong to this tensor
};
static_assert(sizeof(AdamTensor) == 80, "table layout shared with eas_snn_amd/optim.py");

constexpr int kAdamChunk = 4096;      // elements per block: 256 threads x 4 float4

__device__ __forceinline__ double adam_lerp(double a, double b, double t) {      // libstdc++ std::lerp, what torch's kernel calls
    if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if (t == 1) return b;
    const double x = a + t * (b - a);
    return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTensor* __restrict__ tab, int ntensors, double beta1, double beta2, double eps) {
    // the tensor of this block: last entry whose first_block <= blockIdx.x (wave-uniform binary search over <= 4096 entries)
    int lo = 0, hi = ntensors;
    const long long b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid].first_block <= b) lo = mid; else hi = mid;
    }
    const AdamTensor t = tab[lo];
    const long long e0 = (b - t.first_block) * kAdamChunk;
    const float step_count = *t.step + 1.0f;
    const double lr = t.lr_ptr ? (double)*t.lr_ptr : t.lr;
    const float bc1 = (float)(1 - pow(beta1, (double)step_count));
    const float bc2_sqrt = (float)sqrt(1 - pow(beta2, (double)step_count));
    const float step_size = (float)(lr / (double)bc1);
    auto one = [&](float& param, float grad, float& ea, float& eas) {
        if (t.wd != 0) grad = (float)((double)grad + (double)param * t.wd);
        ea = (float)adam_lerp((double)ea, (double)grad, 1 - beta1);
        eas = (float)(beta2 * (double)eas + (1 - beta2) * (double)grad * (double)grad);
        const float denom = (float)((double)(sqrtf(eas) / bc2_sqrt) + eps);
        param -= step_size * ea / denom;
    };
    const bool vec = ((t.n & 3) == 0) && ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = e0 + ((long long)k * 256 + threadIdx.x) * 4;
            if (i >= t.n) break;
            float4 p4 = *reinterpret_cast<float4*>(t.p + i), m4 = *reinterpret_cast<float4*>(t.m + i), v4 = *reinterpret_cast<float4*>(t.v + i);
            const float4 g4 = *reinterpret_cast<const float4*>(t.g + i);
            one(p4.x, g4.x, m4.x, v4.x); one(p4.y, g4.y, m4.y, v4.y); one(p4.z, g4.z, m4.z, v4.z); one(p4.w, g4.w, m4.w, v4.w);
            *reinterpret_cast<float4*>(t.p + i) = p4;
            *reinterpret_cast<float4*>(t.m + i) = m4;
            *reinterpret_cast<float4*>(t.v + i) = v4;
        }
    } else {
        for (long long i = e0 + threadIdx.x; i < e0 + kAdamChunk && i < t.n; i += 256) {
            float p = t.p[i], m = t.m[i], v = t.v[i];
            one(p, t.g[i], m, v);
            t.p[i] = p; t.m[i] = m; t.v[i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void adam_advance_steps_kernel(const AdamTensor* __restrict__ tab, int ntensors) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntensors) *tab[i].step += 1.0f;
}

}  // namespace

extern "C" {

int eas_adam_table_entry_bytes(void) { return (int)sizeof(AdamTensor); }
int eas_adam_chunk(void) { return kAdamChunk; }

// table: device array of ntensors entries (layout above; first_block filled by the caller: prefix sum of ceil(n / eas_adam_chunk())),
// total_blocks = the sum.  Takes the step `*step + 1` on every tensor; the counters themselves are advanced by eas_adam_advance_steps.
int eas_adam_step(const void* table, int ntensors, long long total_blocks, double beta1, double beta2, double eps, eas_stream_t stream) {
    if (!table || ntensors < 1 || total_blocks < 1) return EAS_ERR_INVALID_ARG;
    if (total_blocks >= (1LL << 31) || ntensors > (1 << 20)) return EAS_ERR_UNSUPPORTED;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(adam_step_kernel, dim3((unsigned)total_blocks), dim3(256), 0, eas_s(stream), (const AdamTensor*)table, ntensors, beta1, beta2, eps);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_adam_advance_steps(const void* table, int ntensors, eas_stream_t stream) {
    if (!table || ntensors < 1) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(adam_advance_steps_kernel, dim3((unsigned)((ntensors + 255) / 256)), dim3(256), 0, eas_s(stream), (const AdamTensor*)table, ntensors);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
