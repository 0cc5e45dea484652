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
    float* ema;             // the parameter's twin in the averaged model (ModelEMA.ema), or NULL: no average kept for this tensor
    double lr;              // used when lr_ptr is NULL and group < 0
    double wd;
    long long n;
    long long first_block;  // blocks [first_block, first_block of the next entry) belong to this tensor
    int group;              // >= 0: the learning rate is EasAdamHyper.group_lr[group] (a host number that changes every step without a new table)
    int pad_;
};
static_assert(sizeof(AdamTensor) == 96, "table layout shared with eas_snn_amd/optim.py");
// An entry with g == NULL takes no Adam step: p is a tensor the average follows without an optimizer (BatchNorm running statistics).

constexpr int kAdamChunk = 4096;      // elements per block: 256 threads x 4 float4

__device__ __forceinline__ double adam_lerp(double a, double b, double t) {      // libstdc++ std::lerp, what torch's kernel calls
    if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if (t == 1) return b;
    const double x = a + t * (b - a);
    return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}

// Exponential moving average of the weights inside the same walk (yolox/utils/ema.py:44-60 of the reference, called right after
// optimizer.step() in trainer.py:120-121): d = decay * (1 - exp(-n / ramp)) with n = the number of the update being taken, formed in double
// like the reference's python float, then  v *= d;  v += (1 - d) * p  -- two fp32 roundings of the product terms and one of the sum, as the
// three tensor operators of the reference round (the file is built with -ffp-contract=off).  n comes from a device counter
// (*ema_updates + 1), advanced by eas_adam_advance_steps_ex after this kernel, so a captured step keeps the ramp going.
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTensor* __restrict__ tab, int ntensors, const EasAdamHyper h) {
    const double beta1 = h.beta1, beta2 = h.beta2, eps = h.eps;
    // the tensor of this block: last entry whose first_block <= blockIdx.x (wave-uniform binary search over <= 4096 entries)
    int lo = 0, hi = ntensors;
    const long long b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid].first_block <= b) lo = mid; else hi = mid;
    }
    const AdamTensor t = tab[lo];
    const long long e0 = (b - t.first_block) * kAdamChunk;
    const bool adam = t.g != nullptr;
    const float step_count = adam ? *t.step + 1.0f : 1.0f;
    const double lr = t.lr_ptr ? (double)*t.lr_ptr : (t.group >= 0 ? h.group_lr[t.group] : t.lr);
    float ema_keep = 0.f, ema_take = 0.f;
    if (t.ema) {
        const double d = h.ema_decay * (1.0 - exp(-(*h.ema_updates + 1.0) / h.ema_ramp));
        ema_keep = (float)d;
        ema_take = (float)(1.0 - d);
    }
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
    auto avg = [&](float a, float p) { const float kept = a * ema_keep; const float taken = ema_take * p; return kept + taken; };
    const bool vec = ((t.n & 3) == 0) && ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v | (uintptr_t)t.ema) & 15) == 0);
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = e0 + ((long long)k * 256 + threadIdx.x) * 4;
            if (i >= t.n) break;
            float4 p4 = *reinterpret_cast<float4*>(t.p + i);
            if (adam) {
                float4 m4 = *reinterpret_cast<float4*>(t.m + i), v4 = *reinterpret_cast<float4*>(t.v + i);
                const float4 g4 = *reinterpret_cast<const float4*>(t.g + i);
                one(p4.x, g4.x, m4.x, v4.x); one(p4.y, g4.y, m4.y, v4.y); one(p4.z, g4.z, m4.z, v4.z); one(p4.w, g4.w, m4.w, v4.w);
                *reinterpret_cast<float4*>(t.p + i) = p4;
                *reinterpret_cast<float4*>(t.m + i) = m4;
                *reinterpret_cast<float4*>(t.v + i) = v4;
            }
            if (t.ema) {
                float4 a4 = *reinterpret_cast<float4*>(t.ema + i);
                a4.x = avg(a4.x, p4.x); a4.y = avg(a4.y, p4.y); a4.z = avg(a4.z, p4.z); a4.w = avg(a4.w, p4.w);
                *reinterpret_cast<float4*>(t.ema + i) = a4;
            }
        }
    } else {
        for (long long i = e0 + threadIdx.x; i < e0 + kAdamChunk && i < t.n; i += 256) {
            float p = t.p[i];
            if (adam) {
                float m = t.m[i], v = t.v[i];
                one(p, t.g[i], m, v);
                t.p[i] = p; t.m[i] = m; t.v[i] = v;
            }
            if (t.ema) t.ema[i] = avg(t.ema[i], p);
        }
    }
}

__global__ __launch_bounds__(256) void adam_advance_steps_kernel(const AdamTensor* __restrict__ tab, int ntensors, double* ema_updates) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntensors && tab[i].g) *tab[i].step += 1.0f;
    if (i == 0 && ema_updates) *ema_updates += 1.0;
}

}  // namespace

extern "C" {

int eas_adam_table_entry_bytes(void) { return (int)sizeof(AdamTensor); }
int eas_adam_chunk(void) { return kAdamChunk; }

// table: device array of ntensors entries (layout above; first_block filled by the caller: prefix sum of ceil(n / eas_adam_chunk())),
// total_blocks = the sum.  Takes the step `*step + 1` on every tensor; the counters themselves are advanced by eas_adam_advance_steps.
int eas_adam_step_ex(const void* table, int ntensors, long long total_blocks, const EasAdamHyper* hyper, eas_stream_t stream) {
    if (!table || !hyper || ntensors < 1 || total_blocks < 1) return EAS_ERR_INVALID_ARG;
    if (total_blocks >= (1LL << 31) || ntensors > (1 << 20)) return EAS_ERR_UNSUPPORTED;
    if (hyper->ema_updates && !(hyper->ema_ramp > 0)) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(adam_step_kernel, dim3((unsigned)total_blocks), dim3(256), 0, eas_s(stream), (const AdamTensor*)table, ntensors, *hyper);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_adam_step(const void* table, int ntensors, long long total_blocks, double beta1, double beta2, double eps, eas_stream_t stream) {
    EasAdamHyper h = {};
    h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.ema_ramp = 1.0;
    return eas_adam_step_ex(table, ntensors, total_blocks, &h, stream);
}

int eas_adam_advance_steps_ex(const void* table, int ntensors, double* ema_updates, eas_stream_t stream) {
    if (!table || ntensors < 1) return EAS_ERR_INVALID_ARG;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(adam_advance_steps_kernel, dim3((unsigned)((ntensors + 255) / 256)), dim3(256), 0, eas_s(stream), (const AdamTensor*)table, ntensors, ema_updates);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_adam_advance_steps(const void* table, int ntensors, eas_stream_t stream) { return eas_adam_advance_steps_ex(table, ntensors, nullptr, stream); }

}  // extern "C"
