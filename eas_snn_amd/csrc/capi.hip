// ABI bookkeeping for libeas_hip.so.
#include "eas_common.h"

#include <cxxabi.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

int eas_trace_on = 0;
long long eas_launch_count = 0;
namespace {
std::mutex trace_mu;
std::set<std::string> trace_names;
std::set<const void*> trace_seen;
}  // namespace

void eas_trace_kernel(const void* host_function) {
    std::lock_guard<std::mutex> lock(trace_mu);
    if (!trace_seen.insert(host_function).second) return;
    const char* mangled = hipKernelNameRefByPtr(host_function, nullptr);
    if (!mangled) {
        trace_names.insert("?");
        return;
    }
    int status = 0;
    char* plain = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    trace_names.insert(status == 0 && plain ? plain : mangled);
    free(plain);
}

extern "C" {

// Kernel-instance trace: begin clears the record and switches it on; dump switches it off and writes the distinct device kernel symbols
// launched in between, newline separated, into buf (capacity cap bytes incl. the terminator); returns the bytes needed.
void eas_kernel_trace_begin(void) {
    std::lock_guard<std::mutex> lock(trace_mu);
    trace_names.clear();
    trace_seen.clear();
    eas_trace_on = 1;
}

int64_t eas_kernel_trace_dump(char* buf, int64_t cap) {
    std::lock_guard<std::mutex> lock(trace_mu);
    eas_trace_on = 0;
    std::string all;
    for (const auto& n : trace_names) {
        all += n;
        all += '\n';
    }
    if (buf && cap > 0) {
        const size_t n = all.size() < (size_t)(cap - 1) ? all.size() : (size_t)(cap - 1);
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

// kernel launches the library has issued since it was loaded (bench.py: launches per step)
int64_t eas_launch_counter(void) { return (int64_t)eas_launch_count; }

int eas_abi_version(void) { return 9; }   // 9: eas_adam_step_ex / eas_adam_advance_steps_ex (weight average in the optimizer launch, group learning rates as arguments; table entries 96 bytes); 8: eas_conv_dgrad_small(_supported), eas_pred_dgrad_group, eas_det_decode_eval, eas_adam_step / eas_adam_advance_steps; 7: eas_launch_counter; grouped (multi-problem) launches: eas_conv_fwd_group(_plan), eas_bn_silu_fwd_group / _bwd_group, eas_conv_wgrad_group_partial(_plan), eas_channel_sum_group; 6: eas_smallconv_fwd / eas_smallconv_bwd_weight take x_tm (input read as collated micro-slices, time-major newest first), eas_conv_bn_act_eval; 5: eas_conv_bn_lif_eval (fused eval step), kernel-instance trace, eas_spike_planes_from_f32 takes the tag-violation flag (and eas_conv_fwd / _stats really write theirs); 4: sampler convolutions take weights arranged by eas_smallconv_pack_weights; eas_arsnn_fused_step_fwd, eas_smallconv_bwd_input_dual; 3: eas_conv_fwd_stats, EasBnPending.pitch; 2: eas_bn_lif_fwd_ex gained spikes_u8; eas_conv_fwd_u8 / eas_conv_wgrad_u8, *_patan, eas_stacked_hist_event_sum

const char* eas_status_string(int status) {
    switch (status) {
        case EAS_OK: return "ok";
        case EAS_ERR_INVALID_ARG: return "invalid argument (null/misaligned pointer or bad size)";
        case EAS_ERR_UNSUPPORTED: return "unsupported configuration for the HIP path";
        case EAS_ERR_LAUNCH: return "HIP launch/memset failure";
        default: return "unknown status";
    }
}

}  // extern "C"
