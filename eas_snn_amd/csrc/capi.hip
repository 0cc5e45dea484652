// ABI bookkeeping for libeas_hip.so.
#include "eas_common.h"

extern "C" {

int eas_abi_version(void) { return 4; }   // 4: sampler convolutions take weights arranged by eas_smallconv_pack_weights; eas_arsnn_fused_step_fwd, eas_smallconv_bwd_input_dual; 3: eas_conv_fwd_stats, EasBnPending.pitch; 2: eas_bn_lif_fwd_ex gained spikes_u8; eas_conv_fwd_u8 / eas_conv_wgrad_u8, *_patan, eas_stacked_hist_event_sum

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
