// K3: adaptive-sampling recurrent step (AdaptiveRSNNEmbedding), dense masked form.
// The reference walks fired positions with nonzero() + six gathers/scatters and a host sync per
// step (yolox/models/embedding.py:181-201).  Here every element updates its own state; each
// element adds into agg[seg] at most once per step, in step order, so fp32 sums are identical.
// HBM-bound: per (n,c,hw) per step 16 B conv outputs + 16 B (v, vsum r/w) + 4 B spike + 8 B int state
// (+ 16 B saved for backward when training).
#include "eas_common.h"
#include "smallconv_core.h"

namespace {

constexpr int VEC = 4;

// segment counter (0..Ts) and last-spike step (-1..Tm-1) live in HBM as int8: four elements are one 32-bit word
typedef int8_t eas_state_t;
__device__ __forceinline__ int4 load_q8(const eas_state_t* p) {
    const int w = *reinterpret_cast<const int*>(p);
    return make_int4((int)(int8_t)(w & 0xFF), (int)(int8_t)((w >> 8) & 0xFF), (int)(int8_t)((w >> 16) & 0xFF), (int)(int8_t)((w >> 24) & 0xFF));
}
__device__ __forceinline__ void store_q8(eas_state_t* p, const int4& v) {
    *reinterpret_cast<int*>(p) = (v.x & 0xFF) | ((v.y & 0xFF) << 8) | ((v.z & 0xFF) << 16) | ((v.w & 0xFF) << 24);
}

struct StepCfg {
    int t, Ts, readout, spike_attach, soft_reset;
    float thresh, v_reset, sg_alpha;
    int64_t plane;  // C2 * HW
    int64_t total;  // N * C2 * HW
};

__device__ __forceinline__ void step_elem(float g_in, float c_in, float g_rec, float c_rec, float v, float vsum,
                                          int32_t& seg, int32_t& tl, float* __restrict__ agg, int64_t i,
                                          const StepCfg& c, float& v_out, float& vsum_out, float& spike, float& gate,
                                          float& vn) {
    gate = eas_sigmoidf(g_in + g_rec);
    const float cur = c_in + c_rec;
    vn = gate * v + cur;                                  // embedding.py:133
    spike = (vn - c.thresh) > 0.0f ? 1.0f : 0.0f;         // Rectangle: strict >
    v_out = c.soft_reset ? vn - c.thresh * spike : vn * (1.0f - spike) + c.v_reset * spike;
    const float vs1 = vsum + vn;                          // :179
    if (spike != 0.0f) {
        if (c.readout != 3 && seg < c.Ts) {
            float val = c.readout == 0 ? vs1 : (c.readout == 1 ? v_out : vs1 / (float)(c.t - tl));
            if (c.spike_attach) val = val * spike;
            agg[(int64_t)seg * c.total + i] += val;       // :194
            // the reference filters spike_pos by seg < Ts BEFORE these two updates (:183-184, :195-196):
            // segment counter and last-spike time freeze once an element has filled its Ts frames
            seg += 1;
            tl = c.t;
        }
        vsum_out = c.readout == 3 ? vs1 : 0.0f;           // :197 (every fired element); readout 3 = running sum ("rsnn"/"snn")
    } else {
        vsum_out = vs1;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void arsnn_step_fwd_kernel(
    const float* __restrict__ conv_in, const float* __restrict__ conv_rec, const float* __restrict__ v,
    const float* __restrict__ vsum, eas_state_t* __restrict__ seg, eas_state_t* __restrict__ t_last, float* __restrict__ agg,
    float* __restrict__ v_out, float* __restrict__ vsum_out, float* __restrict__ spike_out,
    float* __restrict__ gate_save, float* __restrict__ vn_save, eas_state_t* __restrict__ seg_before,
    eas_state_t* __restrict__ tl_before, StepCfg c) {
    const int64_t ngroups = c.total / VEC;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = g * VEC;
        const int64_t n = i / c.plane;
        const int64_t r = i - n * c.plane;
        const int64_t cb = n * 2 * c.plane + r;
        const float4 gi = *reinterpret_cast<const float4*>(conv_in + cb);
        const float4 ci = *reinterpret_cast<const float4*>(conv_in + cb + c.plane);
        const float4 gr = *reinterpret_cast<const float4*>(conv_rec + cb);
        const float4 cr = *reinterpret_cast<const float4*>(conv_rec + cb + c.plane);
        // v == NULL: the first step of a sequence -- zero potentials and sums, no segment written yet (seg = 0, t_last = -1): the state
        // tensors need no zero fill before and no read here (seg / t_last are then outputs only)
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f), vs = vv;
        int4 sg = make_int4(0, 0, 0, 0), tl = make_int4(-1, -1, -1, -1);
        if (v) {
            vv = *reinterpret_cast<const float4*>(v + i);
            vs = *reinterpret_cast<const float4*>(vsum + i);
            sg = load_q8(seg + i);
            tl = load_q8(t_last + i);
        }
        if (seg_before) store_q8(seg_before + i, sg);
        if (tl_before) store_q8(tl_before + i, tl);
        float4 vo, vso, sp, ga, vn;
        step_elem(gi.x, ci.x, gr.x, cr.x, vv.x, vs.x, sg.x, tl.x, agg, i + 0, c, vo.x, vso.x, sp.x, ga.x, vn.x);
        step_elem(gi.y, ci.y, gr.y, cr.y, vv.y, vs.y, sg.y, tl.y, agg, i + 1, c, vo.y, vso.y, sp.y, ga.y, vn.y);
        step_elem(gi.z, ci.z, gr.z, cr.z, vv.z, vs.z, sg.z, tl.z, agg, i + 2, c, vo.z, vso.z, sp.z, ga.z, vn.z);
        step_elem(gi.w, ci.w, gr.w, cr.w, vv.w, vs.w, sg.w, tl.w, agg, i + 3, c, vo.w, vso.w, sp.w, ga.w, vn.w);
        *reinterpret_cast<float4*>(v_out + i) = vo;
        *reinterpret_cast<float4*>(vsum_out + i) = vso;
        *reinterpret_cast<float4*>(spike_out + i) = sp;
        store_q8(seg + i, sg);
        store_q8(t_last + i, tl);
        if (gate_save) *reinterpret_cast<float4*>(gate_save + i) = ga;
        if (vn_save) *reinterpret_cast<float4*>(vn_save + i) = vn;
    }
}

// ---- the micro-step with its convolutions (K3 as SURVEY 2.2 defines it) --------------------------------------------------------------
// One launch per micro-step: the SECOND convolution of the input stack and of the gate stack (4 -> 4 channels, k x k; their first
// convolutions + ReLU stay separate launches) and the integrate / fire / reset / segment write.  The two convolution results
// X = conv2_in(a_in) and R = conv2_g(a_g) -- (gate pre-activation | current) planes, 16 B + 16 B per pixel and step that used to be
// written by two launches and read back by the third -- stay in registers: each is accumulated in the order of smallconv_kernel
// (bias, ci, ky, kx), then g = X + R exactly as arsnn_step_fwd_kernel forms it, so every output is bit-identical to the separate launches.
// A block owns a 16 x 64 pixel tile: the a_in window goes through LDS (register prefetch), X is accumulated, the a_g window follows
// through the same LDS buffer, R is accumulated, the state of the tile's two channels is updated in place.
// HAS_G = false: R comes from `r_const` [4][H][W] (step 0: the spike entering it is the constant 0, gate_conv(0) is one image shared by
// the batch) or is zero (no gate stack).
template <int K, bool HAS_G>
__global__ __launch_bounds__(NT, 2) void arsnn_fused_step_fwd_kernel(
    const float* __restrict__ a_in, const float* __restrict__ wr_in, const float* __restrict__ b_in, const float* __restrict__ a_g,
    const float* __restrict__ wr_g, const float* __restrict__ b_g, const float* __restrict__ r_const, const float* __restrict__ v,
    const float* __restrict__ vsum, eas_state_t* __restrict__ seg, eas_state_t* __restrict__ t_last, float* __restrict__ agg,
    float* __restrict__ v_out, float* __restrict__ vsum_out, float* __restrict__ spike_out, float* __restrict__ gate_save,
    float* __restrict__ vn_save, eas_state_t* __restrict__ seg_before, eas_state_t* __restrict__ tl_before, StepCfg c, int N, int H, int W) {
    using SG = Stage<4, K, true>;                 // W % 4 == 0 (launcher): aligned-quad staging
    constexpr int LWF = SG::PITCH, PLANE = SG::PLANE;
    using St = typename SG::St;
    St st;
    st.init(H, W);
    __shared__ __attribute__((aligned(16))) float lds[4 * PLANE];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int HW = H * W;
    float4 pre[St::PER];
    unsigned okm = 0;
    int tile = blockIdx.x;
    if (tile < ntiles)
        okm = st.load(a_in, pre, tile / (tiles_x * tiles_y), H, W, ((tile / tiles_x) % tiles_y) * TH, (tile % tiles_x) * TW);
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int ty0 = ((tile / tiles_x) % tiles_y) * TH, tx0 = (tile % tiles_x) * TW;
        const int nxt = tile + gridDim.x;
        __syncthreads();                               // readers of the previous tile are done
        st.store(lds, pre, okm);
        __syncthreads();
        if (HAS_G) okm = st.load(a_g, pre, n, H, W, ty0, tx0);                   // this tile's gate window flies during the X accumulation
        else if (nxt < ntiles) okm = st.load(a_in, pre, nxt / (tiles_x * tiles_y), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
        // the state of the thread's 2 x 4 elements (and the shared gate image of step 0) is requested in front of the arithmetic
        const int oy = ty0 + ty, ox = tx0 + 4 * tx;
        const bool inside = oy < H && ox < W;          // W % 4 == 0 (checked by the launcher): a thread's four pixels exist together
        const int pix = oy * W + ox;
        float4 vv[2], vs[2], rg[2], rcur[2];
        int4 sg[2], tl[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            vv[c2] = make_float4(0.f, 0.f, 0.f, 0.f); vs[c2] = vv[c2]; rg[c2] = vv[c2]; rcur[c2] = vv[c2];
            sg[c2] = make_int4(0, 0, 0, 0); tl[c2] = make_int4(-1, -1, -1, -1);
            if (inside) {
                const int64_t i = ((int64_t)n * 2 + c2) * HW + pix;
                if (v) {
                    vv[c2] = *reinterpret_cast<const float4*>(v + i);
                    vs[c2] = *reinterpret_cast<const float4*>(vsum + i);
                    sg[c2] = load_q8(seg + i);
                    tl[c2] = load_q8(t_last + i);
                }
                if (!HAS_G && r_const) {
                    rg[c2] = *reinterpret_cast<const float4*>(r_const + (int64_t)c2 * HW + pix);
                    rcur[c2] = *reinterpret_cast<const float4*>(r_const + (int64_t)(2 + c2) * HW + pix);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f2 X[4][2], R[4][2];
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const f2 bx = b_in ? f2{b_in[2 * op], b_in[2 * op + 1]} : f2{0.f, 0.f};
            const f2 bg = (HAS_G && b_g) ? f2{b_g[2 * op], b_g[2 * op + 1]} : f2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) { X[j][op] = bx; R[j][op] = bg; }
        }
        sc_accumulate<4, 4, K, LWF, SG::NQ, SG::OFS>(lds, PLANE, wr_in, X, tx, ty);
        if (HAS_G) {
            __syncthreads();
            st.store(lds, pre, okm);
            __syncthreads();
            if (nxt < ntiles) okm = st.load(a_in, pre, nxt / (tiles_x * tiles_y), H, W, ((nxt / tiles_x) % tiles_y) * TH, (nxt % tiles_x) * TW);
            sc_accumulate<4, 4, K, LWF, SG::NQ, SG::OFS>(lds, PLANE, wr_g, R, tx, ty);
        }
        if (inside) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                // channel c2: gate pre-activation = output c2, current = output 2 + c2 (torch.chunk(.., 2, dim=C), embedding.py:171-174)
                float gi[4], ci[4], gr[4], cr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gi[j] = c2 ? X[j][0].y : X[j][0].x;
                    ci[j] = c2 ? X[j][1].y : X[j][1].x;
                    gr[j] = c2 ? R[j][0].y : R[j][0].x;
                    cr[j] = c2 ? R[j][1].y : R[j][1].x;
                }
                if (!HAS_G && r_const) {
                    gr[0] = rg[c2].x; gr[1] = rg[c2].y; gr[2] = rg[c2].z; gr[3] = rg[c2].w;
                    cr[0] = rcur[c2].x; cr[1] = rcur[c2].y; cr[2] = rcur[c2].z; cr[3] = rcur[c2].w;
                }
                const int64_t i = ((int64_t)n * 2 + c2) * HW + pix;
                if (seg_before) store_q8(seg_before + i, sg[c2]);
                if (tl_before) store_q8(tl_before + i, tl[c2]);
                float4 vo, vso, sp, ga, vn;
                step_elem(gi[0], ci[0], gr[0], cr[0], vv[c2].x, vs[c2].x, sg[c2].x, tl[c2].x, agg, i + 0, c, vo.x, vso.x, sp.x, ga.x, vn.x);
                step_elem(gi[1], ci[1], gr[1], cr[1], vv[c2].y, vs[c2].y, sg[c2].y, tl[c2].y, agg, i + 1, c, vo.y, vso.y, sp.y, ga.y, vn.y);
                step_elem(gi[2], ci[2], gr[2], cr[2], vv[c2].z, vs[c2].z, sg[c2].z, tl[c2].z, agg, i + 2, c, vo.z, vso.z, sp.z, ga.z, vn.z);
                step_elem(gi[3], ci[3], gr[3], cr[3], vv[c2].w, vs[c2].w, sg[c2].w, tl[c2].w, agg, i + 3, c, vo.w, vso.w, sp.w, ga.w, vn.w);
                *reinterpret_cast<float4*>(v_out + i) = vo;
                *reinterpret_cast<float4*>(vsum_out + i) = vso;
                *reinterpret_cast<float4*>(spike_out + i) = sp;
                store_q8(seg + i, sg[c2]);
                store_q8(t_last + i, tl[c2]);
                if (gate_save) *reinterpret_cast<float4*>(gate_save + i) = ga;
                if (vn_save) *reinterpret_cast<float4*>(vn_save + i) = vn;
            }
        }
    }
}

// Backward of one step for one element.
__device__ __forceinline__ void step_elem_bwd(float gv_out, float gvs_out, float gsp, const float* __restrict__ g_agg,
                                              int64_t i, float v_prev, float vsum_prev, float gate, float vn, int32_t seg_b,
                                              int32_t tl_b, const StepCfg& c, float& g_gatepre, float& g_cur,
                                              float& g_v_prev, float& g_vsum_prev) {
    const float spike = (vn - c.thresh) > 0.0f ? 1.0f : 0.0f;
    const bool fired = spike != 0.0f;
    const float vs1 = vsum_prev + vn;
    float d_vs1 = (fired && c.readout != 3) ? 0.0f : gvs_out;   // vsum_out = fired ? 0 : vs1   (running sum: always vs1)
    float d_vout = gv_out;
    float d_s = gsp;
    if (fired && c.readout != 3 && seg_b < c.Ts) {
        const float ga = g_agg[(int64_t)seg_b * c.total + i];
        float d_pre = ga;
        if (c.spike_attach) {
            const float v_out = c.soft_reset ? vn - c.thresh : c.v_reset;  // spike == 1 here
            const float pre = c.readout == 0 ? vs1 : (c.readout == 1 ? v_out : vs1 / (float)(c.t - tl_b));
            d_s += ga * pre;       // val = pre * spike
            d_pre = ga * spike;
        }
        if (c.readout == 0) d_vs1 += d_pre;
        else if (c.readout == 1) d_vout += d_pre;
        else d_vs1 += d_pre / (float)(c.t - tl_b);
    }
    float d_vn = d_vs1;                      // vs1 = vsum_prev + vn
    g_vsum_prev = d_vs1;
    if (c.soft_reset) {                      // v_out = vn - thresh * s
        d_vn += d_vout;
        d_s += d_vout * (-c.thresh);
    } else {                                 // v_out = vn * (1 - s) + v_reset * s
        d_vn += d_vout * (1.0f - spike);
        d_s += d_vout * (c.v_reset - vn);
    }
    d_vn += d_s * eas_surrogate_grad(EAS_SG_RECT, c.sg_alpha, vn - c.thresh);
    g_cur = d_vn;                            // vn = gate * v + cur
    g_v_prev = d_vn * gate;
    g_gatepre = (d_vn * v_prev) * gate * (1.0f - gate);
}

__global__ __launch_bounds__(EAS_BLOCK) void arsnn_step_bwd_kernel(
    const float* __restrict__ g_v_out, const float* __restrict__ g_vsum_out, const float* __restrict__ g_spike,
    const float* __restrict__ g_agg, const float* __restrict__ v_prev, const float* __restrict__ vsum_prev,
    const float* __restrict__ gate_save, const float* __restrict__ vn_save, const eas_state_t* __restrict__ seg_before,
    const eas_state_t* __restrict__ tl_before, float* __restrict__ g_conv, float* __restrict__ g_v_prev,
    float* __restrict__ g_vsum_prev, StepCfg c) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c.total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / c.plane;
        const int64_t r = i - n * c.plane;
        const int64_t cb = n * 2 * c.plane + r;
        float gg, gc, gvp, gvsp;
        step_elem_bwd(g_v_out ? g_v_out[i] : 0.f, g_vsum_out ? g_vsum_out[i] : 0.f, g_spike ? g_spike[i] : 0.f, g_agg, i,
                      v_prev ? v_prev[i] : 0.f, vsum_prev ? vsum_prev[i] : 0.f, gate_save[i], vn_save[i], seg_before[i], tl_before[i], c, gg, gc,
                      gvp, gvsp);
        g_conv[cb] = gg;
        g_conv[cb + c.plane] = gc;
        g_v_prev[i] = gvp;
        g_vsum_prev[i] = gvsp;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void arsnn_tail_fwd_kernel(const float* __restrict__ v, const float* __restrict__ vsum,
                                                                   const float* __restrict__ spike_last,
                                                                   const eas_state_t* __restrict__ seg,
                                                                   const eas_state_t* __restrict__ t_last, float* __restrict__ agg,
                                                                   int Tm, int Ts, int readout, int write_zero, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (spike_last[i] != 0.0f) continue;
        const int s = seg[i];
        if (s >= Ts) continue;
        float val = readout == 0 ? vsum[i] : (readout == 1 ? v[i] : vsum[i] / (float)(Tm - 1 - t_last[i]));
        if (write_zero) val = val * 0.0f;   // RPD (embedding.py:215-216); keeps inf/nan -> nan like the reference
        agg[(int64_t)s * total + i] += val;
    }
}

__global__ __launch_bounds__(EAS_BLOCK) void arsnn_tail_bwd_kernel(const float* __restrict__ g_agg,
                                                                   const float* __restrict__ spike_last,
                                                                   const eas_state_t* __restrict__ seg,
                                                                   const eas_state_t* __restrict__ t_last, float* __restrict__ g_v,
                                                                   float* __restrict__ g_vsum, int Tm, int Ts, int readout,
                                                                   int write_zero, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float gv = 0.f, gvs = 0.f;
        const int s = seg[i];
        if (spike_last[i] == 0.0f && s < Ts && !write_zero) {
            const float ga = g_agg[(int64_t)s * total + i];
            if (readout == 0) gvs = ga;
            else if (readout == 1) gv = ga;
            else gvs = ga / (float)(Tm - 1 - t_last[i]);
        }
        g_v[i] = gv;
        g_vsum[i] = gvs;
    }
}

}  // namespace

extern "C" {

int eas_arsnn_step_fwd(const float* conv_in, const float* conv_rec, const float* v, const float* vsum,
                       int8_t* seg, int8_t* t_last, float* agg, float* v_out, float* vsum_out, float* spike_out,
                       float* gate_save, float* vn_save, int8_t* seg_before, int8_t* t_last_before, int t,
                       int Ts, int readout, int spike_attach, float thresh, float v_reset, int soft_reset, int N,
                       int C2, int HW, eas_stream_t stream) {
    if (!conv_in || !conv_rec || (v == nullptr) != (vsum == nullptr) || !seg || !t_last || !agg || !v_out || !vsum_out || !spike_out)
        return EAS_ERR_INVALID_ARG;
    if (N < 1 || C2 < 1 || HW < 1 || Ts < 1 || t < 0 || readout < 0 || readout > 3) return EAS_ERR_INVALID_ARG;
    if (HW % VEC != 0 || Ts > 127 || t > 126) return EAS_ERR_UNSUPPORTED;       // segment counter / last-spike step are int8
    if (((uintptr_t)conv_in | (uintptr_t)conv_rec | (uintptr_t)v | (uintptr_t)vsum | (uintptr_t)seg | (uintptr_t)t_last |
         (uintptr_t)v_out | (uintptr_t)vsum_out | (uintptr_t)spike_out | (uintptr_t)gate_save | (uintptr_t)vn_save |
         (uintptr_t)seg_before | (uintptr_t)t_last_before) & 15)
        return EAS_ERR_INVALID_ARG;
    StepCfg c{t, Ts, readout, spike_attach, soft_reset, thresh, v_reset, 1.0f, (int64_t)C2 * HW, (int64_t)N * C2 * HW};
    EAS_CLEAR_ERR();
    EAS_LAUNCH(arsnn_step_fwd_kernel, dim3(eas_grid_1d(c.total / VEC)), dim3(EAS_BLOCK), 0, eas_s(stream), conv_in,
                       conv_rec, v, vsum, seg, t_last, agg, v_out, vsum_out, spike_out, gate_save, vn_save, seg_before,
                       t_last_before, c);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

// eas_arsnn_step_fwd with the second convolutions of the input stack and of the gate stack inside the launch (C2 = 2: four hidden, four
// output channels).  a_in [N][4][H][W] = ReLU(conv1_in(events of this step)); a_g [N][4][H][W] = ReLU(conv1_g(spikes entering the
// step)) or NULL, then r_const [4][H][W] (nullable = zero) is the gate stack's contribution for every sample; wr_* = the second
// convolutions' weights packed by eas_smallconv_pack_weights (mode 0, o_total 4), b_* their biases (nullable).
int eas_arsnn_fused_step_fwd(const float* a_in, const float* wr_in, const float* b_in, const float* a_g, const float* wr_g, const float* b_g,
                             const float* r_const, const float* v, const float* vsum, int8_t* seg, int8_t* t_last, float* agg, float* v_out,
                             float* vsum_out, float* spike_out, float* gate_save, float* vn_save, int8_t* seg_before, int8_t* t_last_before,
                             int t, int Ts, int readout, int spike_attach, float thresh, float v_reset, int soft_reset, int N, int H, int W, int k,
                             eas_stream_t stream) {
    if (!a_in || !wr_in || (a_g && !wr_g) || (v == nullptr) != (vsum == nullptr) || !seg || !t_last || !agg || !v_out || !vsum_out || !spike_out)
        return EAS_ERR_INVALID_ARG;
    if (N < 1 || H < 1 || W < 1 || Ts < 1 || t < 0 || readout < 0 || readout > 3) return EAS_ERR_INVALID_ARG;
    if (W % 4 != 0 || (k != 3 && k != 5 && k != 7) || Ts > 127 || t > 126) return EAS_ERR_UNSUPPORTED;
    if (((uintptr_t)a_in | (uintptr_t)a_g | (uintptr_t)wr_in | (uintptr_t)wr_g | (uintptr_t)r_const | (uintptr_t)v | (uintptr_t)vsum | (uintptr_t)seg | (uintptr_t)t_last |
         (uintptr_t)v_out | (uintptr_t)vsum_out | (uintptr_t)spike_out | (uintptr_t)gate_save | (uintptr_t)vn_save | (uintptr_t)seg_before |
         (uintptr_t)t_last_before) & 15)
        return EAS_ERR_INVALID_ARG;
    const int HW = H * W;
    StepCfg c{t, Ts, readout, spike_attach, soft_reset, thresh, v_reset, 1.0f, (int64_t)2 * HW, (int64_t)N * 2 * HW};
    int tiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (tiles > 256 * 6) tiles = 256 * 6;
    hipStream_t st = eas_s(stream);
    EAS_CLEAR_ERR();
#define EAS_FUSED(K_)                                                                                                                        \
    do {                                                                                                                                     \
        if (a_g)                                                                                                                             \
            EAS_LAUNCH((arsnn_fused_step_fwd_kernel<K_, true>), dim3(tiles), dim3(NT), 0, st, a_in, wr_in, b_in, a_g, wr_g, b_g,     \
                               r_const, v, vsum, seg, t_last, agg, v_out, vsum_out, spike_out, gate_save, vn_save, seg_before,               \
                               t_last_before, c, N, H, W);                                                                                    \
        else                                                                                                                                 \
            EAS_LAUNCH((arsnn_fused_step_fwd_kernel<K_, false>), dim3(tiles), dim3(NT), 0, st, a_in, wr_in, b_in, a_g, wr_g, b_g,    \
                               r_const, v, vsum, seg, t_last, agg, v_out, vsum_out, spike_out, gate_save, vn_save, seg_before,               \
                               t_last_before, c, N, H, W);                                                                                    \
    } while (0)
    if (k == 3) EAS_FUSED(3);
    else if (k == 5) EAS_FUSED(5);
    else EAS_FUSED(7);
#undef EAS_FUSED
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_arsnn_step_bwd(const float* g_v_out, const float* g_vsum_out, const float* g_spike, const float* g_agg,
                       const float* v_prev, const float* vsum_prev, const float* gate_save, const float* vn_save,
                       const int8_t* seg_before, const int8_t* t_last_before, float* g_conv, float* g_v_prev,
                       float* g_vsum_prev, int t, int Ts, int readout, int spike_attach, float thresh,
                       float v_reset, int soft_reset, float sg_alpha, int N, int C2, int HW, eas_stream_t stream) {
    if (!g_agg || (v_prev == nullptr) != (vsum_prev == nullptr) || !gate_save || !vn_save || !seg_before || !t_last_before || !g_conv ||
        !g_v_prev || !g_vsum_prev)
        return EAS_ERR_INVALID_ARG;
    if (N < 1 || C2 < 1 || HW < 1 || Ts < 1 || t < 0 || readout < 0 || readout > 3 || !(sg_alpha > 0.f))
        return EAS_ERR_INVALID_ARG;
    StepCfg c{t, Ts, readout, spike_attach, soft_reset, thresh, v_reset, sg_alpha, (int64_t)C2 * HW,
              (int64_t)N * C2 * HW};
    EAS_CLEAR_ERR();
    EAS_LAUNCH(arsnn_step_bwd_kernel, dim3(eas_grid_1d(c.total)), dim3(EAS_BLOCK), 0, eas_s(stream), g_v_out,
                       g_vsum_out, g_spike, g_agg, v_prev, vsum_prev, gate_save, vn_save, seg_before, t_last_before, g_conv,
                       g_v_prev, g_vsum_prev, c);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_arsnn_tail_fwd(const float* v, const float* vsum, const float* spike_last, const int8_t* seg,
                       const int8_t* t_last, float* agg, int Tm, int Ts, int readout, int write_zero, int N,
                       int C2, int HW, eas_stream_t stream) {
    if (!v || !vsum || !spike_last || !seg || !t_last || !agg || N < 1 || C2 < 1 || HW < 1 || Ts < 1 || readout < 0 ||
        readout > 2)
        return EAS_ERR_INVALID_ARG;
    const int64_t total = (int64_t)N * C2 * HW;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(arsnn_tail_fwd_kernel, dim3(eas_grid_1d(total)), dim3(EAS_BLOCK), 0, eas_s(stream), v, vsum,
                       spike_last, seg, t_last, agg, Tm, Ts, readout, write_zero, total);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

int eas_arsnn_tail_bwd(const float* g_agg, const float* spike_last, const int8_t* seg, const int8_t* t_last,
                       float* g_v, float* g_vsum, int Tm, int Ts, int readout, int write_zero, int N, int C2,
                       int HW, eas_stream_t stream) {
    if (!g_agg || !spike_last || !seg || !t_last || !g_v || !g_vsum || N < 1 || C2 < 1 || HW < 1 || Ts < 1 ||
        readout < 0 || readout > 2)
        return EAS_ERR_INVALID_ARG;
    const int64_t total = (int64_t)N * C2 * HW;
    EAS_CLEAR_ERR();
    EAS_LAUNCH(arsnn_tail_bwd_kernel, dim3(eas_grid_1d(total)), dim3(EAS_BLOCK), 0, eas_s(stream), g_agg, spike_last,
                       seg, t_last, g_v, g_vsum, Tm, Ts, readout, write_zero, total);
    EAS_CHECK_LAUNCH();
    return EAS_OK;
}

}  // extern "C"
