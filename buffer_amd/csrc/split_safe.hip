// The split-f16 CNN kernels, safe by construction (round 6).
//
// The split form (csrc/convnet_h3.hip, csrc/costnet_h3.hip) carries every fp32 operand as hi + 2^-11 lo' in two f16 numbers: it needs
// every input and every hidden activation below the f16 range (|v| < 65504).  Rounds 4-5 raised an error AFTER the step when the range
// watch tripped.  Here the decision is made on the device, per patch / per match, in the same stream, without a host round trip:
//   1. the flags (int32 per patch) are cleared, the split kernel runs and sets flags[p] = 1 for every patch whose input or hidden
//      activation left the range (a NaN counts: the watch compares bit patterns / uses !(a < limit));
//   2. the fp32 kernel (csrc/convnet_wg.hip, csrc/costnet.hip) is launched over the same grid with only_if = flags: a workgroup whose
//      flag is clear returns at once, a flagged one recomputes its patch with the fp32 kernel's arithmetic and overwrites the result.
// The outputs are therefore those of the split kernel where it is valid and, bit for bit, those of the fp32 kernel elsewhere; nothing
// is ever returned from behind an overflow.  Cost when nothing trips (the normal case): one memset and np workgroups that exit on
// their first instruction (measured: tools/split_safe_probe.py).  status_dev (nullable) keeps its meaning (bit 0: some launch tripped).
#include "common.h"

// x f32[np,Cin0,140] -> without head: y_or_equi = y f32[np,32,140]; with head_params (DEVICE f32[545], buf_descriptor_head):
// desc f32[np,32] and y_or_equi = equi f32[np,32,140].  wt_split_host / wt_wg_host: the filters in the two kernels' tilings
// (buf_split_tile_filters / buf_winograd_tile_weights), bias_host shared.  flags_ws: DEVICE int32[np] (scratch; after the call
// flags_ws[p] != 0 marks the patches that took the fp32 kernel).
extern "C" int buf_cylindrical_net_split_safe(const float* x, int npatch, const void* const* wt_split_host, const float* const* wt_wg_host,
                                              const float* const* bias_host, const int* cin_host, const int* cout_host, const int* relu_host,
                                              const float* head_params, float* y_or_equi, float* desc, int* status_dev, int* flags_ws,
                                              void* stream)
{
    BUF_REQUIRE(npatch >= 0, BUF_EINVAL, "buf_cylindrical_net_split_safe: npatch=%d", npatch);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(x && y_or_equi && flags_ws && wt_split_host && wt_wg_host && bias_host && cin_host && cout_host && relu_host, BUF_EINVAL,
                "buf_cylindrical_net_split_safe: null argument");
    BUF_REQUIRE(!head_params || desc, BUF_EINVAL, "buf_cylindrical_net_split_safe: head without desc");
    if (int rc = buf_cylindrical_net_wg_supports(cin_host, cout_host)) return rc;       // the re-run must exist for these widths
    hipStream_t s = (hipStream_t)stream;
    BUF_CHECK_HIP(hipMemsetAsync(flags_ws, 0, sizeof(int) * (size_t)npatch, s));
    int rc = h3_launch(x, npatch, wt_split_host, bias_host, cin_host, cout_host, relu_host, head_params ? nullptr : y_or_equi, head_params,
                       desc, head_params ? y_or_equi : nullptr, status_dev, stream, flags_ws);
    if (rc) return rc;
    rc = wg_launch(x, npatch, wt_wg_host, bias_host, cin_host, cout_host, relu_host, y_or_equi, flags_ws, stream);
    if (rc) return rc;
    if (head_params) {
        k_desc_head_masked<<<npatch, DH_THREADS, 0, s>>>(y_or_equi, head_params, desc, y_or_equi, flags_ws);
        BUF_LAUNCH_CHECK();
    }
    return BUF_OK;
}

// The cost net likewise: dense inputs (s_rows == null: s_eq, t_eq f32[m,32,5,20]) or the gathered form (s_eq = t_eq = equi
// f32[rows,32,ele_n,20] with ele_n = 7 and int64 row ids).  wt_split_host: 11 planes (buf_split_tile_gemm), wt_f32_host: the 10 matrices of
// buf_cost_volume_net; the two bias sets as for those entry points.  flags_ws: DEVICE int32[m].
extern "C" int buf_cost_volume_net_split_safe(const float* s_eq, const float* t_eq, int ele_n, const long long* s_rows, const long long* t_rows,
                                              int m, const void* const* wt_split_host, const float* const* bias_split_host,
                                              const float* const* wt_f32_host, const float* const* bias_f32_host, float* ind_out,
                                              int* status_dev, int* flags_ws, void* stream)
{
    BUF_REQUIRE(m >= 0, BUF_EINVAL, "buf_cost_volume_net_split_safe: m=%d", m);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(s_eq && t_eq && wt_split_host && bias_split_host && wt_f32_host && bias_f32_host && ind_out && flags_ws, BUF_EINVAL,
                "buf_cost_volume_net_split_safe: null argument");
    BUF_REQUIRE((s_rows == nullptr) == (t_rows == nullptr), BUF_EINVAL, "buf_cost_volume_net_split_safe: one row list without the other");
    BUF_REQUIRE(!s_rows || ele_n == 7, BUF_EINVAL, "buf_cost_volume_net_split_safe: ele_n=%d (the kernels are built for ele_n = 7)", ele_n);
    hipStream_t s = (hipStream_t)stream;
    BUF_CHECK_HIP(hipMemsetAsync(flags_ws, 0, sizeof(int) * (size_t)m, s));
    int rc = cost_net_h3_launch(s_eq, t_eq, m, wt_split_host, bias_split_host, s_rows, t_rows, ele_n, ind_out, status_dev, stream,
                                "buf_cost_volume_net_split_safe", flags_ws);
    if (rc) return rc;
    return cost_net_launch(s_eq, t_eq, m, wt_f32_host, bias_f32_host, s_rows, t_rows, ele_n, ind_out, stream, "buf_cost_volume_net_split_safe", flags_ws);
}
