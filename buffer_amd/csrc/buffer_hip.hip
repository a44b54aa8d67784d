// Single translation unit of libbuffer_hip.so (gfx950).  Parts are plain includes so that the
// kernels share the static helpers without relocatable device code.
#include <stdlib.h>
#include "core.hip"
#include "radius.hip"
#include "subsample.hip"
#include "pointops.hip"
#include "vn.hip"
#include "voxelize.hip"
#include "registration.hip"
#include "convnet.hip"
#include "convnet_h3.hip"
#include "costnet_h3.hip"
#include "preprocess.hip"
// Packed-fp32 vector instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) are OFF for the whole library (buffer_amd/build.py:
// -target-feature -packed-fp32-ops).  Round 5 finding (tools/race_probe3.py, profiles/r05_packed_fp32_hazard.txt): a wavefront
// executing packed-fp32 instructions returns wrong values in its lanes 0..15 now and then while ANOTHER wavefront of the same SIMD
// issues v_mfma_f32_16x16x32_f16 (k_vn_gather6_lds on the keypoint stream beside k_nn1f_sweep / k_cyl_net_h3 / k_cost_net_h3 of the
// main stream: 40-65 % of its launches differed in 16 lanes; with packed ops off: 0 of 1200; beside the fp32-MFMA kernels: 0).  The
// two fp32-MFMA CNN kernels, whose Winograd transforms the compiler had built on v_pk_add_f32, measured 0.9 % FASTER without them.
#include "convnet_wg.hip"
#include "costnet.hip"
#include "split_safe.hip"
