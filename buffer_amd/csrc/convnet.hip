// A11 (head) -- attention pooling + normalisation of the descriptor CNN's output (models/patch_embedder.py:66-72,81-84).
// (The direct-form Cylindrical_Net kernel that used to share this file is test infrastructure now: tests/native/convnet_direct.hip.)
#include "common.h"

#define CN_POS 140            // 7 elevation x 20 azimuth
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- A11 (head): attention pooling + normalisation (models/patch_embedder.py:81-84, :66-72) ----------------
// Reference: w = pool_layer(y) (Conv2d 1x1 32->16, BN, ReLU, Conv2d 1x1 16->1, BN, ReLU), f = mean(y * w) over the
// 7x20 map, desc = F.normalize(f), equi = F.normalize(y, dim=channel) -- five library launches over [P,32,7,20].
// Here: one workgroup per patch, the 17.9 KB map sits in LDS; one lane per position evaluates the two 1x1
// convolutions (BN folded) and the channel norm, then the map is written back normalised and reduced to the
// 32-vector.  HBM traffic: 17.9 KB in, 17.9 KB + 128 B out per patch.
#define DH_THREADS 256
#define DH_C 32
#define DH_HID 16

#define DH_NPARAM (DH_HID * DH_C + DH_HID + DH_HID + 1)
// params (DEVICE, DH_NPARAM floats): w0[16][32] (pool_layer.0 with pool_layer.1 BatchNorm folded), b0[16],
// w3[16] (pool_layer.3 with pool_layer.4 folded), b3

// The head on a map that already sits in LDS (ys[32][140] fp32; wgt, nrm [140], fs [32], hp [DH_NPARAM + 3] LDS scratch): used by
// k_desc_head below and, fused behind the last layer, by k_cyl_net_h3 (csrc/convnet_h3.hip) -- one body, identical results.
// All DH_THREADS threads of the workgroup call it; the caller has synchronised after filling ys.
typedef __attribute__((address_space(3))) float dh_lds;        // LDS-qualified: ds_read / ds_write in every caller (a generic pointer makes them flat accesses)
typedef __attribute__((address_space(3))) f32x4 dh_lds4;
__device__ __forceinline__ void dh_body(const dh_lds* ys, dh_lds* wgt, dh_lds* nrm, dh_lds* fs, dh_lds* hp, const float* __restrict__ params,
                                        float* __restrict__ desc, float* __restrict__ equi, int patch, int tid)
{
    // w0 [hidden][channel] lands transposed, [channel][hidden]: the 16 weights of a channel are four 16-byte broadcast reads
    // in the loop below (one ds_read_b32 per weight made 512 LDS instructions per lane: 3.4 -> 2.x ms per 320 000 patches)
    for (int i = tid; i < DH_NPARAM; i += DH_THREADS)
        hp[i < DH_HID * DH_C ? (i % DH_C) * DH_HID + i / DH_C : i] = params[i];
    __syncthreads();
    const dh_lds* w0 = hp;
    const dh_lds* b0 = hp + DH_HID * DH_C;
    const dh_lds* w3 = b0 + DH_HID;
    if (tid < CN_POS) {
        float h[DH_HID];
#pragma unroll
        for (int j = 0; j < DH_HID; j++) h[j] = b0[j];
        float ss = 0.f;
        for (int c = 0; c < DH_C; c++) {
            const float v = ys[c * CN_POS + tid];
            ss += v * v;
            const dh_lds4* wr = reinterpret_cast<const dh_lds4*>(w0 + c * DH_HID);    // LDS broadcast reads, 16 bytes each
            const f32x4 wq[4] = { wr[0], wr[1], wr[2], wr[3] };
#pragma unroll
            for (int j = 0; j < DH_HID; j++) h[j] += wq[j >> 2][j & 3] * v;
        }
        float a = w3[DH_HID];                                                     // b3
#pragma unroll
        for (int j = 0; j < DH_HID; j++) a += w3[j] * fmaxf(h[j], 0.f);
        wgt[tid] = fmaxf(a, 0.f);
        nrm[tid] = fmaxf(sqrtf(ss), 1e-12f);                     // F.normalize: x / max(||x||, eps)
    }
    __syncthreads();
    float* eq = equi + (size_t)patch * DH_C * CN_POS;
    static_assert(CN_POS % 4 == 0, "a float4 of the map stays inside one channel");
    for (int i = tid; i < DH_C * CN_POS / 4; i += DH_THREADS) {          // four positions at a time: one index computation, 16-byte accesses
        const f32x4 v = reinterpret_cast<const dh_lds4*>(ys)[i];
        const f32x4 n = reinterpret_cast<const dh_lds4*>(nrm)[i % (CN_POS / 4)];
        reinterpret_cast<f32x4*>(eq)[i] = (f32x4){ v[0] / n[0], v[1] / n[1], v[2] / n[2], v[3] / n[3] };
    }
    {   // f[c] = mean_pos y[c][pos] * w[pos]: 8 lanes per channel
        const int c = tid >> 3, sub = tid & 7;
        float acc = 0.f;
        for (int pos = sub; pos < CN_POS; pos += 8) acc += ys[c * CN_POS + pos] * wgt[pos];
        acc += __shfl_xor(acc, 1, WAVE);
        acc += __shfl_xor(acc, 2, WAVE);
        acc += __shfl_xor(acc, 4, WAVE);
        if (sub == 0) fs[c] = acc / (float)CN_POS;
    }
    __syncthreads();
    if (tid < DH_C) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < DH_C; c++) ss += fs[c] * fs[c];
        desc[(size_t)patch * DH_C + tid] = fs[tid] / fmaxf(sqrtf(ss), 1e-12f);
    }
}

__global__ void __launch_bounds__(DH_THREADS) k_desc_head(const float* __restrict__ y, const float* __restrict__ params,
                                                       float* __restrict__ desc, float* __restrict__ equi)
{
    __shared__ __attribute__((aligned(16))) float ys[DH_C * CN_POS];
    __shared__ __attribute__((aligned(16))) float wgt[CN_POS], nrm[CN_POS], fs[DH_C];
    __shared__ __attribute__((aligned(16))) float hp[DH_NPARAM + 3];
    const int patch = blockIdx.x, tid = threadIdx.x;
    const f32x4* src = reinterpret_cast<const f32x4*>(y + (size_t)patch * DH_C * CN_POS);
    for (int i = tid; i < DH_C * CN_POS / 4; i += DH_THREADS) reinterpret_cast<f32x4*>(ys)[i] = src[i];
    dh_body((dh_lds*)ys, (dh_lds*)wgt, (dh_lds*)nrm, (dh_lds*)fs, (dh_lds*)hp, params, desc, equi, patch, tid);
}

// The head for the patches with only_if[p] != 0 only; y may BE equi (the map is in LDS before the first store): no __restrict__ here.
__global__ void __launch_bounds__(DH_THREADS) k_desc_head_masked(const float* y, const float* __restrict__ params, float* __restrict__ desc,
                                                              float* equi, const int* __restrict__ only_if)
{
    __shared__ __attribute__((aligned(16))) float ys[DH_C * CN_POS];
    __shared__ __attribute__((aligned(16))) float wgt[CN_POS], nrm[CN_POS], fs[DH_C];
    __shared__ __attribute__((aligned(16))) float hp[DH_NPARAM + 3];
    const int patch = blockIdx.x, tid = threadIdx.x;
    if (only_if[patch] == 0) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(y + (size_t)patch * DH_C * CN_POS);
    for (int i = tid; i < DH_C * CN_POS / 4; i += DH_THREADS) reinterpret_cast<f32x4*>(ys)[i] = src[i];
    __syncthreads();
    dh_body((dh_lds*)ys, (dh_lds*)wgt, (dh_lds*)nrm, (dh_lds*)fs, (dh_lds*)hp, params, desc, equi, patch, tid);
}

// y f32[np,32,140] -> desc f32[np,32], equi f32[np,32,140].  params: DEVICE f32[545] = w0 [16][32], b0 [16], w3 [16], b3 (BN folded).
extern "C" int buf_descriptor_head(const float* y, int npatch, const float* params, float* desc, float* equi, void* stream)
{
    BUF_REQUIRE(npatch >= 0, BUF_EINVAL, "buf_descriptor_head: npatch=%d", npatch);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(y && params && desc && equi, BUF_EINVAL, "buf_descriptor_head: null argument");
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, (2.0 * DH_C * CN_POS * 4 + 4.0 * DH_C) * npatch, BUF_TIMED_DESC_HEAD);
    k_desc_head<<<npatch, DH_THREADS, 0, (hipStream_t)stream>>>(y, params, desc, equi);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
