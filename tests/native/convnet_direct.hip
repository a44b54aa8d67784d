// TEST INFRASTRUCTURE -- the DIRECT-form Cylindrical_Net kernel (round 1/2's product kernel, replaced by the Winograd-domain
// kernel of buffer_amd/csrc/convnet_wg.hip): kept as the k-ordered fp32 cross-check of tests/test_model_gpu.py, built into
// tests/native/libbuffer_direct.so (tests/native/build.py), NOT part of libbuffer_hip.so.
//
// A11 dense part -- Cylindrical_Net (models/patchnet.py:15-85) as ONE fused fp32-MFMA kernel.
//
// Reference: Conv3d(16->64, 3x3x3, radial depth 3 -> 1) + 7 x Conv2d 3x3 (64,128,128,64,64,32,32), each
// preceded by torch.cat padding (circular in azimuth, zeros in elevation, utils/common.py:265-310) and
// followed by BatchNorm(affine=False)+ReLU (except the last) -- 8 library convolutions, 16 concatenations
// and 14 element-wise passes over [P,C,7,20] tensors per call.
//
// Here one workgroup owns one patch for the whole stack.  A layer is an implicit GEMM
//     out[n][m] = relu( sum_k A[m][k] * Wt[k][n] + b[n] ),  m = ele*20 + azi (140 positions, 9 tiles of 16),
//     k = s*Cin + c with s = ky*3 + kx,  A[m][k] = in[c][ele+ky-1][(azi+kx-1) mod 20]  (0 outside 0 <= ele < 7)
// on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).  Activations never leave LDS
// (one [128][144] fp32 buffer, 72 KB, rewritten in place after each layer -> two workgroups per CU); A fragments are gathered from LDS with the padding folded into
// the address (one ds_read_b32 per fragment; invalid taps read a zero column), B fragments (BN-folded weights, [K][Cout]
// row-major, 1.7 MB for the whole net, L2-resident) stream from global memory through a register ring.
// HBM traffic per patch: 26.9 KB in, 17.9 KB out.
#include <stdlib.h>
#include "../../buffer_amd/csrc/core.hip"      // error string, LDS grant, timing hooks (this library is loaded on its own)
#include <type_traits>

#define CN_POS 140            // 7 elevation x 20 azimuth
#define CN_STR 144            // LDS channel stride: 144 mod 32 = 16 keeps the 4 channel groups of a fragment on disjoint banks
#define CN_MT 9               // ceil(140 / 16)
#define CN_MAXC 128
#define CN_LAYERS 8
#define CN_THREADS 256
#define CN_PF 4               // k-steps per software-pipeline group
#define CN_BUF (CN_MAXC * CN_STR)                 // one activation buffer
#define CN_TAB 160                                // tap-table row: 10 tiles of 16 output positions (the 5/4 split reaches tile 9)

typedef float f32x4 __attribute__((ext_vector_type(4)));
// The next group's operand loads are woven into the MFMA block of the current group: after every 2 MFMAs up to
// 6 VALU/SALU and 2 memory instructions (fastest measured placement at two workgroups per CU; round-1 experiments
// with loads in front of the block, wave priorities and stubbed operand loads are in the git history).
#define CYL_SCHED_TAIL                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < CN_PF * MT * NT; i_++) {                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x120, 2, 0);                              \
    }

struct CylNetParams {
    const float* wt[CN_LAYERS];     // [9*Cin][Cout], BN folded, in the MFMA B-operand tiling (blocks [K/16][Cout/16] of [lk][li][p])
    const float* bias[CN_LAYERS];   // [Cout]
    int cin[CN_LAYERS], cout[CN_LAYERS], relu[CN_LAYERS];
};

// One layer for the calling wavefront: M-tiles [mt0, mt0+MT) x N-tiles [nt0, nt0+NT) of 16x16.
// The k loop is software-pipelined in groups of CN_PF k-steps: while the MFMAs of group g run, the A
// fragments (LDS) and B fragments (global/L2) of group g+1 are already in flight.  All loads are
// unconditional: a prefetch past the end of a tap is clamped to the tap's last group.
template <int MT, int NT>
__device__ __forceinline__ void cyl_layer(const float* __restrict__ in, float* __restrict__ out_lds, float* __restrict__ out_glb,
                                          const unsigned short* __restrict__ postab, const float* __restrict__ wt,
                                          const float* __restrict__ bias, int cin, int cout, int relu, int mt0, int mt_cnt, int nt0)
{
    int lane = threadIdx.x & (WAVE - 1);
    // keep every lane-derived offset of this layer inside the layer: hoisted out of the layer loop (for all three
    // instantiations at once) they outlive 256 VGPRs and are spilled to scratch, 32 KB of HBM traffic per patch
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    // accumulators start at the bias of their output channel (C/D layout: column = lane & 15): no bias pass afterwards
    f32x4 acc[MT][NT];
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const float bv = bias[(nt0 + u) * 16 + li];
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t][u] = (f32x4){ bv, bv, bv, bv };
    }
    const int groups = cin >> 4;                     // groups of CN_PF (=4) k-steps per kernel tap
    const int ntot = cout >> 4;                      // N-tiles of the layer
    const float* wrow = wt + ((size_t)nt0 * 64 + lane) * 4;      // lane's 16 bytes of block (group 0, N-tile nt0)
    const float* ib = in + lk * CN_STR;              // lane's channel row of k-step 0
    // One kernel tap over the wavefront's M-tiles [T0, T1) (compile-time bounds): ping-pong register sets;
    // sched_barriers pin "issue loads of the next group" in front of "MFMAs of the current group" (the compiler
    // otherwise rotates the loop and exposes the latency).
    auto run_tap = [&](auto t0c, auto t1c, const int (&io)[MT], const float* ws) __attribute__((always_inline)) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
        float a0[CN_PF][MT], b0[CN_PF][NT], a1[CN_PF][MT], b1[CN_PF][NT];
#define CYL_LOAD(A, B, G)                                                                              \
    {                                                                                                  \
        const int g_ = (G) < groups ? (G) : groups - 1;   /* prefetch past the tap re-reads its last group */ \
        const float* wn_ = ws + (size_t)g_ * ntot * 256;      /* MFMA-tiled weights: one 16-byte load per N-tile */ \
        _Pragma("unroll") for (int u = 0; u < NT; u++) {                                               \
            const f32x4 bv_ = *reinterpret_cast<const f32x4*>(wn_ + u * 256);                          \
            _Pragma("unroll") for (int p = 0; p < CN_PF; p++) B[p][u] = bv_[p];             \
        }                                                                                              \
        _Pragma("unroll") for (int p = 0; p < CN_PF; p++) {                                            \
            _Pragma("unroll") for (int t = T0; t < T1; t++) A[p][t] = ib[io[t] + (g_ * CN_PF + p) * 4 * CN_STR]; \
        }                                                                                              \
    }
#define CYL_MMA(A, B)                                                                                  \
    _Pragma("unroll") for (int p = 0; p < CN_PF; p++) {                                                \
        _Pragma("unroll") for (int t = T0; t < T1; t++) {                                              \
            const float av_ = A[p][t];      /* zero padding comes from the address, no select here */ \
            _Pragma("unroll") for (int u = 0; u < NT; u++)                                             \
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, B[p][u], acc[t][u], 0, 0, 0);    \
        }                                                                                              \
    }
        CYL_LOAD(a0, b0, 0)
#pragma unroll 1
        for (int g = 0; g < groups; g += 2) {
            __builtin_amdgcn_sched_barrier(0);
            CYL_LOAD(a1, b1, g + 1)
            CYL_MMA(a0, b0)
            CYL_SCHED_TAIL
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < groups) {
                CYL_LOAD(a0, b0, g + 2)
                CYL_MMA(a1, b1)
                CYL_SCHED_TAIL
            }
        }
#undef CYL_LOAD
#undef CYL_MMA
    };
    using std::integral_constant;
#pragma unroll 1
    for (int s = 0; s < 9; s++) {
        const int ky = s / 3 - 1;
        int io[MT];                                                     // lane's position in its channel row, per tile
#pragma unroll
        for (int t = 0; t < MT; t++) io[t] = postab[s * CN_TAB + (mt0 + t) * 16 + li];      // tiles past mt_cnt: zero column
        const float* ws = wrow + (size_t)s * groups * ntot * 256;
        // M-tiles whose 16 positions all read the zero elevation padding under this tap contribute exactly 0 and are
        // skipped: tile 0 (positions 0..15, elevation row 0) for ky = -1, tile 8 (128..143: row 6 + the 4 padding
        // positions) for ky = +1 -- 6 of 81 (tap, tile) pairs; a 4-tile wavefront of the 32-channel layers also
        // drops its unused fifth tile.
        const bool skip_first = ky < 0 && mt0 == 0;
        const int t1 = mt_cnt - ((ky > 0 && mt0 + mt_cnt == CN_MT) ? 1 : 0);
        if (skip_first) run_tap(integral_constant<int, 1>{}, integral_constant<int, MT>{}, io, ws);
        else if (t1 == MT) run_tap(integral_constant<int, 0>{}, integral_constant<int, MT>{}, io, ws);
        else if (t1 == MT - 1) run_tap(integral_constant<int, 0>{}, integral_constant<int, MT - 1>{}, io, ws);
        else if constexpr (MT == 5) run_tap(integral_constant<int, 0>{}, integral_constant<int, MT - 2>{}, io, ws);
    }
    // The layer's output overwrites its input IN PLACE (one 72 KB LDS buffer per workgroup, so two workgroups
    // fit a CU and one computes while the other loads/stores): every wavefront has finished reading `in` here.
    __syncthreads();
    // epilogue: ReLU + store; C/D layout: col = lane & 15 (n), rows (lane >> 4)*4 + r (m).  Only the last M-tile has
    // rows past position 139 (its lanes with lk == 3); the branches below are wave-uniform except that one.
    const bool last_rows_ok = lk * 4 + 128 < CN_POS;
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const int n = (nt0 + u) * 16 + li;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (t >= mt_cnt) continue;
            const int m = (mt0 + t) * 16 + lk * 4;
            f32x4 v = acc[t][u];
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (mt0 + t == CN_MT - 1 && !last_rows_ok) continue;
            // two typed stores (global_store / ds_write), not one flat store through a selected pointer
            if (out_glb) *reinterpret_cast<f32x4*>(out_glb + (size_t)n * CN_POS + m) = v;
            else         *reinterpret_cast<f32x4*>(out_lds + n * CN_STR + m) = v;
        }
    }
}

__global__ void __launch_bounds__(CN_THREADS, 2) k_cyl_net(const float* __restrict__ x, CylNetParams P, float* __restrict__ y)
{
    extern __shared__ float lds[];                   // [128][144] fp32 (140 positions + bank padding), then the tap table
    float* buf0 = lds;
    // postab[tap][m]: position (within a channel row) that output position m reads under kernel tap (ky,kx): circular
    // azimuth, the zero column 140 for the zero elevation padding and for the tile padding m >= 140.  Built once per
    // workgroup; a tap then costs one ds_read_u16 per M-tile instead of ~12 VALU.
    unsigned short* postab = reinterpret_cast<unsigned short*>(lds + CN_BUF);
    for (int i = threadIdx.x; i < 9 * CN_TAB; i += CN_THREADS) {
        const int s = i / CN_TAB, m = i - s * CN_TAB;
        const int ky = s / 3 - 1, kx = s % 3 - 1;
        int y = m / 20 + ky, x = m % 20 + kx;
        x = x < 0 ? x + 20 : (x >= 20 ? x - 20 : x);
        postab[i] = (unsigned short)((m < CN_POS && y >= 0 && y < 7) ? y * 20 + x : CN_POS);
    }
    const int patch = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);     // wave-uniform: tile ranges branch on it
    {   // input: [16,3,7,20] = 48 folded channels x 140 positions, contiguous
        const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)patch * P.cin[0] * CN_POS);
        for (int i = threadIdx.x; i < P.cin[0] * (CN_POS / 4); i += CN_THREADS) {
            const int c = i / (CN_POS / 4), r = i - c * (CN_POS / 4);
            reinterpret_cast<f32x4*>(buf0 + c * CN_STR)[r] = src[i];
        }
    }
    // columns 140..143 of every channel row (the bank padding) stay zero for the whole kernel: out-of-range
    // elevation taps and unused tile rows read column 140 instead of being masked after the load
    if (threadIdx.x < CN_MAXC)
        *reinterpret_cast<f32x4*>(buf0 + threadIdx.x * CN_STR + CN_POS) = (f32x4){ 0.f, 0.f, 0.f, 0.f };
    __syncthreads();
    float* in = buf0;
    float* out = buf0;
#pragma unroll 1
    for (int l = 0; l < CN_LAYERS; l++) {
        const int cin = P.cin[l], cout = P.cout[l];
        float* glb = l == CN_LAYERS - 1 ? y + (size_t)patch * cout * CN_POS : nullptr;
        if (cout == 128)      cyl_layer<CN_MT, 2>(in, out, glb, postab, P.wt[l], P.bias[l], cin, cout, P.relu[l], 0, CN_MT, 2 * w);
        else if (cout == 64)  cyl_layer<CN_MT, 1>(in, out, glb, postab, P.wt[l], P.bias[l], cin, cout, P.relu[l], 0, CN_MT, w);
        else {                // cout == 32: two wavefronts share an N-tile and split the M-tiles 5 / 4
            const int half = w & 1;
            cyl_layer<5, 1>(in, out, glb, postab, P.wt[l], P.bias[l], cin, cout, P.relu[l], half * 5, half ? 4 : 5, w >> 1);
        }
        __syncthreads();
    }
}

// x f32[np,48,140] (= [np,16,3,7,20]) -> y f32[np,32,140].  wt/bias: DEVICE pointers per layer, passed in host arrays.
extern "C" int buf_test_cylindrical_net_direct(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host,
                                   const int* cin_host, const int* cout_host, const int* relu_host, float* y, void* stream)
{
    BUF_REQUIRE(npatch >= 0, BUF_EINVAL, "buf_test_cylindrical_net_direct: npatch=%d", npatch);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(x && y && wt_host && bias_host && cin_host && cout_host && relu_host, BUF_EINVAL, "buf_test_cylindrical_net_direct: null argument");
    CylNetParams P;
    for (int l = 0; l < CN_LAYERS; l++) {
        P.wt[l] = wt_host[l]; P.bias[l] = bias_host[l];
        P.cin[l] = cin_host[l]; P.cout[l] = cout_host[l]; P.relu[l] = relu_host[l];
        BUF_REQUIRE(P.wt[l] && P.bias[l], BUF_EINVAL, "buf_test_cylindrical_net_direct: null weights for layer %d", l);
        BUF_REQUIRE(P.cin[l] % 16 == 0 && P.cin[l] <= CN_MAXC && (P.cout[l] == 32 || P.cout[l] == 64 || P.cout[l] == 128),
                    BUF_EINVAL, "buf_test_cylindrical_net_direct: layer %d has unsupported widths %d -> %d", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(l == 0 || P.cin[l] == P.cout[l - 1], BUF_EINVAL, "buf_test_cylindrical_net_direct: layer %d width mismatch", l);
    }
    size_t lds = sizeof(float) * CN_BUF + sizeof(unsigned short) * 9 * CN_TAB;
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds((const void*)k_cyl_net, lds, grant)) return rc;
    // algorithmic flops of this launch: 2 * 140 positions * sum over layers of 9*Cin*Cout, per patch (0.1186 GFLOP)
    double macs = 0;
    for (int l = 0; l < CN_LAYERS; l++) macs += 9.0 * P.cin[l] * P.cout[l];
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 2.0 * CN_POS * macs * npatch, BUF_TIMED_CYL_NET);
    k_cyl_net<<<npatch, CN_THREADS, lds, (hipStream_t)stream>>>(x, P, y);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}


