// A11 dense part -- Cylindrical_Net (models/patchnet.py:15-85) as ONE fused fp32-MFMA kernel in the Winograd F(2x2, 3x3)
// domain.
//
// Every layer of the stack is a 3x3 correlation over the 7 x 20 (elevation x azimuth) map (layer 0's radial depth folds
// into 48 input channels), circular in azimuth and zero-padded in elevation (utils/common.py:265-310).  With the map cut
// into 4 x 10 tiles of 2 x 2 outputs,
//     Y = A^T [ sum_c (G g G^T)[c] (.) (B^T d[c] B) ] A          (d: the tile's 4 x 4 input window)
// turns the 9-tap implicit GEMM (9 x 144 rows per channel pair) into 16 component GEMMs over tile rows.  The bottom tile row
// (output row 6; row 7 does not exist and window rows 7, 8 are padding) is the plain two-tap form in elevation for the M-tile
// that holds nothing else: 8 components (wg_round_bottom).  16 + 16 + 8 = 40 matrix instructions per (4 input, 16 output
// channels) against 75 of the direct form.  All arithmetic stays fp32 (v_mfma_f32_16x16x4_f32); the filter transform is done
// once on the host in fp64.
//
// One workgroup owns one patch for the whole stack.  Activations live in ONE LDS buffer [128][160] (rows of 22 = 20 azimuth
// columns + the two circular halo columns, so a window row is one address + immediate offsets; 4 zeros per channel serve
// every read of the elevation padding), rewritten in place after each layer -> two workgroups per CU (2 x 80 KB).
//
// What bounds the kernel (round 3, tools/micro/mfma_group.hip): v_mfma_f32_16x16x4_f32 runs on the f32 vector lanes, so VALU
// work does NOT hide behind it -- with two waves per SIMD a matrix slot costs 32.4 + 3.6 x (VALU instructions per MFMA) cycles
// (alternating single MFMAs and VALU instructions is worse still: 52+; groups of >= 4 MFMAs are enough).  The input transform
// B^T d B costs 8 adds per 4 component operands, i.e. 2.0 VALU per MFMA if every N-tile (16 output channels) transforms for itself.
//   * 128 output channels (52 % of the MFMAs): a wavefront owns an N-tile PAIR and one transform feeds the 8 MFMAs of both
//     (1.06 VALU per MFMA in the loop).  The layer is in place -- the outputs of a pair (80 registers) wait until every
//     wavefront has read its input, in accumulation registers -- and the M-tiles run in two rounds: both Winograd tiles (64
//     accumulator registers), then the bottom row.
//   * 64 output channels: pairs as well, by splitting M -- a wavefront owns an N-tile pair for ONE Winograd M-tile plus the bottom
//     row of one N-tile (32 + 8 MFMAs per k-step for every wavefront, nothing to exchange).  32 output channels: the same inside
//     each half of K (the two wavefronts of a K split hand their partial sums over through the free upper rows of the buffer).
// The filter stream: weights come through a buffer resource with wavefront-uniform offsets (a global_load_dwordx4 with a
// 64-bit VGPR address costs ~50 cycles of SIMD issue beside MFMAs, an SGPR-based one ~10: tools/micro/mfma_vmem.hip), tiled
// so that a wavefront's k-steps are contiguous, two k-steps of weights in registers per N-tile.
// A pass walks K for one row component i of the transform: per (k-step, M-tile) step it reads two window rows (four
// ds_read_b64), forms the four column components (eight plain adds) and issues 4 MFMAs per N-tile; the accumulators run on
// through the four passes and after each pass the column transform of the running sum is folded into the 2 x 2 outputs in
// registers (the C/D layout keeps a tile in one lane).
// LDS banking (ds_read_b64: bank = word mod 64, lanes 0-31 and 32-63 served separately): the 40 tiles are dealt to the
// three M-tiles such that the 16 lanes of a k-step channel cover 32 distinct banks, and the channel stride 160 = 32 mod 64
// puts the second channel of the half-wave on the other 32:
//     M-tile 0: tile row 0 (10) + tile row 1, columns 4..9     M-tile 1: tile row 2 (10) + row 1, columns 0..3 + row 3, columns 8, 9
//     M-tile 2: tile row 3, columns 0..7 (8 lanes idle)
#include "common.h"
#include <type_traits>

#define WG_CS 160             // channel stride in floats: 7 x 22 + 4 zeros + 2 dump words
#define WG_ROW 22             // [col 19][col 0 .. col 19][col 0]
#define WG_ZERO 154           // zeros read by every window row outside 0 <= elevation < 7
#define WG_MAXC 128
#define WG_LAYERS 8
#define WG_THREADS 256
#define WG_BUF (WG_MAXC * WG_CS)
#define WG_KSTEP (16 * WG_CS)  // bytes between k-steps (4 channels)
#ifndef WG_PIN_FOLD
#define WG_PIN_FOLD 1         // pin the folded outputs after every pass (0: k_cyl_net_wg leaves them to the compiler, +0.8 %)
#endif
#define WG_BLOCKS 16          // filter components per (input, output) channel
#ifdef WG_EXP_NOBAR
#define WG_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define WG_SYNC() __syncthreads()
#endif

typedef float wgf4 __attribute__((ext_vector_type(4)));
typedef float wgf2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) wgf2* wg_lds_f2;

struct CylWgParams {
    const float* wt[WG_LAYERS];     // Winograd-domain weights, [i][k-step][N-tile][lane][j] (ops.winograd_tile_weights)
    const float* bias[WG_LAYERS];   // [Cout]
    int cin[WG_LAYERS], cout[WG_LAYERS], relu[WG_LAYERS];
    const int* only_if;             // nullable: workgroup p runs only if only_if[p] != 0 (the fp32 re-run of buf_cylindrical_net_split_safe)
#ifdef WG_STAMP
    long long* stamps;              // development build (-DWG_STAMP): [workgroup][wave][20] s_memtime at the layer boundaries
#endif
};
#ifdef WG_STAMP
__device__ long long* wg_stamp_ptr;
#define WG_STAMP_IN(SLOT) if ((threadIdx.x & 63) == 0) wg_stamp_ptr[((size_t)blockIdx.x * 4 + threadIdx.x / 64) * 20 + (SLOT)] = __builtin_amdgcn_s_memtime();
#define WG_STAMP_AT(SLOT) if ((threadIdx.x & 63) == 0) P.stamps[((size_t)blockIdx.x * 4 + w) * 20 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define WG_STAMP_AT(SLOT)
#define WG_STAMP_IN(SLOT)
#endif

// tile (ty, tx) held by row idx of M-tile t (branch-free)
__device__ __forceinline__ bool wg_tile(int t, int idx, int& ty, int& tx)
{
    if (t == 0) { const bool a = idx < 10; ty = a ? 0 : 1; tx = a ? idx : idx - 6; return true; }
    if (t == 1) { const bool a = idx < 10, b = idx < 14; ty = a ? 2 : (b ? 1 : 3); tx = a ? idx : (b ? idx - 10 : idx - 6); return true; }
    ty = 3; tx = idx & 7;
    return idx < 8;
}

// LDS byte address of window row a (0..3) of the tile that lane (li, lk) holds in M-tile t, for the lane's channel of k-step 0
__device__ __forceinline__ unsigned wg_row_addr(unsigned act_addr, int t, int a, int li, int lk)
{
    int ty, tx;
    const bool ok = wg_tile(t, li, ty, tx);
    const int row = 2 * ty - 1 + a;
    return act_addr + 4u * (unsigned)(lk * WG_CS + ((ok && row >= 0 && row <= 6) ? row * WG_ROW + 2 * tx : WG_ZERO));
}

// Single adds as opaque instructions: left to itself the SLP vectoriser packs the transform into v_pk_add_f32 with a
// v_mov per operand (a packed add costs two plain ones on the issue port that the matrix pipe shares).
__device__ __forceinline__ float wg_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wg_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Window rows (A1, A2) of row component I and the number of M-tiles it runs over in [T0, T1)
__host__ __device__ constexpr int wg_a1(int I) { return I == 0 ? 0 : 1; }
__host__ __device__ constexpr int wg_a2(int I) { return I == 3 ? 3 : 2; }
__host__ __device__ constexpr int wg_te(int I, int T1) { return (I == 3 && T1 == 3) ? 2 : T1; }   // M-tile 2: bottom tile row only, component 3 unused

// Does the pass of row component I (M-tiles [T0, T1)) hand its successor I + 1 the first two LDS steps?  Only where steps 0
// and 1 have the same (k-step, M-tile) shape in both passes.
__host__ __device__ constexpr bool wg_chains(int I, int T0, int T1)
{
    if (I < 0 || I >= 3) return false;
    const int nt = wg_te(I, T1) - T0, ntn = wg_te(I + 1, T1) - T0;
    return nt >= 1 && ntn >= 1 && (1 / nt) == (1 / ntn) && (1 % nt) == (1 % ntn);
}

// LDS reads of one step.  Two ds_read_b64 per row: the empty asm keeps the compiler from fusing them into the half-rate
// ds_read2_b64.
#define WG_LOAD2(DST, A0, A1_, OFS)                                                                       \
    {                                                                                                     \
        DST[0] = *(wg_lds_f2)(size_t)((A0) + (OFS));                                                      \
        asm volatile("" : "+v"(A0));                                                                      \
        DST[1] = *(wg_lds_f2)(size_t)((A0) + (OFS) + 8);                                                  \
        asm volatile("" : "+v"(A0));                                                                      \
        DST[2] = *(wg_lds_f2)(size_t)((A1_) + (OFS));                                                     \
        asm volatile("" : "+v"(A1_));                                                                     \
        DST[3] = *(wg_lds_f2)(size_t)((A1_) + (OFS) + 8);                                                 \
        asm volatile("" : "+v"(A1_));                                                                     \
    }

// one window row only (the bottom-row form)
#define WG_LOAD1(DST, A0, OFS)                                                                            \
    {                                                                                                     \
        DST[0] = *(wg_lds_f2)(size_t)((A0) + (OFS));                                                      \
        asm volatile("" : "+v"(A0));                                                                      \
        DST[1] = *(wg_lds_f2)(size_t)((A0) + (OFS) + 8);                                                  \
        asm volatile("" : "+v"(A0));                                                                      \
    }
#define WG_LOAD(BOT_, DST, A0, A1_, OFS)                                                                  \
    {                                                                                                     \
        if constexpr (BOT_) WG_LOAD1(DST, A0, OFS) else WG_LOAD2(DST, A0, A1_, OFS)                       \
    }

// A lane's 16 bytes of a weight block through a buffer resource: wavefront-uniform byte offset (SGPR) + the lane's 32-bit offset.
// The form matters: global_load_dwordx4 with a 64-bit VGPR address costs ~50 cycles of the SIMD's issue beside MFMAs, an
// SGPR-based address ~10 (tools/micro/mfma_vmem.hip) -- with per-lane pointers the filter stream took 8 % of the kernel.
typedef unsigned wgu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_weights(const float* wt)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, 0x7fffffff, 0x00027000);
}
__device__ __forceinline__ wgf4 wg_ldw(__amdgpu_buffer_rsrc_t rs, unsigned uniform_float_ofs, unsigned lane_byte_ofs)
{
    return __builtin_bit_cast(wgf4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_byte_ofs, uniform_float_ofs * 4, 0));
}

// One row component I of the transform over `niter` x 4 k-steps for the M-tiles [T0, T1) and NN N-tiles that share the
// transformed operand: acc[n][t][j] += V_Ij(tile, c) * U_Ij(c, n).
// The (k-step, M-tile) steps form a three-stage software pipeline held in place by sched_barriers (the compiler otherwise
// hoists the loads to the top of the loop body and waits for them at once): step s issues the LDS reads of step s+2, runs
// the 4 NN MFMAs of step s back to back, then forms the column components of step s+1.  Four k-steps per iteration: their LDS
// offsets ride in the instructions, the weights (W[n][k-step]: one 16-byte load per lane) are reloaded in place half an
// iteration ahead, and the last iteration fetches the weights of whatever runs next (wp_next) instead of its own.  Every
// pass reads its own first two steps.  The N-tiles of a pair are neighbours in the weight tiling (256 floats apart).
// KSTEP: LDS bytes between k-steps (4 channels) -- k_cost_net's Winograd layers run the same passes over other map sizes.
// PRIMED: the previous pass already fetched this pass's first two steps into D; INEXT >= 0: this pass fetches those of the row
// component INEXT in its last iteration (the LDS latency of a pass start is then covered by the previous pass's MFMAs).
// Weights: W[n][k-step & 1], a ring of TWO k-steps per N-tile -- the registers of a k-step are reloaded with the k-step two
// further on (of this pass, or of whatever runs next: wp_next) as soon as its MFMAs are through, so every load is issued two
// k-steps before its use with half the registers of a whole-iteration buffer (the paired layers were spilling their outputs).
template <int I, int NN, int T0, int T1, bool PRIMED, int INEXT, bool BOT = false, unsigned KSTEP = WG_KSTEP, bool DIFF = false>
__device__ __forceinline__ void wg_pass(unsigned (&RA)[3][4], __amdgpu_buffer_rsrc_t rs, unsigned wp, unsigned wp_next, unsigned lofs, int niter,
                                        int wstride, wgf4 (&W)[NN][2], wgf4 (&acc)[NN][3][4], wgf2 (&D)[2][4], wgf4 (*Wb)[2] = nullptr, unsigned bofs = 0)
{
    constexpr int A1 = BOT ? I : wg_a1(I), A2 = BOT ? I : wg_a2(I);      // bottom-row form: I is the filter row = window row, no second row
    constexpr int TE = BOT ? T1 : wg_te(I, T1);
    constexpr int NT = TE > T0 ? TE - T0 : 1;
    if constexpr (TE <= T0) return;
    constexpr int A1N = BOT ? (INEXT < 0 ? 0 : INEXT) : wg_a1(INEXT < 0 ? 0 : INEXT), A2N = BOT ? A1N : wg_a2(INEXT < 0 ? 0 : INEXT);
    constexpr int NTN = (INEXT < 0 || BOT) ? NT : (wg_te(INEXT, T1) - T0);
    static_assert(INEXT < 0 || (NTN >= 1 && (1 / NT) == (1 / NTN) && (1 % NT) == (1 % NTN)), "steps 0 and 1 of the chained pass have this pass's shape");
    float V[2][4];
    unsigned P[NT][2];
#pragma unroll
    for (int t = 0; t < NT; t++) { P[t][0] = RA[T0 + t][A1]; P[t][1] = RA[T0 + t][A2]; }
    if constexpr (!PRIMED) {
#pragma unroll
        for (int g = 0; g < 2; g++) WG_LOAD(BOT, D[g], P[g % NT][0], P[g % NT][1], (g / NT) * KSTEP)
    }
    // row component (d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3), then the four column components
#define WG_XFORM(BUF)                                                                                     \
    {                                                                                                     \
        float r_[4];                                                                                      \
        _Pragma("unroll") for (int b = 0; b < 4; b++) {                                                   \
            const float da_ = D[BUF][b >> 1][b & 1];                                                      \
            if constexpr (BOT) r_[b] = da_;                                                               \
            else {                                                                                        \
                const float db_ = D[BUF][2 + (b >> 1)][b & 1];                                            \
                r_[b] = I == 1 ? wg_add(da_, db_) : (I == 2 ? wg_sub(db_, da_) : wg_sub(da_, db_));       \
            }                                                                                             \
        }                                                                                                 \
        V[BUF][0] = wg_sub(r_[0], r_[2]); V[BUF][1] = wg_add(r_[1], r_[2]);                               \
        V[BUF][2] = wg_sub(r_[2], r_[1]); V[BUF][3] = wg_sub(r_[1], r_[3]);                               \
    }
    WG_XFORM(0)
#pragma unroll 1
    for (int it = 0; it < niter; it++) {
        const bool more = it + 1 < niter;
        const unsigned wcur = wp + 4 * it * wstride;            // this iteration's k-steps 2, 3 ...
        const unsigned wn = more ? wcur + 4 * wstride : wp_next;   // ... and the k-steps 0, 1 of the next one (or of the next pass)
        const unsigned adv = more ? 4u * KSTEP : 0u;         // past the end: the iteration's own first steps again (unused)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4 * NT; s++) {
            const int t = T0 + s % NT, kk = s / NT, g = s + 2;
            if (g < 4 * NT) WG_LOAD(BOT, D[s & 1], P[g % NT][0], P[g % NT][1], (g / NT) * KSTEP)
            else {
                const int t2 = (g - 4 * NT) % NT;                // the next iteration's base from here on (this one no longer reads through it)
                if ((g - 4 * NT) / NT == 0) { P[t2][0] += adv; P[t2][1] += adv; }
                if constexpr (INEXT >= 0) {                      // last iteration: the next pass's first steps instead
                    unsigned q0 = more ? P[t2][0] : RA[T0 + t2][A1N], q1 = more ? P[t2][1] : RA[T0 + t2][A2N];
                    WG_LOAD(BOT, D[s & 1], q0, q1, ((g - 4 * NT) / NT) * KSTEP)
                } else
                    WG_LOAD(BOT, D[s & 1], P[t2][0], P[t2][1], ((g - 4 * NT) / NT) * KSTEP)
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DIFF) {                                // the filter block is the difference of two stored ones (wg_round_bottom)
                wgf4 wd[NN];
#pragma unroll
                for (int n = 0; n < NN; n++)
#pragma unroll
                    for (int j = 0; j < 4; j++) wd[n][j] = wg_sub(W[n][kk & 1][j], Wb[n][kk & 1][j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NN; n++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[n][t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[n][j], V[s & 1][j], acc[n][t][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int n = 0; n < NN; n++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[n][t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[n][kk & 1][j], V[s & 1][j], acc[n][t][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            WG_XFORM((s + 1) & 1)
            if (s % NT == NT - 1) {                              // k-step kk is through: its registers take the k-step two further on
#pragma unroll
                for (int n = 0; n < NN; n++)
                    W[n][kk & 1] = kk < 2 ? wg_ldw(rs, wcur + n * 256 + (kk + 2) * wstride, lofs) : wg_ldw(rs, wn + n * 256 + (kk - 2) * wstride, lofs);
                if constexpr (DIFF) {                            // (past the pass: the same offset behind whatever runs next -- unused, in bounds)
#pragma unroll
                    for (int n = 0; n < NN; n++)
                        Wb[n][kk & 1] = kk < 2 ? wg_ldw(rs, wcur + bofs + n * 256 + (kk + 2) * wstride, lofs) : wg_ldw(rs, wn + bofs + n * 256 + (kk - 2) * wstride, lofs);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 2; t < NT; t++) { P[t][0] += adv; P[t][1] += adv; }     // the first two M-tiles were advanced by the look-ahead steps
    }
#undef WG_XFORM
}

// All 16 components of NN N-tiles over the M-tiles [T0, T1): Y[n][t][u][v] = 2 x 2 outputs of the tiles of M-tile t (C/D
// layout: tile = lane's column).  On entry W holds the weights of the first four k-steps; on exit those at wp_after.
// The accumulators run on through the four passes (cleared once): after pass I they hold the sum of the row components
// 0..I, F_I = its column transform, and with s_i = F_i - F_(i-1)
//     output row 0 = s_0 + s_1 + s_2 = F_2,      output row 1 = s_1 - s_2 - s_3 = -F_0 + 2 F_1 - F_3
// -- three accumulator clears and half of the output-transform adds less per N-tile.  The bias (when this wavefront carries
// it: K-split layers add it once) enters component (1, 1)'s accumulator before pass 1: F_1, F_2, F_3 then carry it once in
// both columns.
template <int NN, int T0, int T1, unsigned KSTEP = WG_KSTEP, bool FENCE = false>
__device__ __forceinline__ void wg_round(unsigned (&RA)[3][4], __amdgpu_buffer_rsrc_t rs, unsigned wp, unsigned wp_after, unsigned lofs, int niter,
                                         int wstride, unsigned pstride, const float* __restrict__ bias_lane, wgf4 (&W)[NN][2], wgf4 (&Y)[NN][3][2][2])
{
    using std::integral_constant;
    wgf2 D[2][4];                                               // window rows A1 (columns 0-1, 2-3) and A2 of two steps in flight
    // the lane's four biases per N-tile: fetched per round and only alive through pass 0, when no partial output exists yet
    wgf4 bv[NN];
#pragma unroll
    for (int n = 0; n < NN; n++) bv[n] = bias_lane ? *reinterpret_cast<const wgf4*>(bias_lane + n * 16) : (wgf4){ 0.f, 0.f, 0.f, 0.f };
    wgf4 acc[NN][3][4];
    // (an opaque zero: folded into the first MFMA as the constant C = 0, the accumulators -- accumulation registers since the
    // paired layers park their outputs there -- become loop-carried copies: 36-68 v_accvgpr moves per iteration of every pass 0)
    float zero = 0.f;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int n = 0; n < NN; n++)
#pragma unroll
        for (int t = T0; t < T1; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[n][t][j] = (wgf4){ zero, zero, zero, zero };
    auto run = [&](auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        // component 3 does nothing for a round that only has M-tile 2: the pass before it hands over to whatever runs next
        constexpr int INEXT = (I == 3 || (I == 2 && wg_te(3, T1) <= T0)) ? 0 : I + 1;
        if constexpr (I == 1) {
#pragma unroll
            for (int n = 0; n < NN; n++)
#pragma unroll
                for (int t = T0; t < T1; t++) acc[n][t][1] += bv[n];
        }
        constexpr int ICHAIN = (INEXT != 0 && wg_chains(I, T0, T1)) ? INEXT : -1;      // the next row component of this round, if any
        wg_pass<I, NN, T0, T1, wg_chains(I - 1, T0, T1), ICHAIN, false, KSTEP>(RA, rs, wp + I * pstride, INEXT == 0 ? wp_after : wp + (I + 1) * pstride, lofs, niter, wstride, W, acc, D);
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);     // k_cost_net: the fold is not mixed into the next pass's start (spills)
#pragma unroll
        for (int n = 0; n < NN; n++)
#pragma unroll
            for (int t = T0; t < T1; t++) {
                if (I == 3 && t == 2) continue;
                if (I == 2 ? false : t == 2) continue;          // M-tile 2 (bottom tile row) only has output row 0 = F_2
                const wgf4 f0 = acc[n][t][0] + acc[n][t][1] + acc[n][t][2];
                const wgf4 f1 = acc[n][t][1] - acc[n][t][2] - acc[n][t][3];
                if constexpr (I == 0) { Y[n][t][1][0] = f0; Y[n][t][1][1] = f1; }                 // F_0 (enters with a minus sign below)
                else if constexpr (I == 1) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        Y[n][t][1][0][r] = __builtin_fmaf(2.f, f0[r], -Y[n][t][1][0][r]);
                        Y[n][t][1][1][r] = __builtin_fmaf(2.f, f1[r], -Y[n][t][1][1][r]);
                    }
                }
                else if constexpr (I == 2) { Y[n][t][0][0] = f0; Y[n][t][0][1] = f1; }
                else { Y[n][t][1][0] -= f0; Y[n][t][1][1] -= f1; }
                // the folded values exist HERE: left free, the compiler sinks these sums below the next pass's loop and carries
                // the accumulators of this pass (twice the registers) through it instead -- copies and, in k_cost_net, spills
                if constexpr (FENCE || WG_PIN_FOLD) {
                    if constexpr (I == 2) asm volatile("" : "+v"(Y[n][t][0][0]), "+v"(Y[n][t][0][1]));
                    else asm volatile("" : "+v"(Y[n][t][1][0]), "+v"(Y[n][t][1][1]));
                }
            }
    };
    run(integral_constant<int, 0>{});
    run(integral_constant<int, 1>{});
    run(integral_constant<int, 2>{});
    run(integral_constant<int, 3>{});
}

// M-tile 2 is the bottom tile row: its window rows 2 and 3 are elevation padding and its second output row (7) does not
// exist, so its one output row is the plain two-tap form in elevation
//     y[6] = A^T-columns of sum_c ( colB(d[5]) (.) colG(g[0]) + colB(d[6]) (.) colG(g[1]) ),
// two passes of four column components (8 tile-components instead of the 12 the row components 0..2 took, and no row
// combination in the input transform).  colG(g[0]) is the Winograd block i = 0; colG(g[1]) = U_1 - U_2 is formed in registers
// from the blocks 1 and 2 as they arrive (4 subtractions per N-tile and k-step, a second weight ring): stored as a fifth
// block it made the filter set 3.8 MB -- more than the 4 MB L2 of an XCD holds beside the activation stream, and the kernel's
// HBM-side traffic went from 47 to 85 KB per patch.  The four accumulators run through both taps; the bias starts in component 1.
template <int NN>
__device__ __forceinline__ void wg_round_bottom(unsigned (&RA)[3][4], __amdgpu_buffer_rsrc_t rs, unsigned wp, unsigned wp_after, unsigned lofs, int niter,
                                                int wstride, unsigned pstride, const float* __restrict__ bias_lane, wgf4 (&W)[NN][2], wgf4 (&Y)[NN][3][2][2])
{
    wgf2 D[2][4];
    wgf4 acc[NN][3][4];
    float zero = 0.f;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int n = 0; n < NN; n++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            acc[n][2][j] = (j == 1 && bias_lane) ? *reinterpret_cast<const wgf4*>(bias_lane + n * 16) : (wgf4){ zero, zero, zero, zero };
    wgf4 Wb[NN][2];                                             // block 2's first two k-steps (block 1's arrive through tap 0's look-ahead)
#pragma unroll
    for (int n = 0; n < NN; n++)
#pragma unroll
        for (int k = 0; k < 2; k++) Wb[n][k] = wg_ldw(rs, wp + 2 * pstride + n * 256 + k * wstride, lofs);
    wg_pass<0, NN, 2, 3, false, 1, true>(RA, rs, wp, wp + pstride, lofs, niter, wstride, W, acc, D);
    wg_pass<1, NN, 2, 3, true, -1, true, WG_KSTEP, true>(RA, rs, wp + pstride, wp_after, lofs, niter, wstride, W, acc, D, Wb, pstride);
#pragma unroll
    for (int n = 0; n < NN; n++) {
        Y[n][2][0][0] = acc[n][2][0] + acc[n][2][1] + acc[n][2][2];
        Y[n][2][0][1] = acc[n][2][1] - acc[n][2][2] - acc[n][2][3];
    }
}

// ReLU + store of one N-tile's outputs: into the activation buffer (with the circular halo copies) or to y[32][140].
// The weights are the MFMA's first operand, so the C/D layout has lane column li = the tile and rows lk * 4 + r = the
// output channel: a store instruction writes 16 different positions of a channel (distinct banks), not 16 channels at one
// position (one bank: the channel stride is a multiple of 32 words).
template <bool GLB, int NU = 2>
__device__ __forceinline__ void wg_store_tile(const wgf4 (&Yt)[2][2], int t, int nt, int relu, float* __restrict__ act, float* __restrict__ out_glb,
                                              int li, int lk)
{
    const int n0 = nt * 16 + lk * 4;
    const float lo = relu ? 0.f : -__builtin_inff();
    {
        int ty, tx;
        const bool valid = wg_tile(t, li, ty, tx);
#pragma unroll
        for (int u = 0; u < NU; u++) {                            // (M-tile 2 is tile row 3: NU = 1, its second output row is row 7)
            const int row = 2 * ty + u;
            const bool ok = valid && row < 7;
            if constexpr (GLB) {
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v0 = fmaxf(Yt[u][0][r], lo), v1 = fmaxf(Yt[u][1][r], lo);
                        // (plain stores: the L2 combines these 8-byte pieces into lines; as non-temporal stores they left as partial
                        // writes, 38 instead of 17.5 KB of WRITE_SIZE per patch)
                        *reinterpret_cast<wgf2*>(out_glb + (size_t)(n0 + r) * 140 + row * 20 + 2 * tx) = (wgf2){ v0, v1 };
                    }
                }
            } else {
                // straight-line code (a branch per store costs more than the store): lanes with nothing to write, and the halo
                // copies of the inner tiles, go to the two dump words behind the channel's zero area
                const int pos = ok ? row * WG_ROW + 2 * tx + 1 : WG_ZERO + 4;
                const int h0 = (ok && tx == 0) ? row * WG_ROW + 21 : WG_ZERO + 4;      // column 0 again behind column 19
                const int h1 = (ok && tx == 9) ? row * WG_ROW : WG_ZERO + 5;           // column 19 again in front of column 0
                float* p = act + n0 * WG_CS;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v0 = fmaxf(Yt[u][0][r], lo), v1 = fmaxf(Yt[u][1][r], lo);
                    p[r * WG_CS + pos] = v0; p[r * WG_CS + pos + 1] = v1;
                    p[r * WG_CS + h0] = v0; p[r * WG_CS + h1] = v1;
                }
            }
        }
    }
}

template <int T0, int T1, bool GLB>
__device__ __forceinline__ void wg_store(const wgf4 (&Y)[3][2][2], int nt, int relu, float* __restrict__ act, float* __restrict__ out_glb,
                                         int li, int lk)
{
#pragma unroll
    for (int t = T0; t < T1; t++) {
        if (t == 2) wg_store_tile<GLB, 1>(Y[t], t, nt, relu, act, out_glb, li, lk);
        else wg_store_tile<GLB, 2>(Y[t], t, nt, relu, act, out_glb, li, lk);
    }
}

// Window-row addresses of the lane: 3 M-tiles x 4 rows, channel lk of k-step k0
__device__ __forceinline__ void wg_addresses(const float* act, int k0, int li, int lk, unsigned (&RA)[3][4])
{
    const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)act + (unsigned)k0 * WG_KSTEP;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int a = 0; a < 4; a++) RA[t][a] = wg_row_addr(act_addr, t, a, li, lk);
}

// The 12 window-row addresses depend on the lane only (the map's geometry is the same in every layer): formed ONCE per kernel and parked in
// accumulation registers (round 6, WG_ADDR_PARK); a layer reads them back (one v_accvgpr_read each, plus one add where its K range does not
// start at channel 0) instead of re-deriving tile coordinates and padding selects -- ~110 vector instructions per layer and wavefront.
#ifndef WG_ADDR_PARK
#define WG_ADDR_PARK 1
#endif
struct WgAddrPark { float a[3][4]; };
__device__ __forceinline__ void wg_park_addresses(const float* act, int li, int lk, WgAddrPark& pk)
{
    const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)act;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const unsigned v = wg_row_addr(act_addr, t, a, li, lk);
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(pk.a[t][a]) : "v"(v));
        }
}
__device__ __forceinline__ unsigned wg_parked(const WgAddrPark& pk, int t, int a)
{
    unsigned v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(pk.a[t][a]));
    return v;
}

template <int NN, int K0, int K1, int KN>
__device__ __forceinline__ void wg_first_weights(__amdgpu_buffer_rsrc_t rs, unsigned wp, unsigned lofs, int wstride, wgf4 (&W)[NN][KN])
{
#pragma unroll
    for (int n = 0; n < NN; n++)
#pragma unroll
        for (int k = K0; k < K1; k++) W[n][k] = wg_ldw(rs, wp + n * 256 + k * wstride, lofs);
}

// One layer with 32 output channels (two N-tiles, four wavefronts): wavefront w owns N-tile w & 1 over ALL three M-tiles and the
// K half w >> 1; the upper half hands its partial sums (40 registers) to the lower one through the rows of the channels >= 64
// (free: Cin <= 64), which adds them and stores.  Round 2 split the M-tiles {0} | {1, 2} over the wavefront pair instead: 16 against
// 28 tile-components, i.e. the layer took 28 / 22 of its balanced time -- these two layers are 6.5 % of the MFMAs and were 11 % of
// the kernel.  The bias rides in the lower half's accumulators.
#define WG_XCH_C 64
template <bool GLB>
__device__ __forceinline__ void wg_layer_ksplit(float* __restrict__ act, float* __restrict__ out_glb, const float* __restrict__ wt,
                                                const float* __restrict__ bias, int cin, int cout, int relu, int w, const WgAddrPark& pk)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const int nt = w & 1, half = w >> 1;
    const int k4 = cin >> 2, kn = k4 >> 1, k0 = half * kn, wstride = 256;
    unsigned RA[3][4];
#if WG_ADDR_PARK
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int a = 0; a < 4; a++) RA[t][a] = wg_parked(pk, t, a) + (unsigned)k0 * WG_KSTEP;
#else
    wg_addresses(act, k0, li, lk, RA);
#endif
    const __amdgpu_buffer_rsrc_t rs = wg_weights(wt);
    const unsigned wp = (unsigned)nt * (WG_BLOCKS * cin * 16) + (unsigned)k0 * wstride;
    const unsigned lofs = lane * 16;
    wgf4 W[1][2];
    wg_first_weights<1, 0, 2>(rs, wp, lofs, wstride, W);
    wgf4 Y[1][3][2][2];
    wg_round<1, 0, 2>(RA, rs, wp, wp, lofs, kn >> 2, wstride, (unsigned)(k4 * wstride), half ? nullptr : bias + nt * 16 + lk * 4, W, Y);
    wg_round_bottom<1>(RA, rs, wp, wp, lofs, kn >> 2, wstride, (unsigned)(k4 * wstride), half ? nullptr : bias + nt * 16 + lk * 4, W, Y);
    wgf4* slot = reinterpret_cast<wgf4*>(act + WG_XCH_C * WG_CS) + nt * 640 + lane;                  // 10 x 64 float4 per N-tile
    if (half) {
#pragma unroll
        for (int q = 0; q < 10; q++) slot[q * 64] = Y[0][q >> 2][(q >> 1) & 1][q & 1];              // t = q / 4, u, column; t = 2 has u = 0 only
    }
    WG_STAMP_IN(18)
    WG_SYNC();                                       // partial sums are in place AND every wavefront has finished reading the input
    WG_STAMP_IN(19)
    if (!half) {
#pragma unroll
        for (int q = 0; q < 10; q++) Y[0][q >> 2][(q >> 1) & 1][q & 1] += slot[q * 64];
        wg_store<0, 3, GLB>(Y[0], nt, relu, act, out_glb, li, lk);
    }
}

// The same 32-channel layer with the M-split pair form inside each K half (WG_KSPLIT_PAIRS): wavefront (h, half) owns BOTH N-tiles
// for the Winograd M-tile h and the bottom row of N-tile h, over its half of K -- one transform per 8 MFMAs instead of 4, the same
// ten quads of partial sums to hand over.
template <bool GLB>
__device__ __forceinline__ void wg_layer_mksplit(float* __restrict__ act, float* __restrict__ out_glb, const float* __restrict__ wt,
                                                 const float* __restrict__ bias, int cin, int cout, int relu, int w, const WgAddrPark& pk)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const int h = w & 1, half = w >> 1;
    const int k4 = cin >> 2, kn = k4 >> 1, k0 = half * kn, wstride = 512;
    unsigned RA[3][4];
#if WG_ADDR_PARK
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const unsigned r0 = wg_parked(pk, 0, a), r1 = wg_parked(pk, 1, a);
        RA[0][a] = (h ? r1 : r0) + (unsigned)k0 * WG_KSTEP; RA[1][a] = RA[0][a]; RA[2][a] = wg_parked(pk, 2, a) + (unsigned)k0 * WG_KSTEP;
    }
#else
    {
        const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)act + (unsigned)k0 * WG_KSTEP;
#pragma unroll
        for (int a = 0; a < 4; a++) { RA[0][a] = wg_row_addr(act_addr, h, a, li, lk); RA[1][a] = RA[0][a]; RA[2][a] = wg_row_addr(act_addr, 2, a, li, lk); }
    }
#endif
    const __amdgpu_buffer_rsrc_t rs = wg_weights(wt);
    const unsigned wp = (unsigned)k0 * wstride;                            // [pair 0][i][k-step][n2][lane][j]
    const unsigned wpb = wp + (unsigned)h * 256;
    const unsigned lofs = lane * 16;
    const unsigned pstride = (unsigned)(k4 * wstride);
    wgf4 Y[2][3][2][2];
    wgf4 W2[2][2];
    wg_first_weights<2, 0, 2>(rs, wp, lofs, wstride, W2);
    wg_round<2, 0, 1>(RA, rs, wp, wpb, lofs, kn >> 2, wstride, pstride, half ? nullptr : bias + lk * 4, W2, Y);               // slot 0 = M-tile h
    wgf4 W1[1][2] = { { W2[0][0], W2[0][1] } };
    wgf4 Yb[1][3][2][2];
    wg_round_bottom<1>(RA, rs, wpb, wpb, lofs, kn >> 2, wstride, pstride, half ? nullptr : bias + h * 16 + lk * 4, W1, Yb);
    wgf4* slot = reinterpret_cast<wgf4*>(act + WG_XCH_C * WG_CS) + h * 640 + lane;                   // 10 x 64 float4 per wavefront pair
    if (half) {
#pragma unroll
        for (int q = 0; q < 8; q++) slot[q * 64] = Y[q >> 2][0][(q >> 1) & 1][q & 1];
        slot[8 * 64] = Yb[0][2][0][0]; slot[9 * 64] = Yb[0][2][0][1];
    }
    WG_SYNC();                                       // partial sums are in place AND every wavefront has finished reading the input
    if (!half) {
#pragma unroll
        for (int q = 0; q < 8; q++) Y[q >> 2][0][(q >> 1) & 1][q & 1] += slot[q * 64];
        Yb[0][2][0][0] += slot[8 * 64]; Yb[0][2][0][1] += slot[9 * 64];
        wg_store_tile<GLB, 2>(Y[0][0], h, 0, relu, act, out_glb, li, lk);
        wg_store_tile<GLB, 2>(Y[1][0], h, 1, relu, act, out_glb, li, lk);
        wg_store_tile<GLB, 1>(Yb[0][2], 2, h, relu, act, out_glb, li, lk);
    }
}

// One layer with 64 output channels (four N-tiles, four wavefronts): wavefront w owns the N-tile pair w & 1 for ONE of the
// M-tiles 0, 1 (w >> 1) -- the transform of a step feeds 8 MFMAs, as in the 128-channel layers -- and the bottom-row form of
// ONE N-tile of its pair: 32 + 8 = 40 matrix instructions per k-step for every wavefront, no partial sums to exchange.
// (One N-tile per wavefront over all M-tiles, the round-2 form, pays 1.8 transform instructions per MFMA instead of 1.0.)
__device__ __forceinline__ void wg_layer_msplit(float* __restrict__ act, const float* __restrict__ wt, const float* __restrict__ bias,
                                                int cin, int cout, int relu, int w, const WgAddrPark& pk)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const int pair = w & 1, h = w >> 1;
    const int k4 = cin >> 2, wstride = 512;
    unsigned RA[3][4];
#if WG_ADDR_PARK
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const unsigned r0 = wg_parked(pk, 0, a), r1 = wg_parked(pk, 1, a);
        RA[0][a] = h ? r1 : r0; RA[1][a] = RA[0][a]; RA[2][a] = wg_parked(pk, 2, a);
    }
#else
    {
        const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)act;
#pragma unroll
        for (int a = 0; a < 4; a++) { RA[0][a] = wg_row_addr(act_addr, h, a, li, lk); RA[1][a] = RA[0][a]; RA[2][a] = wg_row_addr(act_addr, 2, a, li, lk); }
    }
#endif
    const __amdgpu_buffer_rsrc_t rs = wg_weights(wt);
    const unsigned wp = (unsigned)pair * (WG_BLOCKS * cin * 32);           // [pair][i][k-step][n2][lane][j]
    const unsigned wpb = wp + (unsigned)h * 256;                           // the N-tile of the pair whose bottom row is this wavefront's
    const unsigned lofs = lane * 16;
    const unsigned pstride = (unsigned)(k4 * wstride);
    wgf4 Y[2][3][2][2];
    wgf4 W2[2][2];
    wg_first_weights<2, 0, 2>(rs, wp, lofs, wstride, W2);
    wg_round<2, 0, 1>(RA, rs, wp, wpb, lofs, k4 >> 2, wstride, pstride, bias + (2 * pair) * 16 + lk * 4, W2, Y);     // slot 0 = M-tile h
    wgf4 W1[1][2] = { { W2[0][0], W2[0][1] } };
    wgf4 Yb[1][3][2][2];
    wg_round_bottom<1>(RA, rs, wpb, wpb, lofs, k4 >> 2, wstride, pstride, bias + (2 * pair + h) * 16 + lk * 4, W1, Yb);
    WG_SYNC();                                       // every wavefront has finished reading the layer's input
    wg_store_tile<false, 2>(Y[0][0], h, 2 * pair, relu, act, nullptr, li, lk);
    wg_store_tile<false, 2>(Y[1][0], h, 2 * pair + 1, relu, act, nullptr, li, lk);
    wg_store_tile<false, 1>(Yb[0][2], 2, 2 * pair + h, relu, act, nullptr, li, lk);
}

// N-tiles per wavefront of a layer: 2 (the paired form below) or 1.  The filter tiling follows it (buf_winograd_tile_weights).
// Pairs for the 64-channel layers too (two wavefronts per pair, K split between them, partial sums exchanged through the dead
// half of the buffer) measured 1 % slower, before and after the filter stream became cheap: short K loops, a third barrier.
#ifndef WG_PAIR_NT2
#define WG_PAIR_NT2 1         // 128-channel layers: both Winograd M-tiles of a pair in ONE round (0: a round per M-tile, +0.9 %)
#endif
#ifndef WG_KSPLIT_PAIRS
#define WG_KSPLIT_PAIRS 1
#endif
__host__ __device__ constexpr int wg_group(int cin, int cout) { return (cout >= 64 || WG_KSPLIT_PAIRS) ? 2 : 1; }

// Outputs of M-tile T of both N-tiles of a pair -> accumulation registers (see wg_layer_pair)
template <int T>
__device__ __forceinline__ void wg_park(const wgf4 (&Y)[2][3][2][2], float (&park)[2][3][16])
{
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int q = 0; q < (T == 2 ? 8 : 16); q++)
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[n][T][q]) : "v"(Y[n][T][q >> 3][(q >> 2) & 1][q & 3]));
}

// One layer with 128 output channels: wavefront w owns the N-tile pair 2w, 2w+1 over the whole K, and the transform of a step
// feeds 8 MFMAs.  The held outputs (80 registers) wait in accumulation registers; the two Winograd M-tiles share one round
// (64 accumulators; a round per M-tile -- more passes, the pair's filters streamed a third time -- measured 0.9 % slower), the
// bottom row is the second.
__device__ __forceinline__ void wg_layer_pair(float* __restrict__ act, const float* __restrict__ wt, const float* __restrict__ bias,
                                              int cin, int cout, int relu, int pair, const WgAddrPark& pk)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
#ifdef WG_EXP_SAMEW_PAIR
    const int k4 = cin >> 2, wstride = relu >> 4;
#else
    const int k4 = cin >> 2, wstride = 512;
#endif
    unsigned RA[3][4];
#if WG_ADDR_PARK
    (void)li;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int a = 0; a < 4; a++) RA[t][a] = wg_parked(pk, t, a);
#else
    wg_addresses(act, 0, li, lk, RA);
#endif
    const __amdgpu_buffer_rsrc_t rs = wg_weights(wt);
    const unsigned wp = (unsigned)pair * (WG_BLOCKS * cin * 32);           // [pair][i][k-step][n2][lane][j]
    const unsigned lofs = lane * 16;
    const unsigned pstride = (unsigned)(k4 * wstride);
    wgf4 Y[2][3][2][2];
    const float* bv = bias + (2 * pair) * 16 + lk * 4;
    wgf4 W1[2][2];
    wg_first_weights<2, 0, 2>(rs, wp, lofs, wstride, W1);
    // The outputs of a finished round wait for the in-place barrier in ACCUMULATION registers (v_accvgpr_write / _read: the unified
    // file gives a wavefront 256 registers of either kind): left in VGPRs the compiler parked 19 of the 20 quads in scratch memory
    // (264 KB of HBM traffic per patch); the rounds then run in ~130 VGPRs with no scratch at all.
    float park[2][3][16];
#if WG_PAIR_NT2
    wg_round<2, 0, 2>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, pstride, bv, W1, Y);     // both Winograd M-tiles in one round
    wg_park<0>(Y, park);
    wg_park<1>(Y, park);
#else
    wg_round<2, 0, 1>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, pstride, bv, W1, Y);
    wg_park<0>(Y, park);
    wg_round<2, 1, 2>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, pstride, bv, W1, Y);
    wg_park<1>(Y, park);
#endif
    wg_round_bottom<2>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, pstride, bv, W1, Y);
    wg_park<2>(Y, park);
    WG_SYNC();                                       // every wavefront has finished reading the layer's input
    int lane_s = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane_s));                 // the store offsets are formed here, not kept (spilled) from the layer's start
#pragma unroll
    for (int n = 0; n < 2; n++) {
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int q = 0; q < (t == 2 ? 8 : 16); q++)
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(Y[n][t][q >> 3][(q >> 2) & 1][q & 3]) : "a"(park[n][t][q]));
        wg_store<0, 3, false>(Y[n], 2 * pair + n, relu, act, nullptr, lane_s & 15, lane_s >> 4);
    }
}

__device__ __forceinline__ void cyl_net_wg_body(const float* __restrict__ x, const CylWgParams& P, float* __restrict__ y, float* __restrict__ lds)
{
    float* act = lds;                                // [128][160]
    const int patch = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    WG_STAMP_AT(17)
#ifdef WG_STAMP
    if (threadIdx.x == 0) { P.stamps[(size_t)gridDim.x * 80 + blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime(); P.stamps[(size_t)gridDim.x * 80 + blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    {   // input [48][140] -> rows of 22 with the halo columns.  A row is five float4: every load of the patch is issued before the
        // first LDS store (one load at a time, each behind the previous one's stores, took 40 k cycles per patch: 5 % of the kernel)
        const wgf4* src = reinterpret_cast<const wgf4*>(x + (size_t)patch * P.cin[0] * 140);
        const int nq = P.cin[0] * 35;                               // float4 per patch (7 per thread for the 48 channels of this network)
        wgf4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = threadIdx.x + k * WG_THREADS;
            if (i < nq) v[k] = __builtin_nontemporal_load(src + i);    // streamed once: leave the L2 to the filters
        }
        auto put = [&](int i, wgf4 u) __attribute__((always_inline)) {
            const int c = i / 35, q = i - c * 35;
            const int row = q / 5, col = (q - row * 5) * 4;
            float* p = act + c * WG_CS + row * WG_ROW + col + 1;
            p[0] = u[0]; p[1] = u[1]; p[2] = u[2]; p[3] = u[3];
            if (col == 0) p[20] = u[0];
            if (col == 16) p[-17] = u[3];
        };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = threadIdx.x + k * WG_THREADS;
            if (i < nq) put(i, v[k]);
        }
        for (int i = threadIdx.x + 8 * WG_THREADS; i < nq; i += WG_THREADS) put(i, __builtin_nontemporal_load(src + i));   // wider first layers
    }
    for (int i = threadIdx.x; i < WG_MAXC * 4; i += WG_THREADS) act[(i >> 2) * WG_CS + WG_ZERO + (i & 3)] = 0.f;
    __syncthreads();
    WG_STAMP_AT(0)
    WgAddrPark pk;
#if WG_ADDR_PARK
    {
        int lane0 = threadIdx.x & (WAVE - 1);
        asm volatile("" : "+v"(lane0));
        wg_park_addresses(act, lane0 & 15, lane0 >> 4, pk);
    }
#endif
#pragma unroll 1
    for (int l = 0; l < WG_LAYERS; l++) {
        const int cin = P.cin[l], cout = P.cout[l];
        if (cout == 128) wg_layer_pair(act, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
        else if (cout == 64) {
            wg_layer_msplit(act, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
        }
#if WG_KSPLIT_PAIRS
        else if (l < WG_LAYERS - 1) wg_layer_mksplit<false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
        else wg_layer_mksplit<true>(act, y + (size_t)patch * cout * 140, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
#else
        else if (l < WG_LAYERS - 1) wg_layer_ksplit<false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
        else wg_layer_ksplit<true>(act, y + (size_t)patch * cout * 140, P.wt[l], P.bias[l], cin, cout, P.relu[l], w, pk);
#endif
        WG_STAMP_AT(2 * l + 1)
        WG_SYNC();
        WG_STAMP_AT(2 * l + 2)
    }
#ifdef WG_STAMP
    if (threadIdx.x == 0) { P.stamps[(size_t)gridDim.x * 80 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); P.stamps[(size_t)gridDim.x * 80 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

__global__ void __launch_bounds__(WG_THREADS, 2) k_cyl_net_wg(const float* __restrict__ x, CylWgParams P, float* __restrict__ y)
{
    extern __shared__ float lds[];
    cyl_net_wg_body(x, P, y, lds);
}

// The masked re-run of buf_cylindrical_net_split_safe: the same body for the patches the split-f16 kernel flagged (P.only_if[patch] != 0),
// nothing else.  Its own kernel NAME: the launch covers the full grid and is empty in the normal case -- under the name of k_cyl_net_wg
// it would be averaged into that kernel's time, traffic and instruction counts in every profile.
__global__ void __launch_bounds__(WG_THREADS, 2) k_cyl_net_wg_rerun(const float* __restrict__ x, CylWgParams P, float* __restrict__ y)
{
    extern __shared__ float lds[];
    if (P.only_if[blockIdx.x] == 0) return;
    cyl_net_wg_body(x, P, y, lds);
}


// Host helper: filters w [Cout][Cin][3][3] (BN folded) -> U = G g G^T in fp64, rounded once, in the kernel's A-operand tiling
//     out[16 * Cout * Cin] = [N-group][i][k-step][n2][lk][li][j] = U[i][j][16 (NG g + n2) + li][4 ks + lk]
// with N-groups of NG = 2 N-tiles for 128 output channels (a wavefront owns a pair there) and NG = 1 otherwise: the
// k-steps of a wavefront follow each other in memory (round 2 had the N-tile inside the k-step: every 1 KB fetch of a
// wavefront then sat on another 4-8 KB page, and the filter stream cost 8 % of the kernel in translation misses).  No device work.
// General form: N-groups of ng N-tiles; nblk = 4, or 5 with g[1] G^T (= U_1 - U_2, the filter row a window that ends in the
// padding sees: the kernels form it in registers) as block 4; out[4 nblk * Cout * Cin].
extern "C" int buf_winograd_tile_filters(const float* w_host, int cout, int cin, int ng, int nblk, float* out_host)
{
    BUF_REQUIRE(w_host && out_host, BUF_EINVAL, "buf_winograd_tile_filters: null argument");
    BUF_REQUIRE(cout > 0 && cin > 0 && cout % 16 == 0 && cin % 4 == 0, BUF_EINVAL, "buf_winograd_tile_filters: widths %d -> %d", cin, cout);
    BUF_REQUIRE(ng >= 1 && (cout / 16) % ng == 0 && (nblk == 4 || nblk == 5), BUF_EINVAL, "buf_winograd_tile_filters: %d N-tiles in groups of %d, %d blocks",
                cout / 16, ng, nblk);
    static const double G[4][3] = { { 1, 0, 0 }, { .5, .5, .5 }, { .5, -.5, .5 }, { 0, 0, 1 } };
    const int k4 = cin / 4;
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++) {
            const float* g = w_host + ((size_t)o * cin + c) * 9;
            const int n = o / 16;
            for (int i = 0; i < nblk; i++)
                for (int j = 0; j < 4; j++) {
                    double u = 0;
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) u += (i < 4 ? G[i][a] : (a == 1 ? 1.0 : 0.0)) * (double)g[3 * a + b] * G[j][b];
                    const size_t idx = ((((((size_t)(n / ng) * nblk + i) * k4 + c / 4) * ng + n % ng) * 4 + c % 4) * 16 + o % 16);
                    out_host[idx * 4 + j] = (float)u;
                }
        }
    return BUF_OK;
}

extern "C" int buf_winograd_tile_weights(const float* w_host, int cout, int cin, float* out_host)
{
    BUF_REQUIRE(cout > 0 && cin > 0 && cout % 16 == 0 && cin % 4 == 0, BUF_EINVAL, "buf_winograd_tile_weights: widths %d -> %d", cin, cout);
    return buf_winograd_tile_filters(w_host, cout, cin, wg_group(cin, cout), WG_BLOCKS / 4, out_host);
}

// N-tiles per group in the filter tiling of a layer with these widths (the Python side asks instead of restating the rule)
extern "C" int buf_winograd_group(int cin, int cout) { return wg_group(cin, cout); }

// x f32[np,48,140] -> y f32[np,32,140]; weights in the Winograd-domain tiling (see CylWgParams).
static int wg_launch(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host, const int* cin_host,
                     const int* cout_host, const int* relu_host, float* y, const int* only_if, void* stream);

extern "C" int buf_cylindrical_net_wg(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host,
                                      const int* cin_host, const int* cout_host, const int* relu_host, float* y, void* stream)
{
    return wg_launch(x, npatch, wt_host, bias_host, cin_host, cout_host, relu_host, y, nullptr, stream);
}

// 0 when the Winograd fp32 kernel is built for this stack of widths (what buf_cylindrical_net_wg would accept), else BUF_EINVAL with the reason
extern "C" int buf_cylindrical_net_wg_supports(const int* cin_host, const int* cout_host)
{
    BUF_REQUIRE(cin_host && cout_host, BUF_EINVAL, "buf_cylindrical_net_wg_supports: null argument");
    for (int l = 0; l < WG_LAYERS; l++) {
        const int ci = cin_host[l], co = cout_host[l];
        const bool ok = ci > 0 && ci % 16 == 0 && ci <= WG_MAXC && (co == 32 || co == 64 || co == 128) && (co != 128 || ci % 32 == 0) &&
                        (co != 32 || (ci % 32 == 0 && ci <= 64)) && (l == 0 || ci == cout_host[l - 1]) && (l == 0 || cout_host[l - 1] != 32 || co == 32);
        BUF_REQUIRE(ok, BUF_EINVAL, "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d", l, ci, co);
    }
    BUF_REQUIRE(cout_host[WG_LAYERS - 1] == 32, BUF_EINVAL, "buf_cylindrical_net_wg: the last layer must have 32 channels");
    return BUF_OK;
}

static int wg_launch(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host, const int* cin_host,
                     const int* cout_host, const int* relu_host, float* y, const int* only_if, void* stream)
{
    BUF_REQUIRE(npatch >= 0, BUF_EINVAL, "buf_cylindrical_net_wg: npatch=%d", npatch);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(x && y && wt_host && bias_host && cin_host && cout_host && relu_host, BUF_EINVAL, "buf_cylindrical_net_wg: null argument");
    CylWgParams P;
    for (int l = 0; l < WG_LAYERS; l++) {
        P.wt[l] = wt_host[l]; P.bias[l] = bias_host[l];
        P.cin[l] = cin_host[l]; P.cout[l] = cout_host[l]; P.relu[l] = relu_host[l];
        BUF_REQUIRE(P.wt[l] && P.bias[l], BUF_EINVAL, "buf_cylindrical_net_wg: null weights for layer %d", l);
        BUF_REQUIRE(P.cin[l] % 16 == 0 && P.cin[l] <= WG_MAXC && (P.cout[l] == 32 || P.cout[l] == 64 || P.cout[l] == 128),
                    BUF_EINVAL, "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(P.cout[l] != 128 || P.cin[l] % 32 == 0, BUF_EINVAL,
                    "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d (128 output channels need Cin %% 32 == 0)", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(P.cout[l] != 32 || (P.cin[l] % 32 == 0 && P.cin[l] <= 64), BUF_EINVAL,
                    "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d (32 output channels need Cin = 32 or 64)", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(l == 0 || P.cin[l] == P.cout[l - 1], BUF_EINVAL, "buf_cylindrical_net_wg: layer %d width mismatch", l);
        // The 32-output layers hand their K-split partial sums over through the rows of the channels 64..95, zero words included
        // (wg_layer_mksplit / wg_layer_ksplit); those words are written once at kernel start, so no wider layer may follow.
        BUF_REQUIRE(l == 0 || P.cout[l - 1] != 32 || P.cout[l] == 32, BUF_EINVAL,
                    "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d (only 32-output layers may follow a 32-output layer)", l, P.cin[l], P.cout[l]);
    }
    BUF_REQUIRE(P.cout[WG_LAYERS - 1] == 32, BUF_EINVAL, "buf_cylindrical_net_wg: the last layer must have 32 channels");
    P.only_if = only_if;
    size_t lds = sizeof(float) * WG_BUF;
    static LdsGrant grant, grant_rerun;
    if (int rc = only_if ? grant_dynamic_lds((const void*)k_cyl_net_wg_rerun, lds, grant_rerun) : grant_dynamic_lds((const void*)k_cyl_net_wg, lds, grant)) return rc;
    double macs = 0;
    for (int l = 0; l < WG_LAYERS; l++) macs += 9.0 * P.cin[l] * P.cout[l];
    TimedSpan span;
    bool timed = !only_if && timing_begin((hipStream_t)stream, &span, 2.0 * 140 * macs * npatch, BUF_TIMED_CYL_NET);   // (a masked re-run is not a full launch)
#ifdef WG_STAMP
    BUF_CHECK_HIP(hipMalloc(&P.stamps, (size_t)npatch * (4 * 20 + 4) * sizeof(long long)));
    BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(wg_stamp_ptr), &P.stamps, sizeof(P.stamps)));
#endif
    if (only_if) k_cyl_net_wg_rerun<<<npatch, WG_THREADS, lds, (hipStream_t)stream>>>(x, P, y);
    else k_cyl_net_wg<<<npatch, WG_THREADS, lds, (hipStream_t)stream>>>(x, P, y);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
#ifdef WG_STAMP
    {   // per-layer wave cycles (to the wave's arrival at the closing barrier | wait there), second half of the workgroups
        BUF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        long long* h = (long long*)malloc((size_t)npatch * (4 * 20 + 4) * sizeof(long long));
        BUF_CHECK_HIP(hipMemcpy(h, P.stamps, (size_t)npatch * (4 * 20 + 4) * sizeof(long long), hipMemcpyDeviceToHost));
        double comp[WG_LAYERS][4] = {}, wait[WG_LAYERS][4] = {}, tot = 0, pre = 0; long n = 0;
        for (int b = npatch / 2; b < npatch; b++, n++)
            for (int w = 0; w < 4; w++) {
                const long long* q = h + ((size_t)b * 4 + w) * 20;
                for (int l = 0; l < WG_LAYERS; l++) { comp[l][w] += (double)(q[2 * l + 1] - q[2 * l]); wait[l][w] += (double)(q[2 * l + 2] - q[2 * l + 1]); }
                if (w == 0) { tot += (double)(q[2 * WG_LAYERS] - q[0]); pre += (double)(q[0] - q[17]); }
            }
        if (n && npatch >= 1024) {
            double dc = 0, dr = 0;
            for (int b = npatch / 2; b < npatch; b++) { const long long* q = h + (size_t)npatch * 80 + (size_t)b * 4; dc += (double)(q[2] - q[0]); dr += (double)(q[3] - q[1]); }
            fprintf(stderr, "  WG_STAMP in-kernel clock: %.0f shader cycles per workgroup over %.0f ticks of the 100 MHz counter -> %.3f GHz\n", dc / n, dr / n, dc / dr * 0.1);
            double c7[4] = {}, b7[4] = {}, s7[4] = {};
            for (int b = npatch / 2; b < npatch; b++)
                for (int w = 0; w < 4; w++) {
                    const long long* q = h + ((size_t)b * 4 + w) * 20;
                    c7[w] += (double)(q[18] - q[14]); b7[w] += (double)(q[19] - q[18]); s7[w] += (double)(q[15] - q[19]);
                }
            fprintf(stderr, "  last layer per wave: K loops %6.0f %6.0f %6.0f %6.0f | barrier %5.0f %5.0f %5.0f %5.0f | add + store %5.0f %5.0f %5.0f %5.0f\n",
                    c7[0] / n, c7[1] / n, c7[2] / n, c7[3] / n, b7[0] / n, b7[1] / n, b7[2] / n, b7[3] / n, s7[0] / n, s7[1] / n, s7[2] / n, s7[3] / n);
            fprintf(stderr, "WG_STAMP: %ld workgroups, layers total %.0f cycles per patch, input phase %.0f\n", n, tot / n, pre / n);
            for (int l = 0; l < WG_LAYERS; l++) {
                const double mf = 40.0 * (P.cin[l] / 4) * (P.cout[l] / 16) / 4;     // MFMAs per wave
                fprintf(stderr, "  layer %d %3d->%3d: MFMAs/wave %5.0f | to barrier %7.0f %7.0f %7.0f %7.0f | wait %6.0f %6.0f %6.0f %6.0f | cycles per MFMA slot %.1f\n",
                        l, P.cin[l], P.cout[l], mf, comp[l][0] / n, comp[l][1] / n, comp[l][2] / n, comp[l][3] / n, wait[l][0] / n, wait[l][1] / n,
                        wait[l][2] / n, wait[l][3] / n, (comp[l][0] + wait[l][0]) / n / (2 * mf));
            }
        }
        free(h); (void)hipFree(P.stamps);
    }
#endif
    return BUF_OK;
}
