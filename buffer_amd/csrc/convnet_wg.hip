// A11 dense part -- Cylindrical_Net (models/patchnet.py:15-85) as ONE fused fp32-MFMA kernel in the Winograd F(2x2, 3x3)
// domain.
//
// Every layer of the stack is a 3x3 correlation over the 7 x 20 (elevation x azimuth) map (layer 0's radial depth folds
// into 48 input channels), circular in azimuth and zero-padded in elevation (utils/common.py:265-310).  With the map cut
// into 4 x 10 tiles of 2 x 2 outputs,
//     Y = A^T [ sum_c (G g G^T)[c] (.) (B^T d[c] B) ] A          (d: the tile's 4 x 4 input window)
// turns the 9-tap implicit GEMM (9 x 144 rows per channel pair) into 16 component GEMMs over 40 tile rows (16 x 48): 44/75
// of the matrix instructions of the direct form (the fourth row component of the bottom tile row only feeds the discarded
// output row 7 and is skipped for the M-tile that holds nothing else).  All arithmetic stays fp32 (v_mfma_f32_16x16x4_f32);
// the filter transform is done once on the host in fp64.
//
// One workgroup owns one patch for the whole stack.  Activations live in ONE LDS buffer [128][160] (rows of 22 = 20 azimuth
// columns + the two circular halo columns, so a window row is one address + immediate offsets; 6 zeros per channel serve
// every read of the elevation padding), rewritten in place after each layer -> two workgroups per CU (2 x 80 KB).
// A wavefront owns one N-tile (16 output channels) at a time and walks the four row components i of the transform as four
// passes over K: per k-step and M-tile it reads two window rows (four ds_read_b64), forms the four column components on the
// VALU (eight plain adds) and issues four MFMAs; the accumulators run on through the four passes and after each pass the column
// transform of the running sum is folded into the 2 x 2 outputs in registers (the C/D layout keeps a tile in one lane).
// LDS banking (ds_read_b64: bank = word mod 64, lanes 0-31 and 32-63 served separately): the 40 tiles are dealt to the
// three M-tiles such that the 16 lanes of a k-step channel cover 32 distinct banks, and the channel stride 160 = 32 mod 64
// puts the second channel of the half-wave on the other 32:
//     M-tile 0: tile row 0 (10) + tile row 1, columns 4..9     M-tile 1: tile row 2 (10) + row 1, columns 0..3 + row 3, columns 8, 9
//     M-tile 2: tile row 3, columns 0..7 (8 lanes idle)
#include "common.h"
#include <type_traits>

#define WG_CS 160             // channel stride in floats: 7 x 22 + 6 zeros
#define WG_ROW 22             // [col 19][col 0 .. col 19][col 0]
#define WG_ZERO 154           // zeros read by every window row outside 0 <= elevation < 7
#define WG_MAXC 128
#define WG_LAYERS 8
#define WG_THREADS 256
#define WG_BUF (WG_MAXC * WG_CS)
#define WG_KSTEP (16 * WG_CS)  // bytes between k-steps (4 channels)

typedef float wgf4 __attribute__((ext_vector_type(4)));
typedef float wgf2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) wgf2* wg_lds_f2;

struct CylWgParams {
    const float* wt[WG_LAYERS];     // Winograd-domain weights, [i][k-step][N-tile][lane][j] (ops.winograd_tile_weights)
    const float* bias[WG_LAYERS];   // [Cout]
    int cin[WG_LAYERS], cout[WG_LAYERS], relu[WG_LAYERS];
};

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

// LDS reads of step G (M-tile G % NT of k-step G / NT) of a pass.  Two ds_read_b64 per row: the empty asm keeps the
// compiler from fusing them into the half-rate ds_read2_b64.
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

// The first two steps of pass I (fetched by whoever runs before the pass: the layer prologue or the previous pass)
template <int I, int T0, int T1>
__device__ __forceinline__ void wg_first_steps(unsigned (&RA)[3][4], wgf2 (&D)[2][4])
{
    constexpr int NT = wg_te(I, T1) - T0;
    if constexpr (NT > 0) {
#pragma unroll
        for (int g = 0; g < 2; g++) WG_LOAD2(D[g], RA[T0 + g % NT][wg_a1(I)], RA[T0 + g % NT][wg_a2(I)], (g / NT) * WG_KSTEP)
    }
}

// One row component I of the transform over the whole K range for the M-tiles [T0, T1): acc[t][j] += V_Ij(tile, c) * U_Ij(c, n).
// The (k-step, M-tile) steps form a three-stage software pipeline pinned with sched_barriers (the compiler otherwise hoists the
// loads to the top of the loop body and waits for them at once): step s issues the LDS reads of step s+2, forms the column
// components of step s+1 on the VALU and runs the four MFMAs of step s.  VALU instructions do not hide behind the matrix
// pipe on this chip (tools/micro/mfma_coissue: every VALU per MFMA costs 2.5-6 cycles of it), so the loop carries nothing
// but the eight adds of the transform: four k-steps per iteration, their LDS offsets ride in the instructions, the weights
// (Ba: k-steps 0-1, Bb: 2-3; one 16-byte load per k-step) are reloaded in place half an iteration ahead, and the last
// iteration fetches the weights of the pass that follows (wp_next) instead of its own.  Every pass reads its own first two
// steps (handing them over from the previous pass cost registers -- spills -- and bought no time).
template <int I, int T0, int T1>
__device__ __forceinline__ void wg_pass(unsigned (&RA)[3][4], const float* __restrict__ wp, const float* __restrict__ wp_next, int k4, int wstride,
                                        wgf4 (&Ba)[2], wgf4 (&Bb)[2], wgf4 (&acc)[3][4])
{
    constexpr int A1 = wg_a1(I), A2 = wg_a2(I);
    constexpr int TE = wg_te(I, T1);
    constexpr int NT = TE > T0 ? TE - T0 : 1;
    if constexpr (TE <= T0) return;
    wgf2 D[2][4];                                               // window rows A1 (columns 0-1, 2-3) and A2 of two steps in flight
    float V[2][4];
    wg_first_steps<I, T0, T1>(RA, D);
    // row component (d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3), then the four column components
#define WG_XFORM(BUF)                                                                                     \
    {                                                                                                     \
        float r_[4];                                                                                      \
        _Pragma("unroll") for (int b = 0; b < 4; b++) {                                                   \
            const float da_ = D[BUF][b >> 1][b & 1], db_ = D[BUF][2 + (b >> 1)][b & 1];                   \
            r_[b] = I == 1 ? wg_add(da_, db_) : (I == 2 ? wg_sub(db_, da_) : wg_sub(da_, db_));           \
        }                                                                                                 \
        V[BUF][0] = wg_sub(r_[0], r_[2]); V[BUF][1] = wg_add(r_[1], r_[2]);                               \
        V[BUF][2] = wg_sub(r_[2], r_[1]); V[BUF][3] = wg_sub(r_[1], r_[3]);                               \
    }
    const int niter = k4 >> 2;
    unsigned P[NT][2], E[2][2];
    WG_XFORM(0)
#pragma unroll 1
    for (int it = 0; it < niter; it++) {
        const bool more = it + 1 < niter;
        const float* wn = more ? wp + (size_t)(4 * it + 4) * wstride : wp_next;
        const unsigned kb = (unsigned)it * (4 * WG_KSTEP);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; t++) { P[t][0] = RA[T0 + t][A1] + kb; P[t][1] = RA[T0 + t][A2] + kb; }
#pragma unroll
        for (int g = 0; g < 2; g++) {                           // the two steps past this iteration: the next one's or the next pass's
            const unsigned own0 = P[g % NT][0] + (4 + g / NT) * WG_KSTEP, own1 = P[g % NT][1] + (4 + g / NT) * WG_KSTEP;
            E[g][0] = more ? own0 : P[g % NT][0];         // past the end: the iteration's own first steps again (unused)
            E[g][1] = more ? own1 : P[g % NT][1];
        }
#pragma unroll
        for (int s = 0; s < 4 * NT; s++) {
            const int t = T0 + s % NT, kk = s / NT, g = s + 2;
            if (g < 4 * NT) WG_LOAD2(D[s & 1], P[g % NT][0], P[g % NT][1], (g / NT) * WG_KSTEP)
            else WG_LOAD2(D[s & 1], E[g >= 4 * NT ? g - 4 * NT : 0][0], E[g >= 4 * NT ? g - 4 * NT : 0][1], 0)
            WG_XFORM((s + 1) & 1)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk < 2 ? Ba[kk][j] : Bb[kk - 2][j], V[s & 1][j], acc[t][j], 0, 0, 0);
            if (s == 2 * NT - 1) {                              // k-steps 0-1 are through: their registers take the next iteration's
                Ba[0] = *reinterpret_cast<const wgf4*>(wn);
                Ba[1] = *reinterpret_cast<const wgf4*>(wn + wstride);
            }
            if (s == 4 * NT - 1) {
                Bb[0] = *reinterpret_cast<const wgf4*>(wn + 2 * wstride);
                Bb[1] = *reinterpret_cast<const wgf4*>(wn + 3 * wstride);
            }
            // the step's LDS reads go first (they are consumed one step later), the VALU work is woven between the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef WG_XFORM
}

// All 16 components of one N-tile: Y[t][u][v] = 2 x 2 outputs of the tiles of M-tile t (C/D layout: tile = lane's column).
// On entry Ba/Bb hold the weights of the first four k-steps; on exit those of the N-tile at wp_after (the wavefront's next one).
template <int T0, int T1>
__device__ __forceinline__ void wg_ntile(unsigned (&RA)[3][4], const float* __restrict__ wp, const float* __restrict__ wp_after, int k4, int wstride,
                                         wgf4 bv, wgf4 (&Ba)[2], wgf4 (&Bb)[2], wgf4 (&Y)[3][2][2])
{
    const size_t pstride = (size_t)k4 * wstride;
    using std::integral_constant;
    // The accumulators run on through the four passes (cleared once per N-tile): after pass I they hold the sum of the row
    // components 0..I, F_I = its column transform, and with s_i = F_i - F_(i-1)
    //     output row 0 = s_0 + s_1 + s_2 = F_2,      output row 1 = s_1 - s_2 - s_3 = -F_0 + 2 F_1 - F_3
    // -- three accumulator clears and half of the output-transform adds less per N-tile (VALU work is what this kernel is short of).
    // The bias enters component (1, 1)'s accumulator before pass 1: F_1, F_2, F_3 then carry it once in both columns.
    wgf4 acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[t][j] = (wgf4){ 0.f, 0.f, 0.f, 0.f };
    auto run = [&](auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        // component 3 does nothing for a wavefront that only has M-tile 2: the pass before it hands over to component 0
        constexpr int INEXT = (I == 3 || (I == 2 && wg_te(3, T1) <= T0)) ? 0 : I + 1;
        if constexpr (I == 1) {
#pragma unroll
            for (int t = T0; t < T1; t++) acc[t][1] += bv;
        }
        wg_pass<I, T0, T1>(RA, wp + I * pstride, INEXT == 0 ? wp_after : wp + (I + 1) * pstride, k4, wstride, Ba, Bb, acc);
#pragma unroll
        for (int t = T0; t < T1; t++) {
            if (I == 3 && t == 2) continue;
            if (I == 2 ? false : t == 2) continue;              // M-tile 2 (bottom tile row) only has output row 0 = F_2
            const wgf4 f0 = acc[t][0] + acc[t][1] + acc[t][2];
            const wgf4 f1 = acc[t][1] - acc[t][2] - acc[t][3];
            if constexpr (I == 0) { Y[t][1][0] = f0; Y[t][1][1] = f1; }                       // F_0 (enters with a minus sign below)
            else if constexpr (I == 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    Y[t][1][0][r] = __builtin_fmaf(2.f, f0[r], -Y[t][1][0][r]);
                    Y[t][1][1][r] = __builtin_fmaf(2.f, f1[r], -Y[t][1][1][r]);
                }
            }
            else if constexpr (I == 2) { Y[t][0][0] = f0; Y[t][0][1] = f1; }
            else { Y[t][1][0] -= f0; Y[t][1][1] -= f1; }
        }
    };
    run(integral_constant<int, 0>{});
    run(integral_constant<int, 1>{});
    run(integral_constant<int, 2>{});
    run(integral_constant<int, 3>{});
}

// ReLU + store of one N-tile's outputs: into the activation buffer (with the circular halo copies) or to y[32][140].
// The weights are the MFMA's first operand, so the C/D layout has lane column li = the tile and rows lk * 4 + r = the
// output channel: a store instruction writes 16 different positions of a channel (distinct banks), not 16 channels at one
// position (one bank: the channel stride is a multiple of 32 words).
template <int T0, int T1, bool GLB>
__device__ __forceinline__ void wg_store(const wgf4 (&Y)[3][2][2], int nt, int relu, float* __restrict__ act, float* __restrict__ out_glb,
                                         int li, int lk)
{
    const int n0 = nt * 16 + lk * 4;
    const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int t = T0; t < T1; t++) {
        int ty, tx;
        const bool valid = wg_tile(t, li, ty, tx);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (t == 2 && u == 1) continue;                       // M-tile 2 is tile row 3: its second output row is row 7
            const bool ok = valid && (t != 1 || u == 0 || ty < 3);
            const int row = 2 * ty + u;
            if constexpr (GLB) {
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v0 = fmaxf(Y[t][u][0][r], lo), v1 = fmaxf(Y[t][u][1][r], lo);
                        *reinterpret_cast<wgf2*>(out_glb + (size_t)(n0 + r) * 140 + row * 20 + 2 * tx) = (wgf2){ v0, v1 };
                    }
                }
            } else {
                // straight-line code (a branch per store costs more than the store): lanes with nothing to write, and the halo
                // copies of the inner tiles, go to the two spare words behind the channel's zero area
                const int pos = ok ? row * WG_ROW + 2 * tx + 1 : WG_ZERO + 4;
                const int h0 = (ok && tx == 0) ? row * WG_ROW + 21 : WG_ZERO + 4;      // column 0 again behind column 19
                const int h1 = (ok && tx == 9) ? row * WG_ROW : WG_ZERO + 5;           // column 19 again in front of column 0
                float* p = act + n0 * WG_CS;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v0 = fmaxf(Y[t][u][0][r], lo), v1 = fmaxf(Y[t][u][1][r], lo);
                    p[r * WG_CS + pos] = v0; p[r * WG_CS + pos + 1] = v1;
                    p[r * WG_CS + h0] = v0; p[r * WG_CS + h1] = v1;
                }
            }
        }
    }
}

// One layer for the calling wavefront.  NTW N-tiles per wavefront (2 for 128 output channels), M-tiles [T0, T1).
// The layer is written in place, so a wavefront keeps its outputs in registers until every wavefront has finished reading
// the input.  With 128 output channels it owns two N-tiles: the upper one (channels >= 64) first -- when the layer has 64
// input channels its rows are free and it is stored at once (EARLY), otherwise it is held while the lower one is computed.
template <int NTW, int T0, int T1, bool GLB, bool EARLY>
__device__ __forceinline__ void wg_layer(float* __restrict__ act, float* __restrict__ out_glb, const float* __restrict__ wt,
                                         const float* __restrict__ bias, int cin, int cout, int relu, int nt_first)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));                   // lane-derived values are formed per layer: kept across the layers they are spilled
    const int li = lane & 15, lk = lane >> 4;
    unsigned RA[3][4];                               // window-row addresses of the lane: 3 M-tiles x 4 rows, channel lk of k-step 0
    {
        const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)act;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int a = 0; a < 4; a++) RA[t][a] = wg_row_addr(act_addr, t, a, li, lk);
    }
    const int k4 = cin >> 2, wstride = (cout >> 4) * 256;
    wgf4 Ba[2], Bb[2];                               // weights of four k-steps: every pass hands the next one its first four
    const int nt_hi = NTW == 2 ? nt_first + 4 : nt_first, nt_lo = nt_first;
    const float* wp_hi = wt + ((size_t)nt_hi * 64 + lane) * 4;
    const float* wp_lo = wt + ((size_t)nt_lo * 64 + lane) * 4;
    Ba[0] = *reinterpret_cast<const wgf4*>(wp_hi);
    Ba[1] = *reinterpret_cast<const wgf4*>(wp_hi + wstride);
    Bb[0] = *reinterpret_cast<const wgf4*>(wp_hi + 2 * wstride);
    Bb[1] = *reinterpret_cast<const wgf4*>(wp_hi + 3 * wstride);
    wgf4 Y[NTW][3][2][2];
    wg_ntile<T0, T1>(RA, wp_hi, wp_lo, k4, wstride, *reinterpret_cast<const wgf4*>(bias + nt_hi * 16 + lk * 4), Ba, Bb, Y[0]);
    // EARLY: known at compile time for the 64 -> 128 layer; the instantiation of the 128 -> 128 layer tests it at run time (always
    // false there) -- with the test compiled out the register allocator parks every partial result in scratch
    const bool early = NTW == 2 && !GLB && (EARLY || nt_hi * 16 >= cin);
    if constexpr (NTW == 2) {
        if (early) wg_store<T0, T1, false>(Y[0], nt_hi, relu, act, nullptr, li, lk);
        wg_ntile<T0, T1>(RA, wp_lo, wp_lo, k4, wstride, *reinterpret_cast<const wgf4*>(bias + nt_lo * 16 + lk * 4), Ba, Bb, Y[1]);
    }
    __syncthreads();                                 // every wavefront has finished reading the layer's input
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        if (NTW == 2 && q == 0 && early) continue;
        wg_store<T0, T1, GLB>(Y[q], q == 0 ? nt_hi : nt_lo, relu, act, out_glb, li, lk);
    }
}

__global__ void __launch_bounds__(WG_THREADS, 2) k_cyl_net_wg(const float* __restrict__ x, CylWgParams P, float* __restrict__ y)
{
    extern __shared__ float lds[];                   // [128][160]
    float* act = lds;
    const int patch = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    {   // input [48][140] -> rows of 22 with the halo columns
        const float* src = x + (size_t)patch * P.cin[0] * 140;
        for (int i = threadIdx.x; i < P.cin[0] * 140; i += WG_THREADS) {
            const int c = i / 140, pos = i - c * 140;
            const int row = pos / 20, col = pos - row * 20;
            const float v = __builtin_nontemporal_load(src + i);       // streamed once: leave the L2 to the filters
            float* p = act + c * WG_CS + row * WG_ROW + col;
            p[1] = v;
            if (col == 0) p[21] = v;
            if (col == 19) p[-19] = v;
        }
    }
    for (int i = threadIdx.x; i < WG_MAXC * 6; i += WG_THREADS) act[(i / 6) * WG_CS + WG_ZERO + i % 6] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < WG_LAYERS; l++) {
        const int cin = P.cin[l], cout = P.cout[l];
        if (cout == 128) {
            if (cin <= 64)   wg_layer<2, 0, 3, false, true>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w);
            else             wg_layer<2, 0, 3, false, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w);
        } else if (cout == 64) wg_layer<1, 0, 3, false, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w);
        else if (l < WG_LAYERS - 1) {
            if (w & 1)       wg_layer<1, 1, 3, false, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
            else             wg_layer<1, 0, 1, false, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
        } else {
            float* glb = y + (size_t)patch * cout * 140;
            if (w & 1)       wg_layer<1, 1, 3, true, false>(act, glb, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
            else             wg_layer<1, 0, 1, true, false>(act, glb, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
        }
        __syncthreads();
    }
}


// Host helper: filters w [Cout][Cin][3][3] (BN folded) -> U = G g G^T in fp64, rounded once, in the kernel's B-operand tiling
// out[16 * Cout * Cin] = [i][k-step][N-tile][lk][li][j] = U[i][j][16 n + li][4 ks + lk].  No device work.
extern "C" int buf_winograd_tile_weights(const float* w_host, int cout, int cin, float* out_host)
{
    BUF_REQUIRE(w_host && out_host, BUF_EINVAL, "buf_winograd_tile_weights: null argument");
    BUF_REQUIRE(cout > 0 && cin > 0 && cout % 16 == 0 && cin % 4 == 0, BUF_EINVAL, "buf_winograd_tile_weights: widths %d -> %d", cin, cout);
    static const double G[4][3] = { { 1, 0, 0 }, { .5, .5, .5 }, { .5, -.5, .5 }, { 0, 0, 1 } };
    const int k4 = cin / 4, nt = cout / 16;
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++) {
            const float* g = w_host + ((size_t)o * cin + c) * 9;
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double u = 0;
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) u += G[i][a] * (double)g[3 * a + b] * G[j][b];
                    const size_t idx = ((((size_t)i * k4 + c / 4) * nt + o / 16) * 4 + c % 4) * 16 + o % 16;
                    out_host[idx * 4 + j] = (float)u;
                }
        }
    return BUF_OK;
}

// x f32[np,48,140] -> y f32[np,32,140]; weights in the Winograd-domain tiling (see CylWgParams).
extern "C" int buf_cylindrical_net_wg(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host,
                                      const int* cin_host, const int* cout_host, const int* relu_host, float* y, void* stream)
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
        BUF_REQUIRE(l == 0 || P.cin[l] == P.cout[l - 1], BUF_EINVAL, "buf_cylindrical_net_wg: layer %d width mismatch", l);
    }
    BUF_REQUIRE(P.cout[WG_LAYERS - 1] == 32, BUF_EINVAL, "buf_cylindrical_net_wg: the last layer must have 32 channels");
    size_t lds = sizeof(float) * WG_BUF;
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds((const void*)k_cyl_net_wg, lds, grant)) return rc;
    double macs = 0;
    for (int l = 0; l < WG_LAYERS; l++) macs += 9.0 * P.cin[l] * P.cout[l];
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 2.0 * 140 * macs * npatch, BUF_TIMED_CYL_NET);
    k_cyl_net_wg<<<npatch, WG_THREADS, lds, (hipStream_t)stream>>>(x, P, y);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
