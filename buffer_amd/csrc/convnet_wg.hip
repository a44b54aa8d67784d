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
// VALU (packed fp32 adds) and issues four MFMAs; after a pass the column/row output transform is applied in registers (the
// C/D layout keeps a tile in one lane), so only the 2 x 2 outputs are carried between passes.
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
#ifndef WG_EXP
#define WG_EXP 0              // development ablations (tools/wg_variants.sh); 0 in the product
#endif

#ifdef WG_PROF
__device__ unsigned long long wg_prof[16];
#define WG_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WG_ACC(slot, a, b) if ((threadIdx.x & 63) == 0) atomicAdd(&wg_prof[slot], (b) - (a))
#else
#define WG_T(var)
#define WG_ACC(slot, a, b)
#endif
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

// One row component I of the transform over the whole K range for the M-tiles [T0, T1): acc[t][j] += V_Ij(tile, c) * U_Ij(c, n).
// The (k-step, M-tile) steps form a three-stage software pipeline pinned with sched_barriers (the compiler otherwise hoists the
// loads to the top of the loop body and waits for them at once): step s issues the LDS reads of step s+2, forms the column
// components of step s+1 on the VALU and runs the four MFMAs of step s.  Two k-steps per loop iteration; their weights
// (two 16-byte loads per lane, Bc) are fetched one iteration ahead -- the last iteration fetches the first two k-steps
// of the pass that follows (wp_next), so a pass starts with its weights in registers.
template <int I, int T0, int T1>
__device__ __forceinline__ void wg_pass(unsigned act_addr, int li, int lk, const float* __restrict__ wp, const float* __restrict__ wp_next,
                                        int k4, int wstride, wgf4 (&Bc)[2], wgf4 (&acc)[3][4])
{
    constexpr int A1 = I == 0 ? 0 : 1, A2 = I == 3 ? 3 : 2;     // the two window rows of this component
    constexpr int TE = (I == 3 && T1 == 3) ? 2 : T1;            // M-tile 2 holds bottom-row tiles only: component 3 unused
    constexpr int NT = TE > T0 ? TE - T0 : 1;
    if constexpr (TE <= T0) return;
    WG_T(tp0);
    wgf2 D[2][4];                                               // window rows A1 (columns 0-1, 2-3) and A2 of one step
    float V[2][4];
    wgf4 Bn[2];
    unsigned P[NT][2];                                          // LDS byte addresses of the two rows, first k-step of the iteration
    unsigned E[2][2];                                           // the same for the two steps fetched for the NEXT iteration
#pragma unroll
    for (int t = 0; t < NT; t++) {
        P[t][0] = wg_row_addr(act_addr, T0 + t, A1, li, lk);
        P[t][1] = wg_row_addr(act_addr, T0 + t, A2, li, lk);
    }
    const int niter = k4 >> 1;
    // step G of an iteration: M-tile G % NT of k-step G / NT (steps 2 NT, 2 NT + 1: the first two of the next iteration).
    // Two ds_read_b64 per row: the empty asm keeps the compiler from fusing them into the half-rate ds_read2_b64.
#define WG_LOAD2(BUF, A0, A1_, OFS)                                                                       \
    {                                                                                                     \
        D[BUF][0] = *(wg_lds_f2)(size_t)((A0) + (OFS));                                                   \
        asm volatile("" : "+v"(A0));                                                                      \
        D[BUF][1] = *(wg_lds_f2)(size_t)((A0) + (OFS) + 8);                                               \
        asm volatile("" : "+v"(A0));                                                                      \
        D[BUF][2] = *(wg_lds_f2)(size_t)((A1_) + (OFS));                                                  \
        asm volatile("" : "+v"(A1_));                                                                     \
        D[BUF][3] = *(wg_lds_f2)(size_t)((A1_) + (OFS) + 8);                                              \
        asm volatile("" : "+v"(A1_));                                                                     \
    }
#define WG_LOADD(BUF, G)                                                                                  \
    if (!(WG_EXP & 4)) {                                                                                  \
        const int e_ = (G) >= 2 * NT ? (G) - 2 * NT : 0;                                                  \
        if ((G) < 2 * NT) WG_LOAD2(BUF, P[(G) % NT][0], P[(G) % NT][1], ((G) / NT) * WG_KSTEP)            \
        else WG_LOAD2(BUF, E[e_][0], E[e_][1], 0)                                                         \
    }
    // row component (d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3), then the four column components: packed fp32 adds
#define WG_PK_SUB(R, A, B) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R) : "v"(A), "v"(B))
#define WG_PK_ADD(R, A, B) asm("v_pk_add_f32 %0, %1, %2" : "=v"(R) : "v"(A), "v"(B))
#define WG_XFORM(BUF)                                                                                     \
    {                                                                                                     \
        wgf2 r01_, r23_, v03_;                                                                            \
        if (WG_EXP & 16) { V[BUF][0] = D[BUF][0].x; V[BUF][1] = D[BUF][1].x; V[BUF][2] = D[BUF][2].x; V[BUF][3] = D[BUF][3].x; } else { \
        if constexpr (I == 1) { WG_PK_ADD(r01_, D[BUF][0], D[BUF][2]); WG_PK_ADD(r23_, D[BUF][1], D[BUF][3]); } \
        else if constexpr (I == 2) { WG_PK_SUB(r01_, D[BUF][2], D[BUF][0]); WG_PK_SUB(r23_, D[BUF][3], D[BUF][1]); } \
        else { WG_PK_SUB(r01_, D[BUF][0], D[BUF][2]); WG_PK_SUB(r23_, D[BUF][1], D[BUF][3]); }            \
        WG_PK_SUB(v03_, r01_, r23_);                                                                      \
        V[BUF][0] = v03_.x; V[BUF][3] = v03_.y; V[BUF][1] = r01_.y + r23_.x; V[BUF][2] = r23_.x - r01_.y; } \
    }
    if (WG_EXP & 4) { for (int q = 0; q < 2; q++) for (int r = 0; r < 4; r++) D[q][r] = *(wg_lds_f2)(size_t)(P[0][q] + 8 * r); }
    WG_LOADD(0, 0)
    WG_LOADD(1, 1)
    WG_XFORM(0)
    WG_T(tp1);
#pragma unroll 1
    for (int it = 0; it < niter; it++) {
        const bool more = it + 1 < niter;
        const float* wn = more ? wp + (size_t)(2 * it + 2) * wstride : wp_next;
        const unsigned einc = more ? 2 * WG_KSTEP : 0;          // past the end: the two extra steps re-read the iteration (unused)
        __builtin_amdgcn_sched_barrier(0);
        if (WG_EXP & 2) { Bn[0] = Bc[1]; Bn[1] = Bc[0]; } else {
        Bn[0] = *reinterpret_cast<const wgf4*>(wn);
        Bn[1] = *reinterpret_cast<const wgf4*>(wn + wstride); }
#pragma unroll
        for (int g = 0; g < 2; g++) {                           // steps 2 NT + g = M-tile g % NT, k-step 2 + g / NT
            E[g][0] = P[g % NT][0] + einc + (g / NT) * WG_KSTEP;
            E[g][1] = P[g % NT][1] + einc + (g / NT) * WG_KSTEP;
        }
#pragma unroll
        for (int s = 0; s < 2 * NT; s++) {
            const int t = T0 + s % NT, kk = s / NT;
            WG_LOADD(s & 1, s + 2)
            WG_XFORM((s + 1) & 1)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bc[kk][j], V[s & 1][j], acc[t][j], 0, 0, 0);
            // the step's LDS reads go first (they are consumed one step later), the VALU work is woven between the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        Bc[0] = Bn[0]; Bc[1] = Bn[1];
#pragma unroll
        for (int t = 0; t < NT; t++) { P[t][0] += 2 * WG_KSTEP; P[t][1] += 2 * WG_KSTEP; }
    }
    WG_T(tp2);
    WG_ACC(2, tp0, tp1);
    WG_ACC(1, tp1, tp2);
#undef WG_LOADD
#undef WG_LOAD2
#undef WG_XFORM
#undef WG_PK_SUB
#undef WG_PK_ADD
}

// All 16 components of one N-tile: Y[t][u][v] = 2 x 2 outputs of the tiles of M-tile t (C/D layout: tile = lane's rows).
// Bc holds the weights of the first two k-steps on entry and those at wp_after (the wavefront's next N-tile) on exit.
template <int T0, int T1>
__device__ __forceinline__ void wg_ntile(unsigned act_addr, int li, int lk, const float* __restrict__ wp, const float* __restrict__ wp_after,
                                         int k4, int wstride, wgf4 bv, wgf4 (&Bc)[2], wgf4 (&Y)[3][2][2])
{
    const size_t pstride = (size_t)k4 * wstride;
    using std::integral_constant;
    auto run = [&](auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        wgf4 acc[3][4];
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[t][j] = (wgf4){ 0.f, 0.f, 0.f, 0.f };
        wg_pass<I, T0, T1>(act_addr, li, lk, wp + I * pstride, I == 3 ? wp_after : wp + (I + 1) * pstride, k4, wstride, Bc, acc);
        WG_T(to0);
#pragma unroll
        for (int t = T0; t < T1; t++) {
            if (I == 3 && t == 2) continue;
            if ((WG_EXP & 8) && I > 0) { Y[t][I & 1][0] += acc[t][0]; continue; }
            const wgf4 s0 = acc[t][0] + acc[t][1] + acc[t][2];
            const wgf4 s1 = acc[t][1] - acc[t][2] - acc[t][3];
            if constexpr (I == 0) { Y[t][0][0] = s0 + bv; Y[t][0][1] = s1 + bv; }
            else if constexpr (I == 1) { Y[t][0][0] += s0; Y[t][0][1] += s1; Y[t][1][0] = s0 + bv; Y[t][1][1] = s1 + bv; }
            else if constexpr (I == 2) { Y[t][0][0] += s0; Y[t][0][1] += s1; Y[t][1][0] -= s0; Y[t][1][1] -= s1; }
            else { Y[t][1][0] -= s0; Y[t][1][1] -= s1; }
        }
#ifdef WG_PROF
        asm volatile("" :: "v"(Y[T0][0][0]), "v"(Y[T0][1][1]));
#endif
        WG_T(to1);
        WG_ACC(3, to0, to1);
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
            if (ok) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v0 = fmaxf(Y[t][u][0][r], lo), v1 = fmaxf(Y[t][u][1][r], lo);
                    if constexpr (GLB) {
                        *reinterpret_cast<wgf2*>(out_glb + (size_t)(n0 + r) * 140 + row * 20 + 2 * tx) = (wgf2){ v0, v1 };
                    } else if (!(WG_EXP & 1) || n0 == 0) {
                        float* p = act + (n0 + r) * WG_CS + row * WG_ROW + 2 * tx;
                        p[1] = v0; p[2] = v1;
                        if (tx == 0) p[21] = v0;            // column 0 again behind column 19
                        if (tx == 9) p[-18] = v1;           // column 19 again in front of column 0
                    }
                }
            }
        }
    }
}

// One layer for the calling wavefront.  NTW N-tiles per wavefront (2 for 128 output channels), M-tiles [T0, T1).
template <int NTW, int T0, int T1, bool GLB>
__device__ __forceinline__ void wg_layer(float* __restrict__ act, float* __restrict__ out_glb, const float* __restrict__ wt,
                                         const float* __restrict__ bias, int cin, int cout, int relu, int nt_first)
{
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));                   // lane-derived offsets stay inside the layer (see convnet history: spills)
    const int li = lane & 15, lk = lane >> 4;
    const unsigned act_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)act;
    const int k4 = cin >> 2, wstride = (cout >> 4) * 256;
    wgf4 Y[NTW][3][2][2];
    wgf4 Bc[2];                                      // weights of the first two k-steps: every pass hands them to the next one
    bool early = false;
    // the upper N-tile first: when its channel rows lie above the layer's input they are free to be written at once
    const int nts[2] = { NTW == 2 ? nt_first + 4 : nt_first, nt_first };
    const float* wp0 = wt + ((size_t)nts[0] * 64 + lane) * 4;
    Bc[0] = *reinterpret_cast<const wgf4*>(wp0);
    Bc[1] = *reinterpret_cast<const wgf4*>(wp0 + wstride);
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        const float* wp = wt + ((size_t)nts[q] * 64 + lane) * 4;
        const float* wp_after = wt + ((size_t)nts[q + 1 < NTW ? q + 1 : q] * 64 + lane) * 4;
        wg_ntile<T0, T1>(act_addr, li, lk, wp, wp_after, k4, wstride, *reinterpret_cast<const wgf4*>(bias + nts[q] * 16 + lk * 4), Bc, Y[q]);
        if (NTW == 2 && q == 0 && !GLB && nts[0] * 16 >= cin) { wg_store<T0, T1, false>(Y[0], nts[0], relu, act, nullptr, li, lk); early = true; }
    }
    WG_T(tb0);
    __syncthreads();                                 // every wavefront has finished reading the layer's input
    WG_T(tb1);
#pragma unroll
    for (int q = 0; q < NTW; q++) {
        if (NTW == 2 && q == 0 && early) continue;
        wg_store<T0, T1, GLB>(Y[q], nts[q], relu, act, out_glb, li, lk);
    }
    WG_T(tb2);
    WG_ACC(4, tb0, tb1);
    WG_ACC(5, tb1, tb2);
}

__global__ void __launch_bounds__(WG_THREADS, 2) k_cyl_net_wg(const float* __restrict__ x, CylWgParams P, float* __restrict__ y)
{
    extern __shared__ float lds[];                   // [128][160]
    float* act = lds;
    WG_T(tk0);
    const int patch = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
#ifdef WG_STAGGER
    if (blockIdx.x >= 256 && blockIdx.x < 512) for (int i = 0; i < WG_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif
    {   // input [48][140] -> rows of 22 with the halo columns
        const float* src = x + (size_t)patch * P.cin[0] * 140;
        for (int i = threadIdx.x; i < P.cin[0] * 140; i += WG_THREADS) {
            const int c = i / 140, pos = i - c * 140;
            const int row = pos / 20, col = pos - row * 20;
            const float v = src[i];
            float* p = act + c * WG_CS + row * WG_ROW + col;
            p[1] = v;
            if (col == 0) p[21] = v;
            if (col == 19) p[-19] = v;
        }
    }
    for (int i = threadIdx.x; i < WG_MAXC * 6; i += WG_THREADS) act[(i / 6) * WG_CS + WG_ZERO + i % 6] = 0.f;
    __syncthreads();
    WG_T(tk1);
    WG_ACC(7, tk0, tk1);
#pragma unroll 1
    for (int l = 0; l < WG_LAYERS; l++) {
        const int cin = P.cin[l], cout = P.cout[l];
        if (cout == 128)     wg_layer<2, 0, 3, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w);
        else if (cout == 64) wg_layer<1, 0, 3, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w);
        else if (l < WG_LAYERS - 1) {
            if (w & 1)       wg_layer<1, 2, 3, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
            else             wg_layer<1, 0, 2, false>(act, nullptr, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
        } else {
            float* glb = y + (size_t)patch * cout * 140;
            if (w & 1)       wg_layer<1, 2, 3, true>(act, glb, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
            else             wg_layer<1, 0, 2, true>(act, glb, P.wt[l], P.bias[l], cin, cout, P.relu[l], w >> 1);
        }
        WG_T(te0);
        __syncthreads();
        WG_T(te1);
        WG_ACC(6, te0, te1);
    }
    WG_T(tk2);
    WG_ACC(0, tk0, tk2);
}

#ifdef WG_PROF
extern "C" int buf_debug_wg_prof(unsigned long long* out_host, int reset)
{
    BUF_CHECK_HIP(hipDeviceSynchronize());
    BUF_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(wg_prof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = { 0 }; BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(wg_prof), z, sizeof(z))); }
    return BUF_OK;
}
#endif

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
        BUF_REQUIRE(P.cin[l] % 8 == 0 && P.cin[l] <= WG_MAXC && (P.cout[l] == 32 || P.cout[l] == 64 || P.cout[l] == 128),
                    BUF_EINVAL, "buf_cylindrical_net_wg: layer %d has unsupported widths %d -> %d", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(l == 0 || P.cin[l] == P.cout[l - 1], BUF_EINVAL, "buf_cylindrical_net_wg: layer %d width mismatch", l);
    }
    BUF_REQUIRE(P.cout[WG_LAYERS - 1] == 32, BUF_EINVAL, "buf_cylindrical_net_wg: the last layer must have 32 channels");
    size_t lds = sizeof(float) * WG_BUF + (getenv("BUF_WG_LDS_PAD") ? atoi(getenv("BUF_WG_LDS_PAD")) : 0);
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds((const void*)k_cyl_net_wg, lds, grant)) return rc;
    if (getenv("BUF_DEBUG_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_cyl_net_wg, WG_THREADS, lds);
        fprintf(stderr, "k_cyl_net_wg: %d workgroups per CU at %zu B LDS\n", nb, lds);
    }
    double macs = 0;
    for (int l = 0; l < WG_LAYERS; l++) macs += 9.0 * P.cin[l] * P.cout[l];
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 2.0 * 140 * macs * npatch, BUF_TIMED_CYL_NET);
    k_cyl_net_wg<<<npatch, WG_THREADS, lds, (hipStream_t)stream>>>(x, P, y);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
