// A13 -- CostVolume + CostNet (models/BUFFER.py:37-66, models/patchnet.py:88-147) as ONE fused fp32-MFMA kernel.
//
// Reference: gather the 20 circular azimuth shifts of the source map, subtract the target map
// ([M,32,20,5,20] = 256 KB per match), ten unpadded Conv3d (3x3x3, 3x3x3, 7 x (3,1,3), (2,1,2)) with
// BatchNorm(affine=False)+ReLU, softmax over the 20 logits, expected index.
//
// Here one workgroup owns one match and the cost volume is never built.  Layer 0 is LINEAR in
//     cost[c][n][k][l] = S[c][k][(l-n) mod 20] - T[c][k][l],
// so its pre-activation separates exactly (only fp32 re-association, <= 1e-6) into
//     out0[o][n'][k'][l'] = b[o] + Sterm[o][k'][(l'-n') mod 20] - Tterm[o][k'][l'],
//     Sterm[o][k'][j]  = sum_{c,dk,e}  Ws[o][c][dk][e]  S[c][k'+dk][(j+e) mod 20],   Ws[..][e] = sum_{dl-dn=e} W0[o][c][dn][dk][dl]
//     Tterm[o][k'][l'] = sum_{c,dk,dl} Wt[o][c][dk][dl] T[c][k'+dk][l'+dl],          Wt        = sum_dn      W0[o][c][dn][dk][dl]
// (the S-term of output (n',k',l') depends on (k', (l'-n') mod 20) only: 60 positions, K = 480; the T-term does not depend
// on n': 54 positions, K = 288): 1.4 M MAC instead of the 26.9 M of the dense layer 0.  The two small maps are two MFMA GEMMs;
// a layer-0 row is then relu(Smap + (b - Tmap)) formed by VALU, six rows (n') at a time into an LDS chunk that layer 1
// consumes at once, so the 124 KB layer-0 activation never exists either.  Layers 2..6 rewrite ONE 80 KB LDS buffer in place
// (two workgroups per CU).  Layers 1..5 are unpadded 3 x 3 correlations over (n, l) -- layer 1 collapses k' (3 -> 1): its three
// k' planes are 96 input channels; 18 -> 16 -> 14 -> 12 -> 10 -> 8; together 92 % of the kernel's matrix instructions in the
// direct form -- and run in the Winograd F(2x2, 3x3) domain on the pass machinery of csrc/convnet_wg.hip (phase A.2 and cw_layer
// below: 0.49 of their direct-form MFMAs).  -DCV_L1_DIRECT=1 keeps layer 1 in the direct form (three-row chunks, a sliding
// window of accumulator rows: 0.83 of the matrix peak executed, and 14 % slower).
// All GEMMs run on v_mfma_f32_16x16x4_f32 (fp32), weights (BN folded; [K][Cout] MFMA-tiled, or G g G^T in the Winograd tiling)
// stream from L2.
// FLOP accounting: bench.py credits the DENSE algorithmic count of SURVEY 8d (0.160 GFLOP/match); executed: 0.0519 GFLOP.
//
// The LDS maps of the direct-form layers are POSITION-major, [position][channels + 4]: the four k-steps of a 16-channel group that a lane feeds
// to the MFMA A operand (channels 16g + 4lk + 0..3 at its position) are one 16-byte ds_read_b128, and the +4 padding
// (row stride = 4 mod 32 banks) keeps eight neighbouring positions on distinct bank quads.  With one wavefront per SIMD
// the number of memory instructions per MFMA, not their bytes, sets the MFMA duty: 1 LDS + 1-2 global loads per group.
#include "common.h"
#include <type_traits>

#define CV_THREADS 256
#define CV_BUF 20480              // floats of the one LDS buffer (largest map: 128 channels x 160, the 12 x 12 map of layers 3 / 4)
#define CV_LAYERS 10
#define CV_C32 36                 // position-row strides (floats) of maps with 32 / 64 / 128 channels
#define CV_C64 68
#define CV_C128 132

typedef float cvx4 __attribute__((ext_vector_type(4)));

// A group's operand loads are woven into the previous group's MFMAs: after every MFMA up to 4 VALU/SALU and the
// group's memory instructions spread evenly (NMEM over NMFMA, rounded up).
#define CV_SCHED_TAIL(NMFMA, NMEM)                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < (NMFMA); i_++) {                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x120, ((NMEM) + (NMFMA) - 1) / (NMFMA), 0); \
    }
typedef const __attribute__((address_space(1))) cvx4* cv_gptr;     // global (not flat) 16-byte weight loads

struct CostNetParams {
    const float* wt[CV_LAYERS];    // [K][Cout] in the MFMA tiling described at cv_gemm_static; K ordering per layer below
    const float* bias[CV_LAYERS];
    // gathered form (buf_cost_volume_net_gather): match i reads rows s_rows[i] / t_rows[i] of a full [rows,32,ele_n,20] map and
    // takes its elevation rows 1..5 (models/BUFFER.py:291-292) inside the kernel; null rows = dense [m,32,5,20] inputs
    const long long* s_rows;
    const long long* t_rows;
    int row_floats, chan_floats, skip_floats;
    const int* only_if;            // nullable: workgroup i runs only if only_if[i] != 0 (the fp32 re-run of buf_cost_volume_net_split_safe)
};

// ---- tile GEMM: acc[t][u] += sum over groups of 4 k-steps ---------------------------------------------------------
// Loader::load(a, g): fills a[p][t] (p = k-step inside the group, t = M-tile) for group g.
// Weights come TILED for the MFMA B operand: block (group g of 16 K-rows, N-tile n) = 256 floats laid out
// [lane = lk*16 + li][p], so the four B values a lane needs for the four k-steps of a group are ONE 16-byte load and a
// wavefront reads 1 KB contiguously; blocks ordered [g][n].  wl = tiled + (nt0*64 + lane)*4.  K-row of (g, p, lk) is
// 16g + 4lk + p: k-step p of a group takes channel 4lk + p from every lane quarter, matching the A loads above.
// Fully unrolled variant for a compile-time group count: every tap/channel-group index, LDS offset and weight
// offset folds to an immediate, so a group costs no address arithmetic at all (the operand loads of the small
// layer-0/1 tiles otherwise take as many issue cycles as their 4-8 MFMAs).
template <int MT, int NT, int D, int TOTAL, typename Loader>
__device__ __forceinline__ void cv_gemm_static(cvx4 (&acc)[MT][NT], Loader& L, const float* __restrict__ wl, int ntot)
{
    float a[D][4][MT], b[D][4][NT];
#define CVS_LOAD(SLOT, G)                                                                                 \
    {                                                                                                     \
        /* one address per group, pinned here (not hoisted out of the caller's loop) and typed as global  \
           memory so that the loads stay global_load (in-order vmcnt), not flat_load */                   \
        cv_gptr wg_ = (cv_gptr)(wl + (size_t)(G) * ntot * 256);                                           \
        asm volatile("" : "+v"(wg_));                                                                     \
        _Pragma("unroll") for (int u = 0; u < NT; u++) {                                                  \
            const cvx4 bv_ = wg_[u * 64];                                                                 \
            _Pragma("unroll") for (int p = 0; p < 4; p++) b[SLOT][p][u] = bv_[p];                         \
        }                                                                                                 \
        L.load(a[SLOT], (G));                                                                             \
    }
#pragma unroll
    for (int i = 0; i < D - 1; i++) CVS_LOAD(i, i)
#pragma unroll
    for (int g = 0; g < TOTAL; g++) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + D - 1 < TOTAL) CVS_LOAD((g + D - 1) % D, g + D - 1)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int u = 0; u < NT; u++)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % D][p][t], b[g % D][p][u], acc[t][u], 0, 0, 0);
        CV_SCHED_TAIL(4 * MT * NT, MT + NT)
    }
#undef CVS_LOAD
}

// ---- phase A operand loaders (every map position-major with a 36-float row: 32 channels + 4 padding) ---------------
// LDS regions of phase A inside the bufB half (floats):
#define CVA_SP 0          // [5][24][36]  source map, azimuth padded circularly by 2 on both sides (column = l + 2)
#define CVA_TP 4320       // [5][20][36]  target map
#ifndef CV_L1_DIRECT
#define CV_L1_DIRECT 0    // 1: layer 1 in the direct form (round 2 / early round 3: three-row chunks, sliding accumulator window)
#endif
#ifndef CV_L1_PAIRS
#define CV_L1_PAIRS 1     // Winograd layer 1: N-tile pairs with K split over the wavefront pair (0: one N-tile per wavefront over all of K)
#endif
#if CV_L1_DIRECT
#define CVA_SM 7920       // [3][20][36]  Sterm[k'][j][o]
#define CVA_TB 10080      // [3][18][36]  b[o] - Tterm[k'][l'][o]
#else
#define CVA_SM 16376      // the two maps sit at the END of the buffer: the six-row chunk of the Winograd layer 1 takes its start
#define CVA_TB 18536      //   (18536 + 3 * 18 * 36 = 20480 = CV_BUF)
#define CVA_XC 0          // [96 = (k', c)][CW_CSX]: six layer-0 rows n' of 18 columns in rows of CW_RSX, channel-major
#ifndef CW_RSX
#define CW_RSX 18         // (rows of 24 and channels 160 apart -- the two tile rows of an M-tile and the two channels of a half-wave on
#define CW_CSX 112        //   the four quarters of the 64 banks, conflict-free -- measured +-0 against this compact layout)
#endif
#define CVA_XCH (96 * CW_CSX)      // 2 x 8 x 64 float4 of partial sums handed over per chunk (CV_L1_PAIRS)
static_assert(CVA_XCH + 2 * 8 * 64 * 4 <= CVA_SM, "the exchange slots fit between the chunk and the layer-0 maps");
#endif
#define CVA_R0 0          // [3][54][36]  chunk of three layer-0 rows (aliases SP/TP once the two small GEMMs are done)
#define CVA_R1 12024      // second chunk buffer; ends at 17856 <= CV_BUF
#define CVA_RROW (54 * CV_C32)

// S-term: A[m=(k',j)][(dk,e), c] = S[c][k'+dk][(j+e) mod 20], e = -2..2;  K = (dk*5 + e+2)*32 + c  (30 groups of 16)
struct SLoader {
    const float* base;                                   // SP + (k'*24 + j)*36 + lk*4   (column j + (e+2) holds azimuth j + e)
    __device__ __forceinline__ void load(float (&a)[4][1], int g) const
    {
        const int tap = g >> 1, cg = g & 1, dk = tap / 5, e2 = tap - dk * 5;
        const cvx4 v = *reinterpret_cast<const cvx4*>(base + (dk * 24 + e2) * CV_C32 + cg * 16);
#pragma unroll
        for (int p = 0; p < 4; p++) a[p][0] = v[p];
    }
};

// T-term: A[m=(k',l')][(dk,dl), c] = T[c][k'+dk][l'+dl];  K = (dk*3 + dl)*32 + c  (18 groups)
struct TLoader {
    const float* base;                                   // TP + (k'*20 + l')*36 + lk*4
    __device__ __forceinline__ void load(float (&a)[4][1], int g) const
    {
        const int tap = g >> 1, cg = g & 1, dk = tap / 3, dl = tap - dk * 3;
        const cvx4 v = *reinterpret_cast<const cvx4*>(base + (dk * 20 + dl) * CV_C32 + cg * 16);
#pragma unroll
        for (int p = 0; p < 4; p++) a[p][0] = v[p];
    }
};

// layer 1, one dn slab over MT layer-0 rows of the chunk: A[t][m=l''][(dk,dl), c] = R[t][dk*18 + l''+dl][c];  K = (dk*3+dl)*32 + c
template <int MT>
struct L1Loader {
    const float* base;                                   // first row of the chunk this call uses + l''*36 + lk*4
    __device__ __forceinline__ void load(float (&a)[4][MT], int g) const
    {
        const int tap = g >> 1, cg = g & 1, dk = tap / 3, dl = tap - dk * 3;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const cvx4 v = *reinterpret_cast<const cvx4*>(base + t * CVA_RROW + (dk * 18 + dl) * CV_C32 + cg * 16);
#pragma unroll
            for (int p = 0; p < 4; p++) a[p][t] = v[p];
        }
    }
};

// three layer-0 rows n' = r0 .. r0+2 into a chunk buffer: R[t][pos=(k',l')][o] = relu(Smap[k'][(l'-n') mod 20][o] + Tb[pos][o])
__device__ __forceinline__ void form_rows(float* __restrict__ R, const float* __restrict__ SM, const float* __restrict__ TB, int r0)
{
    for (int i = threadIdx.x; i < 3 * 54 * 8; i += CV_THREADS) {
        const int t = i / 432, rem = i - t * 432, pos = rem >> 3, c4 = rem & 7;
        const int kq = pos / 18, lq = pos - kq * 18;
        int j = lq - (r0 + t);
        j = j < 0 ? j + 20 : j;
        const cvx4 sv = *reinterpret_cast<const cvx4*>(SM + (kq * 20 + j) * CV_C32 + c4 * 4);
        const cvx4 tv = *reinterpret_cast<const cvx4*>(TB + pos * CV_C32 + c4 * 4);
        cvx4 v;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = fmaxf(sv[q] + tv[q], 0.f);
        *reinterpret_cast<cvx4*>(R + (t * 54 + pos) * CV_C32 + c4 * 4) = v;
    }
}

// layers 2..9: valid (KW x KW) convolution over an LDS-resident [WIN*WIN][CIN+4] map;  K = (dn*KW + dl)*CIN + c.
// The tap loop is a runtime loop, the CIN/16 channel groups of a tap are unrolled: inside a tap every LDS and
// weight offset is an immediate on one per-tile base register, so a group of 4 k-steps issues only its loads
// (one ds_read_b128 per M-tile, one global_load_dwordx4 per N-tile) and its MFMAs.
// Operands of group g+1 (possibly the first group of the next tap) are loaded while group g multiplies.
// INPLACE: `out` is the buffer `in` lives in (all four wavefronts take part): the accumulators ARE the outputs, so they simply
// wait for the barrier that ends everybody's reads -- one 75 KB buffer instead of two, i.e. two workgroups per CU (round 3).
template <int MT, int NT, int CIN, int COUT, int WIN, int KW, bool INPLACE>
__device__ __forceinline__ void cv_conv_layer(const float* in, float* out, const float* __restrict__ wt,
                                              const float* __restrict__ bias, int nt0, bool relu)
{
    constexpr int WOUT = WIN - KW + 1, P = WOUT * WOUT, GPT = CIN / 16, TAPS = KW * KW, NTOT = COUT / 16;
    constexpr int CSI = CIN + 4, CSO = COUT + 4;                        // position-row strides of the two maps
    static_assert(GPT % 2 == 0, "two pipeline slots alternate per channel group");
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));       // lane-derived addresses are formed per layer (hoisted to the kernel's start they are spilled)
    const int li = lane & 15, lk = lane >> 4;
    const float* pa[MT];                 // lane's 4 channels (4lk..4lk+3 of group 0) of tile t at tap (0,0)
#pragma unroll
    for (int t = 0; t < MT; t++) {
        int m = t * 16 + li;
        m = m < P ? m : P - 1;           // padding rows of the last tile recompute position P-1 (never stored)
        pa[t] = in + ((m / WOUT) * WIN + (m % WOUT)) * CSI + lk * 4;
    }
    const float* wl = wt + ((size_t)nt0 * 64 + lane) * 4;
    cvx4 acc[MT][NT];                    // start at the bias of the output channel (C/D layout: column = lane & 15)
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const float bv = bias[(nt0 + u) * 16 + li];
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t][u] = (cvx4){ bv, bv, bv, bv };
    }
    float a[2][4][MT], b[2][4][NT];
#define CVC_LOAD(SLOT, PTRS, WTAP, CG)                                                                    \
    {                                                                                                     \
        cv_gptr wg_ = (cv_gptr)((WTAP) + (size_t)(CG) * NTOT * 256);                                      \
        asm volatile("" : "+v"(wg_));                                                                     \
        _Pragma("unroll") for (int u = 0; u < NT; u++) {                                                  \
            const cvx4 bv_ = wg_[u * 64];                                                                 \
            _Pragma("unroll") for (int p = 0; p < 4; p++) b[SLOT][p][u] = bv_[p];                         \
        }                                                                                                 \
        _Pragma("unroll") for (int t = 0; t < MT; t++) {                                                  \
            const cvx4 av_ = *reinterpret_cast<const cvx4*>(PTRS[t] + (CG) * 16);                         \
            _Pragma("unroll") for (int p = 0; p < 4; p++) a[SLOT][p][t] = av_[p];                         \
        }                                                                                                 \
    }
    const float* cur[MT];
    const float* nxt[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) cur[t] = pa[t];
    const float* wcur = wl;
    CVC_LOAD(0, cur, wcur, 0)
#pragma unroll 1
    for (int tap = 0; tap < TAPS; tap++) {
        const int tn = tap + 1 < TAPS ? tap + 1 : tap;       // the last tap prefetches itself again (unused)
        const int offn = ((tn / KW) * WIN + (tn % KW)) * CSI;
#pragma unroll
        for (int t = 0; t < MT; t++) nxt[t] = pa[t] + offn;
        const float* wnxt = wl + (size_t)tn * GPT * NTOT * 256;
#pragma unroll
        for (int cg = 0; cg < GPT; cg++) {
            __builtin_amdgcn_sched_barrier(0);
            if (cg + 1 < GPT) CVC_LOAD((cg + 1) & 1, cur, wcur, cg + 1)
            else              CVC_LOAD((cg + 1) & 1, nxt, wnxt, 0)
    #pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int t = 0; t < MT; t++)
#pragma unroll
                    for (int u = 0; u < NT; u++)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cg & 1][p][t], b[cg & 1][p][u], acc[t][u], 0, 0, 0);
            CV_SCHED_TAIL(4 * MT * NT, MT + NT)
        }
#pragma unroll
        for (int t = 0; t < MT; t++) cur[t] = nxt[t];
        wcur = wnxt;
    }
#undef CVC_LOAD
    if constexpr (INPLACE) __syncthreads();          // every wavefront has finished reading the map this one overwrites
    // epilogue: ReLU + store; C/D layout: a lane holds channel n at the four positions m = 16t + 4lk + r
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const int n = (nt0 + u) * 16 + li;
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = t * 16 + lk * 4 + r;
                float v = acc[t][u][r];
                if (relu) v = fmaxf(v, 0.f);
                if (m < P) out[m * CSO + n] = v;
            }
    }
}

// ---- layers 2..5 in the Winograd F(2x2, 3x3) domain (round 3) -------------------------------------------------------------
// The (3,1,3) layers are unpadded 3x3 correlations over the (n, l) map: 16 -> 14 -> 12 -> 10 -> 8.  With the output cut into 2 x 2
// tiles they run as the 16 component GEMMs of csrc/convnet_wg.hip (wg_round: the same passes, input transform, weight ring and
// filter tiling) over M-tiles of whole tile rows -- 64 / 48 / 32 / 16 tile rows instead of 13 / 9 / 7 / 4 M-tiles x 9 taps:
// 0.55 / 0.59 / 0.51 / 0.44 of the direct form's matrix instructions, 69 % of the kernel's before.  These maps are
// CHANNEL-major, [C][CS] with rows of RS floats (a window row of a tile = two ds_read_b64); CS is chosen per map so that the two
// channels a half-wave reads land on different banks.  A wavefront owns an N-tile pair where the layer has 8 N-tiles or can
// split its M-tiles evenly (one transform feeds 8 MFMAs), and holds its outputs in registers until every wavefront has read.
template <int HIN, int TYPM>
__device__ __forceinline__ bool cw_tile(int t, int li, int& ty, int& tx)
{
    constexpr int NTY = (HIN - 2) / 2;                   // tiles per row and per column
    const int q = li / NTY;
    ty = t * TYPM + q; tx = li - q * NTY;
    return q < TYPM && ty < NTY;
}

// outputs of one M-tile of one N-tile: ReLU, then into the channel-major map [C][CSO] (rows of RSO) or, for the layer that
// feeds the direct-form layers again, position-major [(row, col)][CSO]
template <int HIN, int TYPM, int RSO, int CSO, bool POSMAJOR>
__device__ __forceinline__ void cw_store_tile(const wgf4 (&Yt)[2][2], int t, int nt, float* __restrict__ out, int li, int lk)
{
    int ty, tx;
    if (!cw_tile<HIN, TYPM>(t, li, ty, tx)) return;
    const int n0 = nt * 16 + lk * 4;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int row = 2 * ty + u, col = 2 * tx;
        if constexpr (POSMAJOR) {
#pragma unroll
            for (int v = 0; v < 2; v++) {
                wgf4 y;
#pragma unroll
                for (int r = 0; r < 4; r++) y[r] = fmaxf(Yt[u][v][r], 0.f);
                *reinterpret_cast<wgf4*>(out + (row * (HIN - 2) + col + v) * CSO + n0) = y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                *reinterpret_cast<wgf2*>(out + (n0 + r) * CSO + row * RSO + col) = (wgf2){ fmaxf(Yt[u][0][r], 0.f), fmaxf(Yt[u][1][r], 0.f) };
        }
    }
}

// One Winograd layer over the map in `map` (HIN x HIN, rows of RS, channel stride CS), rewritten in place: this wavefront's
// N-group `ng` (NN N-tiles) over the M-tiles t0 .. t0 + MT - 1 in one round and MT2 more in a second one.
template <int HIN, int RS, int CS, int TYPM, int CIN, int NN, int MT, int MT2, int RSO, int CSO, bool POSMAJOR>
__device__ __forceinline__ void cw_layer(float* __restrict__ map, const float* __restrict__ wt, const float* __restrict__ bias, int ng, int t0)
{
    static_assert(MT >= 1 && MT <= 2 && MT2 >= 0 && MT2 <= 1 && RS % 2 == 0 && CS % 2 == 0, "");
    constexpr unsigned KSTEP = 16u * CS;
#ifdef CW_EXP_SAMEW
    constexpr int k4 = CIN / 4;
    const int wstride = (int)(blockDim.x >> 10);              // timing experiment: a run-time zero (every k-step fetches the same KB: L1 hits)
#else
    constexpr int k4 = CIN / 4, wstride = 256 * NN;
#endif
    int lane = threadIdx.x & (WAVE - 1);
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)map;
    auto rows = [&](int t, unsigned (&ra)[4]) __attribute__((always_inline)) {
        int ty, tx;
        if (!cw_tile<HIN, TYPM>(t, li, ty, tx)) { ty = 0; tx = 0; }              // idle lanes recompute tile (0, 0), never stored
#pragma unroll
        for (int a = 0; a < 4; a++) ra[a] = base + 4u * (unsigned)(lk * CS + (2 * ty + a) * RS + 2 * tx);
    };
    unsigned RA[3][4];
    rows(t0, RA[0]);
    rows(t0 + (MT > 1 ? 1 : 0), RA[1]);
    rows(t0, RA[2]);
    const __amdgpu_buffer_rsrc_t rs = wg_weights(wt);
    const unsigned wp = (unsigned)ng * (CIN * NN * 256);          // [N-group][i 0..3][k-step][n2][lane][j]
    const unsigned lofs = lane * 16;
    const float* bl = bias + ng * NN * 16 + lk * 4;
    wgf4 W[NN][2];
    wg_first_weights<NN, 0, 2>(rs, wp, lofs, wstride, W);
    wgf4 Y[NN][3][2][2];
    wg_round<NN, 0, MT, KSTEP, true>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, (unsigned)(k4 * wstride), bl, W, Y);
    wgf4 Y2[NN][3][2][2];
    if constexpr (MT2 > 0) {
        rows(t0 + MT, RA[0]);
        wg_round<NN, 0, 1, KSTEP, true>(RA, rs, wp, wp, lofs, k4 >> 2, wstride, (unsigned)(k4 * wstride), bl, W, Y2);
    }
    __syncthreads();                         // every wavefront has finished reading the map
#pragma unroll
    for (int n = 0; n < NN; n++) {
#pragma unroll
        for (int t = 0; t < MT; t++) cw_store_tile<HIN, TYPM, RSO, CSO, POSMAJOR>(Y[n][t], t0 + t, ng * NN + n, map, li, lk);
        if constexpr (MT2 > 0) cw_store_tile<HIN, TYPM, RSO, CSO, POSMAJOR>(Y2[n][0], t0 + MT, ng * NN + n, map, li, lk);
    }
}
#ifndef CW_NN2
#define CW_NN2 2          // N-tiles per wavefront in layers 2 and 4 (the filter tiling follows: buf_cost_winograd_group)
#endif
#ifndef CW_NN4
#define CW_NN4 2
#endif
#define CW_CS1 272        // channel strides of the maps after layers 1, 2 (= 16 mod 64: the two tile rows of an M-tile leave the
#define CW_CS2 208        //   banks 16..31 / 48..63 to the half-wave's second channel), 3 and 4 (= 32 mod 64)
#define CW_CS3 160

#if !CV_L1_DIRECT
// six layer-0 rows n' = r0 .. r0+5 into the channel-major chunk: X[(k', o)][t * 18 + l'] = relu(Smap[k'][(l'-n') mod 20][o] + Tb[k'][l'][o])
__device__ __forceinline__ void form_rows_cm(float* __restrict__ X, const float* __restrict__ SM, const float* __restrict__ TB, int r0)
{
    for (int i = threadIdx.x; i < 6 * 54 * 8; i += CV_THREADS) {
        const int c4 = i & 7, rem = i >> 3, t = rem / 54, pos = rem - t * 54;
        const int kq = pos / 18, lq = pos - kq * 18;
        int j = lq - (r0 + t);
        j = j < 0 ? j + 20 : j;
        const cvx4 sv = *reinterpret_cast<const cvx4*>(SM + (kq * 20 + j) * CV_C32 + c4 * 4);
        const cvx4 tv = *reinterpret_cast<const cvx4*>(TB + pos * CV_C32 + c4 * 4);
        float* d = X + (kq * 32 + c4 * 4) * CW_CSX + t * CW_RSX + lq;
#pragma unroll
        for (int q = 0; q < 4; q++) d[q * CW_CSX] = fmaxf(sv[q] + tv[q], 0.f);
    }
}
#endif

#ifdef CV_STAMP
__device__ long long* cv_stamp_ptr;       // development build (-DCV_STAMP): s_memtime of wavefront 0 at the phase boundaries
#define CV_STAMP_AT(SLOT) if (threadIdx.x == 0) cv_stamp_ptr[(size_t)blockIdx.x * 16 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define CV_STAMP_AT(SLOT)
#endif

__device__ __forceinline__ void cost_net_body(const float* __restrict__ s_eq, const float* __restrict__ t_eq, const CostNetParams& P,
                                              float* __restrict__ ind_out, float* __restrict__ lds)
{
    // lds: ONE buffer of CV_BUF floats (75 KB): two workgroups per CU
    float* bufA = lds;                       // every map from layer 1 on (layers 2..6 rewrite it in place)
    float* bufB = lds;                       // phase A: the maps of the separated layer 0 and the row chunks live here first
    float* SP = bufB + CVA_SP;
    float* TP = bufB + CVA_TP;
    float* SM = bufB + CVA_SM;
    float* TB = bufB + CVA_TB;
    const int match = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), li = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid / WAVE);        // wavefront-uniform for the compiler too (scalar weight offsets)
    {   // both maps, transposed on the way in: global [c][k][l] -> LDS [k][l][c]; S with its two wrap-around columns per side
        const float* a = s_eq + (P.s_rows ? (size_t)P.s_rows[match] * P.row_floats + P.skip_floats : (size_t)match * 3200);
        const float* b = t_eq + (P.t_rows ? (size_t)P.t_rows[match] * P.row_floats + P.skip_floats : (size_t)match * 3200);
        for (int i = tid; i < 800; i += CV_THREADS) {
            const int c = i / 25, rem = i - c * 25, k = rem / 5, l0 = (rem - k * 5) * 4;
            // (read once: streamed past the L2, which is for the 2.9 MB of filters every workgroup walks)
            const cvx4 sv = __builtin_nontemporal_load(reinterpret_cast<const cvx4*>(a + c * P.chan_floats + rem * 4));      // 100 contiguous floats per channel
            const cvx4 tv = __builtin_nontemporal_load(reinterpret_cast<const cvx4*>(b + c * P.chan_floats + rem * 4));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int l = l0 + q;
                SP[(k * 24 + l + 2) * CV_C32 + c] = sv[q];
                if (l >= 18) SP[(k * 24 + l - 18) * CV_C32 + c] = sv[q];
                if (l < 2) SP[(k * 24 + l + 22) * CV_C32 + c] = sv[q];
                TP[(k * 20 + l) * CV_C32 + c] = tv[q];
            }
        }
    }
    __syncthreads();
    CV_STAMP_AT(0)

    // ---- phase A.1: the two small GEMMs of the separated layer 0; wave w owns M-tile w (16 positions), both N-tiles ----
    {
        const float b0v[2] = { P.bias[0][li], P.bias[0][16 + li] };
        int ms = w * 16 + li, mt = ms;
        ms = ms < 60 ? ms : 59;                              // padding rows recompute the last position (never stored)
        mt = mt < 54 ? mt : 53;
        SLoader LS;
        LS.base = SP + ((ms / 20) * 24 + ms % 20) * CV_C32 + lk * 4;
        TLoader LT;
        LT.base = TP + ((mt / 18) * 20 + mt % 18) * CV_C32 + lk * 4;
        cvx4 accS[1][2] = { { (cvx4){ 0.f, 0.f, 0.f, 0.f }, (cvx4){ 0.f, 0.f, 0.f, 0.f } } };
        cvx4 accT[1][2] = { { (cvx4){ 0.f, 0.f, 0.f, 0.f }, (cvx4){ 0.f, 0.f, 0.f, 0.f } } };
        cv_gemm_static<1, 2, 4, 30>(accS, LS, P.wt[0] + (size_t)lane * 4, 2);
        cv_gemm_static<1, 2, 4, 18>(accT, LT, P.wt[0] + (size_t)30 * 2 * 256 + (size_t)lane * 4, 2);
        // C/D layout: lane holds channel n = 16u + li at the four positions m = 16w + 4lk + r
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = w * 16 + lk * 4 + r;
                if (m < 60) SM[m * CV_C32 + u * 16 + li] = accS[0][u][r];
                if (m < 54) TB[m * CV_C32 + u * 16 + li] = b0v[u] - accT[0][u][r];
            }
    }
    __syncthreads();                         // SM/TB complete; SP/TP dead from here on (chunk buffer R0 takes their place)
    CV_STAMP_AT(1)

#if CV_L1_DIRECT
    // ---- phase A.2: layer 1 over chunks of three layer-0 rows; wave w owns N-tile w (16 of the 64 output channels) ----
    // Row ra = 3j + t of chunk j feeds output row ra - dn through slab dn: accumulators acc5[ra - dn - (3j - 2)] hold the five
    // live output rows 3j-2 .. 3j+2; rows 3j-2, 3j-1, 3j are complete after chunk j.  Tiles whose output row falls outside 0..15
    // (chunk 0: t < dn, chunk 5: t > dn) are not computed.
    {
        const float b1v = P.bias[1][w * 16 + li];
        const float* w1 = P.wt[1] + ((size_t)w * 64 + lane) * 4;          // N-tile w of 4; a dn slab = 18 groups
        cvx4 acc5[5];
        // the 16 output rows of this wavefront's N-tile wait in registers (64) until the last chunk has been read: the layer-1
        // map then takes the place of the chunks
        cvx4 done[16];
#pragma unroll
        for (int i = 0; i < 5; i++) acc5[i] = (cvx4){ 0.f, 0.f, 0.f, 0.f };
        form_rows(bufB + CVA_R0, SM, TB, 0);
        __syncthreads();
#define CV_SLAB(DN, T0, T1)                                                                              \
    {                                                                                                    \
        constexpr int MT_ = (T1) - (T0);                                                                 \
        L1Loader<MT_> L;                                                                                 \
        L.base = Rc + (T0) * CVA_RROW + li * CV_C32 + lk * 4;                                            \
        cvx4 a_[MT_][1];                                                                                 \
        _Pragma("unroll") for (int t = 0; t < MT_; t++) a_[t][0] = acc5[(T0) + t - (DN) + 2];            \
        cv_gemm_static<MT_, 1, 4, 18>(a_, L, w1 + (size_t)(DN) * 18 * 4 * 256, 4);                       \
        _Pragma("unroll") for (int t = 0; t < MT_; t++) acc5[(T0) + t - (DN) + 2] = a_[t][0];            \
    }
        auto chunk = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const float* Rc = bufB + ((j & 1) ? CVA_R1 : CVA_R0);
            if (j < 5) form_rows(bufB + ((j & 1) ? CVA_R0 : CVA_R1), SM, TB, 3 * j + 3);     // next chunk, other buffer
            if constexpr (j == 0)      { CV_SLAB(0, 0, 3) CV_SLAB(1, 1, 3) CV_SLAB(2, 2, 3) }
            else if constexpr (j == 5) { CV_SLAB(0, 0, 1) CV_SLAB(1, 0, 2) CV_SLAB(2, 0, 3) }
            else                       { CV_SLAB(0, 0, 3) CV_SLAB(1, 0, 3) CV_SLAB(2, 0, 3) }
#pragma unroll
            for (int i = 0; i < 3; i++) {                    // completed output rows 3j-2, 3j-1, 3j
                constexpr int nb = 3 * j - 2;
                if (nb + i >= 0 && nb + i < 16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) done[(nb + i) & 15][r] = fmaxf(acc5[i][r] + b1v, 0.f);
                }
            }
            acc5[0] = acc5[3]; acc5[1] = acc5[4];
            acc5[2] = acc5[3] = acc5[4] = (cvx4){ 0.f, 0.f, 0.f, 0.f };
            __syncthreads();                 // the chunk just read may be overwritten; the next one is complete
        };
        chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)                      // channel-major for the Winograd layers: [64][CW_CS1], rows of 16
            *reinterpret_cast<cvx4*>(bufA + (w * 16 + li) * CW_CS1 + n2 * 16 + lk * 4) = done[n2];
        __syncthreads();
    CV_STAMP_AT(2)
#undef CV_SLAB
    }

#else
#if CV_L1_PAIRS
    // ---- phase A.2: layer 1 in the Winograd domain.  Its 3 x 3 x 3 filter collapses k' (3 -> 1), so it is a 3 x 3 correlation over
    // (n', l') with the three k' planes as input channels (96): 8 x 8 tiles of 2 x 2 outputs.  The 124 KB layer-0 map never
    // exists: four chunks of six rows n' = 4j .. 4j+5 (two tile rows = one M-tile of 16 tiles) are formed channel-major one after
    // the other.  Wavefront (p, kh) owns the N-tile PAIR p over half kh of K (48 channels): one input transform feeds 8 MFMAs.  The two
    // K halves of a pair share the stores (kh = 0: chunks 0, 1; kh = 1: chunks 2, 3): the partial sums of a chunk go from the half
    // that does not store it to the one that does, which holds its 2 x 32 output registers until the last chunk has been read.
    // 64 tile rows x 16 components x 96 channels = 0.44 of the direct form's matrix instructions (3 456 -> 1 536 per wavefront).
    {
        int lane_ = threadIdx.x & (WAVE - 1);
        asm volatile("" : "+v"(lane_));
        const int li_ = lane_ & 15, lk_ = lane_ >> 4;
        const int p_ = w & 1, kh = w >> 1;
        float* XC = bufB + CVA_XC;
        constexpr unsigned KSTEP = 16u * CW_CSX;
        const __amdgpu_buffer_rsrc_t rs = wg_weights(P.wt[1]);
        const unsigned wp = (unsigned)p_ * (96 * 512) + (unsigned)(12 * kh) * 512;      // [pair][i 0..3][k-step 0..23][n2][lane][j]
        const unsigned lofs = lane_ * 16;
        unsigned RA[3][4];
        {
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)XC + (unsigned)(12 * kh) * KSTEP;
#pragma unroll
            for (int a = 0; a < 4; a++) RA[0][a] = RA[1][a] = RA[2][a] = base + 4u * (unsigned)(lk_ * CW_CSX + (2 * (li_ >> 3) + a) * CW_RSX + 2 * (li_ & 7));
        }
        const float* bl = kh ? nullptr : P.bias[1] + p_ * 32 + lk_ * 4;               // the bias rides in the lower K half
        wgf4 W[2][2];
        wg_first_weights<2, 0, 2>(rs, wp, lofs, 512, W);
        // kh = 0 stores chunks 0, 1 and kh = 1 chunks 2, 3: after a chunk's round the other half hands its 8 quads over through the
        // 16 KB between the chunk buffer and the layer-0 maps, before the barrier that ends the chunk; the owner adds them behind it
        wgf4* slot = reinterpret_cast<wgf4*>(bufB + CVA_XCH) + (size_t)p_ * 8 * WAVE + lane_;
        wgf4 Yown[2][2][3][2][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            form_rows_cm(XC, SM, TB, 4 * j);
            __syncthreads();
            wgf4 Yt[2][3][2][2];
            wg_round<2, 0, 1, KSTEP, true>(RA, rs, wp, wp, lofs, 3, 512, 24u * 512u, bl, W, Yt);
            const bool own = (j < 2) == (kh == 0);
            if (!own) {
#pragma unroll
                for (int q = 0; q < 8; q++) slot[q * WAVE] = Yt[q >> 2][0][(q >> 1) & 1][q & 1];
            }
            __syncthreads();                 // the chunk may be overwritten; the partial sums handed over are in place
            if (own) {
#pragma unroll
                for (int q = 0; q < 8; q++) Yown[j & 1][q >> 2][0][(q >> 1) & 1][q & 1] = Yt[q >> 2][0][(q >> 1) & 1][q & 1] + slot[q * WAVE];
            }
        }
        // (the last take and the next barrier: the map below covers the slots)
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int n = 0; n < 2; n++)
                cw_store_tile<18, 2, 16, CW_CS1, false>(Yown[jj][n][0], 2 * kh + jj, 2 * p_ + n, bufA, li_, lk_);
        __syncthreads();
        CV_STAMP_AT(2)
    }
#else
    // ---- phase A.2: layer 1 in the Winograd domain.  Its 3 x 3 x 3 filter collapses k' (3 -> 1), so it is a 3 x 3 correlation over
    // (n', l') with the three k' planes as input channels (96): 8 x 8 tiles of 2 x 2 outputs.  The 124 KB layer-0 map never
    // exists: four chunks of six rows n' = 4j .. 4j+5 (two tile rows = one M-tile of 16 tiles) are formed channel-major one after
    // the other; wave w owns N-tile w and holds its 4 x 16 output registers until the last chunk has been read.
    // 64 tile rows x 16 components x 96 channels = 0.44 of the direct form's matrix instructions (3 456 -> 1 536 per wavefront).
    {
        int lane_ = threadIdx.x & (WAVE - 1);
        asm volatile("" : "+v"(lane_));
        const int li_ = lane_ & 15, lk_ = lane_ >> 4;
        float* XC = bufB + CVA_XC;
        constexpr unsigned KSTEP = 16u * CW_CSX;
        const __amdgpu_buffer_rsrc_t rs = wg_weights(P.wt[1]);
        const unsigned wp = (unsigned)w * (96 * 256);                      // [N-tile][i 0..3][k-step 0..23][lane][j]
        const unsigned lofs = lane_ * 16;
        unsigned RA[3][4];
        {
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)XC;
#pragma unroll
            for (int a = 0; a < 4; a++) RA[0][a] = RA[1][a] = RA[2][a] = base + 4u * (unsigned)(lk_ * CW_CSX + (2 * (li_ >> 3) + a) * CW_RSX + 2 * (li_ & 7));
        }
        wgf4 W[1][2];
        wg_first_weights<1, 0, 2>(rs, wp, lofs, 256, W);
        wgf4 Y1[4][1][3][2][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            form_rows_cm(XC, SM, TB, 4 * j);
            __syncthreads();
            wg_round<1, 0, 1, KSTEP, true>(RA, rs, wp, wp, lofs, 6, 256, 24u * 256u, P.bias[1] + w * 16 + lk_ * 4, W, Y1[j]);
            __syncthreads();                 // the chunk may be overwritten
        }
#pragma unroll
        for (int j = 0; j < 4; j++) cw_store_tile<18, 2, 16, CW_CS1, false>(Y1[j][0][0], j, w, bufA, li_, lk_);
        __syncthreads();
        CV_STAMP_AT(2)
    }
#endif
#endif
    // ---- phase B: layers 2..6 rewrite the buffer in place; the three tiny last layers hop through its free parts -----------
    //                                 in: size rows stride | tile rows per M-tile | Cin | N-tiles, M-tiles per wavefront | out
#if CW_NN2 == 4
    cw_layer<16, 16, CW_CS1, 2, 64, 4, 1, 0, 14, CW_CS2, false>(bufA, P.wt[2], P.bias[2], 0, w);                  // 16x16 -> 14x14, 64 ch: M-tile w, all N-tiles
#else
    cw_layer<16, 16, CW_CS1, 2, 64, 2, 2, 0, 14, CW_CS2, false>(bufA, P.wt[2], P.bias[2], w & 1, 2 * (w >> 1));   // 16x16 -> 14x14, 64 ch
#endif
    __syncthreads();
    CV_STAMP_AT(3)
    cw_layer<14, 14, CW_CS2, 2, 64, 2, 2, 1, 12, CW_CS3, false>(bufA, P.wt[3], P.bias[3], w, 0);                  // -> 12x12, 128 ch
    __syncthreads();
    CV_STAMP_AT(4)
#if CW_NN4 == 4
    cw_layer<12, 12, CW_CS3, 3, 128, 4, 1, 0, 12, CW_CS3, false>(bufA, P.wt[4], P.bias[4], w & 1, w >> 1);        // -> 10x10 (rows of 12): N-tile quad, one M-tile
#else
    cw_layer<12, 12, CW_CS3, 3, 128, 2, 2, 0, 12, CW_CS3, false>(bufA, P.wt[4], P.bias[4], w, 0);                 // -> 10x10 (rows of 12)
#endif
    __syncthreads();
    CV_STAMP_AT(5)
    cw_layer<10, 12, CW_CS3, 4, 128, 1, 1, 0, 0, CV_C64, true>(bufA, P.wt[5], P.bias[5], w, 0);                   // -> 8x8, position-major
    __syncthreads();
    CV_STAMP_AT(6)
    cv_conv_layer<3, 1, 64, 64, 8, 3, true>(bufA, bufA, P.wt[6], P.bias[6], w, true);            // -> 6x6 (36 x 68 floats)
    __syncthreads();
    CV_STAMP_AT(7)
    float* hop1 = lds + 4096;
    float* hop2 = lds + 8192;
    float* hop3 = lds + 12288;
    if (w < 2) cv_conv_layer<1, 1, 64, 32, 6, 3, false>(bufA, hop1, P.wt[7], P.bias[7], w, true);    // -> 4x4
    __syncthreads();
    if (w < 2) cv_conv_layer<1, 1, 32, 32, 4, 3, false>(hop1, hop2, P.wt[8], P.bias[8], w, true);    // -> 2x2
    __syncthreads();
    if (w < 2) cv_conv_layer<1, 1, 32, 32, 2, 2, false>(hop2, hop3, P.wt[9], P.bias[9], w, false);   // -> 1x1, 20 (+12 zero) logits
    __syncthreads();
    CV_STAMP_AT(8)
    if (w == 0) {                            // softmax over the 20 logits, expected index (BUFFER.py:63-65)
        float v = lane < 20 ? hop3[lane] : -3.4e38f;               // the 1x1 map: position 0, channels 0..19
        float mx = v;
        for (int d = WAVE / 2; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, WAVE));
        float e = lane < 20 ? expf(v - mx) : 0.f;
        float se = e, sw = e * (float)lane;
        for (int d = WAVE / 2; d > 0; d >>= 1) { se += __shfl_xor(se, d, WAVE); sw += __shfl_xor(sw, d, WAVE); }
        if (lane == 0) ind_out[match] = sw / se;
    }
}

__global__ void __launch_bounds__(CV_THREADS, 2) k_cost_net(const float* __restrict__ s_eq, const float* __restrict__ t_eq, CostNetParams P,
                                                        float* __restrict__ ind_out)
{
    extern __shared__ float lds[];
    cost_net_body(s_eq, t_eq, P, ind_out, lds);
}

// The masked re-run of buf_cost_volume_net_split_safe (matches the split-f16 kernel flagged): its own kernel name, see k_cyl_net_wg_rerun.
__global__ void __launch_bounds__(CV_THREADS, 2) k_cost_net_rerun(const float* __restrict__ s_eq, const float* __restrict__ t_eq, CostNetParams P,
                                                              float* __restrict__ ind_out)
{
    extern __shared__ float lds[];
    if (P.only_if[blockIdx.x] == 0) return;
    cost_net_body(s_eq, t_eq, P, ind_out, lds);
}

// N-tiles per group in the Winograd filter tiling of layer l (2..5): what buf_winograd_tile_filters is to be called with
extern "C" int buf_cost_winograd_group(int layer)
{
    return layer == 1 ? (CV_L1_DIRECT ? 0 : (CV_L1_PAIRS ? 2 : 1)) : layer == 2 ? CW_NN2 : layer == 3 ? 2 : layer == 4 ? CW_NN4 : layer == 5 ? 1 : 0;
}

static int cost_net_launch(const float* s_eq, const float* t_eq, int m, const float* const* wt_host, const float* const* bias_host,
                           const long long* s_rows, const long long* t_rows, int ele_n, float* ind_out, void* stream, const char* who,
                           const int* only_if = nullptr)
{
    CostNetParams P;
    for (int l = 0; l < CV_LAYERS; l++) {
        P.wt[l] = wt_host[l]; P.bias[l] = bias_host[l];
        BUF_REQUIRE(P.wt[l] && P.bias[l], BUF_EINVAL, "%s: null weights for layer %d", who, l);
    }
    P.s_rows = s_rows; P.t_rows = t_rows;
    P.chan_floats = s_rows ? ele_n * 20 : 100;
    P.row_floats = 32 * P.chan_floats;
    P.skip_floats = s_rows ? 20 : 0;                         // elevation row 0 of every channel is not part of the cost volume
    P.only_if = only_if;
    size_t lds = sizeof(float) * CV_BUF;
    static LdsGrant grant, grant_rerun;
    if (int rc = only_if ? grant_dynamic_lds((const void*)k_cost_net_rerun, lds, grant_rerun) : grant_dynamic_lds((const void*)k_cost_net, lds, grant)) return rc;
    // EXECUTED flops per match: layer 0 in its separated form (S-term 60 x 480 x 32, T-term 54 x 288 x 32 MAC instead of the
    // dense 972 x 864 x 32), then the valid convolutions 18x3x18 -> 16x1x16 -> 14 -> 12 -> 10 -> 8 -> 6 -> 4 -> 2 -> 1:
    // 2 * sum(out positions * K * Cout), layers 1..5 with 16 products per 2 x 2 output tile (Winograd) = 0.0519 GFLOP.  The dense algorithmic count of SURVEY 8d is 0.160 GFLOP/match
    // (bench.py reports both; the roofline fraction is taken on the executed count).
    static const double macs_per_match =
        60.0 * 480 * 32 + 54.0 * 288 * 32 + (CV_L1_DIRECT ? 256.0 * 864 * 64 : 16.0 * 64 * 96 * 64) +
        16.0 * (49.0 * 64 * 64 + 36.0 * 64 * 128 + 25.0 * 128 * 128 + 16.0 * 128 * 64) +        // layers 2..5: 16 components per 2 x 2 tile
        36.0 * 576 * 64 + 16.0 * 576 * 32 + 4.0 * 288 * 32 + 1.0 * 128 * 20;
    TimedSpan span;
    bool timed = !only_if && timing_begin((hipStream_t)stream, &span, 2.0 * macs_per_match * m, BUF_TIMED_COST_NET);   // (a masked re-run is not a full launch)
#ifdef CV_STAMP
    long long* stamps = nullptr;
    BUF_CHECK_HIP(hipMalloc(&stamps, (size_t)m * 16 * sizeof(long long)));
    BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cv_stamp_ptr), &stamps, sizeof(stamps)));
#endif
    if (only_if) k_cost_net_rerun<<<m, CV_THREADS, lds, (hipStream_t)stream>>>(s_eq, t_eq, P, ind_out);
    else k_cost_net<<<m, CV_THREADS, lds, (hipStream_t)stream>>>(s_eq, t_eq, P, ind_out);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
#ifdef CV_STAMP
    if (m >= 2048) {
        BUF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        long long* h = (long long*)malloc((size_t)m * 16 * sizeof(long long));
        BUF_CHECK_HIP(hipMemcpy(h, stamps, (size_t)m * 16 * sizeof(long long), hipMemcpyDeviceToHost));
        static const char* name[8] = { "A.1 (layer 0 maps)", "A.2 (layer 1)", "layer 2", "layer 3", "layer 4", "layer 5", "layer 6", "layers 7-9" };
        double d[8] = {};
        for (int b = m / 2; b < m; b++)
            for (int i = 0; i < 8; i++) d[i] += (double)(h[(size_t)b * 16 + i + 1] - h[(size_t)b * 16 + i]);
        for (int i = 0; i < 8; i++) fprintf(stderr, "  CV_STAMP %-20s %8.0f cycles per match\n", name[i], d[i] / (m - m / 2));
        free(h);
    }
    (void)hipFree(stamps);
#endif
    return BUF_OK;
}

// s_eq, t_eq f32[m,32,5,20] (elevation rows 1..5 of the equivariant maps) -> ind f32[m]
extern "C" int buf_cost_volume_net(const float* s_eq, const float* t_eq, int m, const float* const* wt_host,
                                   const float* const* bias_host, float* ind_out, void* stream)
{
    BUF_REQUIRE(m >= 0, BUF_EINVAL, "buf_cost_volume_net: m=%d", m);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(s_eq && t_eq && wt_host && bias_host && ind_out, BUF_EINVAL, "buf_cost_volume_net: null argument");
    return cost_net_launch(s_eq, t_eq, m, wt_host, bias_host, nullptr, nullptr, 7, ind_out, stream, "buf_cost_volume_net");
}

// The same with the gather fused in: equi f32[rows,32,ele_n,20] holds the FULL equivariant maps of all keypoints, match i pairs
// row s_rows[i] with row t_rows[i] (int64, device) and the kernel reads elevation rows 1..5 of each (ele_n = 7) directly.
extern "C" int buf_cost_volume_net_gather(const float* equi, int ele_n, const long long* s_rows, const long long* t_rows, int m,
                                          const float* const* wt_host, const float* const* bias_host, float* ind_out, void* stream)
{
    BUF_REQUIRE(m >= 0 && ele_n == 7, BUF_EINVAL, "buf_cost_volume_net_gather: m=%d ele_n=%d (the kernel is built for ele_n = 7)", m, ele_n);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(equi && s_rows && t_rows && wt_host && bias_host && ind_out, BUF_EINVAL, "buf_cost_volume_net_gather: null argument");
    return cost_net_launch(equi, equi, m, wt_host, bias_host, s_rows, t_rows, ele_n, ind_out, stream, "buf_cost_volume_net_gather");
}
