// A13 -- CostVolume + CostNet (models/BUFFER.py:37-66, models/patchnet.py:88-147) on the f16 matrix pipe with fp32-equivalent
// arithmetic ("split-f16", opt-in beside csrc/costnet.hip's fp32 kernel; the operand split and its accuracy: csrc/convnet_h3.hip).
//
// Same mathematics as csrc/costnet.hip: the [32,20,5,20] cost tensor is never built and layer 0 is separated exactly
// (cost = S(shifted) - T is linear in both maps):
//     out0[o][n'][k'][l'] = relu(Smap[o][k'][(l' - n') mod 20] + (b[o] - Tmap[o][k'][l'])),
// Smap a 3 x 5-tap correlation of the source map over (k, e = dl - dn), Tmap a 3 x 3-tap correlation of the target map.  From
// there on every layer is a direct-form unpadded correlation over (n, l) in the split form: three v_mfma_f32_16x16x32_f16 per
// (16 outputs x 16 positions x 32 channels) into two fp32 accumulators; layer 1 collapses k' (3 -> 1): its three k' planes are
// three k-steps of a 3 x 3 correlation over (n', l').  18 -> 16 -> 14 -> 12 -> 10 -> 8 -> 6 -> 4 -> 2 -> 1.
//
// One workgroup (4 wavefronts) per match, every map a position-major split image in ONE 80 KB LDS buffer
// ([position][C channels hi | C channels lo' | pad]) rewritten in place: two workgroups per CU.  Layer 0's rows are formed by the
// vector ALU from the two small fp32 maps (kept at the end of the buffer) six n'-rows at a time into a chunk image that layer
// 1 consumes at once (four chunks; a wavefront holds its layer-1 outputs, 4 x 32 accumulators, until the last chunk has been
// read: the layer-1 map then takes the place of the chunks).
// Wavefront tiles (position tiles x 16-output tiles): see the table at k_cost_net_h3.  Weights: buf_split_tile_gemm,
// [output group][tap][k-step][16 outputs][hi | lo'][lane][8], streamed through a buffer resource one (tap, k-step) ahead.
#include "common.h"

#define CH_THREADS 256
#define CH_LDS 81920u
#define CH_NW 11                       // weight matrices: layer 0 S-term, layer 0 T-term, layers 1..9
#ifndef CH_PAD
#define CH_PAD 32u                     // position strides 160 / 288 / 400 / 544 (as H3_S: 8 mod 64 words at 128 channels)
#endif
#define CH_S32 (128u + CH_PAD)         // position stride of a 32-channel image (= ch_stride(32))

// region plan (bytes) -- phase A
#define CH_SP 0u                       // source map  [5 k][24 columns: l + 2, circular][32 ch hi | lo' | pad]: 120 positions
#define CH_TP (120u * CH_S32)          // target map  [5 k][20 l]: 100 positions
#define CH_SM 65504u                   // fp32 Smap [3 k'][20 j][36]
#define CH_TB 74144u                   // fp32 b - Tmap [3 k'][18 l'][36]   (ends at 81920)
#define CH_MAPS 36                     // floats per position of the two fp32 maps
#define CH_XC 0u                       // chunk image of layer 0 rows: [6 n'][18 l'][96 ch (k', c) hi | lo' | pad]: 108 x 400
#define CH_SXC 400u
static_assert(CH_TP + 100 * CH_S32 <= CH_SM && CH_XC + 108 * CH_SXC <= CH_SM && CH_TB + 54 * CH_MAPS * 4 == CH_LDS, "phase A regions");

struct CostH3Params {
    const void* wt[CH_NW];
    const float* bias[10];
    const long long* s_rows;
    const long long* t_rows;
    int row_floats, chan_floats, skip_floats;
    int* status;
    int* flags;                    // nullable: int32[m], flags[i] = 1 when a value of match i left the f16 range (caller zero-fills)
};

#ifdef CH_STAMP
__device__ long long* ch_stamp_ptr;       // development build (-DCH_STAMP): s_memtime of wavefront 0 at the phase boundaries
#define CH_STAMP_AT(SLOT) if (threadIdx.x == 0) ch_stamp_ptr[(size_t)blockIdx.x * 16 + (SLOT)] = __builtin_amdgcn_s_memtime();
#define CH_STAMP_ADD(SLOT, T0) if (threadIdx.x == 0) ch_stamp_ptr[(size_t)blockIdx.x * 16 + (SLOT)] += __builtin_amdgcn_s_memtime() - (T0);
#else
#define CH_STAMP_AT(SLOT)
#define CH_STAMP_ADD(SLOT, T0)
#endif

// position stride / lo' offset of an image with C channels
__host__ __device__ constexpr unsigned ch_stride(int c) { return c == 96 ? 400u : 4u * (unsigned)c + CH_PAD; }

// The product sums of a wavefront's tile: NT 16-output tiles x PT position tiles over TH x TW taps x KS k-steps of 32 channels.
// adr[t]: the lane's LDS byte address of its position of tile t at tap (0, 0), k-step 0 (hi plane; + 16 (lane >> 4) included);
// a tap (a, b) adds a * row_bytes + b * col_bytes, the lo' plane lo_ofs.  The blocks f = (tap, k-step) run as ONE software
// pipeline: LDS reads two (block, tile) steps ahead, pinned by sched_barriers, and the weights of D blocks in flight -- a block
// feeds only 3 PT NT matrix instructions (48 PT NT cycles) while a weight fragment takes ~1000 cycles from the L2, so the small
// tiles of this network need a deep ring (D = 2 for 9 x 2 tiles, 6 .. 18 for 2 x 2 .. 1 x 1; with one block ahead the 1 x 1
// layers 7 .. 9 took 64 k cycles per match for 93 matrix instructions).  The ring index is a compile-time number: the block
// loop is unrolled D-fold.
template <int PT, int NT, int KS, int TH, int TW, int D>
__device__ __forceinline__ void ch_gemm(const unsigned (&adr)[PT], unsigned row_bytes, unsigned col_bytes, unsigned lo_ofs,
                                        __amdgpu_buffer_rsrc_t rs, unsigned wofs, unsigned lane, h3f4 (&am)[NT][PT], h3f4 (&ac)[NT][PT])
{
    constexpr int NTAP = TH * TW, F = NTAP * KS;
    constexpr unsigned WBLK = NT * 2048u;                        // weight bytes of one block
    const unsigned lofs = lane * 16u;
    auto blk_ofs = [&](int f) __attribute__((always_inline)) {   // LDS offset of block f (past the end: the last block again, never used)
        const int fc = f < F ? f : F - 1;
        const int tap = fc / KS, ks = fc - tap * KS;
        return (unsigned)(tap / TW) * row_bytes + (unsigned)(tap % TW) * col_bytes + (unsigned)ks * 64u;
    };
    h3u4 W[D][NT][2];
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int q = 0; q < 2 * NT; q++) W[d][q >> 1][q & 1] = h3_ldw(rs, wofs + (unsigned)(d < F ? d : F - 1) * WBLK + q * 1024u, lofs);
    h3u4 X[3][2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const unsigned a = adr[g % PT] + blk_ofs(g / PT);
        X[g][0] = h3_lds128(a); X[g][1] = h3_lds128(a + lo_ofs);
    }
#pragma unroll 1
    for (int f0 = 0; f0 < F; f0 += D) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int f = f0 + d;
            if (f < F) {                                         // (uniform: only the last trip of the block loop can be short)
#pragma unroll
                for (int pt = 0; pt < PT; pt++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int sl = (d * PT + pt) % 3, sn = (d * PT + pt + 2) % 3;
                    {
                        const unsigned a = adr[(pt + 2) % PT] + blk_ofs(f + (pt + 2) / PT);
                        X[sn][0] = h3_lds128(a); X[sn][1] = h3_lds128(a + lo_ofs);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const h3h8 xh = __builtin_bit_cast(h3h8, X[sl][0]), xl = __builtin_bit_cast(h3h8, X[sl][1]);
#pragma unroll
                        for (int n = 0; n < NT; n++) {
                            const h3h8 wh = __builtin_bit_cast(h3h8, W[d][n][0]);
                            am[n][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, am[n][pt], 0, 0, 0);
                            ac[n][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, ac[n][pt], 0, 0, 0);
                        }
#pragma unroll
                        for (int n = 0; n < NT; n++)
                            ac[n][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h3h8, W[d][n][1]), xh, ac[n][pt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (f + D < F) {                                 // the slot's next block
#pragma unroll
                    for (int q = 0; q < 2 * NT; q++) W[d][q >> 1][q & 1] = h3_ldw(rs, wofs + (unsigned)(f + D) * WBLK + q * 1024u, lofs);
                }
            }
        }
        // D PT steps moved the LDS ring by (D PT) mod 3: bring it back so that the next trip starts at slot 0
        if constexpr ((D * PT) % 3 == 1) { h3u4 t0 = X[1][0], t1 = X[1][1]; X[1][0] = X[2][0]; X[1][1] = X[2][1]; X[2][0] = X[0][0]; X[2][1] = X[0][1]; X[0][0] = t0; X[0][1] = t1; }
        if constexpr ((D * PT) % 3 == 2) { h3u4 t0 = X[2][0], t1 = X[2][1]; X[2][0] = X[1][0]; X[2][1] = X[1][1]; X[1][0] = X[0][0]; X[1][1] = X[0][1]; X[0][0] = t0; X[0][1] = t1; }
    }
}

// v = relu(hi-sum + 2^-11 cross-sum) of a tile set (the bias started the hi-sum), split and packed: the C/D layout gives a lane 4
// consecutive output channels of one position = 8 bytes per plane.  amax: running maximum of |v| before the ReLU (range watch)
template <int PT, int NT>
__device__ __forceinline__ void ch_pack(const h3f4 (&am)[NT][PT], const h3f4 (&ac)[NT][PT], float k2048,
                                        h3u2 (&ph)[NT][PT], h3u2 (&pl)[NT][PT], float& amax)
{
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int t = 0; t < PT; t++) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = __builtin_fmaf(ac[n][t][r], 1.f / 2048.f, am[n][t][r]);
            h3_watch2(amax, v[0], v[1]); h3_watch2(amax, v[2], v[3]);
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            unsigned h0, h1, l0, l1;
            h3_split2(v[0], v[1], k2048, h0, l0);
            h3_split2(v[2], v[3], k2048, h1, l1);
            ph[n][t] = (h3u2){ h0, h1 }; pl[n][t] = (h3u2){ l0, l1 };
        }
}

// packed tiles -> the split image `out` ([position][C hi | C lo' | pad], position stride so, lo' plane at lo)
template <int PT, int NT>
__device__ __forceinline__ void ch_put(unsigned out, unsigned so, unsigned lo, const h3u2 (&ph)[NT][PT], const h3u2 (&pl)[NT][PT],
                                       int c0, int p0, int npos, unsigned lane)
{
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int t = 0; t < PT; t++) {
            const int p = p0 + 16 * t + li;
            if (p < npos) {
                const unsigned a = out + (unsigned)p * so + (unsigned)(c0 + 16 * n + 4 * lk) * 2u;
                *(__attribute__((address_space(3))) h3u2*)(size_t)a = ph[n][t];
                *(__attribute__((address_space(3))) h3u2*)(size_t)(a + lo) = pl[n][t];
            }
        }
}

// pack + put tile by tile (the packed form of a whole 9 x 2 tile set beside its 144 accumulators would spill)
template <int PT, int NT>
__device__ __forceinline__ void ch_store(unsigned out, unsigned so, unsigned lo, const h3f4 (&am)[NT][PT], const h3f4 (&ac)[NT][PT],
                                         int c0, int p0, int npos, unsigned lane, float& amax)
{
    const int li = lane & 15, lk = lane >> 4;
    float k2048 = 2048.f;
    asm volatile("" : "+v"(k2048));
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int c = c0 + 16 * n + 4 * lk;
#pragma unroll
        for (int t = 0; t < PT; t++) {
            const int p = p0 + 16 * t + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = __builtin_fmaf(ac[n][t][r], 1.f / 2048.f, am[n][t][r]);
            h3_watch2(amax, v[0], v[1]); h3_watch2(amax, v[2], v[3]);
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            unsigned h0, h1, l0, l1;
            h3_split2(v[0], v[1], k2048, h0, l0);
            h3_split2(v[2], v[3], k2048, h1, l1);
            if (p < npos) {
                const unsigned a = out + (unsigned)p * so + (unsigned)c * 2u;
                *(__attribute__((address_space(3))) h3u2*)(size_t)a = (h3u2){ h0, h1 };
                *(__attribute__((address_space(3))) h3u2*)(size_t)(a + lo) = (h3u2){ l0, l1 };
            }
            __builtin_amdgcn_sched_barrier(0);       // tile by tile: interleaved, the temporaries of several tiles beside 144 accumulators spill
        }
    }
}

// One unpadded 3 x 3 layer HIN x HIN (CIN channels) -> HOUT x HOUT (COUT), in place at the start of the buffer: the wavefront's
// tiles [pt0, pt0 + PT) x outputs [c0, c0 + 16 NT).
template <int PT, int NT, int CIN, int COUT, int HIN, int D>
__device__ __forceinline__ void ch_layer(unsigned lds0, const void* wt, const float* bias, int group, int pt0, unsigned lane, float& amax)
{
    constexpr int TAPS_H = 3, TAPS_W = 3;
    constexpr int HOUT = HIN - TAPS_H + 1, NPOS = HOUT * HOUT, KS = CIN / 32;
    constexpr unsigned SI = ch_stride(CIN), SO = ch_stride(COUT);
    unsigned adr[PT];
#pragma unroll
    for (int t = 0; t < PT; t++) {
        int p = 16 * (pt0 + t) + (int)(lane & 15);
        p = p < NPOS ? p : NPOS - 1;                             // padding rows recompute the last position (never stored)
        const int n = p / HOUT, l = p - n * HOUT;
        adr[t] = lds0 + (unsigned)(n * HIN + l) * SI + (lane >> 4) * 16u;
    }
    h3f4 am[NT][PT], ac[NT][PT];
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const h3f4 b = *reinterpret_cast<const h3f4*>(bias + group * 16 * NT + 16 * n + 4 * (lane >> 4));      // the hi-sum starts at the bias
#pragma unroll
        for (int t = 0; t < PT; t++) { am[n][t] = b; ac[n][t] = (h3f4){ 0.f, 0.f, 0.f, 0.f }; }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, 0x7fffffff, 0x00027000);
    ch_gemm<PT, NT, KS, TAPS_H, TAPS_W, D>(adr, HIN * SI, SI, 2u * CIN, rs, (unsigned)group * (TAPS_H * TAPS_W * KS * NT * 2048u), lane, am, ac);
    __syncthreads();                                             // every wavefront has read the input map
    unsigned lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    ch_store<PT, NT>(lds0, SO, 2u * COUT, am, ac, group * 16 * NT, 16 * pt0, NPOS, lane_s, amax);
}

// Wavefront tiles (w = wavefront, 16-position tiles x 16-output tiles):
//   layer 0 S / T   4 tiles x 2      w: tile w, both output tiles                       (15 / 9 taps, 1 k-step)
//   layer 1         per chunk 4 x 4  w: tile pair w >> 1, output pair w & 1           (9 taps x 3 k-steps; 4 chunks)
//   layer 2         13 x 4           w: tiles 7 | 6 (w >> 1), output pair w & 1
//   layer 3          9 x 8           w: all 9 tiles, output pair w
//   layer 4          7 x 8           w: all 7 tiles, output pair w
//   layer 5          4 x 4           w: tile pair w >> 1, output pair w & 1
//   layer 6          3 x 4           w: tiles 2 | 1 (w >> 1), output pair w & 1
//   layers 7, 8, 9   1 x 2           w < 2: output tile w
__global__ void __launch_bounds__(CH_THREADS, 2) k_cost_net_h3(const float* __restrict__ s_eq, const float* __restrict__ t_eq, CostH3Params P,
                                                           float* __restrict__ ind_out)
{
    extern __shared__ __attribute__((aligned(16))) char ch_smem[];
    const unsigned lds0 = (unsigned)(size_t)(h3_lds_p)ch_smem;
    const int match = blockIdx.x, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const unsigned lane = tid & (WAVE - 1);
    const int li = lane & 15, lk = lane >> 4;
    unsigned amax_in = 0u;                       // inputs: integer watch (catches NaN); inside the stack: float maximum
    float amax = 0.f;
    float k2048 = 2048.f;
    asm volatile("" : "+v"(k2048));
    CH_STAMP_AT(0)
    {   // both maps -> split images, transposed on the way in: global [c][k][l] -> LDS [k][l][c]; S with two wrap-around columns per side
        const float* a = s_eq + (P.s_rows ? (size_t)P.s_rows[match] * P.row_floats + P.skip_floats : (size_t)match * 3200);
        const float* b = t_eq + (P.t_rows ? (size_t)P.t_rows[match] * P.row_floats + P.skip_floats : (size_t)match * 3200);
        for (int i = tid; i < 400; i += CH_THREADS) {            // item = (channel pair, 4 positions)
            const int cp = i % 16, q = i / 16, k = q / 5, l0 = (q - k * 5) * 4;
            h3f4 sv[2], tv[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                sv[u] = __builtin_nontemporal_load(reinterpret_cast<const h3f4*>(a + (2 * cp + u) * P.chan_floats + q * 4));
                tv[u] = __builtin_nontemporal_load(reinterpret_cast<const h3f4*>(b + (2 * cp + u) * P.chan_floats + q * 4));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int l = l0 + j;
                h3h2 sh, sl, th, tl;
                _Float16 h, lo;
                h3_split(sv[0][j], h, lo); sh[0] = h; sl[0] = lo;
                h3_split(sv[1][j], h, lo); sh[1] = h; sl[1] = lo;
                h3_split(tv[0][j], h, lo); th[0] = h; tl[0] = lo;
                h3_split(tv[1][j], h, lo); th[1] = h; tl[1] = lo;
                h3_watch(amax_in, sv[0][j]); h3_watch(amax_in, sv[1][j]); h3_watch(amax_in, tv[0][j]); h3_watch(amax_in, tv[1][j]);
                auto put = [&](unsigned pos_adr, h3h2 vh, h3h2 vl) __attribute__((always_inline)) {
                    *(__attribute__((address_space(3))) unsigned*)(size_t)(pos_adr + 4u * cp) = __builtin_bit_cast(unsigned, vh);
                    *(__attribute__((address_space(3))) unsigned*)(size_t)(pos_adr + 64u + 4u * cp) = __builtin_bit_cast(unsigned, vl);
                };
                put(lds0 + CH_SP + (unsigned)(k * 24 + l + 2) * CH_S32, sh, sl);
                if (l >= 18) put(lds0 + CH_SP + (unsigned)(k * 24 + l - 18) * CH_S32, sh, sl);
                if (l < 2) put(lds0 + CH_SP + (unsigned)(k * 24 + l + 22) * CH_S32, sh, sl);
                put(lds0 + CH_TP + (unsigned)(k * 20 + l) * CH_S32, th, tl);
            }
        }
    }
    __syncthreads();
    CH_STAMP_AT(1)

    // ---- layer 0, separated: the two small correlations -> fp32 maps Smap[k'][j][o], b - Tmap[k'][l'][o] ----
    {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)P.wt[0], 0, 0x7fffffff, 0x00027000);
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)P.wt[1], 0, 0x7fffffff, 0x00027000);
        int ms = w * 16 + li, mt = ms;
        ms = ms < 60 ? ms : 59;
        mt = mt < 54 ? mt : 53;
        const unsigned aS[1] = { lds0 + CH_SP + (unsigned)((ms / 20) * 24 + ms % 20) * CH_S32 + lk * 16u };
        const unsigned aT[1] = { lds0 + CH_TP + (unsigned)((mt / 18) * 20 + mt % 18) * CH_S32 + lk * 16u };
        h3f4 sm[2][1], sc[2][1], tm[2][1], tc[2][1];
#pragma unroll
        for (int n = 0; n < 2; n++) { sm[n][0] = sc[n][0] = tm[n][0] = tc[n][0] = (h3f4){ 0.f, 0.f, 0.f, 0.f }; }
        ch_gemm<1, 2, 1, 3, 5, 8>(aS, 24 * CH_S32, CH_S32, 64u, rsS, 0u, lane, sm, sc);
        ch_gemm<1, 2, 1, 3, 3, 9>(aT, 20 * CH_S32, CH_S32, 64u, rsT, 0u, lane, tm, tc);
        // C/D layout: lane holds outputs 16 n + 4 lk + r at position 16 w + li
        const int m = w * 16 + li;
        float* SM = reinterpret_cast<float*>(ch_smem + CH_SM);
        float* TB = reinterpret_cast<float*>(ch_smem + CH_TB);
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const h3f4 b0 = *reinterpret_cast<const h3f4*>(P.bias[0] + 16 * n + 4 * lk);
            h3f4 vs, vt;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vs[r] = sm[n][0][r] + sc[n][0][r] * (1.f / 2048.f);
                vt[r] = b0[r] - (tm[n][0][r] + tc[n][0][r] * (1.f / 2048.f));
            }
            if (m < 60) *reinterpret_cast<h3f4*>(SM + m * CH_MAPS + 16 * n + 4 * lk) = vs;
            if (m < 54) *reinterpret_cast<h3f4*>(TB + m * CH_MAPS + 16 * n + 4 * lk) = vt;
        }
    }
    __syncthreads();                                             // the two maps are complete; the input images are dead
    CH_STAMP_AT(2)

    // ---- layer 1 over four chunks of six layer-0 rows (rows 4j .. 4j + 5 -> output rows 4j .. 4j + 3) ----
    {
        const int pair = w & 1, half = w >> 1;                   // outputs 32 pair .. + 31; output rows 4j + 2 half, + 1
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)P.wt[2], 0, 0x7fffffff, 0x00027000);
        h3u2 ph[4][2][2], pl[4][2][2];                           // [chunk][output tile][row]: the finished rows, split and packed (64 registers)
        const float* SM = reinterpret_cast<const float*>(ch_smem + CH_SM);
        const float* TB = reinterpret_cast<const float*>(ch_smem + CH_TB);
#ifdef CH_STAMP
        if (threadIdx.x == 0) { ch_stamp_ptr[(size_t)blockIdx.x * 16 + 12] = 0; ch_stamp_ptr[(size_t)blockIdx.x * 16 + 13] = 0; }
#endif
#pragma unroll
        for (int j = 0; j < 4; j++) {
#ifdef CH_STAMP
            const long long tf0 = __builtin_amdgcn_s_memtime();
#endif
            // form: item = (row r of the chunk, l', k', 8 channels): relu(Smap[k'][(l' - n') mod 20] + (b - Tmap)[k'][l']), split
            for (int i = tid; i < 6 * 18 * 12; i += CH_THREADS) {
                const int c8 = i & 3, rest = i >> 2, kq = rest % 3, pos = rest / 3, r = pos / 18, l = pos - r * 18;
                const int n = 4 * j + r;
                int jj = l - n; jj += jj < 0 ? 20 : 0;
                const h3f4* s4 = reinterpret_cast<const h3f4*>(SM + (kq * 20 + jj) * CH_MAPS + 8 * c8);
                const h3f4* t4 = reinterpret_cast<const h3f4*>(TB + (kq * 18 + l) * CH_MAPS + 8 * c8);
                const h3f4 s0 = s4[0], s1 = s4[1], t0 = t4[0], t1 = t4[1];
                h3u4 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    const float v0 = fmaxf((q < 4 ? s0[q & 3] : s1[q & 3]) + (q < 4 ? t0[q & 3] : t1[q & 3]), 0.f);
                    const float v1 = fmaxf((q < 4 ? s0[(q + 1) & 3] : s1[(q + 1) & 3]) + (q < 4 ? t0[(q + 1) & 3] : t1[(q + 1) & 3]), 0.f);
                    h3_watch2(amax, v0, v1);
                    unsigned hh, ll;
                    h3_split2(v0, v1, k2048, hh, ll);
                    hi[q >> 1] = hh; lo[q >> 1] = ll;
                }
                const unsigned ad = lds0 + CH_XC + (unsigned)pos * CH_SXC + (unsigned)(kq * 64 + c8 * 16);
                *(__attribute__((address_space(3))) h3u4*)(size_t)ad = hi;
                *(__attribute__((address_space(3))) h3u4*)(size_t)(ad + 192u) = lo;
            }
            __syncthreads();
            CH_STAMP_ADD(12, tf0)
#ifdef CH_STAMP
            const long long tg0 = __builtin_amdgcn_s_memtime();
#endif
            unsigned adr[2];
#pragma unroll
            for (int t = 0; t < 2; t++) adr[t] = lds0 + CH_XC + (unsigned)((2 * half + t) * 18 + li) * CH_SXC + lk * 16u;
            h3f4 am[2][2], ac[2][2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const h3f4 b1 = *reinterpret_cast<const h3f4*>(P.bias[1] + 32 * pair + 16 * n + 4 * lk);
#pragma unroll
                for (int t = 0; t < 2; t++) { am[n][t] = b1; ac[n][t] = (h3f4){ 0.f, 0.f, 0.f, 0.f }; }
            }
            ch_gemm<2, 2, 3, 3, 3, 6>(adr, 18 * CH_SXC, CH_SXC, 192u, rs1, (unsigned)pair * (9 * 3 * 2 * 2048u), lane, am, ac);
            ch_pack<2, 2>(am, ac, k2048, ph[j], pl[j], amax);
            __syncthreads();                                     // the chunk has been read: the next one (or the layer-1 map) may be written
            CH_STAMP_ADD(13, tg0)
        }
        unsigned lane_s = lane;
        asm volatile("" : "+v"(lane_s));
#pragma unroll
        for (int j = 0; j < 4; j++)
            ch_put<2, 2>(lds0, ch_stride(64), 128u, ph[j], pl[j], 32 * pair, 16 * (4 * j + 2 * half), 256, lane_s);
    }
    __syncthreads();
    CH_STAMP_AT(3)
    // ---- layers 2 .. 9 ----
    if (w < 2) ch_layer<7, 2, 64, 64, 16, 2>(lds0, P.wt[3], P.bias[2], w & 1, 0, lane, amax);             // -> 14 x 14
    else ch_layer<6, 2, 64, 64, 16, 2>(lds0, P.wt[3], P.bias[2], w & 1, 7, lane, amax);
    __syncthreads();
    CH_STAMP_AT(4)
    ch_layer<9, 2, 64, 128, 14, 2>(lds0, P.wt[4], P.bias[3], w, 0, lane, amax);                            // -> 12 x 12, 128 ch
    __syncthreads();
    CH_STAMP_AT(5)
    ch_layer<7, 2, 128, 128, 12, 2>(lds0, P.wt[5], P.bias[4], w, 0, lane, amax);                           // -> 10 x 10
    __syncthreads();
    CH_STAMP_AT(6)
    ch_layer<2, 2, 128, 64, 10, 6>(lds0, P.wt[6], P.bias[5], w & 1, 2 * (w >> 1), lane, amax);             // -> 8 x 8, 64 ch
    __syncthreads();
    CH_STAMP_AT(7)
    if (w < 2) ch_layer<2, 2, 64, 64, 8, 6>(lds0, P.wt[7], P.bias[6], w & 1, 0, lane, amax);              // -> 6 x 6
    else ch_layer<1, 2, 64, 64, 8, 9>(lds0, P.wt[7], P.bias[6], w & 1, 2, lane, amax);
    __syncthreads();
    CH_STAMP_AT(8)
    if (w < 2) ch_layer<1, 1, 64, 32, 6, 18>(lds0, P.wt[8], P.bias[7], w, 0, lane, amax);                  // -> 4 x 4, 32 ch
    else __syncthreads();
    __syncthreads();
    if (w < 2) ch_layer<1, 1, 32, 32, 4, 9>(lds0, P.wt[9], P.bias[8], w, 0, lane, amax);                  // -> 2 x 2
    else __syncthreads();
    __syncthreads();
    // layer 9: (2, 1, 2) taps, 32 -> 20 (+ 12 zero) logits at the one position, no ReLU
    if (w < 2) {
        unsigned adr[1] = { lds0 + lk * 16u };                   // every lane computes position 0 (the 2 x 2 map starts the buffer)
        h3f4 am[1][1] = { { (h3f4){ 0.f, 0.f, 0.f, 0.f } } }, ac[1][1] = { { (h3f4){ 0.f, 0.f, 0.f, 0.f } } };
        const __amdgpu_buffer_rsrc_t rs9 = __builtin_amdgcn_make_buffer_rsrc((void*)P.wt[10], 0, 0x7fffffff, 0x00027000);
        ch_gemm<1, 1, 1, 2, 2, 4>(adr, 2 * CH_S32, CH_S32, 64u, rs9, (unsigned)w * (4 * 1 * 1 * 2048u), lane, am, ac);
        __syncthreads();
        if (li == 0) {
            const h3f4 b = *reinterpret_cast<const h3f4*>(P.bias[9] + 16 * w + 4 * lk);
            float* logits = reinterpret_cast<float*>(ch_smem + 4096);
#pragma unroll
            for (int r = 0; r < 4; r++) logits[16 * w + 4 * lk + r] = (am[0][0][r] + ac[0][0][r] * (1.f / 2048.f)) + b[r];
        }
    } else __syncthreads();
    __syncthreads();
    if (w == 0) {                            // softmax over the 20 logits, expected index (BUFFER.py:63-65)
        const float* logits = reinterpret_cast<const float*>(ch_smem + 4096);
        float v = lane < 20 ? logits[lane] : -3.4e38f;
        float mx = v;
        for (int d = WAVE / 2; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, WAVE));
        float e = lane < 20 ? expf(v - mx) : 0.f;
        float se = e, sw = e * (float)lane;
        for (int d = WAVE / 2; d > 0; d >>= 1) { se += __shfl_xor(se, d, WAVE); sw += __shfl_xor(sw, d, WAVE); }
        if (lane == 0) ind_out[match] = sw / se;
    }
    CH_STAMP_AT(9)
    if ((P.status || P.flags) && __builtin_amdgcn_ballot_w64(amax_in >= H3_F16_LIMIT_BITS || !(amax < 65504.f)) != 0 && lane == 0) {
        if (P.status) atomicOr(P.status, 1);
        if (P.flags) P.flags[match] = 1;                      // this match goes through the fp32 kernel again (buf_cost_volume_net_split_safe)
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// Host helper (general form of buf_split_tile_filters): w [Cout][Cin][ntaps] fp32 -> the two f16 planes for ch_gemm / h3_gemm,
//     out[((((g ntaps + tap) KS + ks) nt + n2) 2 + plane) 512 + (kg 16 + row) 8 + i] = plane(w[16 nt g + 16 n2 + row][32 ks + 8 kg + i][tap]),
// groups of nt 16-output tiles (Cout is padded with zero rows to a multiple of 16 nt, Cin to a multiple of 32).
extern "C" long long buf_split_gemm_count(int cout, int cin, int ntaps, int nt)
{
    const long long groups = (cout + 16 * nt - 1) / (16 * nt);
    return groups * ntaps * ((cin + 31) / 32) * nt * 2 * 512;
}

extern "C" int buf_split_tile_gemm(const float* w_host, int cout, int cin, int ntaps, int nt, unsigned short* out_host)
{
    BUF_REQUIRE(w_host && out_host, BUF_EINVAL, "buf_split_tile_gemm: null argument");
    BUF_REQUIRE(cout > 0 && cin > 0 && ntaps > 0 && (nt == 1 || nt == 2), BUF_EINVAL, "buf_split_tile_gemm: %d -> %d, %d taps, nt %d", cin, cout, ntaps, nt);
    const int KS = (cin + 31) / 32;
    memset(out_host, 0, (size_t)buf_split_gemm_count(cout, cin, ntaps, nt) * sizeof(unsigned short));
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++)
            for (int tap = 0; tap < ntaps; tap++) {
                const float v = w_host[((size_t)o * cin + c) * ntaps + tap];
                BUF_REQUIRE(fabsf(v) < 65504.f, BUF_EINVAL, "buf_split_tile_gemm: weight %g outside the f16 range", (double)v);
                const unsigned short hi = h3_f16_bits(v);
                const unsigned short lo = h3_f16_bits((v - h3_f16_value(hi)) * 2048.f);
                const int g = o / (16 * nt), n2 = (o % (16 * nt)) / 16, row = o % 16, ks = c / 32, kg = (c % 32) / 8, i = c % 8;
                const size_t base = ((((size_t)g * ntaps + tap) * KS + ks) * nt + n2) * 2;
                out_host[(base + 0) * 512 + (kg * 16 + row) * 8 + i] = hi;
                out_host[(base + 1) * 512 + (kg * 16 + row) * 8 + i] = lo;
            }
    return BUF_OK;
}

static int cost_net_h3_launch(const float* s_eq, const float* t_eq, int m, const void* const* wt_host, const float* const* bias_host,
                              const long long* s_rows, const long long* t_rows, int ele_n, float* ind_out, int* status_dev, void* stream,
                              const char* who, int* flags_dev = nullptr)
{
    CostH3Params P;
    for (int l = 0; l < CH_NW; l++) {
        P.wt[l] = wt_host[l];
        BUF_REQUIRE(P.wt[l], BUF_EINVAL, "%s: null weights (matrix %d)", who, l);
    }
    for (int l = 0; l < 10; l++) {
        P.bias[l] = bias_host[l];
        BUF_REQUIRE(P.bias[l], BUF_EINVAL, "%s: null bias for layer %d", who, l);
    }
    P.s_rows = s_rows; P.t_rows = t_rows;
    P.chan_floats = s_rows ? ele_n * 20 : 100;
    P.row_floats = 32 * P.chan_floats;
    P.skip_floats = s_rows ? 20 : 0;
    P.status = status_dev;
    P.flags = flags_dev;
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds((const void*)k_cost_net_h3, CH_LDS, grant)) return rc;
    // work = the dense algorithmic count of SURVEY 8d per match (as id 2 counts the fp32 kernel's executed flops, bench.py
    // converts): 2 x 79 997 440 MAC
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 159994880.0 * m, BUF_TIMED_COST_NET_SPLIT);
#ifdef CH_STAMP
    long long* stamps = nullptr;
    BUF_CHECK_HIP(hipMalloc(&stamps, (size_t)m * 16 * sizeof(long long)));
    BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ch_stamp_ptr), &stamps, sizeof(stamps)));
#endif
    k_cost_net_h3<<<m, CH_THREADS, CH_LDS, (hipStream_t)stream>>>(s_eq, t_eq, P, ind_out);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
#ifdef CH_STAMP
    if (m >= 2048) {
        BUF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        long long* h = (long long*)malloc((size_t)m * 16 * sizeof(long long));
        BUF_CHECK_HIP(hipMemcpy(h, stamps, (size_t)m * 16 * sizeof(long long), hipMemcpyDeviceToHost));
        static const char* name[9] = { "input", "layer 0 maps", "layer 1", "layer 2", "layer 3", "layer 4", "layer 5", "layer 6", "layers 7-9" };
        double d[9] = {}, form = 0, gemm = 0;
        for (int b = m / 2; b < m; b++) {
            for (int i = 0; i < 9; i++) d[i] += (double)(h[(size_t)b * 16 + i + 1] - h[(size_t)b * 16 + i]);
            form += (double)h[(size_t)b * 16 + 12]; gemm += (double)h[(size_t)b * 16 + 13];
        }
        for (int i = 0; i < 9; i++) fprintf(stderr, "  CH_STAMP %-14s %8.0f cycles per match\n", name[i], d[i] / (m - m / 2));
        fprintf(stderr, "  CH_STAMP layer 1: form + barrier %8.0f, gemm + barrier %8.0f\n", form / (m - m / 2), gemm / (m - m / 2));
        free(h);
    }
    (void)hipFree(stamps);
#endif
    return BUF_OK;
}

extern "C" int buf_cost_volume_net_split(const float* s_eq, const float* t_eq, int m, const void* const* wt_host,
                                         const float* const* bias_host, float* ind_out, int* status_dev, void* stream)
{
    BUF_REQUIRE(m >= 0, BUF_EINVAL, "buf_cost_volume_net_split: m=%d", m);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(s_eq && t_eq && wt_host && bias_host && ind_out, BUF_EINVAL, "buf_cost_volume_net_split: null argument");
    return cost_net_h3_launch(s_eq, t_eq, m, wt_host, bias_host, nullptr, nullptr, 7, ind_out, status_dev, stream, "buf_cost_volume_net_split");
}

extern "C" int buf_cost_volume_net_split_gather(const float* equi, int ele_n, const long long* s_rows, const long long* t_rows, int m,
                                                const void* const* wt_host, const float* const* bias_host, float* ind_out,
                                                int* status_dev, void* stream)
{
    BUF_REQUIRE(m >= 0 && ele_n == 7, BUF_EINVAL, "buf_cost_volume_net_split_gather: m=%d ele_n=%d (the kernel is built for ele_n = 7)", m, ele_n);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(equi && s_rows && t_rows && wt_host && bias_host && ind_out, BUF_EINVAL, "buf_cost_volume_net_split_gather: null argument");
    return cost_net_h3_launch(equi, equi, m, wt_host, bias_host, s_rows, t_rows, ele_n, ind_out, status_dev, stream, "buf_cost_volume_net_split_gather");
}
