// A11 dense part -- Cylindrical_Net (models/patchnet.py:15-85) on the f16 matrix pipe with fp32-equivalent arithmetic
// ("split-f16", opt-in beside csrc/convnet_wg.hip's all-fp32 Winograd kernel).
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the chip's f16 / bf16 matrix rate and shares its issue port with the vector ALU.
// Here every fp32 operand is split ONCE into two f16 numbers
//     x = hi + 2^-11 lo',    hi = f16(x) (round to nearest even),    lo' = f16((x - hi) 2^11)
// (|x - hi - 2^-11 lo'| <= 2^-23 |x|, one fp32 ulp, in the worst case and a quarter of that on average: the residual of an RNE
// rounding is signed, so two 11-bit significands carry 23 bits;
// scaling the low part keeps it a NORMAL f16 number wherever x is above 2^-25 -- unscaled it would fall into the f16
// subnormals for every |x| < 2^-3 and lose its bits), and a product sum becomes
//     sum x w = [sum hi_x hi_w] + 2^-11 [sum hi_x lo'_w + sum lo'_x hi_w]        (dropped: 2^-22 sum lo'_x lo'_w)
// -- THREE v_mfma_f32_16x16x32_f16 per (16 outputs x 16 positions x 32 input channels) into TWO fp32 accumulators; the products
// of two f16 numbers are exact in fp32 and the matrix unit adds 32 of them per rounding.  Measured (tools/micro/f16_split.hip,
// profiles/r04_f16_split.txt; K = 1152): 2.8e-7 of the output scale against the float64 dot product, fp32 MFMA / fmaf
// chain 6e-7 .. 9e-7, the bf16 x 6 split 6.5e-7 .. 8.4e-7 at twice the matrix instructions; the matrix unit keeps f16 subnormal
// inputs.  Weights are split on the host (buf_split_tile_filters), activations in each layer's epilogue.
//
// Direct form (9 taps), no arithmetic in the loops: per (tap, 32 channels) a wavefront reads the hi and lo' rows of a 16-position
// tile (two ds_read_b128) and issues 3 MFMAs per 16 output channels.  3 x the dense MAC count at 16 x the fp32 rate:
// 0.37 of the matrix cycles of the Winograd fp32 kernel (which executes 0.508 of the dense count).
//
// One workgroup (4 wavefronts) owns one patch for the whole stack; activations live in ONE LDS image rewritten in place,
//     [140 positions][544 B] = 128 channels hi (256 B) | 128 channels lo' (256 B) | 32 B pad,
// position-major so that a lane's 8 channels of a k-step are 16 contiguous bytes and a tile's 16 lanes sit 136 words apart
// (= 8 mod 64 banks: the four 16-lane groups of a ds_read_b128 cover 64 distinct banks).  Padding is in the ADDRESSES: a
// table tab[tap][lane][tile] in LDS (u16, units of 16 B) holds the row a tap reads for every output position -- the circular
// azimuth wrap folded in, one shared zero row for the elevation padding (utils/common.py:265-310) -- so there are no halo
// copies and the image is 76.7 KB: two workgroups per CU.
// Wavefront tiles: 128 outputs: (32 outputs, all 9 position tiles): 144 accumulators; 64 outputs: (32 outputs, 4 | 5
// position tiles); 32 outputs: (32 outputs, 2 | 2 | 2 | 3 position tiles).  The imbalance inside a workgroup is taken up by
// the other workgroup of the CU, which shares the SIMDs.
// Weights stream through a buffer resource with wavefront-uniform offsets, tiled [32-output group][tap][k-step][16 outputs]
// [hi | lo'][lane][8], 4 KB per (tap, k-step) and wavefront, one step ahead in registers.
#include "common.h"

// Activation-image layouts (round 5; profiles/r05_h3_lds_layouts.txt).  0 (default, round 4): a tile = 16 CONSECUTIVE positions, rows of
// 544 B.  One azimuth-wrapped lane per tile then costs an extra LDS cycle in each of the four 16-lane groups of every ds_read_b128 of
// the dx = +-1 taps (+52 % read cycles; SQ_LDS_BANK_CONFLICT 45 % of SQ_LDS_IDX_ACTIVE).  1: tiles follow the MAP'S ROWS -- tiles
// 0..6 = (y, x = 0..15), tile 7 = (y = 0..3) x (x = 16..19), tile 8 = (y = 4..6) x (x = 16..19) + 4 idle slots --, rows of 512 B, the
// 16-byte chunk XOR-keyed: chunk' = chunk ^ key(y, x), key = (x == 19 ? 14 : 2 (x & 7)) ^ 8 (y & 1): every group of every tap reads
// 16 distinct slots.  2: tiles = (y, x = 1..16) + narrow tiles (x = 17, 18, 19, 0), 544-byte rows, no key, round-4 addressing.
// 1 and 2 remove the read conflicts (conflict cycles -56 %) with bit-identical outputs and do NOT make the kernel faster (LDS 43 %
// busy, matrix pipe 73-76 %): kept as switches, 0 stays the default.
#ifndef H3_ROWTILES
#define H3_ROWTILES 0
#endif
#ifndef H3_S
#if H3_ROWTILES == 1
#define H3_S 512u
#else
#define H3_S 544u                       // bytes per position (136 words = 8 mod 64 banks; 528 measured the same, 560 is 20 % slower)
#endif
#endif
#define H3_LO 256u                      // offset of the lo' plane inside a position's row
#define H3_NPOS 140
#define H3_ZERO (H3_NPOS * H3_S)        // the zero row
#define H3_TAB ((H3_NPOS + 1) * H3_S)   // u16 tab[9 taps][16 lanes][16 tile slots]: row address / 16 of the position a tap reads
#define H3_LDS (H3_TAB + 9 * 16 * 16 * 2)
#define H3_THREADS 256
#define H3_LAYERS 8
#define H3_MAXC 128

typedef _Float16 h3h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h3h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h3h2 __attribute__((ext_vector_type(2)));
typedef float h3f4 __attribute__((ext_vector_type(4)));
typedef unsigned h3u4 __attribute__((ext_vector_type(4)));
typedef unsigned h3u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char* h3_lds_p;

// tile slot (tile t, lane slot m) -> map position (y, x); false: an idle slot of tile 8 (y = 7 is returned: a virtual row)
__device__ __forceinline__ bool h3_pos(int t, int m, int& y, int& x)
{
#if H3_ROWTILES == 1
    if (t < 7) { y = t; x = m; return true; }
    y = (t - 7) * 4 + (m >> 2); x = 16 + (m & 3);
    return y <= 6;
#elif H3_ROWTILES == 2
    // rows shifted by one column: tiles 0..6 = (y, x = 1..16), tiles 7 / 8 = (y = 0..3 | 4..6) x (x = 17, 18, 19, 0).  No tap of a
    // row tile wraps in azimuth (x - 1 .. x + 1 stays inside 0..17), the narrow tiles' four columns are cyclically consecutive, and
    // with the 544-byte row stride 16 consecutive rows already sit on 16 distinct slots: no key, the round-4 addressing unchanged
    if (t < 7) { y = t; x = m + 1; return true; }
    y = (t - 7) * 4 + (m >> 2); x = 17 + (m & 3); x -= x >= 20 ? 20 : 0;
    return y <= 6;
#else
    const int p = 16 * t + m;
    y = p / 20; x = p - y * 20;
    return p < H3_NPOS;
#endif
}
// XOR key of the 16-byte chunk index inside a position's 256-byte plane
__device__ __forceinline__ unsigned h3_key(int y, int x)
{
#if H3_ROWTILES == 1
    return (unsigned)((x == 19 ? 14 : 2 * (x & 7)) ^ (8 * (y & 1)));
#else
    return 0u;
#endif
}

struct CylH3Params {
    const void* wt[H3_LAYERS];          // buf_split_tile_filters
    const float* bias[H3_LAYERS];
    int cin[H3_LAYERS], cout[H3_LAYERS], relu[H3_LAYERS];
    int* status;                        // nullable: bit 0 set when an activation left the f16 range (|v| >= 65504)
    int* flags;                         // nullable: int32[np], flags[p] = 1 when patch p did (caller zero-fills; buf_cylindrical_net_split_safe)
    const float* head;                  // nullable: parameters of the attention-pooling head (csrc/convnet.hip dh_body) -> fused behind the last layer
    float* desc;                        // with head: desc f32[np,32], equi f32[np,32,140] instead of y
    float* equi;
#ifdef H3_STAMP
    long long* stamps;
#endif
};

#ifdef H3_STAMP
__device__ long long* h3_stamp_ptr;      // development build (-DH3_STAMP): s_memtime per wavefront at the layer boundaries
#define H3_STAMP_AT(SLOT) if ((threadIdx.x & 63) == 0) h3_stamp_ptr[((size_t)blockIdx.x * 4 + threadIdx.x / 64) * 32 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define H3_STAMP_AT(SLOT)
#endif

__device__ __forceinline__ h3u4 h3_lds128(unsigned a) { return *(const __attribute__((address_space(3))) h3u4*)(size_t)a; }
__device__ __forceinline__ unsigned h3_lds32(unsigned a) { return *(const __attribute__((address_space(3))) unsigned*)(size_t)a; }

__device__ __forceinline__ h3u4 h3_ldw(__amdgpu_buffer_rsrc_t rs, unsigned uniform_byte_ofs, unsigned lane_byte_ofs)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, lane_byte_ofs, uniform_byte_ofs, 0);
}

// Range watch: running maximum of |v| as an unsigned bit pattern (a NaN compares above every finite value): one AND + one MAX per
// value, no comparison kept alive (a bool OR-ed per value made the compiler park every |v| of a tile set in scratch).
#define H3_F16_LIMIT_BITS 0x477fe000u           // 65504.f
__device__ __forceinline__ void h3_watch(unsigned& amax, float v) { amax = max(amax, __float_as_uint(v) & 0x7fffffffu); }

// x = hi + 2^-11 lo'
__device__ __forceinline__ void h3_split(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.f);
}

// The same for two values at once, packed: 5 instructions instead of 8 (the epilogues are vector-ALU work beside the partner
// workgroup's matrix instructions: 14 % of a workgroup's time).  v_cvt_pk_f16_f32 rounds both to hi; v_fma_mix_f32 forms the exact
// residual v - hi straight from the packed half (no conversion back); v_fma_mixlo / mixhi_f16 scale it by 2^11 and round it into
// the two halves of lo'.  Bit-identical to h3_split (every step is the same exactly rounded operation).
__device__ __forceinline__ void h3_split2(float v0, float v1, float k2048, unsigned& hi, unsigned& lo)
{
    h3h2 h;
    h[0] = (_Float16)v0; h[1] = (_Float16)v1;
    hi = __builtin_bit_cast(unsigned, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(lo) : "v"(r0), "v"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(r1), "v"(k2048));
}
// range watch of the epilogues: running maximum of |v| as a float (one v_max3_f32 with |.| modifiers per value pair).  A NaN
// would slip through a float maximum, but inside the stack a NaN can only follow an infinity of the layer before, which this
// catches; the kernels' INPUTS go through the integer form h3_watch.
__device__ __forceinline__ void h3_watch2(float& amax, float v0, float v1) { amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1))); }

// The product sums of the wavefront's tile: outputs [32 of group ct], positions of the tiles pt0 .. pt0 + PT - 1 (pt0 even), K = 9 taps x
// KS k-steps of 32 channels.  am: sum hi hi, ac: sum hi lo' + lo' hi.  Steps (k-step, tile) of a tap run as a software pipeline
// pinned by sched_barriers: the LDS reads of step s + 2 (across the tap boundary: through the next tap's row addresses), then
// the 6 MFMAs of step s; the weights of the next (tap, k-step) are requested when the current one starts.  Row addresses stay
// PACKED (two u16 per register, as the table holds them) and are expanded at the read: 2 vector instructions per step beside
// 6 MFMAs -- nine expanded addresses for this tap and nine for the next one are 27 registers the 144-accumulator tiles do not have.
template <int PT, int KS>
__device__ __forceinline__ void h3_gemm(unsigned lds0, __amdgpu_buffer_rsrc_t rs, unsigned wofs, int pt0, unsigned lane,
                                        h3f4 (&am)[2][PT], h3f4 (&ac)[2][PT])
{
    constexpr int S = KS * PT;                                   // steps per tap
    constexpr int NDW = (PT + 1) / 2;
    const unsigned lofs = lane * 16u;
    const unsigned tadr = lds0 + H3_TAB + (lane & 15) * 32u + (unsigned)pt0 * 2u;
    unsigned rc[NDW], rn[NDW];
#if H3_ROWTILES == 1
    // table entry = row << 5 | key; the lane's k-group (lane >> 4) is XORed into both halves of a packed pair once per tap, the
    // k-step into the chunk bits at the read: address = lds0 + ((entry ^ k-group ^ 4 k-step) << 4)
    const unsigned kx = (lane >> 4) * 0x10001u;
#pragma unroll
    for (int j = 0; j < NDW; j++) rc[j] = h3_lds32(tadr + 4 * j) ^ kx;
    auto row = [&](const unsigned (&r)[NDW], int t, int ks) __attribute__((always_inline)) {
        return ((((t & 1) ? (r[t >> 1] >> 16) : (r[t >> 1] & 0xffffu)) ^ (unsigned)(ks << 2)) << 4) + lds0;
    };
#else
    const unsigned kgo = lds0 + (lane >> 4) * 16u;
#pragma unroll
    for (int j = 0; j < NDW; j++) rc[j] = h3_lds32(tadr + 4 * j);
    auto row = [&](const unsigned (&r)[NDW], int t, int ks) __attribute__((always_inline)) {
        return (((t & 1) ? (r[t >> 1] >> 16) : (r[t >> 1] & 0xffffu)) << 4) + kgo + (unsigned)ks * 64u;
    };
#endif
    // weights: the blocks (tap, k-step) of 4 KB, WD blocks ahead in registers.  A block feeds 6 PT matrix instructions (96 PT cycles).
    // Two blocks ahead for the 2..5-tile wavefronts measured +-0 (85.0 vs 86.0 ms per 320 000 patches): the partner workgroup of the
    // CU covers the fetch; the kernel sits at 73 % matrix-pipe busy at the ~1.8 GHz the chip holds under f16 MFMA load.
    constexpr int WD = 1;
    h3u4 Wc[2][2], Wn[2][2], Wnn[2][2];
#pragma unroll
    for (int q = 0; q < 4; q++) Wc[q >> 1][q & 1] = h3_ldw(rs, wofs + q * 1024u, lofs);
    if constexpr (WD == 2) {
#pragma unroll
        for (int q = 0; q < 4; q++) Wn[q >> 1][q & 1] = h3_ldw(rs, wofs + (9 * KS > 1 ? 4096u : 0u) + q * 1024u, lofs);
    }
    h3u4 X[3][2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const unsigned a = row(rc, g % PT, (g / PT) % KS);
        X[g][0] = h3_lds128(a); X[g][1] = h3_lds128(a + H3_LO);
    }
#pragma unroll 1
    for (int tap = 0; tap < 9; tap++) {
        const unsigned tn = tadr + (unsigned)(tap < 8 ? tap + 1 : 8) * 512u;
#pragma unroll
        for (int j = 0; j < NDW; j++) {
            rn[j] = h3_lds32(tn + 4 * j);
#if H3_ROWTILES == 1
            rn[j] ^= kx;
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int pt = s % PT, ks = s / PT, g = s + 2;
            if (pt == 0) {                                       // the weights WD blocks on (past the end: the last block again)
                const int nstep = tap * KS + ks + WD;
                const unsigned wn = wofs + (unsigned)(nstep < 9 * KS ? nstep : 9 * KS - 1) * 4096u;
#pragma unroll
                for (int q = 0; q < 4; q++) (WD == 2 ? Wnn : Wn)[q >> 1][q & 1] = h3_ldw(rs, wn + q * 1024u, lofs);
            }
            {
                const unsigned a = g < S ? row(rc, g % PT, g / PT) : row(rn, (g - S) % PT, ((g - S) / PT) % KS);
                X[g % 3][0] = h3_lds128(a); X[g % 3][1] = h3_lds128(a + H3_LO);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const h3h8 xh = __builtin_bit_cast(h3h8, X[s % 3][0]), xl = __builtin_bit_cast(h3h8, X[s % 3][1]);
                const h3h8 w0h = __builtin_bit_cast(h3h8, Wc[0][0]), w0l = __builtin_bit_cast(h3h8, Wc[0][1]);
                const h3h8 w1h = __builtin_bit_cast(h3h8, Wc[1][0]), w1l = __builtin_bit_cast(h3h8, Wc[1][1]);
                am[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xh, am[0][pt], 0, 0, 0);
                ac[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xl, ac[0][pt], 0, 0, 0);
                am[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xh, am[1][pt], 0, 0, 0);
                ac[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xl, ac[1][pt], 0, 0, 0);
                ac[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l, xh, ac[0][pt], 0, 0, 0);
                ac[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l, xh, ac[1][pt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pt == PT - 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    Wc[q >> 1][q & 1] = Wn[q >> 1][q & 1];
                    if constexpr (WD == 2) Wn[q >> 1][q & 1] = Wnn[q >> 1][q & 1];
                }
            }
        }
        // S steps moved the ring by S mod 3: bring it back so that the next tap starts at slot 0
        if constexpr (S % 3 == 1) { h3u4 t0 = X[1][0], t1 = X[1][1]; X[1][0] = X[2][0]; X[1][1] = X[2][1]; X[2][0] = X[0][0]; X[2][1] = X[0][1]; X[0][0] = t0; X[0][1] = t1; }
        if constexpr (S % 3 == 2) { h3u4 t0 = X[2][0], t1 = X[2][1]; X[2][0] = X[1][0]; X[2][1] = X[1][1]; X[1][0] = X[0][0]; X[1][1] = X[0][1]; X[0][0] = t0; X[0][1] = t1; }
#pragma unroll
        for (int j = 0; j < NDW; j++) rc[j] = rn[j];
    }
}

// Epilogue of a wavefront's tile: v = hi-sum + 2^-11 cross-sum + bias, ReLU, then either the split back into the LDS image (the
// C/D layout gives a lane 4 consecutive output channels of one position: one ds_write_b64 per plane) or y[32][140] in fp32.
template <int PT, bool LAST>
__device__ __forceinline__ void h3_store(unsigned lds0, const h3f4 (&am)[2][PT], const h3f4 (&ac)[2][PT], int relu, int ct, int pt0, unsigned lane,
                                         float* __restrict__ y, int* status, int* flag, bool to_lds = false)
{
    const int li = lane & 15, lk = lane >> 4;
    float amax = 0.f;
    float k2048 = 2048.f;
    asm volatile("" : "+v"(k2048));                              // one register for the multiplier of v_fma_mixlo / mixhi_f16
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, 0x7fffffff, 0x00027000);
    unsigned ybase = (unsigned)((32 * ct * H3_NPOS) * 4);
    asm volatile("" : "+s"(ybase));                              // (the per-store scalar offsets are formed here as well)
#pragma unroll
    for (int t = 0; t < PT; t++) {
        int py, px;
        const bool valid = h3_pos(pt0 + t, li, py, px);          // the lane's position in tile pt0 + t
        const unsigned p = (unsigned)(py * 20 + px);
        const unsigned key = h3_key(py, px);
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int c = 32 * ct + 16 * n + 4 * lk;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = __builtin_fmaf(ac[n][t][r], 1.f / 2048.f, am[n][t][r]);     // the bias started the hi-sum
            if constexpr (LAST) {
                if (to_lds) {                                    // fused head: the fp32 map [32][140] at the start of the (now free) image
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            *(__attribute__((address_space(3))) float*)(size_t)(lds0 + ((unsigned)(c + r) * H3_NPOS + p) * 4u) = relu ? fmaxf(v[r], 0.f) : v[r];
                    }
                } else if (valid) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu ? fmaxf(v[r], 0.f) : v[r]), yrs,
                                                              ((unsigned)(4 * lk) * H3_NPOS + p) * 4u, ybase + (unsigned)(((16 * n + r) * H3_NPOS) * 4), 0);
                }
            } else {
                h3_watch2(amax, v[0], v[1]); h3_watch2(amax, v[2], v[3]);       // before the ReLU
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
                }
                unsigned h0, h1, l0, l1;
                h3_split2(v[0], v[1], k2048, h0, l0);
                h3_split2(v[2], v[3], k2048, h1, l1);
                const h3u2 hi = { h0, h1 }, lo = { l0, l1 };
                if (valid) {
                    // 4 channels = 8 bytes of the plane: chunk (c >> 3) keyed by the position, half (c & 4) inside it
                    const unsigned a = lds0 + p * H3_S + ((((unsigned)c >> 3) ^ key) << 4) + ((unsigned)c & 4u) * 2u;
                    *(__attribute__((address_space(3))) h3u2*)(size_t)a = hi;
                    *(__attribute__((address_space(3))) h3u2*)(size_t)(a + H3_LO) = lo;
                }
            }
        }
    }
    if (!LAST && (status || flag) && __builtin_amdgcn_ballot_w64(!(amax < 65504.f)) != 0 && lane == 0) {
        if (status) atomicOr(status, 1);
        if (flag) *flag = 1;                                     // this patch goes through the fp32 kernel again (buf_cylindrical_net_split_safe)
    }
}

template <int PT, int KS>
__device__ __forceinline__ void h3_layer(unsigned lds0, const void* wt, const float* bias, int relu, int ct, int pt0, unsigned lane, bool last,
                                         float* y, int* status, int* flag, bool to_lds)
{
    h3f4 am[2][PT], ac[2][PT];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        // the hi-sum starts at the bias (C/D layout: a lane holds outputs 32 ct + 16 n + 4 (lane >> 4) + r of its position)
        const h3f4 b = *reinterpret_cast<const h3f4*>(bias + 32 * ct + 16 * n + 4 * (lane >> 4));
#pragma unroll
        for (int t = 0; t < PT; t++) { am[n][t] = b; ac[n][t] = (h3f4){ 0.f, 0.f, 0.f, 0.f }; }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wt, 0, 0x7fffffff, 0x00027000);
    h3_gemm<PT, KS>(lds0, rs, (unsigned)ct * (9u * KS * 4096u), pt0, lane, am, ac);
    H3_STAMP_AT(30)
    __syncthreads();                                             // every wavefront has read its input: the image may be rewritten
    H3_STAMP_AT(31)
    unsigned lane_s = lane;
    asm volatile("" : "+v"(lane_s));                             // the store addresses are formed here, not hoisted out of the layer loop (and spilled)
    if (last) h3_store<PT, true>(lds0, am, ac, relu, ct, pt0, lane_s, y, status, flag, to_lds);
    else h3_store<PT, false>(lds0, am, ac, relu, ct, pt0, lane_s, y, status, flag);
}

template <int KS>
__device__ __forceinline__ void h3_dispatch(unsigned lds0, const void* wt, const float* bias, int relu, int cout, int w, unsigned lane, bool last,
                                            float* y, int* status, int* flag, bool to_lds)
{
    if (cout == 128) h3_layer<9, KS>(lds0, wt, bias, relu, w, 0, lane, last, y, status, flag, to_lds);
    else if (cout == 64) {
        if (w < 2) h3_layer<4, KS>(lds0, wt, bias, relu, w & 1, 0, lane, last, y, status, flag, to_lds);
        else h3_layer<5, KS>(lds0, wt, bias, relu, w & 1, 4, lane, last, y, status, flag, to_lds);
    } else {
        if (w == 3) h3_layer<3, KS>(lds0, wt, bias, relu, 0, 6, lane, last, y, status, flag, to_lds);
        else h3_layer<2, KS>(lds0, wt, bias, relu, 0, 2 * w, lane, last, y, status, flag, to_lds);
    }
}

__global__ void __launch_bounds__(H3_THREADS, 2) k_cyl_net_h3(const float* __restrict__ x, CylH3Params P, float* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char h3_smem[];
    const unsigned lds0 = (unsigned)(size_t)(h3_lds_p)h3_smem;
    const int patch = blockIdx.x, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const unsigned lane = tid & (WAVE - 1);
    int* const flag = P.flags ? P.flags + patch : nullptr;
    // address table: tab[tap][p] = row (in 16-byte units) that tap (dy, dx) reads for output position p
    for (int e = tid; e < 9 * 256; e += H3_THREADS) {
        const int tap = e >> 8, t = e & 15, m = (e >> 4) & 15;               // e = (tap, lane m, tile slot t)
        int yy0 = 7, xx0 = 0;
        const bool ok = t < 9 && h3_pos(t, m, yy0, xx0);                      // idle slots read like the virtual row they stand for
        const int yy = yy0 + tap / 3 - 1;
        int xx = xx0 + tap % 3 - 1;
        xx += xx < 0 ? 20 : 0; xx -= xx >= 20 ? 20 : 0;
#if H3_ROWTILES == 1
        (void)ok;
        const unsigned row = (t >= 9 || yy < 0 || yy > 6) ? (unsigned)H3_NPOS : (unsigned)(yy * 20 + xx);
        *(__attribute__((address_space(3))) unsigned short*)(size_t)(lds0 + H3_TAB + 2u * e) = (unsigned short)((row << 5) | h3_key(yy, xx));
#elif H3_ROWTILES == 2
        (void)ok;                                                             // (idle slots read the row they would stand for, or zeros)
        const unsigned row = (t >= 9 || yy < 0 || yy > 6) ? (unsigned)H3_NPOS : (unsigned)(yy * 20 + xx);
        *(__attribute__((address_space(3))) unsigned short*)(size_t)(lds0 + H3_TAB + 2u * e) = (unsigned short)(row * (H3_S / 16));
#else
        const unsigned row = (!ok || yy < 0 || yy > 6) ? (unsigned)H3_NPOS : (unsigned)(yy * 20 + xx);
        *(__attribute__((address_space(3))) unsigned short*)(size_t)(lds0 + H3_TAB + 2u * e) = (unsigned short)(row * (H3_S / 16));
#endif
    }
    for (int i = tid; i < (int)(H3_S / 16); i += H3_THREADS) *(__attribute__((address_space(3))) h3u4*)(size_t)(lds0 + H3_ZERO + 16u * i) = (h3u4){ 0, 0, 0, 0 };
    {   // input x[cin0][140] fp32 -> split rows; a work item = (4 positions, 2 channels), channel pairs fastest (one position's
        // dwords are consecutive banks)
        const int cin0 = P.cin[0], np = cin0 >> 1, cpad = (cin0 + 31) & ~31;
        const float* src = x + (size_t)patch * cin0 * H3_NPOS;
        unsigned amax = 0u;
        for (int i = tid; i < 35 * np; i += H3_THREADS) {
            const int q = i / np, cp = i - q * np;
            const h3f4 a = __builtin_nontemporal_load(reinterpret_cast<const h3f4*>(src + (2 * cp) * H3_NPOS + 4 * q));
            const h3f4 b = __builtin_nontemporal_load(reinterpret_cast<const h3f4*>(src + (2 * cp + 1) * H3_NPOS + 4 * q));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                h3h2 hi, lo;
                _Float16 h, l;
                h3_split(a[j], h, l); hi[0] = h; lo[0] = l;
                h3_split(b[j], h, l); hi[1] = h; lo[1] = l;
                h3_watch(amax, a[j]); h3_watch(amax, b[j]);
                const int pp = 4 * q + j, py = pp / 20;
                const unsigned ad = lds0 + (unsigned)pp * H3_S + ((((unsigned)cp >> 2) ^ h3_key(py, pp - 20 * py)) << 4) + 4u * ((unsigned)cp & 3u);
                *(__attribute__((address_space(3))) unsigned*)(size_t)ad = __builtin_bit_cast(unsigned, hi);
                *(__attribute__((address_space(3))) unsigned*)(size_t)(ad + H3_LO) = __builtin_bit_cast(unsigned, lo);
            }
        }
        if ((P.status || flag) && __builtin_amdgcn_ballot_w64(amax >= H3_F16_LIMIT_BITS) != 0 && lane == 0) {
            if (P.status) atomicOr(P.status, 1);
            if (flag) *flag = 1;
        }
        const int nz = (cpad - cin0) >> 1;                       // zero channel pairs up to the k-step boundary (48 -> 64)
        for (int i = tid; i < H3_NPOS * nz; i += H3_THREADS) {
            const int p = i / nz, cp = (cin0 >> 1) + i - p * nz, py = p / 20;
            const unsigned ad = lds0 + (unsigned)p * H3_S + ((((unsigned)cp >> 2) ^ h3_key(py, p - 20 * py)) << 4) + 4u * ((unsigned)cp & 3u);
            *(__attribute__((address_space(3))) unsigned*)(size_t)ad = 0u;
            *(__attribute__((address_space(3))) unsigned*)(size_t)(ad + H3_LO) = 0u;
        }
    }
    __syncthreads();
    H3_STAMP_AT(0)
#pragma unroll 1
    for (int l = 0; l < H3_LAYERS; l++) {
        const int ks = (P.cin[l] + 31) >> 5, cout = P.cout[l];
        const bool last = l == H3_LAYERS - 1;
        float* yo = y + (size_t)patch * cout * H3_NPOS;
        if (ks == 4) h3_dispatch<4>(lds0, P.wt[l], P.bias[l], P.relu[l], cout, w, lane, last, yo, P.status, flag, last && P.head != nullptr);
        else if (ks == 2) h3_dispatch<2>(lds0, P.wt[l], P.bias[l], P.relu[l], cout, w, lane, last, yo, P.status, flag, last && P.head != nullptr);
        else h3_dispatch<1>(lds0, P.wt[l], P.bias[l], P.relu[l], cout, w, lane, last, yo, P.status, flag, last && P.head != nullptr);
#ifdef H3_STAMP
        if ((threadIdx.x & 63) == 0) {      // gemm end / barrier end of this layer (slots 30, 31) -> per-layer slots
            long long* q = h3_stamp_ptr + ((size_t)blockIdx.x * 4 + threadIdx.x / 64) * 32;
            q[1 + 3 * l] = q[30]; q[2 + 3 * l] = q[31];
        }
#endif
        H3_STAMP_AT(3 + 3 * l)
        __syncthreads();
    }
    if (P.head) {                                                // attention pooling + normalisation on the map the last layer left in LDS
        dh_lds* ys = (dh_lds*)reinterpret_cast<float*>(h3_smem);
        dh_lds* wgt = ys + DH_C * CN_POS;
        dh_body(ys, wgt, wgt + CN_POS, wgt + 2 * CN_POS, wgt + 2 * CN_POS + DH_C, P.head, P.desc, P.equi, patch, tid);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// f32 -> f16 bits, round to nearest even, subnormals kept (the device's v_cvt_f16_f32 in its default mode)
static unsigned short h3_f16_bits(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);                  // NaN
    if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                 // >= 65520 rounds to infinity
    if (u < 0x38800000u) {                                                         // below 2^-14: a multiple of 2^-24
        float a;
        memcpy(&a, &u, 4);
        const float scaled = a * 16777216.f;                                      // exact
        float r = (float)(long long)scaled;                                       // truncation; then round half to even by hand
        const float d = scaled - r;
        if (d > 0.5f || (d == 0.5f && ((long long)r & 1))) r += 1.f;
        return (unsigned short)(sign | (unsigned)(long long)r);                   // 1024 = the smallest normal: the bit pattern carries over
    }
    const unsigned odd = (u >> 13) & 1u;
    u += 0xfffu + odd;                                                             // round the 13 dropped bits to nearest even
    return (unsigned short)(sign | ((u - 0x38000000u) >> 13));
}

static float h3_f16_value(unsigned short h)
{
    const unsigned e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = (float)m * (1.f / 16777216.f);
    else if (e == 31) v = m ? NAN : INFINITY;
    else { const unsigned u = ((e + 112u) << 23) | (m << 13); memcpy(&v, &u, 4); }
    return (h & 0x8000u) ? -v : v;
}

// Host helper: filters w [Cout][Cin][3][3] (BN folded, fp32) -> the two f16 planes in the kernel's tiling,
//     out[((((g 9 + tap) KS + ks) 2 + n2) 2 + plane) 512 + (kg 16 + row) 8 + i]  =  plane(w[32 g + 16 n2 + row][32 ks + 8 kg + i][tap])
// (tap = 3 ky + kx; KS = ceil(Cin / 32), channels beyond Cin are zero; plane 0 = hi, plane 1 = lo' = f16((w - hi) 2^11)):
// 2 * 9 * KS * 32 * Cout u16.  No device work.
extern "C" long long buf_split_filter_count(int cout, int cin) { return 2LL * 9 * ((cin + 31) / 32) * 32 * cout; }

extern "C" int buf_split_tile_filters(const float* w_host, int cout, int cin, unsigned short* out_host)
{
    BUF_REQUIRE(w_host && out_host, BUF_EINVAL, "buf_split_tile_filters: null argument");
    BUF_REQUIRE(cout > 0 && cin > 0 && cout % 32 == 0, BUF_EINVAL, "buf_split_tile_filters: widths %d -> %d", cin, cout);
    const int KS = (cin + 31) / 32;
    memset(out_host, 0, (size_t)buf_split_filter_count(cout, cin) * sizeof(unsigned short));
    for (int o = 0; o < cout; o++)
        for (int c = 0; c < cin; c++)
            for (int tap = 0; tap < 9; tap++) {
                const float v = w_host[((size_t)o * cin + c) * 9 + tap];
                BUF_REQUIRE(fabsf(v) < 65504.f, BUF_EINVAL, "buf_split_tile_filters: weight %g outside the f16 range", (double)v);
                const unsigned short hi = h3_f16_bits(v);
                const unsigned short lo = h3_f16_bits((v - h3_f16_value(hi)) * 2048.f);
                const int g = o / 32, n2 = (o % 32) / 16, row = o % 16, ks = c / 32, kg = (c % 32) / 8, i = c % 8;
                const size_t base = ((((size_t)g * 9 + tap) * KS + ks) * 2 + n2) * 2;
                out_host[(base + 0) * 512 + (kg * 16 + row) * 8 + i] = hi;
                out_host[(base + 1) * 512 + (kg * 16 + row) * 8 + i] = lo;
            }
    return BUF_OK;
}

// x f32[np,Cin0,140] -> y f32[np,32,140] with fp32-equivalent arithmetic on the f16 matrix pipe (header: buf_cylindrical_net_split).
static int h3_launch(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host, const int* cin_host,
                     const int* cout_host, const int* relu_host, float* y, const float* head, float* desc, float* equi, int* status_dev, void* stream,
                     int* flags_dev = nullptr);

extern "C" int buf_cylindrical_net_split(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host,
                                         const int* cin_host, const int* cout_host, const int* relu_host, float* y, int* status_dev, void* stream)
{
    return h3_launch(x, npatch, wt_host, bias_host, cin_host, cout_host, relu_host, y, nullptr, nullptr, nullptr, status_dev, stream);
}

// The same stack with the descriptor head (buf_descriptor_head: attention pooling + normalisation) fused behind the last layer: the
// [32][140] map never goes to HBM.  head_params: DEVICE f32[545] as for buf_descriptor_head -> desc f32[np,32], equi f32[np,32,140],
// bit-identical to buf_cylindrical_net_split followed by buf_descriptor_head.
extern "C" int buf_cylindrical_net_split_head(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host,
                                              const int* cin_host, const int* cout_host, const int* relu_host, const float* head_params,
                                              float* desc, float* equi, int* status_dev, void* stream)
{
    BUF_REQUIRE(npatch <= 0 || (head_params && desc && equi), BUF_EINVAL, "buf_cylindrical_net_split_head: null argument");
    return h3_launch(x, npatch, wt_host, bias_host, cin_host, cout_host, relu_host, nullptr, head_params, desc, equi, status_dev, stream);
}

static int h3_launch(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host, const int* cin_host,
                     const int* cout_host, const int* relu_host, float* y, const float* head, float* desc, float* equi, int* status_dev, void* stream,
                     int* flags_dev)
{
    BUF_REQUIRE(npatch >= 0, BUF_EINVAL, "buf_cylindrical_net_split: npatch=%d", npatch);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(x && (y || head) && wt_host && bias_host && cin_host && cout_host && relu_host, BUF_EINVAL, "buf_cylindrical_net_split: null argument");
    static_assert(DH_THREADS == H3_THREADS && DH_C == 32 && CN_POS == H3_NPOS, "the fused head runs on the kernel's own workgroup and map");
    CylH3Params P;
    for (int l = 0; l < H3_LAYERS; l++) {
        P.wt[l] = wt_host[l]; P.bias[l] = bias_host[l];
        P.cin[l] = cin_host[l]; P.cout[l] = cout_host[l]; P.relu[l] = relu_host[l];
        BUF_REQUIRE(P.wt[l] && P.bias[l], BUF_EINVAL, "buf_cylindrical_net_split: null weights for layer %d", l);
        BUF_REQUIRE(P.cin[l] > 0 && P.cin[l] % 16 == 0 && P.cin[l] <= H3_MAXC && (P.cout[l] == 32 || P.cout[l] == 64 || P.cout[l] == 128),
                    BUF_EINVAL, "buf_cylindrical_net_split: layer %d has unsupported widths %d -> %d", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(((P.cin[l] + 31) / 32) != 3, BUF_EINVAL, "buf_cylindrical_net_split: layer %d has unsupported widths %d -> %d (Cin 65..96)", l, P.cin[l], P.cout[l]);
        BUF_REQUIRE(l == 0 || P.cin[l] == P.cout[l - 1], BUF_EINVAL, "buf_cylindrical_net_split: layer %d width mismatch", l);
    }
    BUF_REQUIRE(P.cout[H3_LAYERS - 1] == 32, BUF_EINVAL, "buf_cylindrical_net_split: the last layer must have 32 channels");
    P.status = status_dev;
    P.flags = flags_dev;
    P.head = head; P.desc = desc; P.equi = equi;
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds((const void*)k_cyl_net_h3, H3_LDS, grant)) return rc;
    double macs = 0;
    for (int l = 0; l < H3_LAYERS; l++) macs += 9.0 * P.cin[l] * P.cout[l];
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 2.0 * 140 * macs * npatch, BUF_TIMED_CYL_NET_SPLIT);
#ifdef H3_STAMP
    long long* stamps = nullptr;
    BUF_CHECK_HIP(hipMalloc(&stamps, (size_t)npatch * 4 * 32 * sizeof(long long)));
    BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(h3_stamp_ptr), &stamps, sizeof(stamps)));
#endif
    k_cyl_net_h3<<<npatch, H3_THREADS, H3_LDS, (hipStream_t)stream>>>(x, P, y);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
#ifdef H3_STAMP
    if (npatch >= 4096) {   // per layer and wavefront: K loops | wait at the barrier | epilogue (split + store); second half of the workgroups
        BUF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        long long* h = (long long*)malloc((size_t)npatch * 4 * 32 * sizeof(long long));
        BUF_CHECK_HIP(hipMemcpy(h, stamps, (size_t)npatch * 4 * 32 * sizeof(long long), hipMemcpyDeviceToHost));
        double g[H3_LAYERS][4] = {}, bw[H3_LAYERS][4] = {}, ep[H3_LAYERS][4] = {}, tot = 0;
        long n = 0;
        for (int b = npatch / 2; b < npatch; b++, n++)
            for (int w = 0; w < 4; w++) {
                const long long* q = h + ((size_t)b * 4 + w) * 32;
                for (int l = 0; l < H3_LAYERS; l++) {
                    const long long start = l == 0 ? q[0] : q[3 + 3 * (l - 1)];
                    g[l][w] += (double)(q[1 + 3 * l] - start); bw[l][w] += (double)(q[2 + 3 * l] - q[1 + 3 * l]); ep[l][w] += (double)(q[3 + 3 * l] - q[2 + 3 * l]);
                }
                if (w == 0) tot += (double)(q[3 + 3 * (H3_LAYERS - 1)] - q[0]);
            }
        fprintf(stderr, "H3_STAMP: %ld workgroups, layers total %.0f cycles per patch\n", n, tot / n);
        for (int l = 0; l < H3_LAYERS; l++) {
            const int ks = (P.cin[l] + 31) / 32;
            const double mf = 9.0 * ks * 27 * (P.cout[l] / 16) / 4;        // MFMAs per wave (mean)
            fprintf(stderr, "  layer %d %3d->%3d: MFMAs/wave %5.0f | K loops %6.0f %6.0f %6.0f %6.0f | barrier wait %5.0f %5.0f %5.0f %5.0f | epilogue %5.0f %5.0f %5.0f %5.0f | cycles per MFMA (wave 0, whole layer) %.1f\n",
                    l, P.cin[l], P.cout[l], mf, g[l][0] / n, g[l][1] / n, g[l][2] / n, g[l][3] / n, bw[l][0] / n, bw[l][1] / n, bw[l][2] / n, bw[l][3] / n,
                    ep[l][0] / n, ep[l][1] / n, ep[l][2] / n, ep[l][3] / n, (g[l][0] + bw[l][0] + ep[l][0]) / n / mf);
        }
        free(h);
    }
    (void)hipFree(stamps);
#endif
    return BUF_OK;
}
