// A9 + A10 + the point MLP of A11, fused (models/patch_embedder.py:123-171,74-79;
// utils/common.py:431-498,501-525).
//
// Reference: axis_align -> normalize -> sphere_query (420 ball queries of 10 samples per patch,
// materialising [P,420,10,3]) -> var_to_invar -> Conv2d1x1(3->16)+BN+ReLU -> max over the 10 samples.
// Here: one workgroup per patch; the aligned, normalised 512-point patch sits in LDS (8 KB).
//   1. hit masks (bit k of centre c <=> |centre_c - point_k|^2 < r^2) for ALL centre x point pairs on the f16 matrix pipe
//      (round 4; rounds 1-3 tested ~30 candidate centres per point found through a 16^3 lookup grid and set the bits with LDS
//      atomics: 42 k of the kernel's 59 k cycles per patch, bound by the number of wavefront-level tests).  One
//      v_mfma_f32_32x32x16_f16 gives D = d^2 - r^2 of 32 points x 32 centres to ~1e-6: both operands are split x = hi + 2^-11 lo'
//      (22 significant bits) and the K = 16 slots hold the terms of |c|^2 - r^2 + |q|^2 - 2 c.q:
//          slot   point side (A)        centre side (B)
//          0-2    q_hi (x,y,z)          X_hi             X = -2c
//          3-5    q_lo'                 2^-11 X_hi
//          6-8    q_hi                  2^-11 X_lo'
//          9,10   |q|^2: hi, lo'        1, 2^-11
//          11-13  1, 2^-11, 2^-22       |c|^2 - r^2 in three pieces
//      The sign bit of D is the hit bit, shifted into the lane's mask piece by one v_alignbit per pair.  The answer has to be
//      the reference's fp32 test `sqdist3(c, q) < r2` bit for bit, and it is: a lane whose |D| comes within eps (hdr[1], > 6x the
//      error bound of the split form, see k_vox_ctab) of zero for any of its 16 pairs redoes those 16 tests with sqdist3 --
//      about one pair in 10^5, a few lanes per patch.  Points that no ball can reach (|q| beyond max|c| + r, NaN) get a vector
//      that makes D = +30000.
//   2. centre-parallel: one lane per voxel centre walks its mask in index order (= pointnet2 ball_query's "first nsample
//      in index order") and runs each hit through the azimuth de-rotation, the 3->16 MLP, BN, ReLU and the running max,
//      all lanes slot by slot.
// HBM traffic: 6 KB in, 26.9 KB out per patch; [P,420,10,3] never exists.
#include "common.h"
#include <type_traits>

#define VOX_THREADS 448     // 7 wavefronts >= 420 centres = 14 centre tiles of 32
#define VOX_MAXPTS 1024
static_assert(VOX_MAXPTS <= 32 * 32, "the non-empty-word summary of a hit mask is one 32-bit register (voxelize phase 2+3)");
#define VOX_CH 16
#define VOX_MAXS 16           // max samples per voxel kept in the hit list
#define VOX_K 16              // K slots of the distance MFMA = f16 values per point / centre vector
#ifndef VOX_EPS_SCALE
#define VOX_EPS_SCALE 1.f           // development builds (-DVOX_EPS_SCALE=...): how far eps can shrink before a hit mask changes
#endif
#define VOX_BIG 30000.f       // D of a pair that cannot hit (unreachable point, lane past the last centre)

typedef _Float16 voxh8 __attribute__((ext_vector_type(8)));
typedef float voxf16 __attribute__((ext_vector_type(16)));

struct VoxMlp {             // Conv2d1x1(3->16) with the eval-mode BatchNorm folded in: 64 scalars, SGPR-resident
    float w[VOX_CH][3];     // s * Desc.pnt_layer.0.weight,  s = gamma / sqrt(var + 1e-5)
    float b[VOX_CH];        // s * Desc.pnt_layer.0.bias + (beta - mean * s)
};

#ifdef VOX_STAMP
__device__ long long* vox_stamp_ptr;      // development build (-DVOX_STAMP): s_memtime of thread 0 at the phase boundaries
#define VOX_STAMP_AT(SLOT) if (threadIdx.x == 0) vox_stamp_ptr[(size_t)blockIdx.x * 8 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define VOX_STAMP_AT(SLOT)
#endif

// A9 (models/patch_embedder.py:131-147): per patch the rotation that takes the keypoint's axis onto e_z and rand_axis; one thread
// per patch, ahead of k_patch_voxelize (inside it this was ~400 serial instructions on one lane in front of a barrier).
__global__ void __launch_bounds__(256) k_patch_rotation(const float* __restrict__ axis, int npatch, float* __restrict__ out_R,
                                                        float* __restrict__ out_rand)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npatch) return;
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    float rx = 1.f, ry = 0.f, rz = 0.f;                          // KITTI/ETH: rand_axis = e_x, R = I (:143-147)
    if (axis) {
        // RodsRotatFormula(z_axis, e_z) (utils/common.py:501-525), returned transposed
        float ax = axis[3 * (size_t)p], ay = axis[3 * (size_t)p + 1], az = axis[3 * (size_t)p + 2];
        float cx = ay, cy = -ax, cz = 0.f;                       // a x e_z
        float na = sqrtf(ax * ax + ay * ay + az * az);
        float cs = az / (fmaxf(na, 1e-8f) * 1.0f);               // cosine_similarity
        cs = fminf(fmaxf(cs, -1.f), 1.f);
        float th = acosf(cs);
        float nc = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);   // F.normalize
        cx /= nc; cy /= nc; cz /= nc;
        rx = cx; ry = cy; rz = cz;                               // rand_axis = normalize(z_axis x e_z) (:138-141)
        float K[9] = { 0, -cz, cy, cz, 0, -cx, -cy, cx, 0 };
        float K2[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
        float sn = sinf(th), oc = 1.f - cosf(th);
        float Rm[9];
        for (int i = 0; i < 9; i++) Rm[i] = (i % 4 == 0 ? 1.f : 0.f) + sn * K[i] + oc * K2[i];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[3 * i + j] = Rm[3 * j + i];
    }
    for (int i = 0; i < 9; i++) out_R[9 * (size_t)p + i] = R[i];
    out_rand[3 * (size_t)p] = rx; out_rand[3 * (size_t)p + 1] = ry; out_rand[3 * (size_t)p + 2] = rz;
}

__global__ void __launch_bounds__(VOX_THREADS) k_patch_voxelize(const float* __restrict__ patches, const float* __restrict__ axis,
                                                             int npts, float des_r, const float* __restrict__ centres,
                                                             int ncentres, int azi_n, const float* __restrict__ azi_cs,
                                                             float voxel_r2, int nsample, VoxMlp M,
                                                             const float* __restrict__ tab_hdr, const unsigned short* __restrict__ tab,
                                                             float* __restrict__ out_x, const float* __restrict__ out_R,
                                                             float* __restrict__ out_patches)
{
    // dynamic LDS, sized by the launch (52 KB at 512 points -> three workgroups per CU; the hit lists never leave the registers):
    extern __shared__ float4 pts[];                              // [npts] aligned, normalised patch
    const int W = (npts + 31) >> 5;                              // mask words per centre = point tiles of 32
    unsigned* mask = reinterpret_cast<unsigned*>(pts + npts);    // [W][VOX_THREADS] hit bits
    voxh8* pvec = reinterpret_cast<voxh8*>(mask + (size_t)W * VOX_THREADS);   // [W][32 rows][2 halves]: point vectors, MFMA row order
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* src = patches + (size_t)p * npts * 3;
    VOX_STAMP_AT(0)
    float Rs[9];                                                 // k_patch_rotation's matrix: uniform loads
#pragma unroll
    for (int i = 0; i < 9; i++) Rs[i] = out_R[9 * (size_t)p + i];
    // centre on the keypoint (= last slot, :124-125), rotate (delta @ R), divide by des_r (:168-171)
    float kx = src[3 * (size_t)(npts - 1)], ky = src[3 * (size_t)(npts - 1) + 1], kz = src[3 * (size_t)(npts - 1) + 2];
    for (int k = tid; k < npts; k += VOX_THREADS) {
        float dx = src[3 * (size_t)k] - kx, dy = src[3 * (size_t)k + 1] - ky, dz = src[3 * (size_t)k + 2] - kz;
        float x = dx * Rs[0] + dy * Rs[3] + dz * Rs[6];
        float y = dx * Rs[1] + dy * Rs[4] + dz * Rs[7];
        float z = dx * Rs[2] + dy * Rs[5] + dz * Rs[8];
        x = x / des_r; y = y / des_r; z = z / des_r;
        pts[k] = make_float4(x, y, z, 0.f);
        if (out_patches) {
            float* d = out_patches + ((size_t)p * npts + k) * 3;
            d[0] = x; d[1] = y; d[2] = z;
        }
    }
    __syncthreads();
    VOX_STAMP_AT(1)

    const int c = tid;
    const bool active = c < ncentres;
    float ca = 1.f, sa = 0.f;
    if (active) {
        int az = c % azi_n;                                      // ordering rad -> ele -> azi (utils/common.py:422-428)
        ca = azi_cs[2 * az]; sa = azi_cs[2 * az + 1];            // cos/sin of -az * 2pi/azi_n (:485-491)
    }
    // Phase 1a: the 16 f16 of every point (slots above), stored at the MFMA row that hands lane half h of the result the points
    // 16h .. 16h+15 of the tile in register order: point 16h + 4a + b -> row 8a + 4h + b.
    {
        const float lim2 = tab_hdr[0];
        for (int k = tid; k < W * 32; k += VOX_THREADS) {
            const float4 q = pts[k < npts ? k : npts - 1];
            const float qq = fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x));
            const bool reach = k < npts && qq <= lim2;           // (NaN: unreachable, as in the reference's `d2 < r2`)
            voxh8 v0 = { 0, 0, 0, 0, 0, 0, 0, 0 }, v1 = { 0, (_Float16)VOX_BIG, 0, (_Float16)1.f, (_Float16)0x1p-11f, (_Float16)0x1p-22f, 0, 0 };
            if (reach) {
                const _Float16 hx = (_Float16)q.x, hy = (_Float16)q.y, hz = (_Float16)q.z;
                const _Float16 lx = (_Float16)((q.x - (float)hx) * 2048.f), ly = (_Float16)((q.y - (float)hy) * 2048.f),
                               lz = (_Float16)((q.z - (float)hz) * 2048.f);
                const _Float16 q0 = (_Float16)qq, q1 = (_Float16)((qq - (float)q0) * 2048.f);
                v0 = voxh8{ hx, hy, hz, lx, ly, lz, hx, hy };
                v1[0] = hz; v1[1] = q0; v1[2] = q1;
            }
            const int p = k & 31, row = ((p >> 2) & 3) * 8 + (p >> 4) * 4 + (p & 3);
            pvec[((k >> 5) * 32 + row) * 2] = v0;
            pvec[((k >> 5) * 32 + row) * 2 + 1] = v1;
        }
    }
    __syncthreads();
    // Phase 1b: wavefront w owns centre tiles w and w + 7 (B operands in registers), walks the point tiles and writes, per tile,
    // the 16-bit mask piece of (centre = lane & 31, points 16 (lane >> 5) ..) -- every piece of every active centre is written,
    // so the masks need no clearing.
    {
        const int wv = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
        const float eps = tab_hdr[1];
        const voxh8* __restrict__ ctab = reinterpret_cast<const voxh8*>(tab);
        unsigned short* mask16 = reinterpret_cast<unsigned short*>(mask);
        const voxf16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int j = 0; j < 2; j++) {
            const int cen = (wv + 7 * j) * 32 + col;
            if ((wv + 7 * j) * 32 >= ncentres) break;
            const voxh8 B = ctab[cen * 2 + half];
            const voxh8* arow = pvec + col * 2 + half;
            unsigned short* mrow = mask16 + (cen << 1) + half;
            const float cx = cen < ncentres ? centres[3 * cen] : 0.f, cy = cen < ncentres ? centres[3 * cen + 1] : 0.f,
                        cz = cen < ncentres ? centres[3 * cen + 2] : 0.f;
            auto tile = [&](const voxh8* a, unsigned short* m, int t) __attribute__((always_inline)) {
                const voxf16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(*a, B, zero, 0, 0, 0);
                unsigned piece = 0u;
                float amin = 3.4e38f;
#pragma unroll
                for (int i = 15; i >= 0; i--) piece = __builtin_amdgcn_alignbit(piece, __float_as_uint(D[i]), 31);   // (piece << 1) | sign
#pragma unroll
                for (int i = 0; i < 16; i += 2) amin = fminf(fminf(amin, fabsf(D[i])), fabsf(D[i + 1]));
                if (!(amin >= eps)) {                            // too close to call from the split form: the reference's own test
                    unsigned fix = 0u;
                    if (cen < ncentres) {
#pragma unroll 1
                        for (int i = 0; i < 16; i++) {
                            const int k = t * 32 + half * 16 + i;
                            if (k < npts) {
                                const float4 q = pts[k];
                                if (sqdist3(cx, cy, cz, q.x, q.y, q.z) < voxel_r2) fix |= 1u << i;
                            }
                        }
                    }
                    piece = fix;
                }
                *m = (unsigned short)piece;
            };
            int t0 = 0;
            for (; t0 + 4 <= W; t0 += 4) {                       // groups of four tiles: constant offsets from one address pair
                const voxh8* a4 = arow + t0 * 64;
                unsigned short* m4 = mrow + t0 * (2 * VOX_THREADS);
#pragma unroll
                for (int u = 0; u < 4; u++) tile(a4 + u * 64, m4 + u * (2 * VOX_THREADS), t0 + u);
            }
            for (; t0 < W; t0++) tile(arow + t0 * 64, mrow + t0 * (2 * VOX_THREADS), t0);
        }
    }
    __syncthreads();
    VOX_STAMP_AT(2)
    // Phases 2 + 3: every centre lane walks its hit mask in index order (= ball_query's "first nsample in index order") and
    // feeds each hit through de-rotation -> 3->16 MLP -> running max; all lanes advance two slots per step (one v_max3 takes
    // both).  max_s relu(w.n_s + b) = relu(max_s(w.n_s) + b) (rounding is monotone), so the bias, BN shift and ReLU are applied
    // once, after the walk: 3 multiply-adds and half a max per channel and sample.
    // `nz` = the mask words that hold a hit; a word is fetched from LDS only when the previous one is used up.
    int cnt = 0;                                                 // accepted samples (incl. a zeroed hit on point 0)
    bool zero_slot = false;
    unsigned nz = 0, bits = 0;
    int wcur = 0;
    if (active)
        for (int w = 0; w < W; w++) nz |= (mask[w * VOX_THREADS + tid] != 0u ? 1u : 0u) << w;
    float acc[VOX_CH];
#pragma unroll
    for (int ch = 0; ch < VOX_CH; ch++) acc[ch] = -3.4e38f;
    auto next_hit = [&]() __attribute__((always_inline)) {        // index of the lane's next hit, -1: none left
        if (bits == 0u && nz != 0u) {
            wcur = __ffs(nz) - 1;
            nz &= nz - 1u;
            bits = mask[wcur * VOX_THREADS + tid];
        }
        if (bits == 0u) return -1;
        const int k = wcur * 32 + __ffs(bits) - 1;
        bits &= bits - 1u;
        cnt++;
        zero_slot = zero_slot || k == 0;                         // utils/common.py:447-449: a hit on point 0 is zeroed
        return k;
    };
    for (int sidx = 0; sidx < nsample && __any(bits != 0u || nz != 0u); sidx += 2) {
        const int ka = next_hit();
        const int kb = sidx + 1 < nsample ? next_hit() : -1;
        // a lone hit stands in for its missing partner; a lane with neither (out of hits, or a hit on point 0) takes the keypoint
        // in the last slot, which is the origin: w.0 = 0 is the value its zero-padded slots contribute anyway
        const bool va = ka > 0, vb = kb > 0;
        const float4 qa = pts[va ? ka : (vb ? kb : npts - 1)], qb = pts[vb ? kb : (va ? ka : npts - 1)];
        const float ax = qa.x * ca - qa.y * sa, ay = qa.x * sa + qa.y * ca, az = qa.z;
        const float bx = qb.x * ca - qb.y * sa, by = qb.x * sa + qb.y * ca, bz = qb.z;
#pragma unroll
        for (int ch = 0; ch < VOX_CH; ch++) {
            const float ha = fmaf(M.w[ch][2], az, fmaf(M.w[ch][1], ay, M.w[ch][0] * ax));
            const float hb = fmaf(M.w[ch][2], bz, fmaf(M.w[ch][1], by, M.w[ch][0] * bx));
            acc[ch] = fmaxf(fmaxf(acc[ch], ha), hb);
        }
    }
    VOX_STAMP_AT(3)
    if (active) {
        const bool padded = cnt < nsample || zero_slot;          // zeroed slots go through the MLP as the origin: w.0 = 0
#pragma unroll
        for (int ch = 0; ch < VOX_CH; ch++) {
            const float v = padded ? fmaxf(acc[ch], 0.f) : acc[ch];
            out_x[((size_t)p * VOX_CH + ch) * ncentres + c] = fmaxf(v + M.b[ch], 0.f);
        }
    }
    VOX_STAMP_AT(4)
}

// Centre side of the distance MFMA: ctab[VOX_THREADS][16] f16 (slots in the file header), hdr[0] = the squared reach
// (max|c| + r)^2 with a margin -- a point beyond it hits no ball --, hdr[1] = eps, the |D| under which a lane falls back on the
// reference's fp32 test.  Error of D against the exact d^2 - r^2, S = (max|c| + reach)^2 bounding every partial sum:
// fp32 accumulation of 14 exact f16 products <= 14 * 2^-24 * S; the 22-bit forms of X and q 2 * 2^-23 * S/2; the dropped
// lo * lo terms 2^-22 * S/2; |q|^2 in fp32 and as two pieces 4 * 2^-24 * S; the fp32 roundings of the reference's own sqdist3 at
// d^2 ~ r^2 4 * 2^-24 * S at most: < 1.5e-6 * S in all.  eps = 1e-5 * max(S, 1).
__global__ void __launch_bounds__(VOX_THREADS) k_vox_ctab(const float* __restrict__ centres, int ncentres, float r, float r2,
                                                          float* __restrict__ hdr, unsigned short* __restrict__ ctab)
{
    __shared__ float smax[VOX_THREADS / WAVE];
    const int c = threadIdx.x;
    double x[3] = { 0.0, 0.0, 0.0 };
    if (c < ncentres)
        for (int a = 0; a < 3; a++) x[a] = (double)centres[3 * c + a];
    const double cc = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    float m = (float)sqrt(cc) * 1.000001f;
    for (int d = WAVE / 2; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    if ((c & (WAVE - 1)) == 0) smax[c / WAVE] = m;
    __syncthreads();
    if (c == 0) {
        float cmax = 0.f;
        for (int i = 0; i < VOX_THREADS / WAVE; i++) cmax = fmaxf(cmax, smax[i]);
        const float reach = (cmax + r) * 1.001f + 1e-3f;
        const float S = (cmax + reach) * (cmax + reach);
        // centre sets too large for the f16 pieces (|c|^2 - r^2 must stay far below 65504): eps = +inf sends EVERY lane to the fp32 test
        hdr[0] = reach * reach; hdr[1] = cmax <= 64.f ? VOX_EPS_SCALE * 1e-5f * fmaxf(S, 1.f) : __builtin_inff(); hdr[2] = cmax; hdr[3] = 0.f;
    }
    _Float16 v[VOX_K];
    for (int i = 0; i < VOX_K; i++) v[i] = (_Float16)0.f;
    v[9] = (_Float16)1.f; v[10] = (_Float16)0x1p-11f;
    if (c < ncentres) {
        for (int a = 0; a < 3; a++) {
            const float X = (float)(-2.0 * x[a]);                               // exact
            const _Float16 Xh = (_Float16)X;
            const _Float16 Xl = (_Float16)((X - (float)Xh) * 2048.f);
            v[a] = Xh;
            v[3 + a] = (_Float16)((float)Xh * 0x1p-11f);
            v[6 + a] = (_Float16)((float)Xl * 0x1p-11f);
        }
        const double e = cc - (double)r2;
        const _Float16 e0 = (_Float16)(float)e;
        const double r1 = e - (double)(float)e0;
        const _Float16 e1 = (_Float16)(float)(r1 * 2048.0);
        const double r2_ = r1 - (double)(float)e1 * (1.0 / 2048.0);
        const _Float16 e2 = (_Float16)(float)(r2_ * 4194304.0);
        v[11] = e0; v[12] = e1; v[13] = e2;
    } else v[11] = (_Float16)VOX_BIG;
    for (int i = 0; i < VOX_K; i++) ctab[c * VOX_K + i] = __builtin_bit_cast(unsigned short, v[i]);
}

extern "C" size_t buf_patch_voxelize_ws_bytes(int ncentres)
{
    if (ncentres <= 0) return 0;
    return 256 + sizeof(unsigned short) * (size_t)VOX_THREADS * VOX_K;
}

extern "C" int buf_patch_voxelize(const float* patches, const float* axis, int npatch, int npts, float des_r,
                                  const float* centres, int ncentres, int azi_n, const float* azi_cs, float voxel_r,
                                  int nsample, const float* mlp_w, const float* mlp_b, const float* bn_scale,
                                  const float* bn_shift, float* out_x, float* out_R, float* out_rand, float* out_patches,
                                  void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(npatch >= 0 && npts > 0 && npts <= VOX_MAXPTS, BUF_EINVAL, "buf_patch_voxelize: npts=%d (max %d)", npts, VOX_MAXPTS);
    BUF_REQUIRE(nsample <= VOX_MAXS, BUF_EINVAL, "buf_patch_voxelize: nsample=%d (max %d)", nsample, VOX_MAXS);
    BUF_REQUIRE(ncentres > 0 && ncentres <= VOX_THREADS && azi_n > 0 && nsample > 0, BUF_EINVAL,
                "buf_patch_voxelize: ncentres=%d (max %d)", ncentres, VOX_THREADS);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(patches && centres && azi_cs && mlp_w && mlp_b && bn_scale && bn_shift && out_x && out_R && out_rand,
                BUF_EINVAL, "buf_patch_voxelize: null argument");
    BUF_REQUIRE(ws && ws_bytes >= buf_patch_voxelize_ws_bytes(ncentres), BUF_EWORKSPACE,
                "buf_patch_voxelize: workspace of %zu bytes, need %zu", ws_bytes, buf_patch_voxelize_ws_bytes(ncentres));
    float* hdr = (float*)ws;
    unsigned short* tab = (unsigned short*)((char*)ws + 256);
    VoxMlp M;   // host copies of the 16x3 MLP (HOST pointers: tiny, passed by value to the kernels)
    for (int ch = 0; ch < VOX_CH; ch++) {
        for (int j = 0; j < 3; j++) M.w[ch][j] = bn_scale[ch] * mlp_w[3 * ch + j];
        M.b[ch] = bn_scale[ch] * mlp_b[ch] + bn_shift[ch];
    }
    k_vox_ctab<<<1, VOX_THREADS, 0, (hipStream_t)stream>>>(centres, ncentres, voxel_r, voxel_r * voxel_r, hdr, tab);
    k_patch_rotation<<<(npatch + 255) / 256, 256, 0, (hipStream_t)stream>>>(axis, npatch, out_R, out_rand);
    BUF_LAUNCH_CHECK();
    const size_t W = (size_t)((npts + 31) / 32);
    const size_t lds = sizeof(float4) * (size_t)npts + sizeof(unsigned) * W * VOX_THREADS + sizeof(_Float16) * VOX_K * 32 * W;
    static LdsGrant grant;
    if (lds > 48 * 1024)
        if (int rc = grant_dynamic_lds((const void*)k_patch_voxelize, lds, grant)) return rc;
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 12.0 * npatch * npts + 4.0 * npatch * VOX_CH * ncentres, BUF_TIMED_PATCH_VOXELIZE);
#ifdef VOX_STAMP
    long long* stamps = nullptr;
    BUF_CHECK_HIP(hipMalloc(&stamps, (size_t)npatch * 8 * sizeof(long long)));
    BUF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(vox_stamp_ptr), &stamps, sizeof(stamps)));
#endif
    k_patch_voxelize<<<npatch, VOX_THREADS, lds, (hipStream_t)stream>>>(patches, axis, npts, des_r, centres, ncentres, azi_n,
                                                                      azi_cs, voxel_r * voxel_r, nsample, M, hdr, tab, out_x,
                                                                      out_R, out_patches);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
#ifdef VOX_STAMP
    if (npatch >= 4096) {
        BUF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        long long* h = (long long*)malloc((size_t)npatch * 8 * sizeof(long long));
        BUF_CHECK_HIP(hipMemcpy(h, stamps, (size_t)npatch * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        static const char* name[4] = { "align + load (incl. thread-0 rotation)", "phase 1 (point -> ball hit masks)", "phase 2+3 (mask walk + MLP)", "output stores" };
        double d[4] = {};
        for (int b = npatch / 2; b < npatch; b++)
            for (int i = 0; i < 4; i++) d[i] += (double)(h[(size_t)b * 8 + i + 1] - h[(size_t)b * 8 + i]);
        for (int i = 0; i < 4; i++) fprintf(stderr, "  VOX_STAMP %-40s %8.0f cycles per patch\n", name[i], d[i] / (npatch - npatch / 2));
        free(h);
    }
    (void)hipFree(stamps);
#endif
    return BUF_OK;
}
