// A9 + A10 + the point MLP of A11, fused (models/patch_embedder.py:123-171,74-79;
// utils/common.py:431-498,501-525).
//
// Reference: axis_align -> normalize -> sphere_query (420 ball queries of 10 samples per patch,
// materialising [P,420,10,3]) -> var_to_invar -> Conv2d1x1(3->16)+BN+ReLU -> max over the 10 samples.
// Here: one workgroup per patch; the aligned, normalised 512-point patch sits in LDS (8 KB).
//   1. point-parallel: every point looks up the voxel balls that can contain it (a 16^3 lookup grid over the
//      unit ball, cell -> candidate centres, built once per call) and sets its bit in the hit mask of each
//      ball that does (exact d^2 test, LDS atomicOr) -- ~30 candidates (14 hits) per point instead of 420 tests.  The table is
//      compact (round 3): a 64-byte row per cell (24 candidates + the descriptor of its overflow run) = 256 KB and one packed
//      overflow array (~40 KB for the 3DMatch grid; round 2: rows of 896 bytes, 3.7 MB).  The phase was 61 k of the kernel's
//      78 k cycles per patch (-DVOX_STAMP) and is bound by the number of wavefront-level tests: it now runs in two passes
//      (rows for every point, overflow runs for the queued 30 % only; see phase 1 below), 42 k of 59 k;
//   2. centre-parallel: one lane per voxel centre walks its mask in index order (= pointnet2 ball_query's "first nsample
//      in index order") and runs each hit through the azimuth de-rotation, the 3->16 MLP, BN, ReLU and the running max,
//      all lanes slot by slot.
// HBM traffic: 6 KB in, 26.9 KB out per patch; [P,420,10,3] never exists.
#include "common.h"
#include <type_traits>

#define VOX_THREADS 448     // 7 wavefronts >= 420 centres
#define VOX_MAXPTS 1024
static_assert(VOX_MAXPTS <= 32 * 32, "the non-empty-word summary of a hit mask is one 32-bit register (voxelize phase 2+3)");
#define VOX_CH 16
#define VOX_MAXS 16           // max samples per voxel kept in the hit list
#define VOX_GRID 16           // lookup grid cells per axis
#ifndef VOX_PP
#define VOX_PP 3              // points per step of a lane group in phase 1a (2..5 measured within 4 %)
#endif
#define VOX_SENT 0xFFFFu      // end-of-list filler of a lookup row

// Lookup table: rows[VOX_CELLS + 1][32] unsigned short -- entries 0..23 the cell's first candidate centres (VOX_SENT beyond the
// list), entries 24, 25 one unsigned = start | count << 21 of the cell's run in the overflow array (candidates 24, 25, ...);
// row VOX_CELLS is empty (points outside the grid).  Overflow runs are claimed with an atomic counter (their order in the array
// is arbitrary, their content is not).
#define VOX_CELLS (VOX_GRID * VOX_GRID * VOX_GRID)
#define VOX_ROW 32
#define VOX_ROW_N 24
__host__ __device__ static inline size_t vox_overflow_capacity(int ncentres) { return (size_t)VOX_CELLS * (size_t)(ncentres > VOX_ROW_N ? ncentres - VOX_ROW_N : 0) + 64; }

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

__global__ void __launch_bounds__(VOX_THREADS) k_patch_voxelize(const float* __restrict__ patches, const float* __restrict__ axis,
                                                             int npts, float des_r, const float* __restrict__ centres,
                                                             int ncentres, int azi_n, const float* __restrict__ azi_cs,
                                                             float voxel_r2, int nsample, VoxMlp M,
                                                             const float* __restrict__ tab_hdr, const unsigned short* __restrict__ tab,
                                                             float* __restrict__ out_x, float* __restrict__ out_R,
                                                             float* __restrict__ out_rand, float* __restrict__ out_patches)
{
    // dynamic LDS, sized by the launch (36.9 KB + the 7 KB centre copy at 512 points -> three workgroups per CU; the hit lists
    // never leave the registers.  Reading the centres from L1/L2 instead would admit a fourth workgroup but measured slower):
    extern __shared__ float4 pts[];                              // [npts] aligned, normalised patch
    const int W = (npts + 31) >> 5;                              // mask words per centre
    unsigned* mask = reinterpret_cast<unsigned*>(pts + npts);    // [W][VOX_THREADS] hit bits
    __shared__ float4 cen[VOX_THREADS];
    __shared__ float Rs[9];
    __shared__ uint2 longq[VOX_MAXPTS];                          // phase 1b: (point, overflow run) of the points with long candidate lists
    __shared__ unsigned nlong;
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* src = patches + (size_t)p * npts * 3;
    VOX_STAMP_AT(0)
    if (tid == 0) {
        float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        float rx = 1.f, ry = 0.f, rz = 0.f;                      // KITTI/ETH: rand_axis = e_x, R = I (:143-147)
        if (axis) {
            // RodsRotatFormula(z_axis, e_z) (utils/common.py:501-525), returned transposed
            float ax = axis[3 * (size_t)p], ay = axis[3 * (size_t)p + 1], az = axis[3 * (size_t)p + 2];
            float cx = ay, cy = -ax, cz = 0.f;                   // a x e_z
            float na = sqrtf(ax * ax + ay * ay + az * az);
            float cs = az / (fmaxf(na, 1e-8f) * 1.0f);           // cosine_similarity
            cs = fminf(fmaxf(cs, -1.f), 1.f);
            float th = acosf(cs);
            float nc = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);   // F.normalize
            cx /= nc; cy /= nc; cz /= nc;
            rx = cx; ry = cy; rz = cz;                           // rand_axis = normalize(z_axis x e_z) (:138-141)
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
        for (int i = 0; i < 9; i++) { Rs[i] = R[i]; out_R[9 * (size_t)p + i] = R[i]; }
        out_rand[3 * (size_t)p] = rx; out_rand[3 * (size_t)p + 1] = ry; out_rand[3 * (size_t)p + 2] = rz;
    }
    __syncthreads();
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
    {
        float4 cc = make_float4(1e30f, 1e30f, 1e30f, 0.f);
        if (active) {
            cc = make_float4(centres[3 * c], centres[3 * c + 1], centres[3 * c + 2], 0.f);
            int az = c % azi_n;                                  // ordering rad -> ele -> azi (utils/common.py:422-428)
            ca = azi_cs[2 * az]; sa = azi_cs[2 * az + 1];        // cos/sin of -az * 2pi/azi_n (:485-491)
        }
        cen[tid] = cc;
    }
    for (int i = tid; i < W * VOX_THREADS; i += VOX_THREADS) mask[i] = 0u;
    if (tid == 0) nlong = 0u;
    __syncthreads();
    // Phase 1: 8 lanes share a point and split its candidate list.  The phase is bound by the NUMBER of wavefront-level tests
    // (~16 instructions each: -DVOX_STAMP experiments -- neither the table's latency nor the LDS atomics moved it), and lists are
    // uneven (mean 30 candidates, 30 % of the points over 24, up to 140 next to the keypoint), so it runs in two passes:
    //   1a. every point: the 24 candidates of its row, 3 per lane (VOX_PP points per step: 4 x VOX_PP row loads in flight per
    //       lane); a point whose cell has an overflow run is queued in LDS;
    //   1b. queued points only: 64 overflow candidates per round, 8 per lane, two points per step.
    // Round 2 walked the long lists inside 1a, 32 candidates per round, whenever ANY of a wavefront's 8 points had one (94 % of
    // the steps): 1 260 wavefront-tests per patch against 380 now.  The d^2 test is the one a full scan would do (same operand
    // order), the lookup grid only removes centres that cannot pass it.
    {
        const float lo = tab_hdr[0], inv_h = tab_hdr[1];
        const int sub = tid & 7;
        const unsigned short* __restrict__ ovf = tab + (size_t)(VOX_CELLS + 1) * VOX_ROW;
        // N candidates of one point: all centre reads first (entries past the list read centre 0 and are masked), then the tests
        auto batch = [&](auto nc, const unsigned short* cj, const float4& q, unsigned* mrow, unsigned bit) __attribute__((always_inline)) {
            constexpr int N = decltype(nc)::value;
            float4 cc[N];
#pragma unroll
            for (int i = 0; i < N; i++) cc[i] = cen[cj[i] == VOX_SENT ? 0 : cj[i]];
#pragma unroll
            for (int i = 0; i < N; i++)
                // (a branch per test: unconditional atomics with a zero operand -- straight-line code -- measured 55 % SLOWER: the
                // LDS atomic of a full wavefront is what costs, not the branch around it)
                if (cj[i] != VOX_SENT && sqdist3(cc[i].x, cc[i].y, cc[i].z, q.x, q.y, q.z) < voxel_r2) atomicOr(&mrow[cj[i]], bit);
        };
        for (int k0 = tid >> 3; k0 < npts; k0 += VOX_PP * (VOX_THREADS / 8)) {
            float4 q[VOX_PP];
            const unsigned short* row[VOX_PP];
            unsigned short e[VOX_PP][3];
            unsigned desc[VOX_PP];
#pragma unroll
            for (int u = 0; u < VOX_PP; u++) {
                const int k = k0 + u * (VOX_THREADS / 8);
                q[u] = pts[k < npts ? k : npts - 1];
                const int ix = (int)floorf((q[u].x - lo) * inv_h), iy = (int)floorf((q[u].y - lo) * inv_h),
                          iz = (int)floorf((q[u].z - lo) * inv_h);
                const bool ok = k < npts && ix >= 0 && iy >= 0 && iz >= 0 && ix < VOX_GRID && iy < VOX_GRID && iz < VOX_GRID;
                row[u] = tab + (size_t)(ok ? (ix * VOX_GRID + iy) * VOX_GRID + iz : VOX_CELLS) * VOX_ROW;   // else: the empty row
            }
#pragma unroll
            for (int u = 0; u < VOX_PP; u++) {
#pragma unroll
                for (int v = 0; v < 3; v++) e[u][v] = row[u][sub + 8 * v];
                desc[u] = *reinterpret_cast<const unsigned*>(row[u] + VOX_ROW_N);
            }
#pragma unroll
            for (int u = 0; u < VOX_PP; u++) {
                const int k = k0 + u * (VOX_THREADS / 8);
                batch(std::integral_constant<int, 3>{}, e[u], q[u], mask + (k >> 5) * VOX_THREADS, 1u << (k & 31));
                if (sub == 0 && (desc[u] >> 21) != 0u) {
                    const unsigned slot = atomicAdd(&nlong, 1u);
                    longq[slot] = make_uint2((unsigned)k, desc[u]);
                }
            }
        }
        __syncthreads();
        const int nq = (int)nlong;
        for (int i0 = tid >> 3; i0 < nq; i0 += 2 * (VOX_THREADS / 8)) {
            uint2 it[2];
            float4 q[2];
            unsigned short f[2][8];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * (VOX_THREADS / 8);
                it[u] = i < nq ? longq[i] : make_uint2(0u, 0u);           // (run of length 0: nothing to test)
                q[u] = pts[it[u].x];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int nov = (int)(it[u].y >> 21);
                const unsigned short* run = ovf + (it[u].y & 0x1FFFFFu);
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const int j = sub + 8 * v;
                    const unsigned short x = *(j < nov ? run + j : tab);
                    f[u][v] = j < nov ? x : (unsigned short)VOX_SENT;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int k = (int)it[u].x;
                const unsigned bit = 1u << (k & 31);
                unsigned* mrow = mask + (k >> 5) * VOX_THREADS;
                batch(std::integral_constant<int, 8>{}, f[u], q[u], mrow, bit);
                const int nov = (int)(it[u].y >> 21);
                const unsigned short* run = ovf + (it[u].y & 0x1FFFFFu);
                for (int j0 = 64; j0 < nov; j0 += 32) {                    // beyond 24 + 64 candidates: 3 % of the points
                    unsigned short g[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int j = j0 + sub + 8 * v;
                        const unsigned short x = run[min(j, nov - 1)];
                        g[v] = j < nov ? x : (unsigned short)VOX_SENT;
                    }
                    batch(std::integral_constant<int, 4>{}, g, q[u], mrow, bit);
                }
            }
        }
    }
    __syncthreads();
    VOX_STAMP_AT(2)
    // Phases 2 + 3: every centre lane walks its hit mask in index order (= ball_query's "first nsample in index order") and
    // feeds each hit straight through de-rotation -> 3->16 MLP -> BN -> ReLU -> running max; all lanes advance slot by slot.
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
    for (int sidx = 0; sidx < nsample && __any(bits != 0u || nz != 0u); sidx++) {
        if (bits == 0u && nz != 0u) {
            wcur = __ffs(nz) - 1;
            nz &= nz - 1u;
            bits = mask[wcur * VOX_THREADS + tid];
        }
        if (bits != 0u) {
            const int k = wcur * 32 + __ffs(bits) - 1;
            bits &= bits - 1u;
            cnt++;
            if (k != 0) {                                        // utils/common.py:447-449: a hit on point 0 is zeroed
                const float4 q = pts[k];
                const float nx = q.x * ca - q.y * sa, ny = q.x * sa + q.y * ca, nz_ = q.z;
#pragma unroll
                for (int ch = 0; ch < VOX_CH; ch++) {
                    const float h = fmaf(M.w[ch][2], nz_, fmaf(M.w[ch][1], ny, fmaf(M.w[ch][0], nx, M.b[ch])));
                    acc[ch] = fmaxf(fmaxf(acc[ch], h), 0.f);     // v_max3: ReLU and the running max
                }
            } else zero_slot = true;
        }
    }
    VOX_STAMP_AT(3)
    if (active) {
        bool padded = cnt < nsample || zero_slot;                // zeroed slots go through the MLP as the origin
#pragma unroll
        for (int ch = 0; ch < VOX_CH; ch++) {
            float v = acc[ch];
            if (padded) v = fmaxf(v, fmaxf(M.b[ch], 0.f));
            out_x[((size_t)p * VOX_CH + ch) * ncentres + c] = v;
        }
    }
    VOX_STAMP_AT(4)
}

// Lookup grid over [-L, L]^3, L = max |centre coordinate| + r: cell -> the centres whose ball can reach the cell
// (box-to-centre distance <= r plus a rounding margin), in centre order: the first 24 in the cell's row, the rest in a run of
// the overflow array claimed from *counter (zeroed by the caller).  One wavefront per cell; block VOX_CELLS writes the empty row.
__global__ void __launch_bounds__(WAVE) k_vox_table(const float* __restrict__ centres, int ncentres, float r,
                                                    float* __restrict__ hdr, unsigned* __restrict__ counter,
                                                    unsigned short* __restrict__ rows, unsigned short* __restrict__ ovf)
{
    const int cell = blockIdx.x, lane = threadIdx.x;
    unsigned short* row = rows + (size_t)cell * VOX_ROW;
    if (cell == VOX_CELLS) {
        if (lane < VOX_ROW) row[lane] = lane < VOX_ROW_N ? (unsigned short)VOX_SENT : (unsigned short)0;
        return;
    }
    float L = 0.f;
    for (int i = lane; i < 3 * ncentres; i += WAVE) L = fmaxf(L, fabsf(centres[i]));
    for (int d = WAVE / 2; d > 0; d >>= 1) L = fmaxf(L, __shfl_xor(L, d, WAVE));
    L += r;
    const float h = 2.f * L / (float)VOX_GRID;
    if (cell == 0 && lane == 0) { hdr[0] = -L; hdr[1] = 1.f / h; hdr[2] = h; hdr[3] = 0.f; }
    const int ix = cell / (VOX_GRID * VOX_GRID), iy = (cell / VOX_GRID) % VOX_GRID, iz = cell % VOX_GRID;
    const float bx = -L + ix * h, by = -L + iy * h, bz = -L + iz * h;
    const float rr = r * 1.001f + 1e-3f * h;
    auto reaches = [&](int c) {
        if (c >= ncentres) return false;
        const float cx = centres[3 * c], cy = centres[3 * c + 1], cz = centres[3 * c + 2];
        const float dx = fmaxf(fmaxf(bx - cx, cx - (bx + h)), 0.f);
        const float dy = fmaxf(fmaxf(by - cy, cy - (by + h)), 0.f);
        const float dz = fmaxf(fmaxf(bz - cz, cz - (bz + h)), 0.f);
        return dx * dx + dy * dy + dz * dz <= rr * rr;
    };
    int total = 0;
    for (int c0 = 0; c0 < ncentres; c0 += WAVE) total += __popcll(__ballot(reaches(c0 + lane)));
    const int nov = total > VOX_ROW_N ? total - VOX_ROW_N : 0;
    unsigned start = 0;
    if (nov > 0) {
        if (lane == 0) start = atomicAdd(counter, (unsigned)nov);
        start = __shfl(start, 0, WAVE);
    }
    int n = 0;
    for (int c0 = 0; c0 < ncentres; c0 += WAVE) {
        const int c = c0 + lane;
        const bool in = reaches(c);
        const unsigned long long m = __ballot(in);
        if (in) {
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < VOX_ROW_N) row[pos] = (unsigned short)c;
            else ovf[start + (unsigned)(pos - VOX_ROW_N)] = (unsigned short)c;
        }
        n += __popcll(m);
    }
    for (int i = n + lane; i < VOX_ROW_N; i += WAVE) row[i] = (unsigned short)VOX_SENT;
    if (lane == 0) *reinterpret_cast<unsigned*>(row + VOX_ROW_N) = start | ((unsigned)nov << 21);
}

extern "C" size_t buf_patch_voxelize_ws_bytes(int ncentres)
{
    if (ncentres <= 0) return 0;
    return 256 + sizeof(unsigned short) * ((size_t)(VOX_CELLS + 1) * VOX_ROW + vox_overflow_capacity(ncentres));
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
    BUF_REQUIRE(vox_overflow_capacity(ncentres) < (1u << 21), BUF_EINVAL, "buf_patch_voxelize: ncentres=%d (overflow runs are addressed with 21 bits)", ncentres);
    float* hdr = (float*)ws;
    unsigned* counter = (unsigned*)((char*)ws + 64);
    unsigned short* tab = (unsigned short*)((char*)ws + 256);
    BUF_CHECK_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), (hipStream_t)stream));
    k_vox_table<<<VOX_CELLS + 1, WAVE, 0, (hipStream_t)stream>>>(centres, ncentres, voxel_r, hdr, counter, tab, tab + (size_t)(VOX_CELLS + 1) * VOX_ROW);
    BUF_LAUNCH_CHECK();
    VoxMlp M;   // host copies of the 16x3 MLP (HOST pointers: tiny, passed by value to the kernel)
    for (int ch = 0; ch < VOX_CH; ch++) {
        for (int j = 0; j < 3; j++) M.w[ch][j] = bn_scale[ch] * mlp_w[3 * ch + j];
        M.b[ch] = bn_scale[ch] * mlp_b[ch] + bn_shift[ch];
    }
    const size_t lds = sizeof(float4) * (size_t)npts + sizeof(unsigned) * (size_t)((npts + 31) / 32) * VOX_THREADS;
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
                                                                      out_R, out_rand, out_patches);
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
