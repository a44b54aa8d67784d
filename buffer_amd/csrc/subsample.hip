// A1 -- batched grid (voxel-barycentre) subsampling
// (replaces cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106,109-211).
//
// The reference accumulates points into an unordered_map keyed by the voxel index and emits the
// barycentres in hash-map order.  Here: a table of key buckets per batch element, counting sort of the
// points by bucket, every point ranks itself by (key, input index) inside its bucket (so each voxel's run
// is contiguous and in INPUT ORDER and the fp32 sum is the reference's, bit for bit), one lane per voxel
// run, rows emitted in ascending voxel-key order.
#include "common.h"

// The table is indexed by BUCKETS of B consecutive voxel keys (B = 1 whenever the bounding box fits `max_cells`,
// which is the normal case: bucket == voxel).  Buckets are contiguous key ranges, so "ascending (bucket, key)" is
// "ascending key"; inside a bucket every point ranks itself by (key, input index).  Keys use the reference's own
// 64-bit wrapping arithmetic, so even its (size_t)floor(negative) corner case lands where the reference puts it
// (such keys are huge and are parked in the element's last bucket).
struct VoxGrid {
    float o[3];                    // originCorner (grid_subsampling.cpp:27)
    float dl;
    unsigned long long NX, NY;     // sampleNX, sampleNY (:30-31)
    double cells;                  // NX*NY*NZ of the bounding box (budgeting only)
    long long nbuckets, table_off; // buckets of this element / first table slot (concatenated)
    int lo, hi;                    // point range
};

struct VoxStatus { int error; unsigned pad; unsigned long long B; };

// Bucket width and table offsets of every element (round 6: one workgroup of 1024 threads; rounds 1-5: a kernel of ONE thread that walked
// the nb descriptors up to 200 times through global memory, 35 us per call at 64 elements -- a quarter of the whole operator.  Run by the last
// workgroup of k_vox_bbox instead, it needed a device-scope fence per element's workgroup: each one writes an XCD's whole L2 back).  Smallest power-of-two bucket width B with sum_b (floor(cells_b / B) + 2) <= max_cells.
__device__ void vox_offsets(VoxGrid* __restrict__ grids, int nb, long long max_cells, VoxStatus* __restrict__ st)
{
    __shared__ double red[16];
    __shared__ long long scan_s[16];
    __shared__ double s_B;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int per = (nb + 1023) / 1024;
    const int b0 = tid * per, b1 = min(b0 + per, nb);
    double Bd = 1.0;
    for (int it = 0; it < 200; it++) {
        double need = 0.0;
        for (int b = b0; b < b1; b++) need += grids[b].hi > grids[b].lo ? floor(grids[b].cells / Bd) + 2.0 : 1.0;
        for (int d = WAVE / 2; d > 0; d >>= 1) need += __shfl_xor(need, d, WAVE);
        __syncthreads();
        if (lane == 0) red[w] = need;
        __syncthreads();
        double tot = 0.0;
        for (int i = 0; i < 16; i++) tot += red[i];
        if (tot <= (double)max_cells) break;                  // (uniform: every thread sees the same sum)
        Bd *= 2.0;
    }
    // exclusive prefix of the bucket counts in element order: thread chunks, then the 16 wave sums
    long long mine = 0;
    for (int b = b0; b < b1; b++) mine += grids[b].hi > grids[b].lo ? (long long)(floor(grids[b].cells / Bd) + 2.0) : 1;
    long long inc = mine;
    for (int d = 1; d < WAVE; d <<= 1) { const long long t = __shfl_up(inc, d, WAVE); if (lane >= d) inc += t; }
    __syncthreads();
    if (lane == WAVE - 1) scan_s[w] = inc;
    __syncthreads();
    long long run = inc - mine, total = 0;
    for (int i = 0; i < 16; i++) { if (i < w) run += scan_s[i]; total += scan_s[i]; }
    for (int b = b0; b < b1; b++) {
        const long long nbk = grids[b].hi > grids[b].lo ? (long long)(floor(grids[b].cells / Bd) + 2.0) : 1;
        grids[b].nbuckets = nbk;
        grids[b].table_off = run;
        run += nbk;
    }
    if (tid == 0) { st->error = total > max_cells ? 1 : 0; st->B = (unsigned long long)Bd; }
    (void)s_B;
}

__global__ void __launch_bounds__(1024) k_vox_offsets(VoxGrid* __restrict__ grids, int nb, long long max_cells, VoxStatus* __restrict__ st)
{
    vox_offsets(grids, nb, max_cells, st);
}

__global__ void __launch_bounds__(1024) k_vox_bbox(const float* __restrict__ pts, const int* __restrict__ off,
                                                 VoxGrid* __restrict__ grids, float dl)
{
    int b = blockIdx.x;
    int lo = off[b], hi = off[b + 1];
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int i = lo + threadIdx.x; i < hi; i += 1024) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v = pts[3 * (size_t)i + c];
            mn[c] = v < mn[c] ? v : mn[c];
            mx[c] = v > mx[c] ? v : mx[c];
        }
    }
    __shared__ float smn[3][16], smx[3][16];
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], z = mx[c];
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            a = fminf(a, __shfl_xor(a, d, WAVE));
            z = fmaxf(z, __shfl_xor(z, d, WAVE));
        }
        if (lane == 0) { smn[c][w] = a; smx[c][w] = z; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        VoxGrid g;
        g.dl = dl;
        g.lo = lo; g.hi = hi;
        float inv = __fdiv_rn(1.0f, dl);                       // (1/sampleDl), fp32
        double N[3];
        for (int c = 0; c < 3; c++) {
            float a = smn[c][0], z = smx[c][0];
            for (int i = 1; i < 16; i++) { a = fminf(a, smn[c][i]); z = fmaxf(z, smx[c][i]); }
            if (hi <= lo) { a = 0.f; z = 0.f; }
            g.o[c] = __fmul_rn(floorf(__fmul_rn(a, inv)), dl);  // floor(min * (1/dl)) * dl
            N[c] = (double)floorf(__fdiv_rn(__fsub_rn(z, g.o[c]), dl)) + 1.0;
        }
        g.NX = (unsigned long long)(long long)N[0];
        g.NY = (unsigned long long)(long long)N[1];
        g.cells = fmax(N[0], 1.0) * fmax(N[1], 1.0) * fmax(N[2], 1.0);
        g.nbuckets = 1; g.table_off = 0;
        grids[b] = g;
    }
}

__global__ void __launch_bounds__(256) k_vox_count(const float* __restrict__ pts, int n, const int* __restrict__ off, int nb,
                                                 const VoxGrid* __restrict__ grids, const VoxStatus* __restrict__ st,
                                                 int* __restrict__ table, int* __restrict__ cell_of,
                                                 unsigned long long* __restrict__ keys)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || st->error) return;
    int b = find_elem(off, nb, i);
    VoxGrid g = grids[b];
    // (size_t)floor((p - origin) / dl), fp32, 64-bit wrapping like the reference (grid_subsampling.cpp:53-56)
    unsigned long long iX = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i], g.o[0]), g.dl));
    unsigned long long iY = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + 1], g.o[1]), g.dl));
    unsigned long long iZ = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * (size_t)i + 2], g.o[2]), g.dl));
    unsigned long long key = iX + g.NX * iY + g.NX * g.NY * iZ;         // mapIdx
    unsigned long long bk = key / st->B;
    if (bk > (unsigned long long)(g.nbuckets - 1)) bk = (unsigned long long)(g.nbuckets - 1);
    int c = (int)(g.table_off + (long long)bk);
    keys[i] = key;
    cell_of[i] = c;
    atomicAdd(&table[c], 1);
}

// rank inside a bucket by (voxel key, input index): runs of equal key end up contiguous and in INPUT ORDER
__global__ void __launch_bounds__(256) k_vox_rank(const int* __restrict__ cell_of, const unsigned long long* __restrict__ keys,
                                                const int* __restrict__ table, const float4* __restrict__ sorted_in, int n,
                                                const VoxStatus* __restrict__ st, float4* __restrict__ sorted_out,
                                                unsigned long long* __restrict__ key_sorted, int* __restrict__ cell_sorted, int* __restrict__ head)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    if (st->error) { head[p] = 0; return; }
    float4 me = sorted_in[p];
    int i = __float_as_int(me.w);
    int c = cell_of[i];
    unsigned long long k = keys[i];
    int s = c == 0 ? 0 : table[c - 1], e = table[c];
    int rank = 0, same_before = 0;
    for (int t = s; t < e; t++) {
        int j = __float_as_int(sorted_in[t].w);
        unsigned long long kj = keys[j];
        rank += (kj < k || (kj == k && j < i)) ? 1 : 0;
        same_before += (kj == k && j < i) ? 1 : 0;
    }
    sorted_out[s + rank] = me;
    key_sorted[s + rank] = k;
    cell_sorted[s + rank] = c;
    head[s + rank] = same_before == 0 ? 1 : 0;                  // the first point of its voxel in input order = the head of the voxel's run (round 6: was k_vox_heads)
}

__device__ __forceinline__ bool vox_is_head(const unsigned long long* __restrict__ key_sorted, const int* __restrict__ cell_sorted, int p)
{
    return p == 0 || cell_sorted[p] != cell_sorted[p - 1] || key_sorted[p] != key_sorted[p - 1];
}

__global__ void __launch_bounds__(256) k_vox_emit(const float4* __restrict__ sorted, const unsigned long long* __restrict__ key_sorted,
                                                const int* __restrict__ cell_sorted, const int* __restrict__ rowidx,
                                                int n, const VoxStatus* __restrict__ st, float* __restrict__ out,
                                                const float* __restrict__ feats, int fdim, float* __restrict__ out_feats,
                                                const int* __restrict__ off, int nb, const int* __restrict__ total_dev, int* __restrict__ counts)
{
    if (blockIdx.x == gridDim.x - 1) {                           // rows per element from the exclusive head scan; counts[nb] = status, [nb + 1] = total (was k_vox_counts)
        const int tot = *total_dev;
        for (int b = threadIdx.x; b <= nb; b += 256) {
            if (b == nb) { counts[nb] = st->error; counts[nb + 1] = tot; continue; }
            const int lo = off[b], hi = off[b + 1];
            counts[b] = (hi < n ? rowidx[hi] : tot) - (lo < n ? rowidx[lo] : tot);
        }
    }
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || st->error) return;
    if (!vox_is_head(key_sorted, cell_sorted, p)) return;
    const unsigned long long k = key_sorted[p];
    const int c = cell_sorted[p];
    int e = p + 1;
    while (e < n && cell_sorted[e] == c && key_sorted[e] == k) e++;
    float sx = 0.f, sy = 0.f, sz = 0.f;      // SampledData.point += p, in input order (grid_subsampling.h:95-100)
    for (int t = p; t < e; t++) {
        float4 q = sorted[t];
        sx = __fadd_rn(sx, q.x); sy = __fadd_rn(sy, q.y); sz = __fadd_rn(sz, q.z);
    }
    float w = (float)(1.0 / (double)(e - p));          // point * (1.0 / count): double -> float (:87)
    int r = rowidx[p];
    out[3 * (size_t)r] = __fmul_rn(sx, w);
    out[3 * (size_t)r + 1] = __fmul_rn(sy, w);
    out[3 * (size_t)r + 2] = __fmul_rn(sz, w);
    if (feats) {                             // features summed in input order, then f / (float)count (:90-96)
        float cf = (float)(e - p);
        for (int d = 0; d < fdim; d++) {
            float acc = 0.f;
            for (int t = p; t < e; t++) acc = __fadd_rn(acc, feats[(size_t)__float_as_int(sorted[t].w) * fdim + d]);
            out_feats[(size_t)r * fdim + d] = __fdiv_rn(acc, cf);
        }
    }
}

// stand-alone form for callers with their own emit kernel (csrc/preprocess.hip)
// per-element row counts from the exclusive head scan; counts[nb] = status word
__global__ void k_vox_counts(const int* __restrict__ rowidx, const int* __restrict__ off, int nb, int n, int total,
                             const int* __restrict__ total_dev, VoxStatus* __restrict__ st, int* __restrict__ counts)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    int tot = *total_dev;
    if (b == nb) { counts[nb] = st->error; counts[nb + 1] = tot; return; }
    int lo = off[b], hi = off[b + 1];
    int rlo = lo < n ? rowidx[lo] : tot;
    int rhi = hi < n ? rowidx[hi] : tot;
    counts[b] = rhi - rlo;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the whole operator for ONE batch element in ONE workgroup, sorted in LDS (elements up to VOXF_MAX_N points; a step's elements
// are fragments of 9-15 k points).  The global-table path above is a counting sort through device-scope atomics: two atomic passes of one
// atomicAdd per point behind eight non-coherent L2s (~20 G atomics/s), three scan launches over the table, three over the head flags --
// 13 launches, 117 us per call at 64 elements (profiles/r06_a1_breakdown.txt).  Here a workgroup holds its element's points in registers
// (16 per lane), computes the bounding box, counts and scatters with LDS atomics on a table of 16-bit counters, keeps the bucket-sorted
// points as ONE 32-bit word each -- (key mod bucket width) << 16 | input index: inside a bucket, word order IS (key, input index) order --
// ranks them inside their buckets in place, and emits the barycentres in ascending key order, rows dealt densely to lanes: the same keys,
// ranks and summation order as above, hence the same rows bit for bit (the bucket width only has to make the element's table fit; rows do
// not depend on it).  Elements whose keys do not pack (bounding boxes over ~2^29 voxels: outliers kilometres away; the reference's wrapped
// keys, which land in the last bucket) take the same steps through global arrays in the same launch (voxf_element_global: slow, exact,
// rare).  k_vox_concat stacks the elements' rows.  One element of 13 k points: 49 us at 3.6 points per voxel, 60 us at 15 (rank 15 / 30 us:
// one compare per pair of bucket mates, on ONE CU; tools/a1_phases.py); a call of 64 elements 221 -> 101 us and 276 -> 110 us host clock.
// Worst case, both forms: the rank is quadratic in a bucket's population -- 16 384 points in ONE voxel (a voxel size far beyond the cloud's
// extent) take 21 ms here and 8 ms on the global-table path (tools/a1_worst.py); a voxel of 2500 points costs 0.4 ms.
#define VOXF_THREADS 1024
#define VOXF_WAVES (VOXF_THREADS / WAVE)
#define VOXF_TABLE 24576          // buckets per element, LDS form: 16-bit counters, two per LDS word (populations and positions are < 2^16)
#define VOXF_TABLE_WORDS (VOXF_TABLE / 2)
#define VOXF_MAX_N 16384          // points per element (16 per lane in registers: 128 registers per lane is all a 1024-lane workgroup gets)
#define VOXF_PT (VOXF_MAX_N / VOXF_THREADS)
#define VOXF_TABLE_G 32768        // buckets of the global-array form (its table takes the LDS of all three arrays of the LDS form)
#define VOXF_LDS_BYTES (sizeof(int) * (VOXF_TABLE_WORDS + 1 + VOXF_MAX_N) + sizeof(unsigned short) * VOXF_MAX_N + 12)
#define VOXF_U 4                  // points per lane whose loads are issued together

// inclusive prefix sum over the 64 lanes through DPP row shifts / broadcasts (__shfl_up is an LDS permute: ~120 cycles a step, six dependent
// steps per 64 entries were most of the scans' 4 us)
__device__ __forceinline__ int wave_scan_incl_i32(int v)
{
#define DPP_ADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, true)
    DPP_ADD(0x111, 0xf);   // row_shr:1
    DPP_ADD(0x112, 0xf);   // row_shr:2
    DPP_ADD(0x114, 0xf);   // row_shr:4
    DPP_ADD(0x118, 0xf);   // row_shr:8   -> inclusive sums inside each row of 16
    DPP_ADD(0x142, 0xa);   // row_bcast:15: row 0's total into row 1, row 2's into row 3
    DPP_ADD(0x143, 0xc);   // row_bcast:31: the total of rows 0-1 into rows 2-3
#undef DPP_ADD
    return v;
}

// exclusive scan of cnt ints (LDS or global) by the whole workgroup, 64 consecutive entries per wavefront step; returns the total
__device__ int voxf_scan(int* __restrict__ a, int cnt, int* __restrict__ wsum)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int seg = ((cnt + VOXF_WAVES - 1) / VOXF_WAVES + WAVE - 1) / WAVE * WAVE;       // a wavefront's contiguous segment
    const int s0 = w * seg, s1 = min(s0 + seg, cnt);
    int sum = 0;
    for (int i = s0 + lane; i < s1; i += WAVE) sum += a[i];
    for (int d = WAVE / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d, WAVE);
    __syncthreads();                                                                      // (wsum may still be read by a previous call)
    if (lane == 0) wsum[w] = sum;
    __syncthreads();
    int carry = 0, total = 0;
    for (int i = 0; i < VOXF_WAVES; i++) { if (i < w) carry += wsum[i]; total += wsum[i]; }
    for (int i0 = s0; i0 < s1; i0 += WAVE) {
        const int i = i0 + lane;
        const int v = i < s1 ? a[i] : 0;
        const int inc = wave_scan_incl_i32(v);
        if (i < s1) a[i] = carry + inc - v;
        carry += __builtin_amdgcn_readlane(inc, WAVE - 1);
    }
    __syncthreads();
    return total;
}

// the same over 2 * cnt 16-bit counters packed two per word (all partial sums < 2^16)
__device__ void voxf_scan16(int* __restrict__ a, int cnt, int* __restrict__ wsum)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int seg = ((cnt + VOXF_WAVES - 1) / VOXF_WAVES + WAVE - 1) / WAVE * WAVE;
    const int s0 = w * seg, s1 = min(s0 + seg, cnt);
    int sum = 0;
    for (int i = s0 + lane; i < s1; i += WAVE) { const unsigned int v = (unsigned int)a[i]; sum += (int)(v & 0xffffu) + (int)(v >> 16); }
    for (int d = WAVE / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d, WAVE);
    __syncthreads();
    if (lane == 0) wsum[w] = sum;
    __syncthreads();
    int carry = 0;
    for (int i = 0; i < w; i++) carry += wsum[i];
    for (int i0 = s0; i0 < s1; i0 += WAVE) {
        const int i = i0 + lane;
        const unsigned int v = i < s1 ? (unsigned int)a[i] : 0u;
        const int c0 = (int)(v & 0xffffu), c1 = (int)(v >> 16);
        const int inc = wave_scan_incl_i32(c0 + c1);
        const int ex = carry + inc - c0 - c1;
        if (i < s1) a[i] = (int)((unsigned int)ex | (unsigned int)(ex + c0) << 16);
        carry += __builtin_amdgcn_readlane(inc, WAVE - 1);
    }
    __syncthreads();
}

// (size_t)floor((p - origin) / dl), fp32, 64-bit wrapping like the reference (grid_subsampling.cpp:53-56)
__device__ __forceinline__ unsigned long long vox_key(const VoxGrid& g, unsigned long long NXY, float x, float y, float z)
{
    unsigned long long iX = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(x, g.o[0]), g.dl));
    unsigned long long iY = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(y, g.o[1]), g.dl));
    unsigned long long iZ = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(z, g.o[2]), g.dl));
    return iX + g.NX * iY + NXY * iZ;         // mapIdx
}

// The same key when the three indices are small non-negative integers and the bounding box has < 2^31 voxels (every point of an ordinary
// element): the float -> 64-bit conversions and 64-bit products of vox_key are ~100 of its ~150 instructions, and one workgroup computes all
// keys of its element on ONE CU.  A lane whose indices are out of that range (the reference's wrapped keys) takes vox_key.
__device__ __forceinline__ unsigned long long vox_key_small(const VoxGrid& g, unsigned long long NXY, bool small_box, float x, float y, float z)
{
    const float fx = floorf(__fdiv_rn(__fsub_rn(x, g.o[0]), g.dl)), fy = floorf(__fdiv_rn(__fsub_rn(y, g.o[1]), g.dl)),
                fz = floorf(__fdiv_rn(__fsub_rn(z, g.o[2]), g.dl));
    // (small_box: NX, NX * NY < 2^31; the indices are integers below 2^23, every term of the sum is exact in fp64: the 64-bit key is below 2^32)
    if (small_box && fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < 8388608.f && fy < 8388608.f && fz < 8388608.f &&
        (double)fx + (double)g.NX * (double)fy + (double)NXY * (double)fz < 4294967296.0)
        return (unsigned long long)((unsigned int)fx + (unsigned int)g.NX * (unsigned int)fy + (unsigned int)NXY * (unsigned int)fz);
    return vox_key(g, NXY, x, y, z);
}

struct VoxfArrays {                 // global scratch, indexed by point (the arrays of the global-table path under other names)
    unsigned long long* keys; int* slot; unsigned long long* bkey; int* bidx; float4* sorted; int* runlen; int* rowidx;
    float* out_tmp; const float* feats; int fdim; float* feat_tmp;
};

// one run of equal keys -> one row; the run's points in input order, VOXF_U gathers in flight
template <typename IDX>
__device__ __forceinline__ void voxf_emit_run(const float* __restrict__ P, const IDX& idx_at, int p, int len, size_t r, const VoxfArrays& A, int lo)
{
    float sx = 0.f, sy = 0.f, sz = 0.f;      // SampledData.point += p, in input order (grid_subsampling.h:95-100)
    for (int t0 = 0; t0 < len; t0 += VOXF_U) {
        float x[VOXF_U], y[VOXF_U], z[VOXF_U];
#pragma unroll
        for (int u = 0; u < VOXF_U; u++) {
            const size_t q = 3 * (size_t)idx_at(p + min(t0 + u, len - 1));
            x[u] = P[q]; y[u] = P[q + 1]; z[u] = P[q + 2];
        }
#pragma unroll
        for (int u = 0; u < VOXF_U; u++)
            if (t0 + u < len) { sx = __fadd_rn(sx, x[u]); sy = __fadd_rn(sy, y[u]); sz = __fadd_rn(sz, z[u]); }
    }
    const float wgt = (float)(1.0 / (double)len);          // point * (1.0 / count): double -> float (:87)
    A.out_tmp[3 * r] = __fmul_rn(sx, wgt);
    A.out_tmp[3 * r + 1] = __fmul_rn(sy, wgt);
    A.out_tmp[3 * r + 2] = __fmul_rn(sz, wgt);
    if (A.feats) {                             // features summed in input order, then f / (float)count (:90-96)
        const float cf = (float)len;
        for (int d = 0; d < A.fdim; d++) {
            float acc = 0.f;
            for (int t = 0; t < len; t++) acc = __fadd_rn(acc, A.feats[((size_t)lo + idx_at(p + t)) * A.fdim + d]);
            A.feat_tmp[r * A.fdim + d] = __fdiv_rn(acc, cf);
        }
    }
}

// The element through global arrays (any key width): table of VOXF_TABLE_G buckets in LDS, everything per point in global memory.
__device__ int voxf_element_global(const float* __restrict__ P, int lo, int np, VoxGrid g, int* __restrict__ table, int* __restrict__ wsum,
                                   const VoxfArrays& A)
{
    const int tid = threadIdx.x;
    double Bd = 1.0;
    int shift = 0;
    while (floor(g.cells / Bd) + 2.0 > (double)VOXF_TABLE_G && shift < 1100) { Bd *= 2.0; shift++; }
    const int nbk = (int)(floor(g.cells / Bd) + 2.0);
    const unsigned long long NXY = g.NX * g.NY;
#define VOXF_BUCKET(key) ((int)min((shift < 64 ? (key) >> shift : 0ull), (unsigned long long)(nbk - 1)))
    __syncthreads();
    for (int i = tid; i <= VOXF_TABLE_G; i += VOXF_THREADS) table[i] = 0;
    __syncthreads();
    for (int i = tid; i < np; i += VOXF_THREADS) {          // ONE returning LDS atomic per point hands out its slot inside the bucket
        const unsigned long long key = vox_key(g, NXY, P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2]);
        A.keys[lo + i] = key;
        A.slot[lo + i] = atomicAdd(&table[VOXF_BUCKET(key)], 1);
    }
    __syncthreads();
    voxf_scan(table, nbk, wsum);
    if (tid == 0) table[nbk] = np;
    __syncthreads();
    for (int i = tid; i < np; i += VOXF_THREADS) {          // (key, input index) pairs, contiguous per bucket
        const unsigned long long k = A.keys[lo + i];
        const int pos = table[VOXF_BUCKET(k)] + A.slot[lo + i];
        A.bkey[lo + pos] = k;
        A.bidx[lo + pos] = i;
    }
    __syncthreads();
    for (int p = tid; p < np; p += VOXF_THREADS) {          // rank by (key, input index) inside the bucket; a run's first point carries its length
        const unsigned long long k = A.bkey[lo + p];
        const int id = A.bidx[lo + p];
        const int c = VOXF_BUCKET(k);
        const int s = table[c], e = table[c + 1];
        int rank = 0, same_before = 0, same = 0;
        for (int t = s; t < e; t++) {
            const unsigned long long kj = A.bkey[lo + t];
            const int j = A.bidx[lo + t];
            rank += (kj < k || (kj == k && j < id)) ? 1 : 0;
            same_before += (kj == k && j < id) ? 1 : 0;
            same += kj == k ? 1 : 0;
        }
        A.sorted[lo + s + rank] = make_float4(0.f, 0.f, 0.f, __int_as_float(id));
        A.runlen[lo + s + rank] = same_before == 0 ? same : 0;
        A.rowidx[lo + s + rank] = same_before == 0 ? 1 : 0;
    }
#undef VOXF_BUCKET
    __syncthreads();
    const int rows = voxf_scan(A.rowidx + lo, np, wsum);
    const float4* srt = A.sorted + lo;
    auto idx_at = [srt](int t) { return __float_as_int(srt[t].w); };
    for (int p = tid; p < np; p += VOXF_THREADS) {
        const int len = A.runlen[lo + p];
        if (len) voxf_emit_run(P, idx_at, p, len, (size_t)lo + A.rowidx[lo + p], A, lo);
    }
    return rows;
}

#ifdef VOXF_TIMING               // development: phase boundaries of workgroup 0 in 10 ns ticks (tools/a1_phases.py)
__device__ long long voxf_ticks[16];
extern "C" int buf_debug_voxf_ticks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(voxf_ticks), sizeof(long long) * 16) == hipSuccess ? 0 : 1; }
#define VOXF_TICK(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) voxf_ticks[k] = wall_clock64(); } while (0)
#else
#define VOXF_TICK(k) do { } while (0)
#endif
__global__ void __launch_bounds__(VOXF_THREADS) k_vox_fused(const float* __restrict__ pts, const int* __restrict__ off, float dl, VoxfArrays A,
                                                           int* __restrict__ counts)
{
    extern __shared__ int vf_lds[];
    int* const table = vf_lds;                                              // [VOXF_TABLE / 2 + 1] packed 16-bit counts -> starts -> ends
    const unsigned short* const table16 = (const unsigned short*)vf_lds;    // bucket b = half (b & 1) of word b >> 1
    unsigned int* const word = (unsigned int*)(vf_lds + VOXF_TABLE_WORDS + 1);    // [VOXF_MAX_N] (key mod B) << 16 | index; later row << 16 | index
    unsigned short* const aux = (unsigned short*)(word + VOXF_MAX_N);       // [VOXF_MAX_N] bucket of a sorted position; later run lengths
    __shared__ float smn[3][VOXF_WAVES], smx[3][VOXF_WAVES];
    __shared__ int wsum[VOXF_WAVES];
    __shared__ VoxGrid sg;
    __shared__ int s_shift;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int lo = off[b], hi = off[b + 1], np = hi - lo;
    if (np <= 0) { if (tid == 0) counts[b] = 0; return; }
    const float* P = pts + 3 * (size_t)lo;
    VOXF_TICK(0);
    // ---- the element's points live in registers from here to the scatter (VOXF_PT per lane, all loads in flight at once)
    float X[VOXF_PT], Y[VOXF_PT], Z[VOXF_PT];
#pragma unroll
    for (int it = 0; it < VOXF_PT; it++) {
        const int i = tid + it * VOXF_THREADS;
        const bool ok = i < np;
        X[it] = ok ? P[3 * (size_t)i] : 0.f; Y[it] = ok ? P[3 * (size_t)i + 1] : 0.f; Z[it] = ok ? P[3 * (size_t)i + 2] : 0.f;
    }
    VOXF_TICK(1);
    // ---- bounding box, origin, NX / NY (k_vox_bbox's arithmetic) and this element's bucket width
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
#pragma unroll
    for (int it = 0; it < VOXF_PT; it++) {
        if (tid + it * VOXF_THREADS < np) {
            mn[0] = X[it] < mn[0] ? X[it] : mn[0]; mx[0] = X[it] > mx[0] ? X[it] : mx[0];
            mn[1] = Y[it] < mn[1] ? Y[it] : mn[1]; mx[1] = Y[it] > mx[1] ? Y[it] : mx[1];
            mn[2] = Z[it] < mn[2] ? Z[it] : mn[2]; mx[2] = Z[it] > mx[2] ? Z[it] : mx[2];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], z = mx[c];
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            a = fminf(a, __shfl_xor(a, d, WAVE));
            z = fmaxf(z, __shfl_xor(z, d, WAVE));
        }
        if (lane == 0) { smn[c][w] = a; smx[c][w] = z; }
    }
    __syncthreads();
    if (tid == 0) {
        VoxGrid g;
        g.dl = dl;
        g.lo = lo; g.hi = hi;
        float inv = __fdiv_rn(1.0f, dl);                       // (1/sampleDl), fp32
        double N[3];
        for (int c = 0; c < 3; c++) {
            float a = smn[c][0], z = smx[c][0];
            for (int i = 1; i < VOXF_WAVES; i++) { a = fminf(a, smn[c][i]); z = fmaxf(z, smx[c][i]); }
            g.o[c] = __fmul_rn(floorf(__fmul_rn(a, inv)), dl);  // floor(min * (1/dl)) * dl
            N[c] = (double)floorf(__fdiv_rn(__fsub_rn(z, g.o[c]), dl)) + 1.0;
        }
        g.NX = (unsigned long long)(long long)N[0];
        g.NY = (unsigned long long)(long long)N[1];
        g.cells = fmax(N[0], 1.0) * fmax(N[1], 1.0) * fmax(N[2], 1.0);
        double Bd = 1.0;
        int shift = 0;
        while (floor(g.cells / Bd) + 2.0 > (double)VOXF_TABLE && shift <= 15) { Bd *= 2.0; shift++; }
        g.nbuckets = (long long)(floor(g.cells / Bd) + 2.0);
        g.table_off = 0;
        sg = g;
        s_shift = shift;
    }
    VOXF_TICK(2);
    for (int i = tid; i <= VOXF_TABLE_WORDS; i += VOXF_THREADS) table[i] = 0;
    __syncthreads();
    const VoxGrid g = sg;
    const int shift = s_shift, nbk = (int)g.nbuckets;
    const unsigned long long NXY = g.NX * g.NY;
    bool packed = shift <= 15;                      // (uniform) a key's position inside its bucket fits 15 bits (words stay below the rank loop's sentinel)
    const bool small_box = g.cells < 2147483648.0;
    const unsigned long long kmask = (1ull << (shift & 63)) - 1ull;
    unsigned int pk[VOXF_PT];                       // bucket << 16 | key mod bucket width
    if (packed) {
        // ---- bucket populations; a key beyond the bounding box (the reference's wrapped keys) is parked in the last bucket
#pragma unroll
        for (int it = 0; it < VOXF_PT; it++) {
            if (tid + it * VOXF_THREADS < np) {
                const unsigned long long key = vox_key_small(g, NXY, small_box, X[it], Y[it], Z[it]);
                const int bk = (int)min(key >> shift, (unsigned long long)(nbk - 1));
                atomicAdd(&table[bk >> 1], 1 << ((bk & 1) * 16));
                pk[it] = (unsigned int)bk << 16 | (unsigned int)(key & kmask);       // all the scatter needs (the points' registers are free from here)
            }
        }
        __syncthreads();
        packed = table16[nbk - 1] == 0;               // in-range keys never reach the last bucket: nbuckets = floor(cells / B) + 2
    }
    VOXF_TICK(3);
    if (!packed) {
        const int rows = voxf_element_global(P, lo, np, g, vf_lds, wsum, A);
        if (tid == 0) counts[b] = rows;
        return;
    }
    VOXF_TICK(4);
    voxf_scan16(table, (nbk + 1) / 2, wsum);
    VOXF_TICK(5);
    // ---- scatter: the cursor of a bucket is its table entry (starts become ends)
#pragma unroll
    for (int it = 0; it < VOXF_PT; it++) {
        const int i = tid + it * VOXF_THREADS;
        if (i < np) {
            const int bk = (int)(pk[it] >> 16);
            const int pos = (atomicAdd(&table[bk >> 1], 1 << ((bk & 1) * 16)) >> ((bk & 1) * 16)) & 0xffff;
            word[pos] = pk[it] << 16 | (unsigned int)i;
            aux[pos] = (unsigned short)bk;
        }
    }
    __syncthreads();
    VOXF_TICK(6);
    // ---- every point ranks itself inside its bucket by its word = by (key, input index): runs of equal key end up contiguous and in INPUT
    //      ORDER; the first point of a run carries the run's length.  In place: the new positions wait in registers for the barrier.
    int npos[VOXF_PT];
    unsigned int nval[VOXF_PT];                     // run length << 16 | input index
#pragma unroll
    for (int it = 0; it < VOXF_PT; it++) {
        const int p = tid + it * VOXF_THREADS;
        npos[it] = -1;
        if (p < np) {
            const unsigned int me = word[p];
            const int bk = aux[p];
            const int s = bk ? table16[bk - 1] : 0, e = table16[bk];
            // three counts per bucket mate, two instructions each: words below mine, words of a smaller key, words of a key <= mine
            // (lanes past the end of the bucket compare the sentinel ~0, which counts nowhere: words are < 2^31)
            const unsigned int klo = me & 0xffff0000u, khi = me | 0xffffu;
            int rank = 0, ltk = 0, lek = 0;
            for (int t0 = s; t0 < e; t0 += VOXF_U) {
                unsigned int o[VOXF_U];
#pragma unroll
                for (int u = 0; u < VOXF_U; u++) o[u] = word[min(t0 + u, e - 1)];
#pragma unroll
                for (int u = 0; u < VOXF_U; u++) {
                    const unsigned int v = t0 + u < e ? o[u] : 0xffffffffu;
                    rank += v < me ? 1 : 0;
                    ltk += v < klo ? 1 : 0;
                    lek += v <= khi ? 1 : 0;
                }
            }
            const int same_before = rank - ltk, same = lek - ltk;
            npos[it] = s + rank;
            nval[it] = (unsigned int)(same_before == 0 ? same : 0) << 16 | (me & 0xffffu);
        }
    }
    __syncthreads();
    VOXF_TICK(7);
#pragma unroll
    for (int it = 0; it < VOXF_PT; it++)
        if (npos[it] >= 0) { word[npos[it]] = nval[it] & 0xffffu; aux[npos[it]] = (unsigned short)(nval[it] >> 16); }
    __syncthreads();
    VOXF_TICK(8);
    // ---- rows before a run's head (ballot + popcount per 64 positions)
    {
        const int seg = ((np + VOXF_WAVES - 1) / VOXF_WAVES + WAVE - 1) / WAVE * WAVE;
        const int s0 = w * seg, s1 = min(s0 + seg, np);
        int heads = 0;
        for (int p0 = s0; p0 < s1; p0 += WAVE) heads += __popcll(__ballot(p0 + lane < s1 && aux[p0 + lane] != 0));
        if (lane == 0) wsum[w] = heads;
        __syncthreads();
        int carry = 0, total = 0;
        for (int i = 0; i < VOXF_WAVES; i++) { if (i < w) carry += wsum[i]; total += wsum[i]; }
        for (int p0 = s0; p0 < s1; p0 += WAVE) {
            const bool hd = p0 + lane < s1 && aux[p0 + lane] != 0;
            const unsigned long long m = __ballot(hd);
            if (hd) A.rowidx[lo + carry + __popcll(m & ((1ull << lane) - 1ull))] = (p0 + lane) | (int)aux[p0 + lane] << 16;      // row -> (head position, run length)
            carry += __popcll(m);
        }
        if (tid == 0) counts[b] = total;
        __syncthreads();
    }
    VOXF_TICK(9);
    // ---- one lane per voxel run, rows dealt to lanes densely (a lane per POSITION that works only where a run starts kept 1 lane in 14 busy
    //      at 14 points per voxel, and every wavefront paid its longest run: 40-70 us of the kernel; staging the window's points in LDS did
    //      not change that).  VOXF_U gathers in flight per lane.
    int rows = 0;
    for (int i = 0; i < VOXF_WAVES; i++) rows += wsum[i];
    auto idx_at = [word](int t) { return (int)(word[t] & 0xffffu); };
    for (int r = tid; r < rows; r += VOXF_THREADS) {
        const int info = A.rowidx[lo + r];
        voxf_emit_run(P, idx_at, info & 0xffff, info >> 16, (size_t)lo + r, A, lo);
    }
    VOXF_TICK(10);
}

// rows of element b (at row off[b] of the element-local buffers) -> the stacked output, max_p rows per element at most (grid_subsampling.cpp:186-200)
__global__ void __launch_bounds__(256) k_vox_concat(const float* __restrict__ out_tmp, const float* __restrict__ feat_tmp, const int* __restrict__ counts,
                                                  const int* __restrict__ off, int max_p, int fdim, float* __restrict__ out, float* __restrict__ out_feats)
{
    __shared__ int red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int before = 0;
    for (int i = tid; i < b; i += 256) { const int c = counts[i]; before += max_p > 0 && c > max_p ? max_p : c; }
    for (int d = WAVE / 2; d > 0; d >>= 1) before += __shfl_xor(before, d, WAVE);
    if ((tid & (WAVE - 1)) == 0) red[tid / WAVE] = before;
    __syncthreads();
    const size_t dst = (size_t)red[0] + red[1] + red[2] + red[3], src = (size_t)off[b];
    const int c = counts[b], keep = max_p > 0 && c > max_p ? max_p : c;
    for (int i = tid; i < 3 * keep; i += 256) out[3 * dst + i] = out_tmp[3 * src + i];
    if (fdim > 0)
        for (long long i = tid; i < (long long)fdim * keep; i += 256) out_feats[dst * fdim + i] = feat_tmp[src * fdim + i];
}

struct VoxWs {
    VoxGrid* grids; VoxStatus* st; int* off; int* table; int* cell_of; float4* sorted_tmp; float4* sorted;
    int* order; int* head; int* scan_tmp; int* total; int* counts; float* out_tmp; float* feat_tmp;
    unsigned long long* keys; unsigned long long* key_sorted; int* cell_sorted;
};

static VoxWs carve_vox(WsCarver& w, int n, int nb, int64_t max_cells, int fdim)
{
    VoxWs v;
    size_t nn = (size_t)(n > 0 ? n : 1);
    v.grids = w.take<VoxGrid>((size_t)nb);
    v.off = w.take<int>((size_t)nb + 1);
    v.st = w.take<VoxStatus>(1);
    v.table = w.take<int>((size_t)max_cells);
    v.cell_of = w.take<int>(nn);
    v.sorted_tmp = w.take<float4>(nn);
    v.sorted = w.take<float4>(nn);
    v.order = w.take<int>(nn);
    v.head = w.take<int>(nn);
    v.scan_tmp = w.take<int>(scan_tmp_ints());
    v.total = w.take<int>(1);
    v.counts = w.take<int>((size_t)nb + 2);
    v.out_tmp = w.take<float>(3 * nn);
    v.feat_tmp = w.take<float>((size_t)(fdim > 0 ? fdim : 0) * nn + 1);
    v.keys = w.take<unsigned long long>(nn);
    v.key_sorted = w.take<unsigned long long>(nn);
    v.cell_sorted = w.take<int>(nn);
    return v;
}

extern "C" size_t buf_grid_subsample_ws_bytes(int n, int nb, int64_t max_cells, int fdim)
{
    WsCarver w(nullptr, 0);
    carve_vox(w, n, nb, max_cells, fdim);
    return w.used();
}

extern "C" int buf_grid_subsample_batch(const float* pts, int n, const int* batches_host, int nb, float dl,
                                        int max_p, const float* feats, int fdim, float* out_pts, float* out_feats,
                                        int* out_batches_host, int* out_m_host,
                                        int64_t max_cells, void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(batches_host && out_batches_host && out_m_host && ws, BUF_EINVAL, "buf_grid_subsample_batch: null argument");
    BUF_REQUIRE(n >= 0 && nb > 0, BUF_EINVAL, "buf_grid_subsample_batch: n=%d nb=%d", n, nb);
    BUF_REQUIRE(dl > 0.f, BUF_EINVAL, "buf_grid_subsample_batch: sampleDl=%g must be > 0", dl);
    BUF_REQUIRE(max_cells >= 2LL * nb && max_cells < 0x7fffffffLL, BUF_EINVAL, "buf_grid_subsample_batch: max_cells=%lld (need >= 2 per batch element)", (long long)max_cells);
    BUF_REQUIRE(n == 0 || (pts && out_pts), BUF_EINVAL, "buf_grid_subsample_batch: null points");
    WsCarver w(ws, ws_bytes);
    BUF_REQUIRE(fdim >= 0 && (fdim == 0 || (feats && out_feats)), BUF_EINVAL, "buf_grid_subsample_batch: features");
    if (!feats) fdim = 0;
    VoxWs v = carve_vox(w, n, nb, max_cells, fdim);
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_grid_subsample_batch: workspace %zu < %zu bytes", ws_bytes, w.used());
    int rc = upload_offsets(v.off, batches_host, nb, n, "buf_grid_subsample_batch", s);
    if (rc) return rc;
    if (n == 0) {
        for (int b = 0; b < nb; b++) out_batches_host[b] = 0;
        *out_m_host = 0;
        return BUF_OK;
    }
    TimedSpan span;      // the whole kernel sequence of one call (bbox .. emit); M is bounded by N in the byte count
    bool timed = timing_begin(s, &span, 24.0 * n + 4.0 * nb, BUF_TIMED_GRID_SUBSAMPLE);
    // one workgroup per element with its table in LDS when every element fits (BUF_VOX_FUSED=0: the global-table path, for tests)
    bool fused = true;
    for (int b = 0; b < nb; b++) fused = fused && batches_host[b] <= VOXF_MAX_N;
    { const char* e_ = getenv("BUF_VOX_FUSED"); if (e_ && e_[0] == '0') fused = false; }
    if (fused) {
        static LdsGrant grant;
        const size_t lds = VOXF_LDS_BYTES;
        if (int rc_ = grant_dynamic_lds((const void*)k_vox_fused, lds, grant)) return rc_;
        VoxfArrays A;                                              // (arrays of the other path under new names)
        A.keys = v.keys; A.slot = v.cell_of; A.bkey = v.key_sorted; A.bidx = v.order; A.sorted = v.sorted; A.runlen = v.head; A.rowidx = v.cell_sorted;
        A.out_tmp = v.out_tmp; A.feats = fdim > 0 ? feats : nullptr; A.fdim = fdim; A.feat_tmp = v.feat_tmp;
        k_vox_fused<<<nb, VOXF_THREADS, lds, s>>>(pts, v.off, dl, A, v.counts);
        k_vox_concat<<<nb, 256, 0, s>>>(v.out_tmp, v.feat_tmp, v.counts, v.off, max_p, fdim, out_pts, out_feats);
    } else {
    BUF_CHECK_HIP(hipMemsetAsync(v.table, 0, sizeof(int) * (size_t)max_cells, s));
    k_vox_bbox<<<nb, 1024, 0, s>>>(pts, v.off, v.grids, dl);
    k_vox_offsets<<<1, 1024, 0, s>>>(v.grids, nb, (long long)max_cells, v.st);
    int blocks = cdiv(n, 256);
    k_vox_count<<<blocks, 256, 0, s>>>(pts, n, v.off, nb, v.grids, v.st, v.table, v.cell_of, v.keys);
    rc = exclusive_scan_i32(v.table, (long long)max_cells, v.scan_tmp, nullptr, s);
    if (rc) return rc;
    k_cell_scatter<<<blocks, 256, 0, s>>>(pts, n, v.cell_of, &v.st->error, v.table, v.sorted_tmp);
    k_vox_rank<<<blocks, 256, 0, s>>>(v.cell_of, v.keys, v.table, v.sorted_tmp, n, v.st, v.sorted, v.key_sorted, v.cell_sorted, v.head);   // + head flags
    rc = exclusive_scan_i32(v.head, n, v.scan_tmp, v.total, s);
    if (rc) return rc;
    float* dst = max_p > 0 ? v.out_tmp : out_pts;
    float* fdst = max_p > 0 ? v.feat_tmp : out_feats;
    k_vox_emit<<<blocks, 256, 0, s>>>(v.sorted, v.key_sorted, v.cell_sorted, v.head, n, v.st, dst, fdim > 0 ? feats : nullptr, fdim, fdst,
                                      v.off, nb, v.total, v.counts);                                  // + rows per element
    }
    if (timed) timing_end(s, &span);
    BUF_LAUNCH_CHECK();
    int stackc[66];
    int* hc = nb + 2 <= 66 ? stackc : (int*)malloc(sizeof(int) * ((size_t)nb + 2));
    hipError_t e = hipMemcpyAsync(hc, v.counts, sizeof(int) * ((size_t)nb + (fused ? 0 : 2)), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int err = e == hipSuccess && !fused ? hc[nb] : 0;          // hc[] is valid only after a successful copy + sync (the LDS path has no capacity to exceed)
    int m = 0;
    rc = BUF_OK;
    if (e != hipSuccess) { buf_set_error("buf_grid_subsample_batch: %s", hipGetErrorString(e)); rc = BUF_EHIP; }
    else if (err) { buf_set_error("buf_grid_subsample_batch: bucket table does not fit max_cells=%lld", (long long)max_cells); rc = BUF_ECAPACITY; }
    else {
        int src = 0;
        for (int b = 0; b < nb; b++) {
            int mb = hc[b];
            int keep = (max_p > 0 && mb > max_p) ? max_p : mb;      // grid_subsampling.cpp:186-200, in OUR row order
            if (max_p > 0 && keep > 0 && !fused) {                  // (k_vox_concat has stacked the rows already)
                hipError_t e2 = hipMemcpyAsync(out_pts + 3 * (size_t)m, v.out_tmp + 3 * (size_t)src, sizeof(float) * 3 * (size_t)keep,
                                               hipMemcpyDeviceToDevice, s);
                if (e2 == hipSuccess && fdim > 0)
                    e2 = hipMemcpyAsync(out_feats + (size_t)fdim * m, v.feat_tmp + (size_t)fdim * src, sizeof(float) * (size_t)fdim * keep,
                                        hipMemcpyDeviceToDevice, s);
                if (e2 != hipSuccess) { buf_set_error("buf_grid_subsample_batch: %s", hipGetErrorString(e2)); rc = BUF_EHIP; break; }
            }
            out_batches_host[b] = keep;
            m += keep;
            src += mb;
        }
        *out_m_host = m;
    }
    if (hc != stackc) free(hc);
    return rc;
}
