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
    if (timed) timing_end(s, &span);
    BUF_LAUNCH_CHECK();
    int stackc[66];
    int* hc = nb + 2 <= 66 ? stackc : (int*)malloc(sizeof(int) * ((size_t)nb + 2));
    hipError_t e = hipMemcpyAsync(hc, v.counts, sizeof(int) * ((size_t)nb + 2), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int err = e == hipSuccess ? hc[nb] : 0;          // hc[] is valid only after a successful copy + sync
    int m = 0;
    rc = BUF_OK;
    if (e != hipSuccess) { buf_set_error("buf_grid_subsample_batch: %s", hipGetErrorString(e)); rc = BUF_EHIP; }
    else if (err) { buf_set_error("buf_grid_subsample_batch: bucket table does not fit max_cells=%lld", (long long)max_cells); rc = BUF_ECAPACITY; }
    else {
        int src = 0;
        for (int b = 0; b < nb; b++) {
            int mb = hc[b];
            int keep = (max_p > 0 && mb > max_p) ? max_p : mb;      // grid_subsampling.cpp:186-200, in OUR row order
            if (max_p > 0 && keep > 0) {
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
