// A2 -- batched radius neighbours on a uniform cell grid (replaces
// cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332, the nanoflann KD-tree search).
//
// Build (per call, all batch elements at once):
//   k_grid_bbox    one workgroup per batch element: bounding box, cell edge >= radius coarsened
//                  until the dense table fits `cells_per_elem`
//   k_grid_count   one thread per support: cell id (fp64 cell coordinates), atomic histogram
//   exclusive scan over the concatenated tables -> cell starts in the stacked order
//   k_grid_scatter pos = atomicAdd(table[cell],1): float4 (x,y,z,bitcast index) in cell order;
//                  the table ends up holding inclusive cell ends
// Query:
//   k_grid_query   one lane per query, 9 contiguous x-runs of cells (3 y * 3 z), candidates read as
//                  float4 from the cell-ordered array (lanes of a wavefront that sit in the same
//                  cell read the same addresses -> one transaction), accept d2 < r2, keep the KL
//                  smallest (d2,index) keys in a lane-private LDS column, emit ascending.
//                  Rows longer than KL are produced in further sweeps (keys > last emitted).
#include "common.h"

struct CellGrid {
    float mn[3];
    int   table_off;     // first table slot of this element (global, concatenated)
    double inv_cell;
    int   dim[3];
    int   s_off;
};

#define BBOX_THREADS 1024

__global__ void __launch_bounds__(BBOX_THREADS) k_grid_bbox(const float* __restrict__ pts, const int* __restrict__ s_off,
                                                         CellGrid* __restrict__ grids, double radius,
                                                         long long cells_per_elem)
{
    int b = blockIdx.x;
    int lo = s_off[b], hi = s_off[b + 1];
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int i = lo + threadIdx.x; i < hi; i += BBOX_THREADS) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
    __shared__ float smn[3][BBOX_THREADS / WAVE], smx[3][BBOX_THREADS / WAVE];
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
        CellGrid g;
        double ext[3];
        for (int c = 0; c < 3; c++) {
            float a = smn[c][0], z = smx[c][0];
            for (int i = 1; i < BBOX_THREADS / WAVE; i++) { a = fminf(a, smn[c][i]); z = fmaxf(z, smx[c][i]); }
            if (hi <= lo) { a = 0.f; z = 0.f; }
            g.mn[c] = a;
            ext[c] = (double)z - (double)a;
        }
        // cell edge strictly larger than the radius so that |dx| < r never spans two cells + 1
        double cell = radius > 0 ? radius * 1.00001 : 1.0;
        for (int it = 0; it < 200; it++) {
            double tot = 1.0;
            for (int c = 0; c < 3; c++) tot *= floor(ext[c] / cell) + 1.0;
            if (tot <= (double)cells_per_elem) break;
            cell *= 1.25;
        }
        for (int c = 0; c < 3; c++) g.dim[c] = (int)(floor(ext[c] / cell) + 1.0);
        g.inv_cell = 1.0 / cell;
        g.table_off = (int)((long long)b * cells_per_elem);
        g.s_off = lo;
        grids[b] = g;
    }
}

__device__ __forceinline__ int cell_coord(float v, float mn, double inv_cell, int dim)
{
    double c = floor(((double)v - (double)mn) * inv_cell);
    c = fmin(fmax(c, 0.0), (double)(dim - 1));
    return (int)c;
}

__global__ void __launch_bounds__(256) k_grid_count(const float* __restrict__ pts, int ns, const int* __restrict__ s_off,
                                                  int nb, const CellGrid* __restrict__ grids,
                                                  int* __restrict__ table, int* __restrict__ cell_of)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    int b = find_elem(s_off, nb, i);
    CellGrid g = grids[b];
    int cx = cell_coord(pts[3 * (size_t)i], g.mn[0], g.inv_cell, g.dim[0]);
    int cy = cell_coord(pts[3 * (size_t)i + 1], g.mn[1], g.inv_cell, g.dim[1]);
    int cz = cell_coord(pts[3 * (size_t)i + 2], g.mn[2], g.inv_cell, g.dim[2]);
    int c = g.table_off + cx + g.dim[0] * (cy + g.dim[1] * cz);
    cell_of[i] = c;
    atomicAdd(&table[c], 1);
}

template <int KL>
__global__ void __launch_bounds__(WAVE) k_grid_query(const CellGrid* __restrict__ grids, const int* __restrict__ table,
                                                   const float4* __restrict__ sorted, const float* __restrict__ queries,
                                                   int nq, const int* __restrict__ q_off, int nb,
                                                   const int* __restrict__ q_order, float r2, int k_out, int shadow,
                                                   int* __restrict__ nbr_out, int* __restrict__ counts_out,
                                                   int* __restrict__ max_count_out, const int* __restrict__ nq_dyn)
{
    __shared__ unsigned long long col[KL][WAVE];
    const int lane = threadIdx.x;
    int t = blockIdx.x * WAVE + lane;
    if (nq_dyn) {                       // fallback pass: the list length lives in device memory
        nq = min(nq, *nq_dyn);
        if (blockIdx.x * WAVE >= nq) return;
    }
    bool active = t < nq;
    int qi = active ? (q_order ? q_order[t] : t) : 0;
    int rs[9], re[9];
    float qx = 0, qy = 0, qz = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { rs[j] = 0; re[j] = 0; }
    if (active) {
        int b = find_elem(q_off, nb, qi);
        CellGrid g = grids[b];
        qx = queries[3 * (size_t)qi]; qy = queries[3 * (size_t)qi + 1]; qz = queries[3 * (size_t)qi + 2];
        double fx = floor(((double)qx - (double)g.mn[0]) * g.inv_cell);
        double fy = floor(((double)qy - (double)g.mn[1]) * g.inv_cell);
        double fz = floor(((double)qz - (double)g.mn[2]) * g.inv_cell);
        // clamp far-away queries before the int conversion; they simply find no cell
        fx = fmin(fmax(fx, -2.0), (double)g.dim[0] + 1.0);
        fy = fmin(fmax(fy, -2.0), (double)g.dim[1] + 1.0);
        fz = fmin(fmax(fz, -2.0), (double)g.dim[2] + 1.0);
        int cx = (int)fx, cy = (int)fy, cz = (int)fz;
        int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
#pragma unroll
        for (int j = 0; j < 9; j++) {
            int y = cy + (j % 3) - 1, z = cz + (j / 3) - 1;
            if (x0 <= x1 && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) {
                int g0 = g.table_off + x0 + g.dim[0] * (y + g.dim[1] * z);
                int g1 = g0 + (x1 - x0);
                rs[j] = g0 == 0 ? 0 : table[g0 - 1];
                re[j] = table[g1];
            }
        }
    }

    int total = 0;
    unsigned long long lb = 0;      // keys must be > lb (exclusive) in sweeps after the first
    bool first = true;
    int written = 0;
    int* row = nbr_out + (size_t)qi * k_out;
    // sweeps are wave-uniform in count only through the ballot below
    for (;;) {
        int n = 0;
        unsigned long long worst = ~0ull;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            for (int p = rs[j]; p < re[j]; p++) {
                float4 c = sorted[p];
                float d2 = sqdist3(qx, qy, qz, c.x, c.y, c.z);
                if (d2 < r2) {
                    if (first) total++;
                    unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) |
                                             (unsigned int)__float_as_int(c.w);
                    if ((first || key > lb) && (n < KL || key < worst) && k_out > 0) {
                        int i = n < KL ? n : KL - 1;
                        while (i > 0 && col[i - 1][lane] > key) { col[i][lane] = col[i - 1][lane]; i--; }
                        col[i][lane] = key;
                        if (n < KL) n++;
                        if (n == KL) worst = col[KL - 1][lane];
                    }
                }
            }
        }
        if (active) {
            int lim = min(n, k_out - written);
            for (int i = 0; i < lim; i++) row[written + i] = (int)(unsigned int)(col[i][lane] & 0xffffffffu);
            written += lim;
            if (n == KL) lb = col[KL - 1][lane];
        }
        first = false;
        bool more = active && n == KL && written < k_out && written < total;
        if (!__any(more)) break;
        if (!more) {  // this lane is done: make further sweeps empty for it
#pragma unroll
            for (int j = 0; j < 9; j++) re[j] = rs[j];
        }
    }
    if (active) {
        for (int i = written; i < k_out; i++) row[i] = shadow;
        if (counts_out) counts_out[qi] = total;
    }
    if (max_count_out) {
        int m = active ? total : 0;
        for (int d = WAVE / 2; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, WAVE));
        if (lane == 0 && m > 0) atomicMax(max_count_out, m);
    }
}

// Fast path: a 16-lane group per query, four queries per wavefront (a neighbourhood holds 10-35 points, so a
// full wavefront per query would leave most lanes idle in the scan and in the sort).  The kernel is VALU-issue bound
// (DESIGN.md section 5), so every phase is shaped to cost few instructions per wavefront:
//   - the grid is 2-D: blockIdx.y = batch element, so the element's grid descriptor and offsets are wave-uniform
//     (scalar registers, no per-query search), blockIdx.x = chunk of 16 queries of that element;
//   - cell coordinates stay in fp64 (a query must land in exactly the cell its coordinates round to: the cell edge
//     exceeds the radius by 1e-5 only), but lane d of a group computes dimension d only and a group shuffle shares them;
//   - lanes 0..8 of a group fetch the 9 cell-run bounds and pass them to the group through LDS (five ds_read_b128);
//     the runs are flattened and scanned 16 candidates at a time (float4 loads, contiguous inside a run), candidate c
//     maps to its run through per-run offsets (compare + select per run);
//   - accepted (d2,index) keys are compacted in arrival order through ballot + popcount prefix, and counted into 16
//     distance buckets per query (bucket = floor(d2 * 16 / r2), monotone in d2);
//   - sorting = exclusive scan of the bucket counts over the group's 16 lanes (DPP), a scatter of the keys into bucket
//     order, then every key ranks itself against the keys of ITS bucket only (2-3 instead of 35): an exact sort by
//     (d2, index) for rows of up to QW_CAP neighbours.  Longer rows are pushed on a todo list and redone by
//     k_grid_query (lane-per-query, unbounded rows) in the same stream.
#define QW_WAVES 4
#define QG 16                       // lanes per query
#define QPW (WAVE / QG)             // queries per wavefront
#define QW_CAP 64                   // row capacity of the LDS slice: template parameter CAP = 64, or 128 for k_out > 32 (the 30-NN
//                                     candidate search of the normals asks for 40 and meets 5 % of rows longer than 64)
#define QW_QPB (QW_WAVES * QPW)     // queries per workgroup

// exclusive prefix sum over each row of 16 lanes (DPP row shifts, zero fill)
__device__ __forceinline__ int row16_excl_scan(int v)
{
    int inc = v;
#define ROW_SHR_ADD(ctrl) inc += __builtin_amdgcn_update_dpp(0, inc, ctrl, 0xf, 0xf, true)
    ROW_SHR_ADD(0x111);   // row_shr:1
    ROW_SHR_ADD(0x112);   // row_shr:2
    ROW_SHR_ADD(0x114);   // row_shr:4
    ROW_SHR_ADD(0x118);   // row_shr:8
#undef ROW_SHR_ADD
    return inc - v;
}

// XCD-aware (chunk, element) of a workgroup of the 2-D query launches (grid = chunks x elements).  The hardware deals workgroups to the
// 8 XCDs round-robin in launch order (x fastest): with the plain blockIdx mapping the chunks of ONE element land on all eight XCDs in turn and
// every L2 fetches that element's whole cell-ordered array and table (measured, round 6: FETCH_SIZE of the self query = 8 x its algorithmic
// input; profiles/r06_a2_floor.txt).  Here XCD x takes the elements x, x + 8, ...: an element's rows, cells and records stay in ONE L2
// (FETCH_SIZE -36 % / -83 % for the cell-centric / query-centric kernel, WRITE_SIZE -24 %: the 68-byte rows of neighbouring queries merge in
// one L2).  The price is the imbalance between the XCDs' element sums (+2 % / +7 % time at 64 elements of 9-12 k points).  Also tried: XCD x
// takes the x-th eighth of EVERY element's chunk range (no division, balanced by construction -- but the launch is as wide as the LARGEST
// element, so the last XCD's slab is mostly empty for the others: +16 %).  A bijection on the grid whenever the element count is a multiple
// of 8 (pairs: always); other grids keep the plain mapping.
__device__ __forceinline__ void grid2d_xcd_block(int& chunk, int& elem)
{
    chunk = blockIdx.x; elem = blockIdx.y;
    if ((gridDim.y & 7u) == 0u) {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, x = L & 7u, idx = L >> 3;
        elem = (int)(8u * (idx / gridDim.x) + x);
        chunk = (int)(idx % gridDim.x);
    }
}

template <int CAP>
__global__ void __launch_bounds__(QW_WAVES * WAVE) k_grid_query_wave(const CellGrid* __restrict__ grids, const int* __restrict__ table,
                                                                  const float4* __restrict__ sorted, const float* __restrict__ queries,
                                                                  const int* __restrict__ q_off,
                                                                  const int* __restrict__ q_order, int self_query, float r2, float bin_scale,
                                                                  int k_out, int shadow,
                                                                  int* __restrict__ nbr_out, int* __restrict__ counts_out,
                                                                  int* __restrict__ max_count_out, int* __restrict__ todo,
                                                                  int* __restrict__ todo_n)
{
    __shared__ __attribute__((aligned(16))) unsigned long long keys[QW_WAVES][QPW][CAP];       // arrival order
    __shared__ __attribute__((aligned(16))) unsigned long long bkeys[QW_WAVES][QPW][CAP];      // bucket order
    __shared__ __attribute__((aligned(16))) int runs[QW_WAVES][QPW][20];                           // 9 starts | 9 lengths
    __shared__ int hist[QW_WAVES][QPW][QG];                                                        // bucket counts
    __shared__ int pref[QW_WAVES][QPW][QG + 4];                                                    // exclusive prefix, [16] = total
    __shared__ int cur[QW_WAVES][QPW][QG];                                                         // scatter cursors
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int grp = lane / QG, l16 = lane & (QG - 1);
    int bx, b;                                                            // chunk, batch element: wave-uniform
    grid2d_xcd_block(bx, b);
    const int e_lo = q_off[b], e_n = q_off[b + 1] - e_lo;                 // scalar loads
    if (bx * QW_QPB >= e_n) return;                                       // chunk past this element's queries
    const int tl = (bx * QW_WAVES + w) * QPW + grp;                       // query number inside the element
    bool active = tl < e_n;
    const int t = e_lo + tl;
    int qi = 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
        if (self_query) {                                                 // index AND coordinates from the cell-ordered array
            const float4 s = sorted[t];
            qi = __float_as_int(s.w); qx = s.x; qy = s.y; qz = s.z;
        } else {
            qi = q_order ? q_order[t] : t;
            // The block takes its batch element from blockIdx.y, i.e. from the SLOT t.  A caller's q_order may put a query of
            // another element into this slot (any permutation of 0..nq-1 is allowed): that query goes to the todo list -- the
            // lane-per-query pass behind this kernel finds the element of every query it handles from the query's index.
            if (qi < e_lo || qi >= e_lo + e_n) {
                if (l16 == 0) todo[atomicAdd(todo_n, 1)] = qi;
                active = false;
            } else {
                qx = queries[3 * (size_t)qi]; qy = queries[3 * (size_t)qi + 1]; qz = queries[3 * (size_t)qi + 2];
            }
        }
    }
    const CellGrid g = grids[b];                                          // uniform address: scalar loads
    hist[w][grp][l16] = 0;
    // cell coordinate of dimension d = l16 % 3 in fp64, shared through a width-16 shuffle
    const int d = l16 % 3;
    const float qd = d == 0 ? qx : (d == 1 ? qy : qz);
    const float mnd = d == 0 ? g.mn[0] : (d == 1 ? g.mn[1] : g.mn[2]);
    const int dimd = d == 0 ? g.dim[0] : (d == 1 ? g.dim[1] : g.dim[2]);
    double fd = floor(((double)qd - (double)mnd) * g.inv_cell);
    fd = fmin(fmax(fd, -2.0), (double)dimd + 1.0);                        // far-away queries simply find no cell
    const int cd = (int)fd;
    const int cx = __shfl(cd, 0, QG), cy = __shfl(cd, 1, QG), cz = __shfl(cd, 2, QG);
    {
        int rs = 0, len = 0;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        const int y = cy + (l16 % 3) - 1, z = cz + (l16 / 3) - 1;
        if (active && l16 < 9 && x0 <= x1 && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) {
            const int g0 = g.table_off + x0 + g.dim[0] * (y + g.dim[1] * z);
            rs = g0 == 0 ? 0 : table[g0 - 1];
            len = table[g0 + (x1 - x0)] - rs;
        }
        if (l16 < 10) { runs[w][grp][l16] = rs; runs[w][grp][10 + l16] = len; }          // slots 9 / 19: zero padding
    }
    wave_sync();
    int st[9], pre[9], total = 0;
    {
        const int4* R4 = reinterpret_cast<const int4*>(runs[w][grp]);
        const int4 a0 = R4[0], a1 = R4[1], a2 = R4[2], a3 = R4[3], a4 = R4[4];
        const int sv[9] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x };
        const int lv[9] = { a2.z, a2.w, a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z };
#pragma unroll
        for (int j = 0; j < 9; j++) { pre[j] = total; st[j] = sv[j] - total; total += lv[j]; }   // st = start - prefix
    }
    unsigned long long* K = keys[w][grp];
    const unsigned long long gmask_lo = (1ull << l16) - 1ull;
    int m = 0;
    for (int c0 = 0; __any(c0 < total); c0 += 4 * QG) {
        float4 cand[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {               // all four chunk loads of the pass are in flight together
            const int c = c0 + u * QG + l16;
            int off = st[0];
#pragma unroll
            for (int j = 1; j < 9; j++) off = c >= pre[j] ? st[j] : off;
            cand[u] = c < total ? sorted[c + off] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c0 + u * QG + l16;
            const float d2 = sqdist3(qx, qy, qz, cand[u].x, cand[u].y, cand[u].z);
            const bool hit = c < total && d2 < r2;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)__float_as_int(cand[u].w);
            const unsigned long long mask = (__ballot(hit) >> (grp * QG)) & 0xffffull;
            const int pos = m + __popcll(mask & gmask_lo);
            if (hit && pos < CAP) {
                K[pos] = key;
                atomicAdd(&hist[w][grp][min((int)(d2 * bin_scale), QG - 1)], 1);
            }
            m += __popcll(mask);
        }
    }
    if (active && l16 == 0) {
        if (counts_out) counts_out[qi] = m;
        if (max_count_out && m > 0) atomicMax(max_count_out, m);
    }
    if (k_out == 0) return;
    if (m > CAP) {                    // rare: row longer than the LDS slice -> lane-per-query fallback
        if (active && l16 == 0) todo[atomicAdd(todo_n, 1)] = qi;
        m = 0;
    }
    int* row = nbr_out + (size_t)qi * k_out;
    int mmax = m;
    for (int dd = QG; dd < WAVE; dd <<= 1) mmax = max(mmax, __shfl_xor(mmax, dd, WAVE));      // max over the 4 groups
    mmax = __builtin_amdgcn_readfirstlane(mmax);
    // bucket offsets: exclusive scan of the 16 counts over the group's lanes
    wave_sync();
    {
        const int cnt = hist[w][grp][l16];
        const int ex = row16_excl_scan(cnt);
        pref[w][grp][l16] = ex;
        cur[w][grp][l16] = ex;
        if (l16 == QG - 1) pref[w][grp][QG] = ex + cnt;
    }
    wave_sync();
    unsigned long long* B = bkeys[w][grp];
    const int rounds = (mmax + QG - 1) / QG;
    // scatter into bucket order; every key remembers the extent of its bucket
    unsigned long long mine[CAP / QG];
    int blo[CAP / QG], bhi[CAP / QG];
#pragma unroll
    for (int r = 0; r < CAP / QG; r++) {
        mine[r] = 0ull; blo[r] = 0; bhi[r] = 0;
        if (r < rounds) {
            const int i = r * QG + l16;
            if (i < m) {
                const unsigned long long key = K[i];
                const int bk = min((int)(__uint_as_float((unsigned int)(key >> 32)) * bin_scale), QG - 1);
                B[atomicAdd(&cur[w][grp][bk], 1)] = key;
                mine[r] = key; blo[r] = pref[w][grp][bk]; bhi[r] = pref[w][grp][bk + 1];
            }
        }
    }
    wave_sync();
    // rank inside the bucket (the keys of lower buckets are all smaller), then write the row
#pragma unroll
    for (int r = 0; r < CAP / QG; r++) {
        if (r < rounds) {
            int rank = blo[r];
            for (int tt = 0; __any(blo[r] + tt < bhi[r]); tt++) {
                const int p = blo[r] + tt;
                const unsigned long long o = B[p < bhi[r] ? p : 0];
                rank += (p < bhi[r] && o < mine[r]) ? 1 : 0;
            }
            if (r * QG + l16 < m && rank < k_out) row[rank] = (int)(unsigned int)(mine[r] & 0xffffffffu);
        }
    }
    // shadow padding (skipped for rows handed to the fallback: it rewrites the whole row)
    if (active) {
        for (int i = m + l16; i < k_out; i += QG) row[i] = shadow;
    }
}

// Cell-centric self query (round 4): the queries ARE the grid's supports in cell order, so consecutive queries of the
// stream share their cell and with it their 27-cell candidate set.  A wavefront takes 16 consecutive rows of the cell-ordered
// stream; whenever the cell changes (scell[] holds the cell of every row) it derives the 9 run bounds ONCE -- integer cell
// coordinates from the cell number, no fp64, no per-query search -- and stages the candidates of the 27 cells ONCE in LDS as
// float4 (contiguous runs: coalesced 16-byte loads); every query of the cell then tests them from LDS with all 64 lanes
// (its coordinates are wave-uniform: scalar operands), compacts the accepted (d2, index) keys into its LDS slice by ballot +
// mbcnt and counts them into the 16 distance buckets.  After every four queries the bucket sort of k_grid_query_wave runs
// on the four slices at once (a 16-lane group per query).  Per query: one LDS pass over ~50 candidates instead of ~50 float4
// loads through L2 and three dependent memory round trips; rows are identical to the query-centric kernel (exact order by
// (d2, index)).  Cells whose 27-cell set exceeds CAPC candidates, and rows longer than CAP, go to the todo list.
#define QC_WAVES 4
#define QC_QPW 16                   // queries per wavefront
#ifndef QC_CAPC
// candidates of a 27-cell set the LDS stage holds (mean ~50; larger sets go to the lane-per-query pass): 160 and 192 measure the
// same at 64 pairs per launch (28 vs 24 wavefronts per CU); 192 leaves fewer rows to that pass
#define QC_CAPC 192
#endif

// N queries of ONE cell against its staged candidates, 64 candidates per step: N independent chains of distance test, ballot
// compaction and bucket count per LDS read
template <int CAP, int N>
__device__ __forceinline__ void qc_pass(const float4* __restrict__ cd_lds, int total, const float (&q)[N][3], unsigned long long* const (&K)[N],
                                        int* const (&H)[N], int (&m)[N], float r2, float bin_scale, int lane)
{
#pragma unroll
    for (int u = 0; u < N; u++) m[u] = 0;
    for (int c0 = 0; c0 < total; c0 += WAVE) {
        const int ci = c0 + lane;
        const float4 cd = cd_lds[ci < total ? ci : 0];
        float d2[N];
        unsigned long long mask[N];
#pragma unroll
        for (int u = 0; u < N; u++) d2[u] = sqdist3(q[u][0], q[u][1], q[u][2], cd.x, cd.y, cd.z);
#pragma unroll
        for (int u = 0; u < N; u++) mask[u] = __ballot(ci < total && d2[u] < r2);
#pragma unroll
        for (int u = 0; u < N; u++) {
            const bool hit = ci < total && d2[u] < r2;
            const int pos = m[u] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[u], 0u));
            if (hit && pos < CAP) {
                K[u][pos] = ((unsigned long long)__float_as_uint(d2[u]) << 32) | (unsigned int)__float_as_int(cd.w);
                atomicAdd(&H[u][min((int)(d2[u] * bin_scale), QG - 1)], 1);
            }
            m[u] += __popcll(mask[u]);
        }
    }
}

template <int CAP, int CAPC>
__global__ void __launch_bounds__(QC_WAVES * WAVE) k_grid_query_cell(const CellGrid* __restrict__ grids, const int* __restrict__ table,
                                                                  const float4* __restrict__ sorted, const int* __restrict__ scell,
                                                                  const int* __restrict__ cruns,
                                                                  const int* __restrict__ q_off, float r2, float bin_scale, int k_out, int shadow,
                                                                  int* __restrict__ nbr_out, int* __restrict__ counts_out,
                                                                  int* __restrict__ max_count_out, int* __restrict__ todo,
                                                                  int* __restrict__ todo_n)
{
    __shared__ __attribute__((aligned(16))) float4 cand[QC_WAVES][CAPC];
    __shared__ __attribute__((aligned(16))) unsigned long long keys[QC_WAVES][QPW][CAP];       // arrival order, then bucket order in place
    __shared__ __attribute__((aligned(16))) int runs[QC_WAVES][20];                                // 9 starts | 9 lengths of the staged cell
    __shared__ int hist[QC_WAVES][QPW][QG];
    __shared__ int pref[QC_WAVES][QPW][QG + 4];
    __shared__ int cur[QC_WAVES][QPW][QG];
    const int lane = threadIdx.x & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int grp = lane / QG, l16 = lane & (QG - 1);
    int bx, b;
    grid2d_xcd_block(bx, b);
    const int e_lo = q_off[b], e_n = q_off[b + 1] - e_lo;
    const int base = (bx * QC_WAVES + w) * QC_QPW;
    if (base >= e_n) return;
    const int nq_w = min(QC_QPW, e_n - base);
    (void)grids; (void)table;            // (round 6: the neighbourhood of a cell comes from its record, not from the dense table)
    // lanes 0..15 (and their copies in the other rows) hold the wavefront's 16 queries
    const bool have = l16 < nq_w;
    const float4 me = have ? sorted[e_lo + base + l16] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int mycell = have ? scell[e_lo + base + l16] : -1;
    int staged = -2, total = 0;

    // the 27-cell candidate set of the cell whose run starts at row `c` -> cand[w][0 .. total): its record (k_cell_rank) is ONE 80-byte read
    auto stage = [&](int c) __attribute__((always_inline)) {
        int rec = 0;
        if (lane < 20) rec = cruns[20 * (size_t)c + lane];
        wave_sync();                         // the previous cell's candidates and runs have been read
        if (lane < 20) runs[w][lane] = rec;
        wave_sync();
        int st[9], pre[9];
        int tot = 0;
        {
            const int4* R4 = reinterpret_cast<const int4*>(runs[w]);
            const int4 a0 = R4[0], a1 = R4[1], a2 = R4[2], a3 = R4[3], a4 = R4[4];
            const int sv[9] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x };
            const int lv[9] = { a2.z, a2.w, a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z };
#pragma unroll
            for (int jj = 0; jj < 9; jj++) { pre[jj] = tot; st[jj] = sv[jj] - tot; tot += lv[jj]; }
        }
        total = __builtin_amdgcn_readfirstlane(tot);
        if (total <= CAPC) {
            for (int c0 = 0; c0 < total; c0 += 2 * WAVE) {
                float4 cv[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int ci = c0 + u * WAVE + lane;
                    int off = st[0];
#pragma unroll
                    for (int jj = 1; jj < 9; jj++) off = ci >= pre[jj] ? st[jj] : off;
                    cv[u] = ci < total ? sorted[ci + off] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int ci = c0 + u * WAVE + lane;
                    if (ci < total) cand[w][ci] = cv[u];
                }
            }
        }
        wave_sync();
    };

#pragma unroll 1
    for (int r = 0; r < QC_QPW / QPW; r++) {
        if (r * QPW >= nq_w) break;
        hist[w][grp][l16] = 0;
        wave_sync();
        int mq[QPW];
        unsigned stage_ovf = 0;                      // bit j: the candidate set of query j's cell does not fit the LDS stage
#pragma unroll
        for (int j = 0; j < QPW; j++) mq[j] = 0;
#pragma unroll
        for (int j = 0; j < QPW; j += 2) {           // a pair of queries per pass where they share their cell (the common case)
            const int q0 = r * QPW + j;
            if (q0 >= nq_w) continue;
            const int c0 = __builtin_amdgcn_readlane(mycell, q0);
            const int c1 = q0 + 1 < nq_w ? __builtin_amdgcn_readlane(mycell, q0 + 1) : -3;
            float qq[2][3];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                qq[u][0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.x), (q0 + u) & (QC_QPW - 1)));
                qq[u][1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.y), (q0 + u) & (QC_QPW - 1)));
                qq[u][2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(me.z), (q0 + u) & (QC_QPW - 1)));
            }
            if (c0 != staged) { staged = c0; stage(c0); }
            if (c1 == c0) {
                if (total > CAPC) { stage_ovf |= 3u << j; continue; }
                unsigned long long* const KK[2] = { keys[w][j], keys[w][j + 1] };
                int* const HH[2] = { hist[w][j], hist[w][j + 1] };
                int mm[2];
                qc_pass<CAP, 2>(cand[w], total, qq, KK, HH, mm, r2, bin_scale, lane);
                mq[j] = mm[0]; mq[j + 1] = mm[1];
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (u == 1) {
                        if (c1 == -3) break;
                        staged = c1; stage(c1);
                    }
                    if (total > CAPC) { stage_ovf |= 1u << (j + u); continue; }
                    const float q1[1][3] = { { qq[u][0], qq[u][1], qq[u][2] } };
                    unsigned long long* const KK[1] = { keys[w][j + u] };
                    int* const HH[1] = { hist[w][j + u] };
                    int mm[1];
                    qc_pass<CAP, 1>(cand[w], total, q1, KK, HH, mm, r2, bin_scale, lane);
                    mq[j + u] = mm[0];
                }
            }
        }
        // ---- the four slices: counts, bucket sort, rows (a 16-lane group per query, as k_grid_query_wave) ----
        const int qn = r * QPW + grp;
        const bool active = qn < nq_w;
        int m = grp == 0 ? mq[0] : (grp == 1 ? mq[1] : (grp == 2 ? mq[2] : mq[3]));
        const int qi = __float_as_int(__shfl(me.w, qn & (QC_QPW - 1), WAVE));
        const bool sovf = (stage_ovf >> grp) & 1u;
        if (active && l16 == 0 && !sovf) {
            if (counts_out) counts_out[qi] = m;
            if (max_count_out && m > 0) atomicMax(max_count_out, m);
        }
        const bool redo = sovf || (m > CAP && k_out > 0);   // stage or row overflow: redone (counts included) by the lane-per-query pass
        if (redo) {
            if (active && l16 == 0) todo[atomicAdd(todo_n, 1)] = qi;
            m = 0;
        }
        if (k_out == 0) { wave_sync(); continue; }
        int* row = nbr_out + (size_t)qi * k_out;
        int mmax = 0;
#pragma unroll
        for (int j = 0; j < QPW; j++) mmax = max(mmax, mq[j] > CAP ? 0 : mq[j]);
        wave_sync();
        {
            const int cnt = hist[w][grp][l16];
            const int ex = row16_excl_scan(cnt);
            pref[w][grp][l16] = ex;
            cur[w][grp][l16] = ex;
            if (l16 == QG - 1) pref[w][grp][QG] = ex + cnt;
        }
        wave_sync();
        unsigned long long* K = keys[w][grp];
        unsigned long long* B = K;                   // bucket order IN PLACE: every lane holds its keys in registers before the first scatter store
        const int rounds = (mmax + QG - 1) / QG;
        unsigned long long mine[CAP / QG];
        int blo[CAP / QG], bhi[CAP / QG], bks[CAP / QG];
#pragma unroll
        for (int rr = 0; rr < CAP / QG; rr++) {
            mine[rr] = 0ull; blo[rr] = 0; bhi[rr] = 0; bks[rr] = -1;
            if (rr < rounds) {
                const int i = rr * QG + l16;
                if (i < m) {
                    const unsigned long long key = K[i];
                    const int bk = min((int)(__uint_as_float((unsigned int)(key >> 32)) * bin_scale), QG - 1);
                    mine[rr] = key; bks[rr] = bk; blo[rr] = pref[w][grp][bk]; bhi[rr] = pref[w][grp][bk + 1];
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int rr = 0; rr < CAP / QG; rr++)
            if (rr < rounds && bks[rr] >= 0) B[atomicAdd(&cur[w][grp][bks[rr]], 1)] = mine[rr];
        wave_sync();
#pragma unroll
        for (int rr = 0; rr < CAP / QG; rr++) {
            if (rr < rounds) {
                int rank = blo[rr];
                for (int tt = 0; __any(blo[rr] + tt < bhi[rr]); tt++) {
                    const int p = blo[rr] + tt;
                    const unsigned long long o = B[p < bhi[rr] ? p : 0];
                    rank += (p < bhi[rr] && o < mine[rr]) ? 1 : 0;
                }
                if (rr * QG + l16 < m && rank < k_out) row[rank] = (int)(unsigned int)(mine[rr] & 0xffffffffu);
            }
        }
        if (active && !redo) {
            for (int i = m + l16; i < k_out; i += QG) row[i] = shadow;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------
extern "C" int64_t buf_grid_default_cells(int ns, int nb)
{
    int64_t per = nb > 0 ? ((int64_t)ns + nb - 1) / nb : ns;
    int64_t c = 16 * per + 65536;
    // keep the concatenated table addressable with int32
    int64_t cap = nb > 0 ? (int64_t)0x7fff0000 / nb : 0x7fff0000;
    return c < cap ? c : cap;
}

static void carve_grid(buf_grid_t* g, WsCarver& w, int ns, int nb, int64_t cells)
{
    g->desc = w.take<CellGrid>((size_t)nb);
    g->s_off = w.take<int>((size_t)nb + 1);
    g->table = w.take<int>((size_t)nb * (size_t)cells);
    g->sorted = w.take<float4>((size_t)(ns > 0 ? ns : 1));
    g->order = w.take<int>((size_t)(ns > 0 ? ns : 1));
    g->scan_tmp = w.take<int>(scan_tmp_ints());
}

struct GridExtra { float4* sorted_tmp; int* cell_of; int* q_off; int* todo_n; int* scell; int* cruns; };

static GridExtra carve_extra(WsCarver& w, int ns, int nb)
{
    GridExtra e;
    e.sorted_tmp = w.take<float4>((size_t)(ns > 0 ? ns : 1));
    e.cell_of = w.take<int>((size_t)(ns > 0 ? ns : 1));
    e.q_off = w.take<int>((size_t)nb + 1);
    e.todo_n = w.take<int>(64);
    e.scell = w.take<int>((size_t)(ns > 0 ? ns : 1));          // cell of every row of the cell-ordered stream (= first row of the cell's run)
    e.cruns = w.take<int>(20 * (size_t)(ns > 0 ? ns : 1));     // 27-cell neighbourhood records, one per occupied cell at 20 * its first row (80 B, written by k_cell_rank)
    return e;
}

extern "C" size_t buf_grid_ws_bytes(int ns, int nb, int64_t cells_per_elem)
{
    if (cells_per_elem <= 0) cells_per_elem = buf_grid_default_cells(ns, nb);
    buf_grid_t g;
    WsCarver w(nullptr, 0);
    carve_grid(&g, w, ns, nb, cells_per_elem);
    carve_extra(w, ns, nb);
    return w.used();
}

extern "C" int buf_grid_build(buf_grid_t* g, const float* supports, int ns, const int* s_batches_host, int nb,
                              float radius, int64_t cells_per_elem, void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(g && s_batches_host && ws, BUF_EINVAL, "buf_grid_build: null argument");
    BUF_REQUIRE(ns >= 0 && nb > 0, BUF_EINVAL, "buf_grid_build: ns=%d nb=%d", ns, nb);
    BUF_REQUIRE(ns == 0 || supports, BUF_EINVAL, "buf_grid_build: null supports");
    BUF_REQUIRE(radius == radius, BUF_EINVAL, "buf_grid_build: radius is NaN");
    if (cells_per_elem <= 0) cells_per_elem = buf_grid_default_cells(ns, nb);
    BUF_REQUIRE((int64_t)nb * cells_per_elem < 0x7fffffffLL, BUF_EINVAL, "buf_grid_build: table too large");
    memset(g, 0, sizeof(*g));
    WsCarver w(ws, ws_bytes);
    carve_grid(g, w, ns, nb, cells_per_elem);
    GridExtra ex = carve_extra(w, ns, nb);
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_grid_build: workspace %zu < %zu bytes", ws_bytes, w.used());
    g->ws = ws; g->ws_bytes = ws_bytes; g->ns = ns; g->nb = nb; g->cells_per_elem = cells_per_elem; g->radius = radius;
    g->supports = supports;

    int rc = upload_offsets(g->s_off, s_batches_host, nb, ns, "buf_grid_build", s);
    if (rc) return rc;
    size_t table_n = (size_t)nb * (size_t)cells_per_elem;
    BUF_CHECK_HIP(hipMemsetAsync(g->table, 0, sizeof(int) * table_n, s));
    k_grid_bbox<<<nb, BBOX_THREADS, 0, s>>>(supports, g->s_off, (CellGrid*)g->desc, (double)radius, (long long)cells_per_elem);
    if (ns > 0) {
        k_grid_count<<<cdiv(ns, 256), 256, 0, s>>>(supports, ns, g->s_off, nb, (const CellGrid*)g->desc, g->table, ex.cell_of);
        rc = exclusive_scan_i32(g->table, (long long)table_n, (int*)g->scan_tmp, nullptr, s);
        if (rc) return rc;
        k_cell_scatter<<<cdiv(ns, 256), 256, 0, s>>>(supports, ns, ex.cell_of, nullptr, g->table, ex.sorted_tmp);
        k_cell_rank<<<cdiv(ns, 256), 256, 0, s>>>(ex.cell_of, g->table, ex.sorted_tmp, ns, nullptr, (float4*)g->sorted, g->order, ex.scell,
                                                  ((const CellGrid*)g->desc)->dim, (long long)cells_per_elem, (int)(sizeof(CellGrid) / sizeof(int)), ex.cruns);
    }
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_grid_query(const buf_grid_t* g, const float* queries, int nq, const int* q_batches_host,
                              const int* q_order, float radius, int k_out, int* nbr_out, int* counts_out,
                              int* max_count_out, void* todo_ws, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(g && g->ws && q_batches_host, BUF_EINVAL, "buf_grid_query: null argument");
    // self query in cell order: index and coordinates of query t come from the cell-ordered array itself
    const int self_query = (q_order == g->order && queries == g->supports && nq == g->ns) ? 1 : 0;
    BUF_REQUIRE(nq >= 0 && k_out >= 0, BUF_EINVAL, "buf_grid_query: nq=%d k_out=%d", nq, k_out);
    BUF_REQUIRE(k_out == 0 || nbr_out, BUF_EINVAL, "buf_grid_query: null nbr_out");
    BUF_REQUIRE(radius <= g->radius, BUF_EINVAL, "buf_grid_query: radius %g > grid radius %g", radius, g->radius);
    if (nq == 0) return BUF_OK;
    BUF_REQUIRE(queries, BUF_EINVAL, "buf_grid_query: null queries");
    // q_off lives behind the grid in the same workspace
    WsCarver w(g->ws, g->ws_bytes);
    buf_grid_t tmp;
    carve_grid(&tmp, w, g->ns, g->nb, g->cells_per_elem);
    GridExtra ex = carve_extra(w, g->ns, g->nb);
    int rc = upload_offsets(ex.q_off, q_batches_host, g->nb, nq, "buf_grid_query", s);
    if (rc) return rc;
    float r2 = radius * radius;     // neighbors.cpp:228
    int blocks = cdiv(nq, WAVE);
    BUF_REQUIRE(todo_ws || k_out == 0, BUF_EINVAL, "buf_grid_query: todo workspace missing");
    // a caller-supplied order other than the grid's own may cross batch elements: those queries need the todo list
    const bool foreign_order = q_order && !self_query;
    BUF_REQUIRE(todo_ws || !foreign_order, BUF_EINVAL, "buf_grid_query: a q_order other than the grid's own needs the todo workspace");
    int* todo = (int*)todo_ws;
    BUF_CHECK_HIP(hipMemsetAsync(ex.todo_n, 0, sizeof(int), s));
    // algorithmic bytes of this launch: queries + supports + the index table
    TimedSpan span;
    bool timed = timing_begin(s, &span, 12.0 * nq + 12.0 * g->ns + 4.0 * (double)nq * k_out);
    // todo list (query ids of rows longer than QW_CAP, and -- cell-centric self queries, the count-only k_out == 0 pass included --
    // queries whose 27-cell candidate set exceeds the LDS stage): caller's workspace of nq ints; invariant: whenever q_order is the
    // grid's own order the workspace must hold nq ints, at k_out == 0 too (include/buffer_hip.h, todo_ws).
    // 2-D launch: y = batch element, x = chunks of 16 queries of the longest element (chunks past an element's end exit at once)
    int qmax = 0;
    for (int bb = 0; bb < g->nb; bb++) qmax = q_batches_host[bb] > qmax ? q_batches_host[bb] : qmax;
    if (qmax > 0) {
        BUF_REQUIRE(g->nb <= 65535, BUF_EINVAL, "buf_grid_query: %d batch elements (at most 65535 per call)", g->nb);
        dim3 grid2(cdiv(qmax, QW_QPB), g->nb);
        const float bin_scale = r2 > 0.f ? (float)QG / r2 : 0.f;           // distance bucket = floor(d2 * 16 / r2), monotone in d2
        static const bool query_centric = getenv("BUF_A2_QUERY_CENTRIC") != nullptr;      // development switch: the round-1..3 kernel for self queries too
        if (self_query && todo && !query_centric) {
            // cell-centric: 16 rows of the cell-ordered stream per wavefront, the candidates of a cell staged once in LDS
            dim3 gridc(cdiv(qmax, QC_WAVES * QC_QPW), g->nb);
            if (k_out > 32)
                k_grid_query_cell<2 * QW_CAP, 512><<<gridc, QC_WAVES * WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted, ex.scell,
                                                                                 ex.cruns, g->s_off, r2, bin_scale, k_out, g->ns, nbr_out, counts_out,
                                                                                 max_count_out, todo, ex.todo_n);
            else
                k_grid_query_cell<QW_CAP, QC_CAPC><<<gridc, QC_WAVES * WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted, ex.scell,
                                                                             ex.cruns, g->s_off, r2, bin_scale, k_out, g->ns, nbr_out, counts_out,
                                                                             max_count_out, todo, ex.todo_n);
        } else if (k_out > 32)
            k_grid_query_wave<2 * QW_CAP><<<grid2, QW_WAVES * WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted,
                                                                             queries, ex.q_off, q_order, self_query, r2, bin_scale, k_out, g->ns,
                                                                             nbr_out, counts_out, max_count_out, todo, ex.todo_n);
        else
            k_grid_query_wave<QW_CAP><<<grid2, QW_WAVES * WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted, queries,
                                                                         ex.q_off, q_order, self_query, r2, bin_scale, k_out, g->ns, nbr_out,
                                                                         counts_out, max_count_out, todo, ex.todo_n);
    }
    if (k_out > 0 || foreign_order || (self_query && todo)) {
        // fallback pass over the (normally empty) todo list; exits at once when the device-side count is 0.  It rewrites whole
        // rows and counts (the same values where the wave kernel already counted a long row)
        if (k_out <= 32)
            k_grid_query<32><<<blocks, WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted, queries, nq,
                                                    ex.q_off, g->nb, todo, r2, k_out, g->ns, nbr_out, counts_out, max_count_out, ex.todo_n);
        else
            k_grid_query<64><<<blocks, WAVE, 0, s>>>((const CellGrid*)g->desc, g->table, (const float4*)g->sorted, queries, nq,
                                                    ex.q_off, g->nb, todo, r2, k_out, g->ns, nbr_out, counts_out, max_count_out, ex.todo_n);
    }
    if (timed) timing_end(s, &span);          // the span covers the fallback pass too (stage / row overflows of the fast kernels are part of the operator's cost)
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_radius_neighbors(const float* queries, int nq, const float* supports, int ns,
                                    const int* q_batches_host, const int* s_batches_host, int nb, float radius,
                                    int k_out, int* nbr_out, int* counts_out, int* max_count_out,
                                    void* ws, size_t ws_bytes, void* stream)
{
    buf_grid_t g;
    int rc = buf_grid_build(&g, supports, ns, s_batches_host, nb, radius, 0, ws, ws_bytes, stream);
    if (rc) return rc;
    // the todo list of the query pass goes behind the grid in the same workspace
    size_t need = buf_grid_ws_bytes(ns, nb, 0);
    BUF_REQUIRE(ws_bytes >= need + sizeof(int) * (size_t)nq, BUF_EWORKSPACE, "buf_radius_neighbors: workspace %zu < %zu",
                ws_bytes, need + sizeof(int) * (size_t)nq);
    return buf_grid_query(&g, queries, nq, q_batches_host, nullptr, radius, k_out, nbr_out, counts_out, max_count_out,
                          (char*)ws + need, stream);
}
