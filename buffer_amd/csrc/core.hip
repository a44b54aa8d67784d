// Error state, device probe and the int32 exclusive scan shared by the cell-grid builders.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void buf_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* buf_last_error(void) { return g_err; }
extern "C" int buf_version(void) { return 600; }   // 600: round 6 (split-f16 kernels safe by construction, cell records of the A2 self query, A1 with 13 launches); 500: round 5 (no packed-fp32 instructions, per-element scaled 1-NN planes); 200: round 2 (batched entry points, Winograd descriptor CNN); 300: round 3 (filter tilings: N-tile
                                                   // groups for 32 / 64 channels, Winograd layers 1-5 of the cost net, compact voxel lookup table)

// ------------------------------------------------------------------------------------------
// Optional per-kernel timing for bench.py's roofline objects: HIP events recorded on the launch stream
// directly around the kernels below.  Off by default; profiling state only.
// kernel ids: BUF_TIMED_* of include/buffer_hip.h
#include <mutex>
#include <vector>
struct TimedSpan { hipEvent_t a, b; double bytes; int id; };   // bytes = algorithmic bytes (HBM kernels) or flops (MFMA kernels)
static std::mutex g_time_mu;
static std::vector<TimedSpan> g_spans;
static int g_timing_on = 0;

extern "C" void buf_timing_enable(int on) { std::lock_guard<std::mutex> l(g_time_mu); g_timing_on = on; }

static bool timing_begin(hipStream_t s, TimedSpan* sp, double bytes, int id = BUF_TIMED_GRID_QUERY)
{
    if (!g_timing_on) return false;
    sp->id = id;
    if (hipEventCreate(&sp->a) != hipSuccess || hipEventCreate(&sp->b) != hipSuccess) return false;
    sp->bytes = bytes;
    (void)hipEventRecord(sp->a, s);
    return true;
}

static void timing_end(hipStream_t s, TimedSpan* sp)
{
    (void)hipEventRecord(sp->b, s);
    std::lock_guard<std::mutex> l(g_time_mu);
    g_spans.push_back(*sp);
}

// Synchronises on the recorded events of kernel `id`; returns the number of launches collected and drops them.
extern "C" long long buf_timing_collect_kernel(int id, double* total_ms, double* total_work)
{
    std::lock_guard<std::mutex> l(g_time_mu);
    double ms = 0, by = 0;
    long long n = 0;
    std::vector<TimedSpan> keep;
    for (auto& sp : g_spans) {
        if (sp.id != id) { keep.push_back(sp); continue; }
        float t = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms += t; by += sp.bytes; n++; }
        (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
    }
    g_spans.swap(keep);
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = by;
    return n;
}

extern "C" long long buf_timing_collect(double* total_ms, double* total_bytes)
{
    return buf_timing_collect_kernel(BUF_TIMED_GRID_QUERY, total_ms, total_bytes);
}

static std::mutex g_grant_mu;
int grant_dynamic_lds(const void* kernel, size_t bytes, LdsGrant& g)
{
    int dev = 0;
    BUF_CHECK_HIP(hipGetDevice(&dev));
    BUF_REQUIRE(dev >= 0 && dev < BUF_MAX_DEVICES, BUF_EINVAL, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> l(g_grant_mu);
    if (g.bytes[dev] >= bytes) return BUF_OK;
    BUF_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g.bytes[dev] = bytes;
    return BUF_OK;
}

extern "C" int buf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        buf_set_error("hipGetDeviceCount -> %s", hipGetErrorString(e));
        return BUF_ENODEVICE;
    }
    return n;
}

// ------------------------------------------------------------------------------------------
// Exclusive scan: <=SCAN_BLOCKS workgroups each own a contiguous chunk; pass 1 reduces the
// chunks, one workgroup scans the chunk sums, pass 2 rescans each chunk with its offset.
// Reads the data twice and writes it once (12 B per element of HBM traffic).
// (Round 6 tried the single-launch chained form with decoupled look-back: on this chip every block's ticket is a same-address
// device-scope atomic, ~170 ns apiece behind eight L2s -- 1024 tickets cost more than the two launches they save (measured: 3 ms per
// scan with a CAS ticket); without tickets the form leans on in-order workgroup dispatch, a hang if that ever fails.  Three launches stay.)
#define SCAN_BLOCKS 1024
#define SCAN_THREADS 256

size_t scan_tmp_ints() { return SCAN_BLOCKS + 8; }

__device__ __forceinline__ int wave_incl_scan(int v)
{
    int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// inclusive scan across a 256-thread workgroup; returns inclusive value, *total = block sum
__device__ __forceinline__ int block_incl_scan(int v, int* total)
{
    __shared__ int wsum[SCAN_THREADS / WAVE];
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    int inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / WAVE; i++) {
        int s = wsum[i];
        if (i < w) add += s;
        tot += s;
    }
    *total = tot;
    return inc + add;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const int* __restrict__ data, long long n,
                                                            long long chunk, int* __restrict__ sums)
{
    long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > n) hi = n;
    int acc = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += SCAN_THREADS) acc += data[i];
    int tot;
    block_incl_scan(acc, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(int* __restrict__ sums, int nblocks,
                                                          int* __restrict__ total_out)
{
    // nblocks <= SCAN_BLOCKS = 4 * SCAN_THREADS
    int v[4], acc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int i = threadIdx.x * 4 + j;
        v[j] = i < nblocks ? sums[i] : 0;
        acc += v[j];
    }
    int tot;
    int inc = block_incl_scan(acc, &tot);
    int run = inc - acc;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int i = threadIdx.x * 4 + j;
        if (i < nblocks) sums[i] = run;
        run += v[j];
    }
    if (threadIdx.x == 0) {
        sums[SCAN_BLOCKS] = tot;
        if (total_out) *total_out = tot;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(int* __restrict__ data, long long n,
                                                           long long chunk, const int* __restrict__ sums)
{
    long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > n) hi = n;
    int run = sums[blockIdx.x];
    for (long long base = lo; base < hi; base += SCAN_THREADS) {
        long long i = base + threadIdx.x;
        int v = i < hi ? data[i] : 0;
        int tot;
        int inc = block_incl_scan(v, &tot);
        if (i < hi) data[i] = run + inc - v;
        run += tot;
        __syncthreads();
    }
}

int exclusive_scan_i32(int* data, long long n, int* tmp, int* total_out, hipStream_t stream)
{
    if (n <= 0) {
        if (total_out) BUF_CHECK_HIP(hipMemsetAsync(total_out, 0, sizeof(int), stream));
        return BUF_OK;
    }
    long long chunk = (n + SCAN_BLOCKS - 1) / SCAN_BLOCKS;
    chunk = (chunk + SCAN_THREADS - 1) / SCAN_THREADS * SCAN_THREADS;
    int nblocks = (int)((n + chunk - 1) / chunk);
    k_scan_reduce<<<nblocks, SCAN_THREADS, 0, stream>>>(data, n, chunk, tmp);
    k_scan_sums<<<1, SCAN_THREADS, 0, stream>>>(tmp, nblocks, total_out);
    k_scan_apply<<<nblocks, SCAN_THREADS, 0, stream>>>(data, n, chunk, tmp);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------
// Counting-sort tail shared by the cell grid (radius.hip) and the voxel grid (subsample.hip).
// `table` holds exclusive cell starts on entry to k_cell_scatter and inclusive cell ends after.
__global__ void __launch_bounds__(256) k_cell_scatter(const float* __restrict__ pts, int n, const int* __restrict__ cell_of,
                                                    const int* __restrict__ err, int* __restrict__ table,
                                                    float4* __restrict__ sorted)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (err && *err)) return;
    int pos = atomicAdd(&table[cell_of[i]], 1);
    sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

// The scatter order inside a cell depends on atomic arrival; every point ranks itself by input
// index inside its (small) cell run, which makes the cell-ordered stream deterministic and, for the
// voxel grid, puts every run in INPUT ORDER.
__global__ void __launch_bounds__(256) k_cell_rank(const int* __restrict__ cell_of, const int* __restrict__ table,
                                                 const float4* __restrict__ sorted_in, int n, const int* __restrict__ err,
                                                 float4* __restrict__ sorted_out, int* __restrict__ order_out,
                                                 int* __restrict__ scell_out = nullptr, const int* __restrict__ dims = nullptr,
                                                 long long cells_per_elem = 0, int dims_stride = 0, int* __restrict__ cruns = nullptr)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || (err && *err)) return;
    float4 me = sorted_in[p];
    int i = __float_as_int(me.w);
    int c = cell_of[i];
    int s = c == 0 ? 0 : table[c - 1], e = table[c];
    int rank = 0;
    for (int t = s; t < e; t++) rank += (__float_as_int(sorted_in[t].w) < i) ? 1 : 0;
    sorted_out[s + rank] = me;
    order_out[s + rank] = i;
    if (scell_out) {
        // the cell of every row of the cell-ordered stream, named by the FIRST ROW of the cell's run (round 6; rounds 4-5: packed cell
        // coordinates, from which k_grid_query_cell derived 18 scattered reads of the dense table at every cell change -- most of the
        // operator's HBM traffic).  The row that opens the run writes the cell's 27-cell neighbourhood ONCE as a record of 9 run starts |
        // 0 | 9 run lengths | 0 at cruns[20 * first row]: the query kernel stages a cell from one contiguous 80-byte read.
        scell_out[s + rank] = s;
        if (rank == 0) {
            const int b = (int)(c / cells_per_elem), cl = (int)(c - (long long)b * cells_per_elem);
            const int dx = dims[b * dims_stride], dy = dims[b * dims_stride + 1], dz = dims[b * dims_stride + 2];
            const int cx = cl % dx, t2 = cl / dx, cy = t2 % dy, cz = t2 / dy;
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, dx - 1);
            const long long toff = (long long)b * cells_per_elem;
            int rs[9], len[9];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int y = cy + (j % 3) - 1, z = cz + (j / 3) - 1;
                rs[j] = 0; len[j] = 0;
                if (y >= 0 && y < dy && z >= 0 && z < dz) {
                    const long long g0 = toff + x0 + (long long)dx * (y + (long long)dy * z);
                    rs[j] = g0 == 0 ? 0 : table[g0 - 1];
                    len[j] = table[g0 + (x1 - x0)] - rs[j];
                }
            }
            int4* rec = reinterpret_cast<int4*>(cruns + 20 * (size_t)s);
            rec[0] = make_int4(rs[0], rs[1], rs[2], rs[3]);
            rec[1] = make_int4(rs[4], rs[5], rs[6], rs[7]);
            rec[2] = make_int4(rs[8], 0, len[0], len[1]);
            rec[3] = make_int4(len[2], len[3], len[4], len[5]);
            rec[4] = make_int4(len[6], len[7], len[8], 0);
        }
    }
}

// Prefix offsets of a host batch-length array -> device, validated against the expected total.  The values travel
// BY VALUE in the kernel-argument block of a one-workgroup store kernel (128 per launch): no pageable-memory staging,
// no host-blocking copy, safe from several host threads on several streams.
#define OFFS_PER_LAUNCH 128
struct OffsetChunk { int v[OFFS_PER_LAUNCH]; };

__global__ void __launch_bounds__(OFFS_PER_LAUNCH) k_store_offsets(OffsetChunk c, int count, int* __restrict__ dst)
{
    if ((int)threadIdx.x < count) dst[threadIdx.x] = c.v[threadIdx.x];
}

static int upload_offsets(int* dev, const int* lens_host, int nb, int expect_total, const char* what, hipStream_t s)
{
    long long run = 0;
    for (int b = 0; b < nb; b++) {
        if (lens_host[b] < 0) { buf_set_error("%s: negative batch length", what); return BUF_EINVAL; }
        run += lens_host[b];
    }
    if (run != expect_total) {
        buf_set_error("%s: batch lengths sum to %lld, expected %d", what, run, expect_total);
        return BUF_EINVAL;
    }
    int off = 0;                                            // off = offsets[i0]
    for (int i0 = 0; i0 <= nb; i0 += OFFS_PER_LAUNCH) {
        OffsetChunk c;
        int cnt = nb + 1 - i0 < OFFS_PER_LAUNCH ? nb + 1 - i0 : OFFS_PER_LAUNCH;
        for (int j = 0; j < cnt; j++) { c.v[j] = off; if (i0 + j < nb) off += lens_host[i0 + j]; }
        k_store_offsets<<<1, OFFS_PER_LAUNCH, 0, s>>>(c, cnt, dev + i0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { buf_set_error("%s: offsets upload -> %s", what, hipGetErrorString(e)); return BUF_EHIP; }
    return BUF_OK;
}
