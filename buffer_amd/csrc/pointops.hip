// pointnet2_ops / knn_cuda operator surface on gfx950 (A6, A7, A8, A12, A18).
// Upstream CUDA sources are not part of the BUFFER repository (README.md:30-35); the semantics below
// are the documented/recalled ones restated in oracle/buffer_oracle.c.
#include "common.h"

// ------------------------------------------------------------------------------------------ A6
// furthest_point_sample: one 1024-lane workgroup per cloud, the cloud lives in registers
// (PPT points per lane), one workgroup barrier per round.
//
// Selection order reproduces the upstream kernel: temp starts at 1e10, points with
// x^2+y^2+z^2 <= 1e-3 (double compare) neither update nor compete, the arg-max prefers, on equal
// distance, the smaller (k mod T) and then the smaller k, T = min(512, 2^floor(log2 n)) being the
// upstream block size (per-thread strict '>' over k = t, t+T, ...; pairwise tree keeps the lower
// thread).  Keys are packed as (d2 bits << 32) | ~((k mod T) << 22 | k) and max-reduced.
#define FPS_THREADS 1024

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        unsigned int lo = __shfl_xor((unsigned int)v, d, WAVE);
        unsigned int hi = __shfl_xor((unsigned int)(v >> 32), d, WAVE);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

template <int PPT>
__global__ void __launch_bounds__(FPS_THREADS) k_fps(const float* __restrict__ xyz, int n, int m, int T,
                                                    int* __restrict__ idx_out)
{
    const float* P = xyz + (size_t)blockIdx.x * n * 3;
    int* out = idx_out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    float px[PPT], py[PPT], pz[PPT], temp[PPT];
    unsigned int tie[PPT];      // ~((k mod T) << 22 | k), 0 marks "never competes"
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        int k = tid + j * FPS_THREADS;
        bool ok = k < n;
        px[j] = ok ? P[3 * (size_t)k] : 0.f;
        py[j] = ok ? P[3 * (size_t)k + 1] : 0.f;
        pz[j] = ok ? P[3 * (size_t)k + 2] : 0.f;
        temp[j] = 1e10f;
        float mag = __fadd_rn(__fadd_rn(__fmul_rn(px[j], px[j]), __fmul_rn(py[j], py[j])), __fmul_rn(pz[j], pz[j]));
        ok = ok && !((double)mag <= 1e-3);
        tie[j] = ok ? ~((((unsigned int)(k % T)) << 22) | (unsigned int)k) : 0u;
    }
    __shared__ unsigned long long skey[2][FPS_THREADS / WAVE];
    __shared__ float sxyz[2][FPS_THREADS / WAVE][3];
    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = 0;
    for (int r = 1; r < m; r++) {
        unsigned long long best = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            float d = sqdist3(px[j], py[j], pz[j], x1, y1, z1);
            float d2 = fminf(d, temp[j]);
            if (tie[j]) {
                temp[j] = d2;
                unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | tie[j];
                if (key > best) { best = key; bx = px[j]; by = py[j]; bz = pz[j]; }
            }
        }
        unsigned long long wbest = wave_max_u64(best);
        int buf = r & 1;
        // exactly one lane holds the wave maximum (keys are unique per point) unless it is 0
        if (best == wbest && (wbest != 0 || lane == 0)) {
            skey[buf][w] = wbest;
            sxyz[buf][w][0] = bx; sxyz[buf][w][1] = by; sxyz[buf][w][2] = bz;
        }
        __syncthreads();
        unsigned long long g = 0;
        int gw = 0;
#pragma unroll
        for (int i = 0; i < FPS_THREADS / WAVE; i++) {
            unsigned long long v = skey[buf][i];
            if (v > g) { g = v; gw = i; }
        }
        int old;
        if (g == 0) {             // nobody competes: upstream returns index 0
            old = 0;
            x1 = P[0]; y1 = P[1]; z1 = P[2];
        } else {
            old = (int)((~(unsigned int)g) & 0x3fffffu);
            x1 = sxyz[buf][gw][0]; y1 = sxyz[buf][gw][1]; z1 = sxyz[buf][gw][2];
        }
        if (tid == 0) out[r] = old;
    }
}

// generic fallback for clouds that do not fit the register-resident kernel
__global__ void __launch_bounds__(FPS_THREADS) k_fps_global(const float* __restrict__ xyz, int n, int m, int T,
                                                           float* __restrict__ temp_all, int* __restrict__ idx_out)
{
    const float* P = xyz + (size_t)blockIdx.x * n * 3;
    float* temp = temp_all + (size_t)blockIdx.x * n;
    int* out = idx_out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    for (int k = tid; k < n; k += FPS_THREADS) temp[k] = 1e10f;
    __shared__ unsigned long long skey[2][FPS_THREADS / WAVE];
    int old = 0;
    if (tid == 0) out[0] = 0;
    for (int r = 1; r < m; r++) {
        float x1 = P[3 * (size_t)old], y1 = P[3 * (size_t)old + 1], z1 = P[3 * (size_t)old + 2];
        unsigned long long best = 0;
        for (int k = tid; k < n; k += FPS_THREADS) {
            float x = P[3 * (size_t)k], y = P[3 * (size_t)k + 1], z = P[3 * (size_t)k + 2];
            float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
            if ((double)mag <= 1e-3) continue;
            float d2 = fminf(sqdist3(x, y, z, x1, y1, z1), temp[k]);
            temp[k] = d2;
            unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) |
                                     (unsigned int)~((((unsigned int)(k % T)) << 22) | (unsigned int)k);
            best = key > best ? key : best;
        }
        unsigned long long wbest = wave_max_u64(best);
        int buf = r & 1;
        if (lane == 0) skey[buf][w] = wbest;
        __syncthreads();
        unsigned long long g = 0;
#pragma unroll
        for (int i = 0; i < FPS_THREADS / WAVE; i++) g = skey[buf][i] > g ? skey[buf][i] : g;
        old = g == 0 ? 0 : (int)((~(unsigned int)g) & 0x3fffffu);
        if (tid == 0) out[r] = old;
    }
}

static int fps_upstream_threads(int n)
{
    int p = 1;
    while (p * 2 <= n && p * 2 <= 512) p *= 2;
    return p;
}

extern "C" size_t buf_fps_ws_bytes(int b, int n) { return n > 32 * FPS_THREADS ? sizeof(float) * (size_t)b * n : 256; }

extern "C" int buf_fps(const float* xyz, int b, int n, int m, int* idx_out, void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(b >= 0 && n > 0 && m >= 0, BUF_EINVAL, "buf_fps: b=%d n=%d m=%d", b, n, m);
    BUF_REQUIRE(n < (1 << 22), BUF_EINVAL, "buf_fps: n=%d exceeds 2^22 points per cloud", n);
    if (b == 0 || m == 0) return BUF_OK;
    BUF_REQUIRE(xyz && idx_out, BUF_EINVAL, "buf_fps: null argument");
    int T = fps_upstream_threads(n);
    if (n <= 4 * FPS_THREADS) k_fps<4><<<b, FPS_THREADS, 0, s>>>(xyz, n, m, T, idx_out);
    else if (n <= 8 * FPS_THREADS) k_fps<8><<<b, FPS_THREADS, 0, s>>>(xyz, n, m, T, idx_out);
    else if (n <= 16 * FPS_THREADS) k_fps<16><<<b, FPS_THREADS, 0, s>>>(xyz, n, m, T, idx_out);
    else if (n <= 32 * FPS_THREADS) k_fps<32><<<b, FPS_THREADS, 0, s>>>(xyz, n, m, T, idx_out);
    else {
        BUF_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)b * n, BUF_EWORKSPACE, "buf_fps: workspace too small");
        k_fps_global<<<b, FPS_THREADS, 0, s>>>(xyz, n, m, T, (float*)ws, idx_out);
    }
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A7
// gather_operation: out[b,c,j] = feat[b,c,idx[b,j]];  grouping_operation: out[b,c,j,s] = feat[b,c,idx[b,j,s]]
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ feat, const int* __restrict__ idx, int c, int n,
                                              long long m, long long total, float* __restrict__ out)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    long long j = t % m;
    long long bc = t / m;
    long long b = bc / c;
    int k = idx[b * m + j];
    out[t] = feat[bc * n + k];
}

extern "C" int buf_gather(const float* feat, const int* idx, int b, int c, int n, int m, float* out, void* stream)
{
    BUF_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, BUF_EINVAL, "buf_gather: negative size");
    long long total = (long long)b * c * m;
    if (total == 0) return BUF_OK;
    BUF_REQUIRE(feat && idx && out, BUF_EINVAL, "buf_gather: null argument");
    k_gather<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(feat, idx, c, n, m, total, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_group(const float* feat, const int* idx, int b, int c, int n, int m, int nsample, float* out, void* stream)
{
    BUF_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0 && nsample >= 0, BUF_EINVAL, "buf_group: negative size");
    long long ms = (long long)m * nsample, total = (long long)b * c * ms;
    if (total == 0) return BUF_OK;
    BUF_REQUIRE(feat && idx && out, BUF_EINVAL, "buf_group: null argument");
    k_gather<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(feat, idx, c, n, ms, total, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A8
// ball_query: one wavefront per query scans the cloud 64 points at a time; accepted lanes get
// their output slot from ballot + popcount prefix, so hits land in INDEX ORDER; early exit at
// nsample; remaining slots = first hit; rows without a hit stay zero.
#define BQ_WAVES 4

__device__ __forceinline__ int lane_prefix(unsigned long long mask, int lane)
{
    return __popcll(mask & ((1ull << lane) - 1ull));
}

__global__ void __launch_bounds__(BQ_WAVES * WAVE) k_ball_query(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                              int n, int m, float r2, int nsample, int* __restrict__ idx)
{
    int b = blockIdx.y;
    int q = blockIdx.x * BQ_WAVES + threadIdx.x / WAVE;
    if (q >= m) return;
    int lane = threadIdx.x & (WAVE - 1);
    const float* P = xyz + (size_t)b * n * 3;
    const float* Q = new_xyz + ((size_t)b * m + q) * 3;
    int* row = idx + ((size_t)b * m + q) * nsample;
    float qx = Q[0], qy = Q[1], qz = Q[2];
    int cnt = 0, first = 0;
    for (int base = 0; base < n && cnt < nsample; base += WAVE) {
        int k = base + lane;
        bool hit = false;
        if (k < n) {
            float d2 = sqdist3(qx, qy, qz, P[3 * (size_t)k], P[3 * (size_t)k + 1], P[3 * (size_t)k + 2]);
            hit = d2 < r2;
        }
        unsigned long long mask = __ballot(hit);
        if (mask) {
            if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
            int slot = cnt + lane_prefix(mask, lane);
            if (hit && slot < nsample) row[slot] = k;
            cnt += __popcll(mask);
        }
    }
    if (cnt > nsample) cnt = nsample;
    int fill = cnt == 0 ? 0 : first;
    for (int j = cnt + lane; j < nsample; j += WAVE) row[j] = fill;
}

extern "C" int buf_ball_query(const float* xyz, const float* new_xyz, int b, int n, int m, float radius, int nsample,
                              int* idx, void* stream)
{
    BUF_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, BUF_EINVAL, "buf_ball_query: negative size");
    if ((long long)b * m * nsample == 0) return BUF_OK;
    BUF_REQUIRE(new_xyz && idx && (n == 0 || xyz), BUF_EINVAL, "buf_ball_query: null argument");
    float r2 = radius * radius;
    dim3 grid(cdiv(m, BQ_WAVES), b);
    k_ball_query<<<grid, BQ_WAVES * WAVE, 0, (hipStream_t)stream>>>(xyz, new_xyz, n, m, r2, nsample, idx);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// Fused select_patches (models/patch_embedder.py:93-121): ball query over the (already permuted)
// support cloud + grouping + the keypoint substitution, without materialising idx:
//   slot j < hits        -> j-th hit in index order
//   slot j >= hits       -> the keypoint   (reference: padding slots equal the first hit -> masked)
//   slot nsample-1       -> the keypoint, always
//   no hit at all        -> slot 0 = point 0 of the cloud (the zero-initialised index row)
__global__ void __launch_bounds__(BQ_WAVES * WAVE) k_select_patches(const float* __restrict__ pts, const float* __restrict__ kpts,
                                                                  int n, int m, float r2, int nsample,
                                                                  float* __restrict__ patches)
{
    int q = blockIdx.x * BQ_WAVES + threadIdx.x / WAVE;
    if (q >= m) return;
    int lane = threadIdx.x & (WAVE - 1);
    float qx = kpts[3 * (size_t)q], qy = kpts[3 * (size_t)q + 1], qz = kpts[3 * (size_t)q + 2];
    float* row = patches + (size_t)q * nsample * 3;
    int cnt = 0;
    for (int base = 0; base < n && cnt < nsample; base += WAVE) {
        int k = base + lane;
        bool hit = false;
        float x = 0, y = 0, z = 0;
        if (k < n) {
            x = pts[3 * (size_t)k]; y = pts[3 * (size_t)k + 1]; z = pts[3 * (size_t)k + 2];
            hit = sqdist3(qx, qy, qz, x, y, z) < r2;
        }
        unsigned long long mask = __ballot(hit);
        if (mask) {
            int slot = cnt + lane_prefix(mask, lane);
            if (hit && slot < nsample - 1) { row[3 * slot] = x; row[3 * slot + 1] = y; row[3 * slot + 2] = z; }
            cnt += __popcll(mask);
        }
    }
    if (cnt > nsample - 1) cnt = nsample - 1;
    if (cnt == 0 && nsample > 1) {
        if (lane == 0 && n > 0) { row[0] = pts[0]; row[1] = pts[1]; row[2] = pts[2]; }
        cnt = n > 0 ? 1 : 0;
    }
    for (int j = cnt + lane; j < nsample; j += WAVE) { row[3 * j] = qx; row[3 * j + 1] = qy; row[3 * j + 2] = qz; }
}

extern "C" int buf_select_patches(const float* pts, const float* kpts, int n, int m, float radius, int nsample,
                                  float* patches, void* stream)
{
    BUF_REQUIRE(n >= 0 && m >= 0 && nsample >= 1, BUF_EINVAL, "buf_select_patches: n=%d m=%d nsample=%d", n, m, nsample);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(kpts && patches && (n == 0 || pts), BUF_EINVAL, "buf_select_patches: null argument");
    float r2 = radius * radius;
    k_select_patches<<<cdiv(m, BQ_WAVES), BQ_WAVES * WAVE, 0, (hipStream_t)stream>>>(pts, kpts, n, m, r2, nsample, patches);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A18
__global__ void __launch_bounds__(256) k_three_nn(const float* __restrict__ unknown, const float* __restrict__ known,
                                                int n, int m, float* __restrict__ dist, int* __restrict__ idx)
{
    int b = blockIdx.y;
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* u = unknown + ((size_t)b * n + j) * 3;
    const float* K = known + (size_t)b * m * 3;
    float ux = u[0], uy = u[1], uz = u[2];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; k++) {
        float d = sqdist3(ux, uy, uz, K[3 * (size_t)k], K[3 * (size_t)k + 1], K[3 * (size_t)k + 2]);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
        else if (d < b3) { b3 = d; i3 = k; }
    }
    float* D = dist + ((size_t)b * n + j) * 3;
    int* I = idx + ((size_t)b * n + j) * 3;
    D[0] = sqrtf((float)b1); D[1] = sqrtf((float)b2); D[2] = sqrtf((float)b3);
    I[0] = i1; I[1] = i2; I[2] = i3;
}

extern "C" int buf_three_nn(const float* unknown, const float* known, int b, int n, int m, float* dist, int* idx, void* stream)
{
    BUF_REQUIRE(b >= 0 && n >= 0 && m >= 0, BUF_EINVAL, "buf_three_nn: negative size");
    if ((long long)b * n == 0) return BUF_OK;
    BUF_REQUIRE(unknown && dist && idx && (m == 0 || known), BUF_EINVAL, "buf_three_nn: null argument");
    dim3 grid(cdiv(n, 256), b);
    k_three_nn<<<grid, 256, 0, (hipStream_t)stream>>>(unknown, known, n, m, dist, idx);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A12
// knn_cuda.KNN(k, transpose_mode=True): brute force, ssd accumulated over the feature dims in order,
// ascending, ties keep the smaller reference index, Euclidean (sqrt) distances, int64 indices.
// One lane per query; reference rows are staged through LDS in tiles and read as broadcasts;
// the k best (ssd bits << 32 | index) keys sit in a lane-private LDS column.
#define KNN_TILE 64
#define KNN_MAXD 64

template <int KL>
__global__ void __launch_bounds__(WAVE) k_knn(const float* __restrict__ ref, const float* __restrict__ query, int n, int nq,
                                            int d, int k, float* __restrict__ dist, long long* __restrict__ idx)
{
    __shared__ float tile[KNN_TILE][KNN_MAXD];
    __shared__ unsigned long long col[KL][WAVE];
    int b = blockIdx.y;
    int lane = threadIdx.x;
    int q = blockIdx.x * WAVE + lane;
    bool active = q < nq;
    const float* R = ref + (size_t)b * n * d;
    const float* Q = query + ((size_t)b * nq + (active ? q : 0)) * d;
    float qv[KNN_MAXD];
#pragma unroll
    for (int c = 0; c < KNN_MAXD; c++) qv[c] = (c < d) ? Q[c] : 0.f;
    int have = 0;
    unsigned long long worst = ~0ull;
    for (int base = 0; base < n; base += KNN_TILE) {
        int cntt = min(KNN_TILE, n - base);
        __syncthreads();
        for (int t = lane; t < cntt * d; t += WAVE) tile[t / d][t % d] = R[(size_t)base * d + t];
        __syncthreads();
        for (int i = 0; i < cntt; i++) {
            float ssd = 0.f;
#pragma unroll
            for (int c = 0; c < KNN_MAXD; c++) {
                if (c < d) {
                    float t = __fsub_rn(tile[i][c], qv[c]);
                    ssd = __fadd_rn(ssd, __fmul_rn(t, t));
                }
            }
            unsigned long long key = ((unsigned long long)__float_as_uint(ssd) << 32) | (unsigned int)(base + i);
            if (have < k || key < worst) {
                int p = have < k ? have : k - 1;
                while (p > 0 && col[p - 1][lane] > key) { col[p][lane] = col[p - 1][lane]; p--; }
                col[p][lane] = key;
                if (have < k) have++;
                if (have == k) worst = col[k - 1][lane];
            }
        }
    }
    if (active) {
        for (int t = 0; t < k; t++) {
            size_t o = ((size_t)b * nq + q) * k + t;
            if (t < have) {
                unsigned long long key = col[t][lane];
                dist[o] = sqrtf(__uint_as_float((unsigned int)(key >> 32)));
                idx[o] = (long long)(key & 0xffffffffu);
            } else {
                dist[o] = __uint_as_float(0x7f800000u);
                idx[o] = 0;
            }
        }
    }
}

extern "C" int buf_knn(const float* ref, const float* query, int b, int n, int nq, int d, int k, float* dist,
                       long long* idx, void* stream)
{
    BUF_REQUIRE(b >= 0 && n >= 0 && nq >= 0 && d > 0 && k > 0, BUF_EINVAL, "buf_knn: b=%d n=%d q=%d d=%d k=%d", b, n, nq, d, k);
    BUF_REQUIRE(d <= KNN_MAXD, BUF_EINVAL, "buf_knn: feature dim %d > %d", d, KNN_MAXD);
    BUF_REQUIRE(k <= 64, BUF_EINVAL, "buf_knn: k=%d > 64", k);
    if ((long long)b * nq == 0) return BUF_OK;
    BUF_REQUIRE(query && dist && idx && (n == 0 || ref), BUF_EINVAL, "buf_knn: null argument");
    dim3 grid(cdiv(nq, WAVE), b);
    if (k <= 8) k_knn<8><<<grid, WAVE, 0, (hipStream_t)stream>>>(ref, query, n, nq, d, k, dist, idx);
    else k_knn<64><<<grid, WAVE, 0, (hipStream_t)stream>>>(ref, query, n, nq, d, k, dist, idx);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
