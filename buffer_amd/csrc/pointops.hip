// pointnet2_ops / knn_cuda operator surface on gfx950 (A6, A7, A8, A12, A18).
// Upstream CUDA sources are not part of the BUFFER repository (README.md:30-35); the semantics below
// are the documented/recalled ones restated in oracle/buffer_oracle.c.
#include "common.h"

// ------------------------------------------------------------------------------------------ A6
// furthest_point_sample: one 1024-lane workgroup per cloud, the cloud lives in registers
// (PPT points per lane), one workgroup barrier per round.
//
// Selection order reproduces the upstream kernel: temp starts at 1e10, points with
// x^2+y^2+z^2 <= 1e-3 (double compare) neither update nor compete.  T = min(512, 2^floor(log2 n)) is the
// upstream block size: thread t scans k = t, t+T, ... with a strict '>' (its first maximum wins), then the
// shared-memory tree folds slot t+s into slot t for s = T/2 .. 1 and keeps slot t on equality.  The LAST fold
// (s = 1) separates even from odd threads, the one before it bit 1, ...: among equal maxima the thread that is
// smallest in BIT-REVERSED order wins (T = 4, threads 1 and 2 tied -> thread 2).  Tie key, smaller wins:
// bitrev_log2T(k mod T) << 22 | k.
#define FPS_THREADS 512           // fallback kernel + the >16k-point register-resident variants
#define FPS_WAVES (FPS_THREADS / WAVE)
#define FPS_LDS_POINTS 13600      // clouds up to this size keep an xyz copy in LDS for the winner lookup (163 200 + 128 of the CU's 163 840 bytes)
#define FPS_MAXB 64

struct FpsBatch { int off[FPS_MAXB]; int n[FPS_MAXB]; };   // per-cloud row offset and length (ragged batch)

// wave64 max-reduction of a non-negative-or-(-1) float through DPP row shifts/broadcasts (no LDS traffic)
__device__ __forceinline__ unsigned int fps_tie_rank(unsigned int t, int T)      // bit reversal of t in log2(T) bits
{
    return T > 1 ? __brev(t) >> (__clz(T) + 1) : 0u;       // T = 2^b: clz = 31 - b, shift = 32 - b
}

// One instruction per step: the DPP operand rides on the v_max itself (the builtin form compiles to v_mov_dpp + a canonicalising v_max v, v, v +
// the v_max: 18 dependent instructions on the round's critical path instead of 6).  Lanes a step does not address are not written and keep v;
// the s_nop covers the two wait states between a vector write and a DPP read of the same register (the hazard recogniser does not see
// into inline assembly).
__device__ __forceinline__ float wave_max_f32(float v)
{
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// fminf() whose operands the compiler cannot prove quiet costs a canonicalising v_max_f32 v, v, v in front of the v_min_f32 (1 of the 13 vector
// instructions a slot of k_fps costs per round; the running minimum comes round the loop through a phi).  The instruction itself already
// returns the other operand for a NaN: same results.
__device__ __forceinline__ float min_f32_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float max_f32_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float max3_f32_raw(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// maximum of N registers as a tree of v_max3_f32 (25 values: 8 + 3 + 1 instructions, depth 3)
template <int N>
__device__ __forceinline__ float max_tree_f32(const float (&v)[N])
{
    if constexpr (N == 1) return v[0];
    else if constexpr (N == 2) return max_f32_raw(v[0], v[1]);
    else {
        constexpr int M = (N + 2) / 3;
        float t[M];
#pragma unroll
        for (int i = 0; i < M; i++)
            t[i] = 3 * i + 2 < N ? max3_f32_raw(v[3 * i], v[3 * i + 1], v[3 * i + 2])
                 : 3 * i + 1 < N ? max_f32_raw(v[3 * i], v[3 * i + 1]) : v[3 * i];
        return max_tree_f32<M>(t);
    }
}

// Per lane the candidates k = tid + j*FPS_THREADS share (k mod T) because T divides FPS_THREADS, so
// inside a lane the upstream tie rule reduces to "first j wins" = strict '>' in ascending j.
// Points the upstream kernel skips carry temp = -1, which never beats the initial best of -1.
// Cross-wave stage of a round: with the LDS copy of the cloud, one ordered 64-bit key per wavefront and a depth-3 tree of maxima (11 k points:
// 1.31 -> 1.18 us per round); WITHOUT the copy (13 600 < n <= 16 384: the winner's coordinates are a scalar load) the tree measured 1.85 us per
// round against 1.52 for the compare-and-branch chain over (distance, tie key) pairs, which stays there (tools/bench_ops.py fps, n = 15 000).
#ifndef FPS_KEY_REDUCE
#define FPS_KEY_REDUCE(lds) (lds)
#endif
template <int THREADS, int PPT, bool LDSPTS>
__global__ void __launch_bounds__(THREADS) k_fps(const float* __restrict__ xyz, FpsBatch B, int m, int* __restrict__ idx_out)
{
    constexpr int NW = THREADS / WAVE;
    extern __shared__ float spts[];                  // LDSPTS: xyz copy of the cloud (3*n floats)
    const int n = B.n[blockIdx.x];
    const float* P = xyz + (size_t)B.off[blockIdx.x] * 3;
    int* out = idx_out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    int T = 1;
    while (T * 2 <= n && T * 2 <= 512) T *= 2;
    float px[PPT], py[PPT], pz[PPT], temp[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        int k = tid + j * THREADS;
        bool ok = k < n;
        px[j] = ok ? P[3 * (size_t)k] : 0.f;
        py[j] = ok ? P[3 * (size_t)k + 1] : 0.f;
        pz[j] = ok ? P[3 * (size_t)k + 2] : 0.f;
        float mag = __fadd_rn(__fadd_rn(__fmul_rn(px[j], px[j]), __fmul_rn(py[j], py[j])), __fmul_rn(pz[j], pz[j]));
        ok = ok && !((double)mag <= 1e-3);
        temp[j] = ok ? 1e10f : -1.0f;
    }
    if (LDSPTS) {
        for (int i = tid; i < 3 * n; i += THREADS) spts[i] = P[i];
    }
    // a wavefront's round result as ONE ordered key: distance bits << 32 | ~tie key (distances are >= +0: their bit patterns order like the
    // values; 0 = no candidate).  The cross-wave arg-max is a depth-3 tree of 64-bit maxima instead of eight dependent compare-and-branch steps.
    __shared__ __attribute__((aligned(16))) unsigned long long skey[2][NW];
    __shared__ float sbest[2][NW];
    __shared__ unsigned int stie[2][NW];
    float x1 = n > 0 ? P[0] : 0.f, y1 = n > 0 ? P[1] : 0.f, z1 = n > 0 ? P[2] : 0.f;
    if (tid == 0 && m > 0) out[0] = 0;
    const unsigned int tmod = fps_tie_rank((unsigned int)(tid % T), T) << 22;
    for (int r = 1; r < m; r++) {
        // distances of all PPT points are independent; the arg-max is a balanced tree (max, then the
        // first j that attains it), so no serial compare/select chain sits on the critical path
        float d2[PPT];
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            d2[j] = min_f32_raw(sqdist3(px[j], py[j], pz[j], x1, y1, z1), temp[j]);
            temp[j] = d2[j];
        }
        const float best = max_tree_f32<PPT>(d2);
        int bj = PPT;
#pragma unroll
        for (int j = PPT - 1; j >= 0; j--) bj = d2[j] == best ? j : bj;   // independent compares, short select chain
        float wmax = wave_max_f32(best);
        const int buf = r & 1;
        if (wmax >= 0.f) {
            // smaller tie key wins among equal distances: bitrev(k mod T) << 22 | k
            unsigned int tk = tmod | (unsigned int)(tid + bj * THREADS);
            unsigned long long cand = __ballot(best == wmax);
            int win;
            if (__popcll(cand) == 1) win = __ffsll((long long)cand) - 1;
            else {
                unsigned int v = best == wmax ? tk : 0xffffffffu;
                for (int d = WAVE / 2; d > 0; d >>= 1) v = min(v, (unsigned int)__shfl_xor((int)v, d, WAVE));
                win = __ffsll((long long)__ballot(best == wmax && tk == v)) - 1;
            }
            if (lane == win) {
                if (FPS_KEY_REDUCE(LDSPTS)) skey[buf][w] = ((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned int)~tk;
                else { sbest[buf][w] = wmax; stie[buf][w] = tk; }
            }
        } else if (lane == 0) {
            if (FPS_KEY_REDUCE(LDSPTS)) skey[buf][w] = 0ull;
            else { sbest[buf][w] = -1.0f; stie[buf][w] = 0xffffffffu; }
        }
        __syncthreads();
        int old;
        if (FPS_KEY_REDUCE(LDSPTS)) {
            unsigned long long kk[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) kk[i] = skey[buf][i];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2 <<= 1)
#pragma unroll
                for (int i = 0; i + w2 < NW; i += 2 * w2) kk[i] = kk[i] > kk[i + w2] ? kk[i] : kk[i + w2];
            old = kk[0] == 0ull ? 0 : (int)(~(unsigned int)kk[0] & 0x3fffffu);      // nobody competing -> index 0
        } else {
            float g = -1.0f;
            unsigned int gt = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < NW; i++) {
                float v = sbest[buf][i];
                unsigned int t = stie[buf][i];
                bool better = v > g || (v == g && t < gt);
                g = better ? v : g; gt = better ? t : gt;
            }
            old = g < 0.f ? 0 : (int)(gt & 0x3fffffu);
        }
        // winner's coordinates: one broadcast read of the cloud (LDS copy, or L2 resident through the scalar cache)
        if (LDSPTS) { const unsigned int o3 = __umul24((unsigned int)old, 3u); x1 = spts[o3]; y1 = spts[o3 + 1]; z1 = spts[o3 + 2]; }   // (full-rate multiply)
        else {      // the index is wavefront-uniform: said so, the lookup is a scalar load (as a vector load it cost 0.3 us more per round)
            const size_t o3 = 3 * (size_t)__builtin_amdgcn_readfirstlane(old);
            x1 = P[o3]; y1 = P[o3 + 1]; z1 = P[o3 + 2];
        }
        if (tid == 0) out[r] = old;
    }
}

// generic fallback for clouds that do not fit the register-resident kernel (temp in global memory)
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

__global__ void __launch_bounds__(FPS_THREADS) k_fps_global(const float* __restrict__ xyz, FpsBatch B, int m,
                                                           float* __restrict__ temp_all, int* __restrict__ idx_out)
{
    const int n = B.n[blockIdx.x];
    const float* P = xyz + (size_t)B.off[blockIdx.x] * 3;
    float* temp = temp_all + (size_t)B.off[blockIdx.x];
    int* out = idx_out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    int T = 1;
    while (T * 2 <= n && T * 2 <= 512) T *= 2;
    for (int k = tid; k < n; k += FPS_THREADS) temp[k] = 1e10f;
    __shared__ unsigned long long skey[2][FPS_WAVES];
    int old = 0;
    if (tid == 0 && m > 0) out[0] = 0;
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
                                     (unsigned int)~((fps_tie_rank((unsigned int)(k % T), T) << 22) | (unsigned int)k);
            best = key > best ? key : best;
        }
        unsigned long long wbest = wave_max_u64(best);
        int buf = r & 1;
        if (lane == 0) skey[buf][w] = wbest;
        __syncthreads();
        unsigned long long g = 0;
#pragma unroll
        for (int i = 0; i < FPS_WAVES; i++) g = skey[buf][i] > g ? skey[buf][i] : g;
        old = g == 0 ? 0 : (int)((~(unsigned int)g) & 0x3fffffu);
        if (tid == 0) out[r] = old;
    }
}

#define FPS_MAX_RESIDENT (32 * FPS_THREADS)

extern "C" size_t buf_fps_ws_bytes(int b, int n) { return n > FPS_MAX_RESIDENT ? sizeof(float) * (size_t)b * n : 256; }

// Ragged batch: clouds stacked in xyz f32[sum(n),3], lengths_host int[b] -> idx int32[b,m] (indices local to a cloud).
extern "C" int buf_fps_ragged(const float* xyz, const int* lengths_host, int b, int m, int* idx_out, void* ws, size_t ws_bytes,
                              void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(b >= 0 && m >= 0, BUF_EINVAL, "buf_fps: b=%d m=%d", b, m);
    if (b == 0 || m == 0) return BUF_OK;
    BUF_REQUIRE(xyz && idx_out && lengths_host, BUF_EINVAL, "buf_fps: null argument");
    for (int c0 = 0, row0 = 0; c0 < b; c0 += FPS_MAXB) {
        FpsBatch B;
        int nb = b - c0 < FPS_MAXB ? b - c0 : FPS_MAXB, nmax = 0, row = row0;
        for (int i = 0; i < nb; i++) {
            int n = lengths_host[c0 + i];
            BUF_REQUIRE(n > 0 && n < (1 << 22), BUF_EINVAL, "buf_fps: cloud %d has %d points (need 1..2^22-1)", c0 + i, n);
            B.off[i] = row; B.n[i] = n; row += n;
            nmax = n > nmax ? n : nmax;
        }
        int* out = idx_out + (size_t)c0 * m;
        TimedSpan span;
        bool timed = timing_begin(s, &span, (double)m, BUF_TIMED_FPS);            // work = rounds; bytes follow from the lengths
        // <= FPS_LDS_POINTS points: an xyz copy in LDS serves the winner lookup (one broadcast ds_read instead of an L2 round trip)
        const size_t lds = sizeof(float) * 3 * (size_t)nmax;
#define FPS_LAUNCH(TH, PPT_, L) \
    do { if (L) { static LdsGrant grant_; if (int rc_ = grant_dynamic_lds((const void*)k_fps<TH, PPT_, L>, FPS_LDS_POINTS * 12, grant_)) return rc_; } \
         k_fps<TH, PPT_, L><<<nb, TH, (L) ? lds : 0, s>>>(xyz, B, m, out); } while (0)
        if (nmax <= 4 * FPS_THREADS) FPS_LAUNCH(FPS_THREADS, 4, true);
        else if (nmax <= 8 * FPS_THREADS) FPS_LAUNCH(FPS_THREADS, 8, true);
        else if (nmax <= 16 * FPS_THREADS) FPS_LAUNCH(FPS_THREADS, 16, true);
        else if (nmax <= 25 * FPS_THREADS) FPS_LAUNCH(FPS_THREADS, 25, true);
        else if (nmax <= FPS_LDS_POINTS) FPS_LAUNCH(FPS_THREADS, 32, true);
        else if (nmax <= 32 * FPS_THREADS) FPS_LAUNCH(FPS_THREADS, 32, false);
#undef FPS_LAUNCH
        else {
            BUF_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)row, BUF_EWORKSPACE, "buf_fps: workspace too small");
            k_fps_global<<<nb, FPS_THREADS, 0, s>>>(xyz, B, m, (float*)ws, out);
        }
        if (timed) timing_end(s, &span);
        row0 = row;
    }
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_fps(const float* xyz, int b, int n, int m, int* idx_out, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(b >= 0 && n > 0 && m >= 0, BUF_EINVAL, "buf_fps: b=%d n=%d m=%d", b, n, m);
    if (b == 0) return BUF_OK;
    int stackl[64];
    int* lens = b <= 64 ? stackl : (int*)malloc(sizeof(int) * (size_t)b);
    for (int i = 0; i < b; i++) lens[i] = n;
    int rc = buf_fps_ragged(xyz, lens, b, m, idx_out, ws, ws_bytes, stream);
    if (lens != stackl) free(lens);
    return rc;
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
struct __attribute__((packed, aligned(4))) Pt3 { float x, y, z; };     // 12-byte point, dword aligned

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
            const Pt3 p = *reinterpret_cast<const Pt3*>(pts + 3 * (size_t)k);      // one global_load_dwordx3
            x = p.x; y = p.y; z = p.z;
            hit = sqdist3(qx, qy, qz, x, y, z) < r2;
        }
        unsigned long long mask = __ballot(hit);
        if (mask) {
            int slot = cnt + lane_prefix(mask, lane);
            if (hit && slot < nsample - 1) { Pt3 o; o.x = x; o.y = y; o.z = z; *reinterpret_cast<Pt3*>(row + 3 * slot) = o; }
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
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 12.0 * n + 12.0 * m + 12.0 * m * nsample, BUF_TIMED_SELECT_PATCHES);
    k_select_patches<<<cdiv(m, BQ_WAVES), BQ_WAVES * WAVE, 0, (hipStream_t)stream>>>(pts, kpts, n, m, r2, nsample, patches);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ---- A8 batched: every cloud of a step in ONE launch ----------------------------------------------------------------
// A keypoint's ball (r = 0.3 m on a 2.5 cm cloud) holds 450-900 of ~23 000 points: often fewer than the 511 the patch
// keeps, so the index-ordered brute-force scan above reads the WHOLE cloud for such queries.  Here the ball is found
// through the A2 cell grid built over the stacked (permuted) clouds: the candidates of the 27 cells around the keypoint
// (~2 000) are tested and every in-ball point sets ITS bit in a per-query LDS bitmask over the cloud's indices; the mask is
// then walked in index order (popcount prefix over the lanes), which restores exactly the "first 511 in index order"
// semantics.  Where the ball is so dense that the ordered scan would stop early anyway (KITTI: des_r = 3 m), the kernel
// keeps the brute-force scan; the choice is per query, made from the candidate count, and changes speed only.
#define SPB_WAVES 4
#define SPB_MAX_MASK_BYTES 16384            // per query: clouds up to 131 072 points take the mask path

// pseudo-random permutation of [0, n): 4-round Feistel network on an even number of bits >= log2 n, cycle-walked into range
__device__ __forceinline__ unsigned int prp_hash(unsigned int x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ unsigned int prp_index(unsigned int i, unsigned int n, int half_bits, unsigned long long key)
{
    const unsigned int hm = (1u << half_bits) - 1u;
    unsigned int v = i;
    do {
        unsigned int l = v >> half_bits, r = v & hm;
#pragma unroll
        for (int rd = 0; rd < 4; rd++) {
            const unsigned int f = prp_hash(r ^ (unsigned int)(key >> (16 * rd)) ^ (0x9e3779b9u * (rd + 1))) & hm;
            const unsigned int nl = r;
            r = l ^ f;
            l = nl;
        }
        v = (l << half_bits) | r;
    } while (v >= n);
    return v;
}

#define PERM_MAXC 64
struct PermBatch { const float* src[PERM_MAXC]; int off[PERM_MAXC + 1]; unsigned long long key[PERM_MAXC]; };

// out[off[c] + j] = src_c[prp_c(j)]: cloud c shuffled by its own keyed permutation (patch_embedder.py:97-98's randperm)
__global__ void __launch_bounds__(256) k_permute_clouds(PermBatch B, int nc, float* __restrict__ out)
{
    const int c = blockIdx.y;
    const int n = B.off[c + 1] - B.off[c];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int hb = 1;
    while ((1u << (2 * hb)) < (unsigned int)n) hb++;
    const unsigned int k = prp_index((unsigned int)j, (unsigned int)n, hb, B.key[c]);
    const Pt3 p = *reinterpret_cast<const Pt3*>(B.src[c] + 3 * (size_t)k);
    *reinterpret_cast<Pt3*>(out + 3 * ((size_t)B.off[c] + j)) = p;
}

// clouds_host: nc HOST pointers to DEVICE clouds f32[n_c,3]; keys_host: nc 64-bit permutation keys -> out f32[sum n_c, 3]
extern "C" int buf_permute_clouds(const float* const* clouds_host, const int* lengths_host, const unsigned long long* keys_host,
                                  int nc, float* out, void* stream)
{
    BUF_REQUIRE(nc >= 0, BUF_EINVAL, "buf_permute_clouds: nc=%d", nc);
    if (nc == 0) return BUF_OK;
    BUF_REQUIRE(clouds_host && lengths_host && keys_host && out, BUF_EINVAL, "buf_permute_clouds: null argument");
    long long base = 0;
    for (int c0 = 0; c0 < nc; c0 += PERM_MAXC) {
        PermBatch B;
        const int cnt = nc - c0 < PERM_MAXC ? nc - c0 : PERM_MAXC;
        int nmax = 0;
        B.off[0] = 0;
        for (int i = 0; i < cnt; i++) {
            const int n = lengths_host[c0 + i];
            BUF_REQUIRE(n >= 0 && n < (1 << 30) && (n == 0 || clouds_host[c0 + i]), BUF_EINVAL, "buf_permute_clouds: cloud %d (n=%d)", c0 + i, n);
            B.src[i] = clouds_host[c0 + i]; B.key[i] = keys_host[c0 + i]; B.off[i + 1] = B.off[i] + n;
            nmax = n > nmax ? n : nmax;
        }
        if (nmax > 0) k_permute_clouds<<<dim3(cdiv(nmax, 256), cnt), 256, 0, (hipStream_t)stream>>>(B, cnt, out + 3 * base);
        base += B.off[cnt];
    }
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

#define SPB_SCAN 4
// One wavefront per keypoint; blockIdx.y = cloud.  kpts: m keypoints per cloud, stacked; patches f32[nc*m, nsample, 3].
__global__ void __launch_bounds__(SPB_WAVES * WAVE) k_select_patches_grid(const CellGrid* __restrict__ grids, const int* __restrict__ table,
                                                                      const float4* __restrict__ sorted, const float* __restrict__ pts,
                                                                      const int* __restrict__ s_off, const float* __restrict__ kpts, int m,
                                                                      float r2, int nsample, int mask_words, float* __restrict__ patches)
{
    extern __shared__ unsigned long long spb_mask[];             // [SPB_WAVES][mask_words]
    // XCD-contiguous block order (common.h): the keypoints of one cloud go to ONE XCD, whose L2 then holds that cloud's
    // cell-ordered points (every keypoint tests ~2000 of them) instead of all eight L2s fetching every cloud
    const int lb = xcd_contiguous_block(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int c = lb / (int)gridDim.x;
    const int w = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    const int q = (lb - c * (int)gridDim.x) * SPB_WAVES + w;
    if (q >= m) return;
    const int lo = s_off[c], n = s_off[c + 1] - lo;              // scalar loads: the cloud's rows in the stacked array
    const float* P = pts + 3 * (size_t)lo;
    const float* Q = kpts + 3 * ((size_t)c * m + q);
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    float* row = patches + ((size_t)c * m + q) * nsample * 3;
    const int keep = nsample - 1;
    const CellGrid g = grids[c];
    // the 9 cell runs around the keypoint (lanes 0..8), flattened
    int rs = 0, len = 0;
    {
        double fx = floor(((double)qx - (double)g.mn[0]) * g.inv_cell);
        double fy = floor(((double)qy - (double)g.mn[1]) * g.inv_cell);
        double fz = floor(((double)qz - (double)g.mn[2]) * g.inv_cell);
        fx = fmin(fmax(fx, -2.0), (double)g.dim[0] + 1.0);
        fy = fmin(fmax(fy, -2.0), (double)g.dim[1] + 1.0);
        fz = fmin(fmax(fz, -2.0), (double)g.dim[2] + 1.0);
        const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        const int y = cy + (lane % 3) - 1, z = cz + (lane / 3) - 1;
        if (lane < 9 && x0 <= x1 && y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) {
            const int g0 = g.table_off + x0 + g.dim[0] * (y + g.dim[1] * z);
            rs = g0 == 0 ? 0 : table[g0 - 1];
            len = table[g0 + (x1 - x0)] - rs;
        }
    }
    int st[9], pre[9], total = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int sj = __builtin_amdgcn_readlane(rs, j), lj = __builtin_amdgcn_readlane(len, j);
        pre[j] = total; st[j] = sj - total; total += lj;
    }
    int cnt = 0;
    // Path choice (speed only).  Expected in-ball points ~ 0.4 * candidates on surface-like clouds; the ordered scan stops
    // after about keep / that * n indices at ~25 instructions per 64 indices, the grid path tests every candidate at ~35 per 64.
    const float in_ball = 0.4f * (float)total;
    const bool grid_path = mask_words > 0 && (in_ball <= (float)keep || (float)keep / in_ball * (float)n * 25.f > (float)total * 35.f + 64.f * 300.f);
    if (grid_path) {
        unsigned long long* M = spb_mask + (size_t)w * mask_words;
        const int nw = (n + 63) >> 6;
        for (int i = lane; i < nw; i += WAVE) M[i] = 0ull;
        wave_sync();
        // run by run (each a contiguous piece of the cell-ordered array): no per-candidate search for the run a flat candidate
        // number falls into (8 compare + select pairs of the ~35 instructions per 64 candidates), at the price of a partly
        // filled last step per run
#pragma unroll 1
        for (int j = 0; j < 9; j++) {
            const int sj = __builtin_amdgcn_readlane(rs, j), lj = __builtin_amdgcn_readlane(len, j);
            for (int c0 = lane; c0 < lj; c0 += WAVE) {
                const float4 p = sorted[sj + c0];
                if (sqdist3(qx, qy, qz, p.x, p.y, p.z) < r2) {
                    const int k = __float_as_int(p.w) - lo;                  // index inside the cloud
                    atomicOr(&M[k >> 6], 1ull << (k & 63));
                }
            }
        }
        wave_sync();
        // ordered emission: lane l owns the words [l*wpl, (l+1)*wpl); slot of its first hit = hits in the lanes below
        const int wpl = (nw + WAVE - 1) / WAVE;
        int mine = 0;
        for (int i = 0; i < wpl; i++) { const int wi = lane * wpl + i; mine += wi < nw ? __popcll(M[wi]) : 0; }
        int incl = mine;
        for (int dd = 1; dd < WAVE; dd <<= 1) { const int o = __shfl_up(incl, dd, WAVE); if (lane >= dd) incl += o; }
        int slot = incl - mine;
        cnt = __builtin_amdgcn_readlane(incl, WAVE - 1);
        for (int i = 0; i < wpl && slot < keep; i++) {
            const int wi = lane * wpl + i;
            unsigned long long bits = wi < nw ? M[wi] : 0ull;
            while (bits && slot < keep) {
                const int bpos = __ffsll((long long)bits) - 1;
                bits &= bits - 1ull;
                const Pt3 p = *reinterpret_cast<const Pt3*>(P + 3 * (size_t)(wi * 64 + bpos));
                *reinterpret_cast<Pt3*>(row + 3 * slot) = p;
                slot++;
            }
        }
    } else {
        // index-ordered scan with early exit, SPB_SCAN blocks of 64 indices per step: their loads are independent of the count and
        // go out together (one block per step left every step waiting on its own L2 round trip)
        for (int base = 0; base < n && cnt < keep; base += SPB_SCAN * WAVE) {
            Pt3 p[SPB_SCAN];
            bool hit[SPB_SCAN];
#pragma unroll
            for (int u = 0; u < SPB_SCAN; u++) {
                const int k = base + u * WAVE + lane;
                p[u] = Pt3{ 0.f, 0.f, 0.f };
                if (k < n) p[u] = *reinterpret_cast<const Pt3*>(P + 3 * (size_t)k);
            }
#pragma unroll
            for (int u = 0; u < SPB_SCAN; u++) {
                const int k = base + u * WAVE + lane;
                hit[u] = k < n && sqdist3(qx, qy, qz, p[u].x, p[u].y, p[u].z) < r2;
            }
#pragma unroll
            for (int u = 0; u < SPB_SCAN; u++) {
                const unsigned long long mask = __ballot(hit[u]);
                if (mask) {
                    const int slot = cnt + lane_prefix(mask, lane);
                    if (hit[u] && slot < keep) *reinterpret_cast<Pt3*>(row + 3 * slot) = p[u];
                    cnt += __popcll(mask);
                }
            }
        }
    }
    if (cnt > keep) cnt = keep;
    if (cnt == 0 && nsample > 1) {                                           // empty ball: slot 0 = point 0 of the cloud
        if (lane == 0 && n > 0) { row[0] = P[0]; row[1] = P[1]; row[2] = P[2]; }
        cnt = n > 0 ? 1 : 0;
    }
    for (int j = cnt + lane; j < nsample; j += WAVE) { row[3 * j] = qx; row[3 * j + 1] = qy; row[3 * j + 2] = qz; }
}

extern "C" size_t buf_select_patches_batched_ws_bytes(int n_total, int nc) { return buf_grid_ws_bytes(n_total, nc > 0 ? nc : 1, 0) + 256; }

// pts f32[sum n_c, 3]: the (already permuted) support clouds stacked; lengths_host int[nc]; kpts f32[nc*m,3]: m keypoints per
// cloud -> patches f32[nc*m, nsample, 3].  One grid build + one launch for all clouds.
extern "C" int buf_select_patches_batched(const float* pts, const int* lengths_host, int nc, const float* kpts, int m, float radius,
                                          int nsample, float* patches, void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(nc >= 0 && m >= 0 && nsample >= 1 && radius > 0.f, BUF_EINVAL, "buf_select_patches_batched: nc=%d m=%d nsample=%d radius=%g", nc, m, nsample, radius);
    if (nc == 0 || m == 0) return BUF_OK;
    BUF_REQUIRE(lengths_host && kpts && patches && ws, BUF_EINVAL, "buf_select_patches_batched: null argument");
    BUF_REQUIRE(nc <= 65535, BUF_EINVAL, "buf_select_patches_batched: %d clouds (at most 65535 per call)", nc);
    long long ntot = 0;
    int nmax = 0;
    for (int c = 0; c < nc; c++) {
        BUF_REQUIRE(lengths_host[c] >= 0, BUF_EINVAL, "buf_select_patches_batched: negative length");
        ntot += lengths_host[c];
        nmax = lengths_host[c] > nmax ? lengths_host[c] : nmax;
    }
    BUF_REQUIRE(ntot < 0x7fffffffLL && (ntot == 0 || pts), BUF_EINVAL, "buf_select_patches_batched: points");
    BUF_REQUIRE(ws_bytes >= buf_select_patches_batched_ws_bytes((int)ntot, nc), BUF_EWORKSPACE, "buf_select_patches_batched: workspace %zu < %zu",
                ws_bytes, buf_select_patches_batched_ws_bytes((int)ntot, nc));
    buf_grid_t g;
    int rc = buf_grid_build(&g, pts, (int)ntot, lengths_host, nc, radius, 0, ws, ws_bytes, stream);
    if (rc) return rc;
    int mask_words = (nmax + 63) / 64;
    if ((size_t)mask_words * 8 > SPB_MAX_MASK_BYTES) mask_words = 0;         // huge clouds: ordered scan only
    const size_t lds = (size_t)SPB_WAVES * mask_words * 8;
    static LdsGrant grant;
    if (lds > 48 * 1024)
        if (int rc2 = grant_dynamic_lds((const void*)k_select_patches_grid, lds, grant)) return rc2;
    TimedSpan span;
    bool timed = timing_begin(s, &span, 12.0 * ntot + (12.0 + 12.0 * nsample) * (double)nc * m, BUF_TIMED_SELECT_PATCHES);
    k_select_patches_grid<<<dim3(cdiv(m, SPB_WAVES), nc), SPB_WAVES * WAVE, lds, s>>>((const CellGrid*)g.desc, g.table, (const float4*)g.sorted, pts,
                                                                                 g.s_off, kpts, m, radius * radius, nsample, mask_words, patches);
    if (timed) timing_end(s, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A18
// Four lanes per query, each scanning every fourth known point of an LDS tile; the three best of the four lanes are merged by (distance,
// index) order, which is what the upstream sequential scan with its strict '<' keeps (round 6: one lane per query reading the known points
// from global memory, fp64 compares and three-way branches took 275 us for 10 055 queries x 1500 points -- 157 wavefronts on the whole chip).
#define TNN_PARTS 4
#define TNN_TILE 2048
__device__ __forceinline__ void tnn_insert(float d, int k, bool lt1, bool lt2, bool lt3, float& b1, float& b2, float& b3, int& i1, int& i2, int& i3)
{
    // the list is sorted: lt1 implies lt2 implies lt3
    b3 = lt2 ? b2 : (lt3 ? d : b3); i3 = lt2 ? i2 : (lt3 ? k : i3);
    b2 = lt1 ? b1 : (lt2 ? d : b2); i2 = lt1 ? i1 : (lt2 ? k : i2);
    b1 = lt1 ? d : b1;              i1 = lt1 ? k : i1;
}

__global__ void __launch_bounds__(256) k_three_nn(const float* __restrict__ unknown, const float* __restrict__ known,
                                                int n, int m, float* __restrict__ dist, int* __restrict__ idx)
{
    __shared__ float tile[3 * TNN_TILE];
    const int b = blockIdx.y, tid = threadIdx.x, part = tid & (TNN_PARTS - 1);
    const int j = blockIdx.x * (256 / TNN_PARTS) + tid / TNN_PARTS;
    const bool active = j < n;
    const float* u = unknown + ((size_t)b * n + (active ? j : 0)) * 3;
    const float* K = known + (size_t)b * m * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float inf = __int_as_float(0x7f800000);          // (the upstream 1e40 as a float: a distance has to be below it to enter the list)
    float b1 = inf, b2 = inf, b3 = inf;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += TNN_TILE) {
        const int cnt = min(TNN_TILE, m - base);
        __syncthreads();
        for (int t = tid; t < 3 * cnt; t += 256) tile[t] = K[3 * (size_t)base + t];
        __syncthreads();
#pragma unroll 4
        for (int k = part; k < cnt; k += TNN_PARTS) {
            const float d = sqdist3(ux, uy, uz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]);
            tnn_insert(d, base + k, d < b1, d < b2, d < b3, b1, b2, b3, i1, i2, i3);
        }
    }
#pragma unroll
    for (int mask = 1; mask < TNN_PARTS; mask <<= 1) {      // the partner's three, in its order, by (distance, index)
        const float o1 = __shfl_xor(b1, mask), o2 = __shfl_xor(b2, mask), o3 = __shfl_xor(b3, mask);
        const int k1 = __shfl_xor(i1, mask), k2 = __shfl_xor(i2, mask), k3 = __shfl_xor(i3, mask);
#define TNN_LEX(d, k, bb, ii) ((d) < (bb) || ((d) == (bb) && (k) < (ii)))
        tnn_insert(o1, k1, TNN_LEX(o1, k1, b1, i1), TNN_LEX(o1, k1, b2, i2), TNN_LEX(o1, k1, b3, i3), b1, b2, b3, i1, i2, i3);
        tnn_insert(o2, k2, TNN_LEX(o2, k2, b1, i1), TNN_LEX(o2, k2, b2, i2), TNN_LEX(o2, k2, b3, i3), b1, b2, b3, i1, i2, i3);
        tnn_insert(o3, k3, TNN_LEX(o3, k3, b1, i1), TNN_LEX(o3, k3, b2, i2), TNN_LEX(o3, k3, b3, i3), b1, b2, b3, i1, i2, i3);
#undef TNN_LEX
    }
    if (!active || part != 0) return;
    float* D = dist + ((size_t)b * n + j) * 3;
    int* I = idx + ((size_t)b * n + j) * 3;
    D[0] = sqrtf(b1); D[1] = sqrtf(b2); D[2] = sqrtf(b3);
    I[0] = i1; I[1] = i2; I[2] = i3;
}

extern "C" int buf_three_nn(const float* unknown, const float* known, int b, int n, int m, float* dist, int* idx, void* stream)
{
    BUF_REQUIRE(b >= 0 && n >= 0 && m >= 0, BUF_EINVAL, "buf_three_nn: negative size");
    if ((long long)b * n == 0) return BUF_OK;
    BUF_REQUIRE(unknown && dist && idx && (m == 0 || known), BUF_EINVAL, "buf_three_nn: null argument");
    dim3 grid(cdiv(n, 256 / TNN_PARTS), b);
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

// k = 1 fast path (the mutual-matching call sites, models/BUFFER.py:347,352): grid = query tiles x reference
// splits; 4 wavefronts of a workgroup share 64 queries (lane = query, vector in registers) and each scans a
// quarter of the split's LDS-staged reference rows (float4 broadcast reads).  ssd is accumulated over the
// feature dims in order without contraction, exactly like the generic kernel; the per-query winner
// (ssd bits << 32 | index) is min-reduced through LDS and one 64-bit atomicMin per query and split.
#define NN1_SPLIT 256

template <int D>
__global__ void __launch_bounds__(256) k_nn1(const float* __restrict__ ref, const float* __restrict__ query, int n, int nq,
                                           unsigned long long* __restrict__ best)
{
    __shared__ float4 tile[NN1_SPLIT * (D / 4)];
    __shared__ unsigned long long red[4][WAVE];
    // XCD-contiguous block order: the blocks of one batch element (a cloud pair's descriptors, 640 KB) share an L2
    const int gxy = gridDim.x * gridDim.y;
    const int lb = xcd_contiguous_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gxy * gridDim.z);
    const int b = lb / gxy, by = (lb - b * gxy) / (int)gridDim.x, bx = lb - b * gxy - by * (int)gridDim.x;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int q = bx * WAVE + lane;
    const int base = by * NN1_SPLIT;
    const int cnt = min(NN1_SPLIT, n - base);
    const float4* R4 = reinterpret_cast<const float4*>(ref + ((size_t)b * n + base) * D);
    for (int t = threadIdx.x; t < cnt * (D / 4); t += 256) tile[t] = R4[t];
    float qv[D];
    const float* Q = query + ((size_t)b * nq + (q < nq ? q : 0)) * D;
#pragma unroll
    for (int c = 0; c < D; c++) qv[c] = Q[c];
    __syncthreads();
    unsigned long long mine = ~0ull;
    for (int i = w; i < cnt; i += 4) {
        float ssd = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; c4++) {
            float4 r = tile[i * (D / 4) + c4];
            float t0 = __fsub_rn(r.x, qv[4 * c4]);     ssd = __fadd_rn(ssd, __fmul_rn(t0, t0));
            float t1 = __fsub_rn(r.y, qv[4 * c4 + 1]); ssd = __fadd_rn(ssd, __fmul_rn(t1, t1));
            float t2 = __fsub_rn(r.z, qv[4 * c4 + 2]); ssd = __fadd_rn(ssd, __fmul_rn(t2, t2));
            float t3 = __fsub_rn(r.w, qv[4 * c4 + 3]); ssd = __fadd_rn(ssd, __fmul_rn(t3, t3));
        }
        unsigned long long key = ((unsigned long long)__float_as_uint(ssd) << 32) | (unsigned int)(base + i);
        mine = key < mine ? key : mine;
    }
    red[w][lane] = mine;
    __syncthreads();
    if (w == 0 && q < nq) {
        unsigned long long m = red[0][lane];
        for (int i = 1; i < 4; i++) m = red[i][lane] < m ? red[i][lane] : m;
        atomicMin(&best[(size_t)b * nq + q], m);
    }
}

__global__ void __launch_bounds__(256) k_nn1_finish(const unsigned long long* __restrict__ best, long long total,
                                                  float* __restrict__ dist, long long* __restrict__ idx)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    unsigned long long key = best[t];
    if (key == ~0ull) { dist[t] = __uint_as_float(0x7f800000u); idx[t] = 0; return; }
    dist[t] = sqrtf(__uint_as_float((unsigned int)(key >> 32)));
    idx[t] = (long long)(key & 0xffffffffu);
}

// ---- k = 1, d = 32 through the f16 matrix pipe (round 4) ---------------------------------------------------------------------
// The exact kernel above spends ~100 vector instructions per (query, reference) pair.  Here a split-f16 MFMA form ranks the
// pairs and only the ones it cannot separate from the best go through the exact sum:
//   S(q, r) = |r|^2 - 2 q.r   (|q|^2 is the same for every r)  ~  ssd(q, r) - |q|^2  to within eps,
//   three v_mfma_f32_16x16x32_f16 per 16 references x 16 queries (hi.hi, hi.lo, lo.hi of x = hi + lo, both f16; -2 folded into
//   the query planes, |r|^2 -- summed in fp64 -- is the accumulator's start value),
//   pass 1: min_r S per query (half a v_min3 per pair and lane); pass 2: every pair with S <= min + 2 eps is a CANDIDATE and gets
//   the reference's own sum (same operations in the same order as k_nn1) and the same 64-bit atomicMin key.
// The reference's arg-min r* has S(r*) <= S(r) + 2 eps for every r (eps >= error of S + error of the fp32 sum), so it is among the
// candidates together with everything that ties with it; the key then decides exactly as k_nn1 does.
// Round 5: the planes hold sc x with sc an exact power of two per BATCH ELEMENT that brings the element's largest norm
// (max over its queries and references, k_nn1f_norm) into [0.7, 1.42): the ranking S' = sc^2 S is that of S, and the bound no longer
// depends on the magnitude of the data (unscaled, the f16 planes of inputs with norms below ~1e-2 sank into the f16 subnormals:
// absolute plane error 2^-25 per component against an eps that shrinks like the norm squared).
// eps = 8e-6 (|q|max + |r|max)^2 + 6e-7 (|q|max + |r|max) in the scaled domain: the fp32 sum's own rounding (34 x 2^-24 d^2), the
// 22-bit operand forms (2^-22 |q||r| incl. the dropped lo.lo) and 99 products + 3 partial sums accumulated in fp32 are the quadratic
// term; the f16 low parts (subnormal below 2^-14) and any high part of a small row carry an ABSOLUTE error <= 2^-25 per component,
// i.e. <= 2^-25 sqrt(32) (|r| + 2 |q|) in S: the linear term (rows much smaller than the element's largest).
// Inputs that are not finite or beyond 1e15 in norm (flag `bad`), and query groups with more candidates than NNF_BUDGET
// (clouds of duplicated descriptors), go to k_nn1f_fallback, the exact scan for just those groups.
typedef _Float16 nnh8 __attribute__((ext_vector_type(8)));
typedef float nnf4 __attribute__((ext_vector_type(4)));
#define NNF_QT 4                       // query tiles of 16 per wavefront
#define NNF_QW (16 * NNF_QT)           // queries per wavefront
#define NNF_BUDGET 2048                // exact sums per wavefront before its group is handed to the fallback kernel

struct NnfHdr {                        // device header at the head of the workspace (zeroed per call)
    unsigned bad;                      // a non-finite input, or a norm beyond 1e15
    unsigned nfall;                    // query groups handed to the fallback kernel
};
// per batch element, behind the header: max |q|^2 and max |r|^2 as float bits (non-negative floats order as integers)
struct NnfMax { unsigned q_bits, r_bits; };

// the element's plane scale: the power of two 2^k with (2^k)^2 max(|q|^2max, |r|^2max) in [0.5, 2)
__device__ __forceinline__ float nnf_scale(const NnfMax mx)
{
    const float m = fmaxf(__uint_as_float(mx.q_bits), __uint_as_float(mx.r_bits));
    if (!(m > 0.f)) return 1.f;
    int ex;
    frexpf(m, &ex);                                              // m = f 2^ex, f in [0.5, 1)
    int k = -(ex >> 1);                                          // floor(ex / 2)
    k = k > 60 ? 60 : (k < -60 ? -60 : k);
    return ldexpf(1.f, k);
}

// one thread per row of the references and of the queries: |x|^2 (fp64 sum, rounded once) -> the element's maxima
__global__ void __launch_bounds__(256) k_nn1f_norm(const float* __restrict__ ref, const float* __restrict__ query, int b, int n, int nq,
                                                  NnfMax* __restrict__ emax, unsigned* __restrict__ bad)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nr = (long long)b * n, total = nr + (long long)b * nq;
    if (t - (threadIdx.x & (WAVE - 1)) >= total) return;         // (whole wavefronts past the end; the last one keeps its idle lanes)
    const bool valid = t < total;
    const long long tt = valid ? t : total - 1;
    const bool isq = tt >= nr;
    const long long row = isq ? tt - nr : tt;
    const int e = (int)(row / (isq ? nq : n));
    const float4* src = reinterpret_cast<const float4*>((isq ? query : ref) + (size_t)row * 32);
    double nn = 0.0;
#pragma unroll
    for (int c4 = 0; c4 < 8; c4++) {
        const float4 v = src[c4];
        nn = fma((double)v.x, (double)v.x, nn); nn = fma((double)v.y, (double)v.y, nn);      // (exact products: rounds like mul + add)
        nn = fma((double)v.z, (double)v.z, nn); nn = fma((double)v.w, (double)v.w, nn);
    }
    const float nf = (float)nn;
    const bool isbad = !(nf <= 1e30f);                           // also catches NaN / infinity
    if (__any(isbad)) { if (isbad) atomicOr(bad, 1u); }
    // one atomic per wavefront, not per row: 10 000 same-address device-scope atomics per element serialise behind the L2s
    // (measured: 1.7 ms for 32 x (5000 + 5000) rows); the rows of a wavefront nearly always share their (element, side) slot
    unsigned* slot = isq ? &emax[e].q_bits : &emax[e].r_bits;
    unsigned v = isbad ? 0u : __float_as_uint(nf);
    const unsigned long long mine = (unsigned long long)(size_t)slot;
    const unsigned long long slot0 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mine) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mine >> 32)) << 32);
    if (__all(mine == slot0)) {
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, d, WAVE));
        if ((threadIdx.x & (WAVE - 1)) == 0 && v) atomicMax(slot, v);
    } else if (v) atomicMax(slot, v);
}

// one thread per row: x f32[32] -> planes hi = f16(s x), lo = f16(s x - hi) with s = scale * nnf_scale(element) (row-major, 64 B per
// row), sc^2 |x|^2 (fp64 sum; rows >= n of the padded tail: zero planes and a norm no minimum will pick)
__global__ void __launch_bounds__(256) k_nn1f_prep(const float* __restrict__ x, int b, int n, int npad, float scale,
                                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                  float* __restrict__ norm, const NnfMax* __restrict__ emax)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)b * npad) return;
    const int e = (int)(t / npad), i = (int)(t - (long long)e * npad);
    uint4 h4[4], l4[4];
    double nn = 0.0;
    const float sc = nnf_scale(emax[e]);
    if (i < n) {
        const float4* src = reinterpret_cast<const float4*>(x + ((size_t)e * n + i) * 32);
        _Float16 h[32], l[32];
        const float s2 = scale * sc;
#pragma unroll
        for (int c4 = 0; c4 < 8; c4++) {
            const float4 v = src[c4];
            const float a[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                nn = fma((double)a[j], (double)a[j], nn);
                const float sv = a[j] * s2;                      // (power of two: exact unless the product leaves the fp32 range -- `bad` rows)
                h[4 * c4 + j] = (_Float16)sv;
                l[4 * c4 + j] = (_Float16)(sv - (float)h[4 * c4 + j]);
            }
        }
        memcpy(h4, h, 64); memcpy(l4, l, 64);
    } else {
        for (int j = 0; j < 4; j++) { h4[j] = make_uint4(0, 0, 0, 0); l4[j] = make_uint4(0, 0, 0, 0); }
    }
    uint4* ho = reinterpret_cast<uint4*>(hi + (size_t)t * 32);
    uint4* lw = reinterpret_cast<uint4*>(lo + (size_t)t * 32);
    for (int j = 0; j < 4; j++) { ho[j] = h4[j]; lw[j] = l4[j]; }
    if (norm) norm[t] = i < n ? (float)(nn * (double)sc * (double)sc) : 3e38f;
}

// the reference's sum, exactly as k_nn1 forms it (two 16-byte pieces of each row in flight: the register budget of the sweep)
__device__ __forceinline__ unsigned long long nn1_exact_key(const float* __restrict__ r, const float* __restrict__ q, int idx)
{
    const float4* r4 = reinterpret_cast<const float4*>(r);
    const float4* q4 = reinterpret_cast<const float4*>(q);
    float ssd = 0.f;
#pragma unroll 2
    for (int c4 = 0; c4 < 8; c4++) {
        const float4 a = r4[c4], b = q4[c4];
        float t0 = __fsub_rn(a.x, b.x); ssd = __fadd_rn(ssd, __fmul_rn(t0, t0));
        float t1 = __fsub_rn(a.y, b.y); ssd = __fadd_rn(ssd, __fmul_rn(t1, t1));
        float t2 = __fsub_rn(a.z, b.z); ssd = __fadd_rn(ssd, __fmul_rn(t2, t2));
        float t3 = __fsub_rn(a.w, b.w); ssd = __fadd_rn(ssd, __fmul_rn(t3, t3));
    }
    return ((unsigned long long)__float_as_uint(ssd) << 32) | (unsigned int)idx;
}

__global__ void __launch_bounds__(256) k_nn1f_sweep(const float* __restrict__ ref, const float* __restrict__ query, int n, int nq,
                                                   int npad, int nqpad, const unsigned short* __restrict__ rhi,
                                                   const unsigned short* __restrict__ rlo, const float* __restrict__ rnorm,
                                                   const unsigned short* __restrict__ qhi, const unsigned short* __restrict__ qlo,
                                                   NnfHdr* __restrict__ hdr, const NnfMax* __restrict__ emax, unsigned* __restrict__ fall,
                                                   unsigned long long* __restrict__ best, int b)
{
    if (hdr->bad) return;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int groups = nqpad / NNF_QW;
    const int task = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + w;     // (batch element, query group): the groups of one element on ONE XCD (its planes stay in that L2: -2 % per call, round 6)
    const int e = task / groups, g = task - e * groups;
    if (e >= b) return;
    const int col = lane & 15, kg = lane >> 4;
#ifndef NNF_EPS_SCALE
#define NNF_EPS_SCALE 1.f          // development builds (-DNNF_EPS_SCALE=...): how far the bound can shrink before a result changes
#endif
    const NnfMax mx = emax[e];
    const float sc = nnf_scale(mx);
    const float nsum = sc * (sqrtf(__uint_as_float(mx.q_bits)) + sqrtf(__uint_as_float(mx.r_bits)));     // in the scaled domain: [0.7, 2.9)
    const float eps = NNF_EPS_SCALE * (8e-6f * nsum * nsum + 6e-7f * nsum);
    nnh8 Bh[NNF_QT], Bl[NNF_QT];
#pragma unroll
    for (int j = 0; j < NNF_QT; j++) {
        const size_t row = (size_t)e * nqpad + (size_t)g * NNF_QW + 16 * j + col;
        Bh[j] = *reinterpret_cast<const nnh8*>(qhi + row * 32 + 8 * kg);
        Bl[j] = *reinterpret_cast<const nnh8*>(qlo + row * 32 + 8 * kg);
    }
    const nnh8* Ah = reinterpret_cast<const nnh8*>(rhi + ((size_t)e * npad + col) * 32 + 8 * kg);
    const nnh8* Al = reinterpret_cast<const nnh8*>(rlo + ((size_t)e * npad + col) * 32 + 8 * kg);
    const nnf4* Nr = reinterpret_cast<const nnf4*>(rnorm + (size_t)e * npad) + kg;
    const int nt = npad / 16;
    // reference tiles two ahead in registers (A planes + norms: 12 registers per tile; the loads are L2 round trips)
    nnh8 ah[3], al[3];
    nnf4 nr[3];
    auto fetch = [&](int t, int slot) __attribute__((always_inline)) {
        const int tt = t < nt ? t : nt - 1;
        ah[slot] = Ah[(size_t)tt * 64]; al[slot] = Al[(size_t)tt * 64];      // 16 rows x 64 B = 64 nnh8 per tile
        nr[slot] = Nr[(size_t)tt * 4];
    };
    auto tile = [&](int slot, nnf4 (&acc)[NNF_QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NNF_QT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[slot], Bh[j], nr[slot], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NNF_QT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[slot], Bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NNF_QT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[slot], Bh[j], acc[j], 0, 0, 0);
    };
    // pass 1: the smallest S per query (a lane sees references 4 kg + r of every tile for query column `col`)
    float m[NNF_QT];
#pragma unroll
    for (int j = 0; j < NNF_QT; j++) m[j] = 3.4e38f;
    fetch(0, 0); fetch(1, 1);
    for (int t0 = 0; t0 < nt; t0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (t0 + u >= nt) break;
            fetch(t0 + u + 2, (u + 2) % 3);
            nnf4 acc[NNF_QT];
            tile(u, acc);
#pragma unroll
            for (int j = 0; j < NNF_QT; j++) m[j] = fminf(fminf(fminf(m[j], acc[j][0]), acc[j][1]), fminf(acc[j][2], acc[j][3]));
        }
    }
#pragma unroll
    for (int j = 0; j < NNF_QT; j++) {
        m[j] = fminf(m[j], __shfl_xor(m[j], 16, WAVE));
        m[j] = fminf(m[j], __shfl_xor(m[j], 32, WAVE));
        m[j] += 2.f * eps;                                       // the candidate threshold of query 16 j + col
    }
    // pass 2: candidates (query, reference) into a queue of the wavefront; whenever 64 are waiting every lane forms one reference
    // sum (same operations, same order as k_nn1) and its 64-bit key
    __shared__ unsigned queue[4][128];
    unsigned* qw = queue[w];
    int nqueued = 0, rounds = 0;
    auto drain = [&](int count) __attribute__((always_inline)) {          // the first `count` (<= 64) entries
        if (lane < count) {
            const unsigned c = qw[lane];
            const int qi = g * NNF_QW + (int)(c >> 26), ri = (int)(c & 0x3ffffffu);
            if (ri < n) {
            const unsigned long long key = nn1_exact_key(ref + ((size_t)e * n + ri) * 32, query + ((size_t)e * nq + qi) * 32, ri);
            atomicMin(&best[(size_t)e * nq + qi], key);
            }
        }
        rounds++;
    };
    bool over = false;
    unsigned qvalid = 0u;                                        // queries of the padded tail (zero planes: every reference ties) take no part
#pragma unroll
    for (int j = 0; j < NNF_QT; j++) qvalid |= (g * NNF_QW + 16 * j + col < nq ? 0xfu : 0u) << (4 * j);
    fetch(0, 0); fetch(1, 1);
    for (int t0 = 0; t0 < nt && !over; t0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int t = t0 + u;
            if (t >= nt) break;
            fetch(t + 2, (u + 2) % 3);
            nnf4 acc[NNF_QT];
            tile(u, acc);
            unsigned hits = 0u;                                  // bit 4 j + r: reference 16 t + 4 kg + r is a candidate of query 16 j + col
#pragma unroll
            for (int j = 0; j < NNF_QT; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) hits |= (acc[j][r] <= m[j] ? 1u : 0u) << (4 * j + r);
            hits &= qvalid;
            while (__any(hits != 0u)) {                          // one candidate per lane and round into the queue
                const bool has = hits != 0u;
                const unsigned long long bal = __ballot(has);
                if (has) {
                    const int bi = __ffs(hits) - 1;
                    hits &= hits - 1u;
                    const int qi = 16 * (bi >> 2) + col, ri = 16 * t + 4 * kg + (bi & 3);
                    const int slot = nqueued + __popcll(bal & ((1ull << lane) - 1ull));
                    if (g * NNF_QW + qi < nq && ri < n) qw[slot] = ((unsigned)qi << 26) | (unsigned)ri;
                    else qw[slot] = ((unsigned)(16 * (bi >> 2) + col) << 26) | 0x3ffffffu;          // padding rows: dropped at the drain
                }
                nqueued += __popcll(bal);
                if (nqueued >= 64) {
                    wave_sync();                                 // queue stores of other lanes before the drain reads them
                    drain(64);
                    unsigned mv = 0u;
                    if (lane < nqueued - 64) mv = qw[64 + lane];
                    wave_sync();
                    if (lane < nqueued - 64) qw[lane] = mv;
                    wave_sync();
                    nqueued -= 64;
                }
            }
            over = rounds > NNF_BUDGET / 64;                      // (uniform)
        }
    }
    wave_sync();
    if (!over && nqueued > 0) drain(nqueued);
    if (over && lane == 0) fall[atomicAdd(&hdr->nfall, 1u)] = (unsigned)task;
}

// the exact scan for the query groups the filter gave up on (or all of them when the inputs were `bad`): a workgroup takes work
// items (group, reference split) off the list; same sums, same keys as k_nn1
__global__ void __launch_bounds__(256) k_nn1f_fallback(const float* __restrict__ ref, const float* __restrict__ query, int b, int n, int nq,
                                                      int nqpad, const NnfHdr* __restrict__ hdr, const unsigned* __restrict__ fall,
                                                      unsigned long long* __restrict__ best)
{
    __shared__ float4 tile[NN1_SPLIT * 8];
    __shared__ unsigned long long red[4][WAVE];
    const int groups = nqpad / NNF_QW, splits = (n + NN1_SPLIT - 1) / NN1_SPLIT;
    const bool all = hdr->bad != 0;
    const long long items = (long long)(all ? b * groups : (int)hdr->nfall) * splits;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int task = all ? (int)(it / splits) : (int)fall[it / splits], sp = (int)(it % splits);
        const int e = task / groups, g = task - e * groups;
        const int q = g * NNF_QW + lane, base = sp * NN1_SPLIT, cnt = min(NN1_SPLIT, n - base);
        __syncthreads();
        const float4* R4 = reinterpret_cast<const float4*>(ref + ((size_t)e * n + base) * 32);
        for (int t = threadIdx.x; t < cnt * 8; t += 256) tile[t] = R4[t];
        float qv[32];
        const float* Q = query + ((size_t)e * nq + (q < nq ? q : 0)) * 32;
#pragma unroll
        for (int c = 0; c < 32; c++) qv[c] = Q[c];
        __syncthreads();
        unsigned long long mine = ~0ull;
        for (int i = w; i < cnt; i += 4) {
            float ssd = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < 8; c4++) {
                const float4 r = tile[i * 8 + c4];
                float t0 = __fsub_rn(r.x, qv[4 * c4]);     ssd = __fadd_rn(ssd, __fmul_rn(t0, t0));
                float t1 = __fsub_rn(r.y, qv[4 * c4 + 1]); ssd = __fadd_rn(ssd, __fmul_rn(t1, t1));
                float t2 = __fsub_rn(r.z, qv[4 * c4 + 2]); ssd = __fadd_rn(ssd, __fmul_rn(t2, t2));
                float t3 = __fsub_rn(r.w, qv[4 * c4 + 3]); ssd = __fadd_rn(ssd, __fmul_rn(t3, t3));
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(ssd) << 32) | (unsigned int)(base + i);
            mine = key < mine ? key : mine;
        }
        red[w][lane] = mine;
        __syncthreads();
        if (w == 0 && q < nq) {
            unsigned long long mm = red[0][lane];
            for (int i = 1; i < 4; i++) mm = red[i][lane] < mm ? red[i][lane] : mm;
            atomicMin(&best[(size_t)e * nq + q], mm);
        }
    }
}

static size_t nnf_round(size_t x) { return (x + 255) & ~(size_t)255; }
// workspace of the matrix-pipe 1-NN: best keys | header | fallback list | reference planes + norms | query planes
extern "C" size_t buf_knn1_ws_bytes(int b, int n, int q)
{
    if (b <= 0 || n <= 0 || q <= 0) return 256;
    const size_t npad = ((size_t)n + 15) / 16 * 16, qpad = ((size_t)q + NNF_QW - 1) / NNF_QW * NNF_QW;
    return nnf_round(8 * (size_t)b * q) + 256 + nnf_round(8 * (size_t)b) + nnf_round(4 * (size_t)b * (qpad / NNF_QW)) + 2 * nnf_round(64 * (size_t)b * npad) +
           nnf_round(4 * (size_t)b * npad) + 2 * nnf_round(64 * (size_t)b * qpad) + 256;
}

extern "C" size_t buf_knn_ws_bytes(int b, int q, int k) { return k == 1 ? sizeof(unsigned long long) * (size_t)b * q + 256 : 256; }

extern "C" int buf_knn(const float* ref, const float* query, int b, int n, int nq, int d, int k, float* dist,
                       long long* idx, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(b >= 0 && n >= 0 && nq >= 0 && d > 0 && k > 0, BUF_EINVAL, "buf_knn: b=%d n=%d q=%d d=%d k=%d", b, n, nq, d, k);
    BUF_REQUIRE(d <= KNN_MAXD, BUF_EINVAL, "buf_knn: feature dim %d > %d", d, KNN_MAXD);
    BUF_REQUIRE(k <= 64, BUF_EINVAL, "buf_knn: k=%d > 64", k);
    if ((long long)b * nq == 0) return BUF_OK;
    BUF_REQUIRE(query && dist && idx && (n == 0 || ref), BUF_EINVAL, "buf_knn: null argument");
    static const bool nn1_exact_scan = getenv("BUF_NN1_EXACT_SCAN") != nullptr;        // development switch: the round-1..3 kernel
    if (k == 1 && d == 32 && n > 0 && ws && ws_bytes >= buf_knn1_ws_bytes(b, n, nq) && !nn1_exact_scan) {
        hipStream_t s = (hipStream_t)stream;
        const int npad = (n + 15) / 16 * 16, qpad = (nq + NNF_QW - 1) / NNF_QW * NNF_QW, groups = qpad / NNF_QW;
        char* p = (char*)ws;
        unsigned long long* best = (unsigned long long*)p;  p += nnf_round(8 * (size_t)b * nq);
        NnfHdr* hdr = (NnfHdr*)p;                            p += 256;
        NnfMax* emax = (NnfMax*)p;                           p += nnf_round(8 * (size_t)b);
        unsigned* fall = (unsigned*)p;                       p += nnf_round(4 * (size_t)b * groups);
        unsigned short* rhi = (unsigned short*)p;            p += nnf_round(64 * (size_t)b * npad);
        unsigned short* rlo = (unsigned short*)p;            p += nnf_round(64 * (size_t)b * npad);
        float* rnorm = (float*)p;                            p += nnf_round(4 * (size_t)b * npad);
        unsigned short* qhi = (unsigned short*)p;            p += nnf_round(64 * (size_t)b * qpad);
        unsigned short* qlo = (unsigned short*)p;
        BUF_CHECK_HIP(hipMemsetAsync(best, 0xff, sizeof(unsigned long long) * (size_t)b * nq, s));
        BUF_CHECK_HIP(hipMemsetAsync(hdr, 0, 256 + nnf_round(8 * (size_t)b), s));          // header + per-element maxima
        TimedSpan span;
        bool timed = timing_begin(s, &span, 2.0 * b * nq * (double)n * 32, BUF_TIMED_NN1);
        k_nn1f_norm<<<cdiv((long long)b * (n + nq), 256), 256, 0, s>>>(ref, query, b, n, nq, emax, &hdr->bad);
        k_nn1f_prep<<<cdiv((long long)b * npad, 256), 256, 0, s>>>(ref, b, n, npad, 1.f, rhi, rlo, rnorm, emax);
        k_nn1f_prep<<<cdiv((long long)b * qpad, 256), 256, 0, s>>>(query, b, nq, qpad, -2.f, qhi, qlo, nullptr, emax);
        k_nn1f_sweep<<<cdiv((long long)b * groups, 4), 256, 0, s>>>(ref, query, n, nq, npad, qpad, rhi, rlo, rnorm, qhi, qlo, hdr, emax, fall, best, b);
        k_nn1f_fallback<<<256, 256, 0, s>>>(ref, query, b, n, nq, qpad, hdr, fall, best);
        if (timed) timing_end(s, &span);
        long long total = (long long)b * nq;
        k_nn1_finish<<<cdiv(total, 256), 256, 0, s>>>(best, total, dist, idx);
        BUF_LAUNCH_CHECK();
        return BUF_OK;
    }
    if (k == 1 && d == 32 && n > 0 && ws && ws_bytes >= sizeof(unsigned long long) * (size_t)b * nq) {
        hipStream_t s = (hipStream_t)stream;
        unsigned long long* best = (unsigned long long*)ws;
        BUF_CHECK_HIP(hipMemsetAsync(best, 0xff, sizeof(unsigned long long) * (size_t)b * nq, s));
        dim3 g1(cdiv(nq, WAVE), cdiv(n, NN1_SPLIT), b);
        TimedSpan span;
        bool timed = timing_begin(s, &span, 2.0 * b * nq * (double)n * 32, BUF_TIMED_NN1);
        k_nn1<32><<<g1, 256, 0, s>>>(ref, query, n, nq, best);
        if (timed) timing_end(s, &span);
        long long total = (long long)b * nq;
        k_nn1_finish<<<cdiv(total, 256), 256, 0, s>>>(best, total, dist, idx);
        BUF_LAUNCH_CHECK();
        return BUF_OK;
    }
    dim3 grid(cdiv(nq, WAVE), b);
    if (k <= 8) k_knn<8><<<grid, WAVE, 0, (hipStream_t)stream>>>(ref, query, n, nq, d, k, dist, idx);
    else k_knn<64><<<grid, WAVE, 0, (hipStream_t)stream>>>(ref, query, n, nq, d, k, dist, idx);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A5b
// Ordered stream compaction of `score > threshold` (torch.where at models/BUFFER.py:255-259): indices come out
// ascending.  Pass 1 counts hits per 256-lane workgroup (ballot + popcount), an exclusive scan turns the counts
// into offsets, pass 2 places every hit at offset + (hits in lower wavefronts) + (popcount of lower lanes).
__global__ void __launch_bounds__(256) k_compact_count(const float* __restrict__ x, int stride, int n, float thr, int* __restrict__ counts)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    bool hit = i < n && x[(size_t)i * stride] > thr;
    unsigned long long m = __ballot(hit);
    __shared__ int wc[4];
    if ((threadIdx.x & (WAVE - 1)) == 0) wc[threadIdx.x / WAVE] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ void __launch_bounds__(256) k_compact_scatter(const float* __restrict__ x, int stride, int n, float thr,
                                                       const int* __restrict__ offsets, int* __restrict__ idx_out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    bool hit = i < n && x[(size_t)i * stride] > thr;
    unsigned long long m = __ballot(hit);
    __shared__ int wc[4];
    if (lane == 0) wc[w] = __popcll(m);
    __syncthreads();
    int base = offsets[blockIdx.x];
    for (int k = 0; k < w; k++) base += wc[k];
    if (hit) idx_out[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

extern "C" size_t buf_compact_ws_bytes(int n) { return sizeof(int) * ((size_t)cdiv(n > 0 ? n : 1, 256) + scan_tmp_ints() + 64); }

// x f32[n] read with element stride `stride`; idx_out int32[n] capacity; count_out int32[1] (device).
extern "C" int buf_compact_greater(const float* x, int stride, int n, float threshold, int* idx_out, int* count_out,
                                   void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(n >= 0 && stride >= 1, BUF_EINVAL, "buf_compact_greater: n=%d stride=%d", n, stride);
    BUF_REQUIRE(count_out && ws, BUF_EINVAL, "buf_compact_greater: null argument");
    if (n == 0) { BUF_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int), s)); return BUF_OK; }
    BUF_REQUIRE(x && idx_out, BUF_EINVAL, "buf_compact_greater: null argument");
    int blocks = cdiv(n, 256);
    WsCarver w(ws, ws_bytes);
    int* counts = w.take<int>((size_t)blocks);
    int* tmp = w.take<int>(scan_tmp_ints());
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_compact_greater: workspace %zu < %zu", ws_bytes, w.used());
    k_compact_count<<<blocks, 256, 0, s>>>(x, stride, n, threshold, counts);
    int rc = exclusive_scan_i32(counts, blocks, tmp, count_out, s);
    if (rc) return rc;
    k_compact_scatter<<<blocks, 256, 0, s>>>(x, stride, n, threshold, counts, idx_out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
