// A14-A17 -- pose recovery (models/BUFFER.py:295-333,382-464; utils/common.py:709-726).
//   buf_svd3x3_batched   one lane per 3x3 matrix, one-sided Jacobi (torch_batch_svd.svd surface)
//   buf_hypotheses_score R,t per match from the predicted azimuth shift + all-vs-all inlier counting
//                        without the [M,M,3] tensor of the reference
//   buf_ransac_kabsch    deterministic 3-point RANSAC on the winning inlier set (replaces the open3d
//                        CPU call, BUFFER.py:314-326): edge-length + distance pre-checks, Kabsch, fitness/rmse
//   buf_post_refine      the <=20 rounds of weighted Kabsch in ONE launch (reference: 20 host round
//                        trips for a 3x3 SVD)
#include "common.h"

// ------------------------------------------------------------------------------------------ SVD
// A = U diag(S) V^T, S descending.  One-sided (Hestenes) Jacobi: rotate column pairs of W = A V
// until mutually orthogonal; S = column norms, U = W / S.  Rank-deficient columns of U are
// completed to a right-handed orthonormal basis.
__device__ void svd3(const float A[9], float U[9], float S[3], float V[9])
{
    float W[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i % 4 == 0) ? 1.f : 0.f; }
    for (int sweep = 0; sweep < 12; sweep++) {
        float off = 0.f;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            float al = W[p] * W[p] + W[3 + p] * W[3 + p] + W[6 + p] * W[6 + p];
            float be = W[q] * W[q] + W[3 + q] * W[3 + q] + W[6 + q] * W[6 + q];
            float ga = W[p] * W[q] + W[3 + p] * W[3 + q] + W[6 + p] * W[6 + q];
            float lim = 1e-7f * sqrtf(al * be);
            if (fabsf(ga) > lim && fabsf(ga) > 1e-30f) {
                off = fmaxf(off, fabsf(ga) / fmaxf(sqrtf(al * be), 1e-30f));
                float zeta = (be - al) / (2.f * ga);
                float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                float c = 1.f / sqrtf(1.f + t * t), s = c * t;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    float wp = W[3 * r + p], wq = W[3 * r + q];
                    W[3 * r + p] = c * wp - s * wq;
                    W[3 * r + q] = s * wp + c * wq;
                    float vp = V[3 * r + p], vq = V[3 * r + q];
                    V[3 * r + p] = c * vp - s * vq;
                    V[3 * r + q] = s * vp + c * vq;
                }
            }
        }
        if (off < 1e-7f) break;
    }
    float n[3];
#pragma unroll
    for (int j = 0; j < 3; j++) n[j] = sqrtf(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
    // sort columns by descending norm (3 compare-exchanges)
#define SWAPCOL(a, b)                                                                      \
    if (n[a] < n[b]) {                                                                     \
        float tn = n[a]; n[a] = n[b]; n[b] = tn;                                           \
        for (int r = 0; r < 3; r++) {                                                      \
            float tw = W[3 * r + a]; W[3 * r + a] = W[3 * r + b]; W[3 * r + b] = tw;       \
            float tv = V[3 * r + a]; V[3 * r + a] = V[3 * r + b]; V[3 * r + b] = tv;       \
        }                                                                                  \
    }
    SWAPCOL(0, 1) SWAPCOL(0, 2) SWAPCOL(1, 2)
#undef SWAPCOL
    float tiny = fmaxf(n[0], 1e-30f) * 1e-6f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        S[j] = n[j];
        float inv = n[j] > tiny ? 1.f / n[j] : 0.f;
        for (int r = 0; r < 3; r++) U[3 * r + j] = W[3 * r + j] * inv;
    }
    // complete a rank-deficient U
    if (n[0] <= 1e-30f) { for (int i = 0; i < 9; i++) U[i] = (i % 4 == 0) ? 1.f : 0.f; return; }
    if (n[1] <= tiny) {
        // any unit vector orthogonal to u0
        float ux = U[0], uy = U[3], uz = U[6];
        float ax = fabsf(ux) < 0.9f ? 1.f : 0.f, ay = fabsf(ux) < 0.9f ? 0.f : 1.f;
        float vx = uy * 0.f - uz * ay, vy = uz * ax - ux * 0.f, vz = ux * ay - uy * ax;
        float nv = rsqrtf(vx * vx + vy * vy + vz * vz);
        U[1] = vx * nv; U[4] = vy * nv; U[7] = vz * nv;
    }
    if (n[2] <= tiny) {
        float sgn = 1.f;   // keep det(U) sign free: choose u2 = u0 x u1
        U[2] = sgn * (U[3] * U[7] - U[6] * U[4]);
        U[5] = sgn * (U[6] * U[1] - U[0] * U[7]);
        U[8] = sgn * (U[0] * U[4] - U[3] * U[1]);
    }
}

__device__ __forceinline__ float det3(const float M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// R = V diag(1,1,det(V U^T)) U^T   (models/BUFFER.py:455-461)
__device__ void kabsch_rotation(const float H[9], float R[9])
{
    float U[9], S[3], V[9];
    svd3(H, U, S, V);
    float VUt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) VUt[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + V[3 * i + 2] * U[3 * j + 2];
    float d = det3(VUt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = V[3 * i] * U[3 * j] + V[3 * i + 1] * U[3 * j + 1] + d * V[3 * i + 2] * U[3 * j + 2];
}

__global__ void __launch_bounds__(256) k_svd3(const float* __restrict__ A, int n, float* __restrict__ U, float* __restrict__ S,
                                            float* __restrict__ V)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a[9], u[9], s[3], v[9];
    for (int k = 0; k < 9; k++) a[k] = A[9 * (size_t)i + k];
    svd3(a, u, s, v);
    for (int k = 0; k < 9; k++) { U[9 * (size_t)i + k] = u[k]; V[9 * (size_t)i + k] = v[k]; }
    for (int k = 0; k < 3; k++) S[3 * (size_t)i + k] = s[k];
}

extern "C" int buf_svd3x3_batched(const float* a, int n, float* u, float* s, float* v, void* stream)
{
    BUF_REQUIRE(n >= 0, BUF_EINVAL, "buf_svd3x3_batched: n=%d", n);
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(a && u && s && v, BUF_EINVAL, "buf_svd3x3_batched: null argument");
    k_svd3<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, n, u, s, v);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A14
__global__ void __launch_bounds__(256) k_hypotheses(const float* __restrict__ ind, const float* __restrict__ ss, const float* __restrict__ tt,
                                                  const float* __restrict__ ssR, const float* __restrict__ ttR, int m, float azi_n,
                                                  float* __restrict__ R_out, float* __restrict__ t_out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    // angle = ind * 2 * pi / azi_n + 1e-6 (BUFFER.py:295); kornia Rodrigues about z for theta^2 > 1e-6
    float angle = ind[i] * 2.f * 3.14159265358979323846f / azi_n + 1e-6f;
    float c = cosf(angle), s = sinf(angle);
    const float* A = ttR + 9 * (size_t)i;
    const float* B = ssR + 9 * (size_t)i;
    float Rz[9] = { c, -s, 0, s, c, 0, 0, 0, 1 };
    float T[9], R[9];
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) T[3 * r + k] = A[3 * r] * Rz[k] + A[3 * r + 1] * Rz[3 + k] + A[3 * r + 2] * Rz[6 + k];
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) R[3 * r + k] = T[3 * r] * B[3 * k] + T[3 * r + 1] * B[3 * k + 1] + T[3 * r + 2] * B[3 * k + 2];
    float sx = ss[3 * (size_t)i], sy = ss[3 * (size_t)i + 1], sz = ss[3 * (size_t)i + 2];
    for (int k = 0; k < 9; k++) R_out[9 * (size_t)i + k] = R[k];
    for (int r = 0; r < 3; r++) t_out[3 * (size_t)i + r] = tt[3 * (size_t)i + r] - (R[3 * r] * sx + R[3 * r + 1] * sy + R[3 * r + 2] * sz);
}

// one workgroup per hypothesis h (stacked row h, pair rows [lo, hi)): count j with
//   ||R_h ss_j + t_h - tt_j|| < ||ss_j|| * pi/azi_n * inlier_th                       (models/BUFFER.py:302-309)
__device__ __forceinline__ void score_hypothesis(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ ss,
                                                 const float* __restrict__ tt, int h, int lo, int hi, float azi_n, float inlier_th,
                                                 int* __restrict__ inlier_num)
{
    float r[9], tv[3];
    for (int k = 0; k < 9; k++) r[k] = R[9 * (size_t)h + k];
    for (int k = 0; k < 3; k++) tv[k] = t[3 * (size_t)h + k];
    int cnt = 0;
    for (int j = lo + threadIdx.x; j < hi; j += 256) {
        float x = ss[3 * (size_t)j], y = ss[3 * (size_t)j + 1], z = ss[3 * (size_t)j + 2];
        float dx = r[0] * x + r[1] * y + r[2] * z + tv[0] - tt[3 * (size_t)j];
        float dy = r[3] * x + r[4] * y + r[5] * z + tv[1] - tt[3 * (size_t)j + 1];
        float dz = r[6] * x + r[7] * y + r[8] * z + tv[2] - tt[3 * (size_t)j + 2];
        float diff = sqrtf(dx * dx + dy * dy + dz * dz);
        float thr = sqrtf(x * x + y * y + z * z) * 3.14159265358979323846f / azi_n * inlier_th;
        cnt += diff < thr ? 1 : 0;
    }
    __shared__ int sc[4];
    for (int d = WAVE / 2; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) sc[threadIdx.x / WAVE] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) inlier_num[h] = sc[0] + sc[1] + sc[2] + sc[3];
}

__global__ void __launch_bounds__(256) k_score(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ ss,
                                             const float* __restrict__ tt, int m, float azi_n, float inlier_th,
                                             int* __restrict__ inlier_num)
{
    score_hypothesis(R, t, ss, tt, blockIdx.x, 0, m, azi_n, inlier_th, inlier_num);
}

// argmax (first maximum, torch.argmax) over the m hypotheses of rows [lo, lo+m) + inlier mask of the winner (1024 threads)
__device__ __forceinline__ void best_and_mask(const int* __restrict__ inlier_num, const float* __restrict__ R, const float* __restrict__ t,
                                              const float* __restrict__ ss, const float* __restrict__ tt, int lo, int m, float azi_n,
                                              float inlier_th, int* __restrict__ best_out, unsigned char* __restrict__ mask)
{
    __shared__ unsigned long long sk[16];
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < m; i += 1024) {
        unsigned long long key = ((unsigned long long)(unsigned int)(inlier_num[lo + i] + 1) << 32) | (unsigned int)(0x7fffffff - i);
        best = key > best ? key : best;
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        unsigned int l = __shfl_xor((unsigned int)best, d, WAVE), hgh = __shfl_xor((unsigned int)(best >> 32), d, WAVE);
        unsigned long long o = ((unsigned long long)hgh << 32) | l;
        best = o > best ? o : best;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) sk[threadIdx.x / WAVE] = best;
    __syncthreads();
    best = 0;
    for (int i = 0; i < 16; i++) best = sk[i] > best ? sk[i] : best;
    if (m == 0) { if (threadIdx.x == 0) *best_out = 0; return; }
    const int h = 0x7fffffff - (int)(unsigned int)(best & 0xffffffffu);
    if (threadIdx.x == 0) *best_out = h;                         // index inside the pair
    float r[9], tv[3];
    for (int k = 0; k < 9; k++) r[k] = R[9 * (size_t)(lo + h) + k];
    for (int k = 0; k < 3; k++) tv[k] = t[3 * (size_t)(lo + h) + k];
    for (int j = threadIdx.x; j < m; j += 1024) {
        const size_t g = (size_t)lo + j;
        float x = ss[3 * g], y = ss[3 * g + 1], z = ss[3 * g + 2];
        float dx = r[0] * x + r[1] * y + r[2] * z + tv[0] - tt[3 * g];
        float dy = r[3] * x + r[4] * y + r[5] * z + tv[1] - tt[3 * g + 1];
        float dz = r[6] * x + r[7] * y + r[8] * z + tv[2] - tt[3 * g + 2];
        float diff = sqrtf(dx * dx + dy * dy + dz * dz);
        float thr = sqrtf(x * x + y * y + z * z) * 3.14159265358979323846f / azi_n * inlier_th;
        mask[g] = diff < thr ? 1 : 0;
    }
}

__global__ void __launch_bounds__(1024) k_best_mask(const int* __restrict__ inlier_num, const float* __restrict__ R, const float* __restrict__ t,
                                                  const float* __restrict__ ss, const float* __restrict__ tt, int m, float azi_n,
                                                  float inlier_th, int* __restrict__ best_out, unsigned char* __restrict__ mask)
{
    best_and_mask(inlier_num, R, t, ss, tt, 0, m, azi_n, inlier_th, best_out, mask);
}

extern "C" int buf_hypotheses_score(const float* ind, const float* ss_kpts, const float* tt_kpts, const float* ss_R,
                                    const float* tt_R, int m, int azi_n, float inlier_th, float* R_out, float* t_out,
                                    int* inlier_num, int* best_out, unsigned char* best_mask, void* stream)
{
    BUF_REQUIRE(m >= 0 && azi_n > 0, BUF_EINVAL, "buf_hypotheses_score: m=%d azi_n=%d", m, azi_n);
    if (m == 0) return BUF_OK;
    BUF_REQUIRE(ind && ss_kpts && tt_kpts && ss_R && tt_R && R_out && t_out && inlier_num && best_out && best_mask,
                BUF_EINVAL, "buf_hypotheses_score: null argument");
    hipStream_t s = (hipStream_t)stream;
    k_hypotheses<<<cdiv(m, 256), 256, 0, s>>>(ind, ss_kpts, tt_kpts, ss_R, tt_R, m, (float)azi_n, R_out, t_out);
    k_score<<<m, 256, 0, s>>>(R_out, t_out, ss_kpts, tt_kpts, m, (float)azi_n, inlier_th, inlier_num);
    k_best_mask<<<1, 1024, 0, s>>>(inlier_num, R_out, t_out, ss_kpts, tt_kpts, m, (float)azi_n, inlier_th, best_out, best_mask);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A15
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// One lane per hypothesis: 3 distinct correspondences from the candidate list, edge-length checker,
// Kabsch, distance checker, then fitness (inlier count) and sum of squared inlier distances over
// all candidates.  key = (count << 32) | ~bits(mean squared error): larger is better.
// ncorr_dev (nullable): the candidate count lives on the device (buf_ransac_kabsch_masked: no host round trip); fewer
// than 3 candidates leave every key 0, which k_ransac_pick turns into the identity (open3d's result in that case).
__device__ __forceinline__ void ransac_hypothesis(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ corr,
                                                  int ncorr, int nhyp, unsigned long long seed, float max_dist, float edge_sim, int h,
                                                  unsigned long long* __restrict__ keys, float* __restrict__ Ts)
{
    if (h >= nhyp) return;
    unsigned long long key = 0;
    float T[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    if (ncorr < 3) {
        keys[h] = 0;
        for (int k = 0; k < 12; k++) Ts[12 * (size_t)h + k] = T[k];
        return;
    }
    int i0 = (int)(splitmix64(seed + 3ull * h) % (unsigned long long)ncorr);
    int i1 = (int)(splitmix64(seed + 3ull * h + 1) % (unsigned long long)(ncorr - 1));
    int i2 = (int)(splitmix64(seed + 3ull * h + 2) % (unsigned long long)(ncorr - 2));
    if (i1 >= i0) i1++;
    int lo = min(i0, i1), hi = max(i0, i1);
    if (i2 >= lo) i2++;
    if (i2 >= hi) i2++;
    int id[3] = { corr[i0], corr[i1], corr[i2] };
    float a[3][3], b[3][3];
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) { a[k][c] = src[3 * (size_t)id[k] + c]; b[k][c] = tgt[3 * (size_t)id[k] + c]; }
    bool ok = true;
    for (int p = 0; p < 3 && ok; p++) {
        int q = (p + 1) % 3;
        float ds = sqrtf(sqdist3(a[p][0], a[p][1], a[p][2], a[q][0], a[q][1], a[q][2]));
        float dt = sqrtf(sqdist3(b[p][0], b[p][1], b[p][2], b[q][0], b[q][1], b[q][2]));
        ok = ds >= dt * edge_sim && dt >= ds * edge_sim;
    }
    if (ok) {
        float ca[3], cb[3];
        for (int c = 0; c < 3; c++) { ca[c] = (a[0][c] + a[1][c] + a[2][c]) / 3.f; cb[c] = (b[0][c] + b[1][c] + b[2][c]) / 3.f; }
        float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int k = 0; k < 3; k++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) H[3 * r + c] += (a[k][r] - ca[r]) * (b[k][c] - cb[c]);
        float R[9];
        kabsch_rotation(H, R);
        for (int r = 0; r < 3; r++) {
            T[4 * r] = R[3 * r]; T[4 * r + 1] = R[3 * r + 1]; T[4 * r + 2] = R[3 * r + 2];
            T[4 * r + 3] = cb[r] - (R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]);
        }
        for (int k = 0; k < 3 && ok; k++) {
            float x = T[0] * a[k][0] + T[1] * a[k][1] + T[2] * a[k][2] + T[3] - b[k][0];
            float y = T[4] * a[k][0] + T[5] * a[k][1] + T[6] * a[k][2] + T[7] - b[k][1];
            float z = T[8] * a[k][0] + T[9] * a[k][1] + T[10] * a[k][2] + T[11] - b[k][2];
            ok = sqrtf(x * x + y * y + z * z) <= max_dist;
        }
    }
    if (ok) {
        int cnt = 0;
        float err2 = 0.f;
        for (int j = 0; j < ncorr; j++) {
            int k = corr[j];
            float sx = src[3 * (size_t)k], sy = src[3 * (size_t)k + 1], sz = src[3 * (size_t)k + 2];
            float x = T[0] * sx + T[1] * sy + T[2] * sz + T[3] - tgt[3 * (size_t)k];
            float y = T[4] * sx + T[5] * sy + T[6] * sz + T[7] - tgt[3 * (size_t)k + 1];
            float z = T[8] * sx + T[9] * sy + T[10] * sz + T[11] - tgt[3 * (size_t)k + 2];
            float d2 = x * x + y * y + z * z;
            if (sqrtf(d2) < max_dist) { cnt++; err2 += d2; }
        }
        if (cnt > 0) key = ((unsigned long long)(unsigned int)cnt << 32) | (unsigned int)~__float_as_uint(err2 / (float)cnt);
    }
    keys[h] = key;
    for (int k = 0; k < 12; k++) Ts[12 * (size_t)h + k] = T[k];
}

__global__ void __launch_bounds__(WAVE) k_ransac(const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ corr,
                                              int ncorr, const int* __restrict__ ncorr_dev, int nhyp, unsigned long long seed,
                                              float max_dist, float edge_sim, unsigned long long* __restrict__ keys, float* __restrict__ Ts)
{
    // one wavefront per workgroup: 4096 hypotheses spread over 64 CUs
    ransac_hypothesis(src, tgt, corr, ncorr_dev ? *ncorr_dev : ncorr, nhyp, seed, max_dist, edge_sim, blockIdx.x * WAVE + threadIdx.x, keys, Ts);
}

__device__ __forceinline__ void ransac_pick(const unsigned long long* __restrict__ keys, const float* __restrict__ Ts, int nhyp,
                                            float* __restrict__ T_out, int* __restrict__ info)
{
    __shared__ unsigned long long sk[16];
    __shared__ int si[16];
    unsigned long long best = 0;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nhyp; i += 1024) {
        unsigned long long k = keys[i];
        if (k > best || (k == best && i < bi)) { best = k; bi = i; }
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        unsigned int lo = __shfl_xor((unsigned int)best, d, WAVE), hi = __shfl_xor((unsigned int)(best >> 32), d, WAVE);
        int oi = __shfl_xor(bi, d, WAVE);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        if (o > best || (o == best && oi < bi)) { best = o; bi = oi; }
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) { sk[threadIdx.x / WAVE] = best; si[threadIdx.x / WAVE] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        best = 0; bi = 0x7fffffff;
        for (int i = 0; i < 16; i++) if (sk[i] > best || (sk[i] == best && si[i] < bi)) { best = sk[i]; bi = si[i]; }
        float T[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        if (best != 0) for (int k = 0; k < 12; k++) T[k] = Ts[12 * (size_t)bi + k];
        for (int k = 0; k < 16; k++) T_out[k] = T[k];
        if (info) { info[0] = (int)(best >> 32); info[1] = best != 0 ? bi : -1; }
    }
}

__global__ void __launch_bounds__(1024) k_ransac_pick(const unsigned long long* __restrict__ keys, const float* __restrict__ Ts, int nhyp,
                                                    float* __restrict__ T_out, int* __restrict__ info)
{
    ransac_pick(keys, Ts, nhyp, T_out, info);
}

extern "C" size_t buf_ransac_ws_bytes(int nhyp) { return 256 + (size_t)nhyp * (8 + 48) + 512; }

extern "C" int buf_ransac_kabsch(const float* src, const float* tgt, const int* corr, int ncorr, int nhyp,
                                 unsigned long long seed, float max_dist, float edge_similarity, float* T_out,
                                 int* info_out, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(nhyp > 0 && ncorr >= 0, BUF_EINVAL, "buf_ransac_kabsch: ncorr=%d nhyp=%d", ncorr, nhyp);
    BUF_REQUIRE(T_out && ws, BUF_EINVAL, "buf_ransac_kabsch: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver w(ws, ws_bytes);
    unsigned long long* keys = w.take<unsigned long long>((size_t)nhyp);
    float* Ts = w.take<float>(12 * (size_t)nhyp);
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_ransac_kabsch: workspace %zu < %zu", ws_bytes, w.used());
    if (ncorr < 3) {   // open3d returns the identity when fewer than ransac_n correspondences exist
        BUF_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)nhyp, s));
    } else {
        BUF_REQUIRE(src && tgt && corr, BUF_EINVAL, "buf_ransac_kabsch: null argument");
        k_ransac<<<cdiv(nhyp, WAVE), WAVE, 0, s>>>(src, tgt, corr, ncorr, nullptr, nhyp, seed, max_dist, edge_similarity, keys, Ts);
    }
    k_ransac_pick<<<1, 1024, 0, s>>>(keys, Ts, nhyp, T_out, info_out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ascending indices of the set bytes of mask[0..m) and their count: one wavefront, ballot + popcount per 64 entries
__device__ __forceinline__ void mask_compact(const unsigned char* __restrict__ mask, int m, int* __restrict__ idx, int* __restrict__ count)
{
    const int lane = threadIdx.x;
    int cnt = 0;
    for (int base = 0; base < m; base += WAVE) {
        const int i = base + lane;
        const bool hit = i < m && mask[i] != 0;
        const unsigned long long b = __ballot(hit);
        if (hit) idx[cnt + lane_prefix(b, lane)] = i;
        cnt += __popcll(b);
    }
    if (lane == 0) *count = cnt;
}

__global__ void __launch_bounds__(WAVE) k_mask_compact(const unsigned char* __restrict__ mask, int m, int* __restrict__ idx,
                                                    int* __restrict__ count)
{
    mask_compact(mask, m, idx, count);
}

extern "C" size_t buf_ransac_masked_ws_bytes(int m, int nhyp) { return buf_ransac_ws_bytes(nhyp) + sizeof(int) * ((size_t)(m > 0 ? m : 0) + 64); }

// buf_ransac_kabsch on the correspondences selected by mask uint8[m] (e.g. best_mask of buf_hypotheses_score): the
// index list and its length stay on the device, so a whole pose recovery is enqueued without a host round trip.
extern "C" int buf_ransac_kabsch_masked(const float* src, const float* tgt, const unsigned char* mask, int m, int nhyp,
                                        unsigned long long seed, float max_dist, float edge_similarity, float* T_out,
                                        int* info_out, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(nhyp > 0 && m >= 0, BUF_EINVAL, "buf_ransac_kabsch_masked: m=%d nhyp=%d", m, nhyp);
    BUF_REQUIRE(T_out && ws && (m == 0 || (src && tgt && mask)), BUF_EINVAL, "buf_ransac_kabsch_masked: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver w(ws, ws_bytes);
    unsigned long long* keys = w.take<unsigned long long>((size_t)nhyp);
    float* Ts = w.take<float>(12 * (size_t)nhyp);
    int* count = w.take<int>(1);
    int* idx = w.take<int>((size_t)(m > 0 ? m : 1));
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_ransac_kabsch_masked: workspace %zu < %zu", ws_bytes, w.used());
    k_mask_compact<<<1, WAVE, 0, s>>>(mask, m, idx, count);
    k_ransac<<<cdiv(nhyp, WAVE), WAVE, 0, s>>>(src, tgt, idx, 0, count, nhyp, seed, max_dist, edge_similarity, keys, Ts);
    k_ransac_pick<<<1, 1024, 0, s>>>(keys, Ts, nhyp, T_out, info_out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A16
#define REF_THREADS 256

__device__ __forceinline__ float block_sum(float v, float* sh)
{
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < REF_THREADS / WAVE; i++) t += sh[i];
    return t;
}

// post_refinement (BUFFER.py:382-418) + rigid_transform_3d (:424-464) for one pair per workgroup.
__device__ __forceinline__ void post_refine(const float* __restrict__ T_init, const float* __restrict__ src, const float* __restrict__ tgt,
                                            int m, float thr, int iters, float* __restrict__ T_out, int* __restrict__ info)
{
    __shared__ float sh[REF_THREADS / WAVE];
    __shared__ float Ts[12];
    if (threadIdx.x < 12) Ts[threadIdx.x] = T_init[threadIdx.x];
    __syncthreads();
    int prev = 0, rounds = 0;
    for (int it = 0; it < iters; it++) {
        float T[12];
        for (int k = 0; k < 12; k++) T[k] = Ts[k];
        // pass 1: inlier count, weight sum, weighted centroids
        float cnt = 0.f, sw = 0.f, sa[3] = { 0, 0, 0 }, sb[3] = { 0, 0, 0 };
        for (int j = threadIdx.x; j < m; j += REF_THREADS) {
            float x = src[3 * (size_t)j], y = src[3 * (size_t)j + 1], z = src[3 * (size_t)j + 2];
            float bx = tgt[3 * (size_t)j], by = tgt[3 * (size_t)j + 1], bz = tgt[3 * (size_t)j + 2];
            float dx = T[0] * x + T[1] * y + T[2] * z + T[3] - bx;
            float dy = T[4] * x + T[5] * y + T[6] * z + T[7] - by;
            float dz = T[8] * x + T[9] * y + T[10] * z + T[11] - bz;
            float dis = sqrtf(dx * dx + dy * dy + dz * dz);
            if (dis < thr) {
                float q = dis / thr;
                float w = 1.f / (1.f + q * q);
                cnt += 1.f; sw += w;
                sa[0] += w * x; sa[1] += w * y; sa[2] += w * z;
                sb[0] += w * bx; sb[1] += w * by; sb[2] += w * bz;
            }
        }
        int num = (int)(block_sum(cnt, sh) + 0.5f);
        if (abs(num - prev) < 1) break;                          // :406
        prev = num;
        rounds++;
        sw = block_sum(sw, sh);
        float ca[3], cb[3];
        for (int c = 0; c < 3; c++) { ca[c] = block_sum(sa[c], sh) / (sw + 1e-6f); cb[c] = block_sum(sb[c], sh) / (sw + 1e-6f); }
        // pass 2: H = Am^T W Bm
        float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int j = threadIdx.x; j < m; j += REF_THREADS) {
            float x = src[3 * (size_t)j], y = src[3 * (size_t)j + 1], z = src[3 * (size_t)j + 2];
            float bx = tgt[3 * (size_t)j], by = tgt[3 * (size_t)j + 1], bz = tgt[3 * (size_t)j + 2];
            float dx = T[0] * x + T[1] * y + T[2] * z + T[3] - bx;
            float dy = T[4] * x + T[5] * y + T[6] * z + T[7] - by;
            float dz = T[8] * x + T[9] * y + T[10] * z + T[11] - bz;
            float dis = sqrtf(dx * dx + dy * dy + dz * dz);
            if (dis < thr) {
                float q = dis / thr;
                float w = 1.f / (1.f + q * q);
                float am[3] = { x - ca[0], y - ca[1], z - ca[2] }, bm[3] = { bx - cb[0], by - cb[1], bz - cb[2] };
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) H[3 * r + c] += am[r] * w * bm[c];
            }
        }
        float Hs[9];
        for (int k = 0; k < 9; k++) Hs[k] = block_sum(H[k], sh);
        __syncthreads();
        if (threadIdx.x == 0) {
            float R[9];
            kabsch_rotation(Hs, R);
            for (int r = 0; r < 3; r++) {
                Ts[4 * r] = R[3 * r]; Ts[4 * r + 1] = R[3 * r + 1]; Ts[4 * r + 2] = R[3 * r + 2];
                Ts[4 * r + 3] = cb[r] - (R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 12) T_out[threadIdx.x] = Ts[threadIdx.x];
    if (threadIdx.x >= 12 && threadIdx.x < 16) T_out[threadIdx.x] = threadIdx.x == 15 ? 1.f : 0.f;
    if (threadIdx.x == 0 && info) { info[0] = prev; info[1] = rounds; }
}

__global__ void __launch_bounds__(REF_THREADS) k_post_refine(const float* __restrict__ T_init, const float* __restrict__ src,
                                                           const float* __restrict__ tgt, int m, float thr, int iters,
                                                           float* __restrict__ T_out, int* __restrict__ info)
{
    post_refine(T_init, src, tgt, m, thr, iters, T_out, info);
}

extern "C" int buf_post_refine(const float* T_init, const float* src, const float* tgt, int m, float inlier_threshold,
                               int iters, float* T_out, int* info_out, void* stream)
{
    BUF_REQUIRE(m >= 0 && iters >= 0, BUF_EINVAL, "buf_post_refine: m=%d iters=%d", m, iters);
    BUF_REQUIRE(T_init && T_out && (m == 0 || (src && tgt)), BUF_EINVAL, "buf_post_refine: null argument");
    k_post_refine<<<1, REF_THREADS, 0, (hipStream_t)stream>>>(T_init, src, tgt, m, inlier_threshold, iters, T_out, info_out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------ A14-A16, batched
// The pose recovery of EVERY pair of a step in one set of launches (round 1: seven launches per pair on side streams).
// The matches of the pairs are stacked; seg[p] .. seg[p+1] are the rows of pair p (device int32[nb+1]).  Per pair the
// arithmetic -- thread assignment, reduction order, sampler -- is that of the single-pair kernels above, so the poses are
// bit-identical to buf_hypotheses_score + buf_ransac_kabsch_masked + buf_post_refine pair by pair.
#define RB_MAXB 64
struct RecoverSeeds { unsigned long long s[RB_MAXB]; };

__global__ void __launch_bounds__(256) k_score_b(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ ss,
                                               const float* __restrict__ tt, const int* __restrict__ seg, int nb, float azi_n,
                                               float inlier_th, int* __restrict__ inlier_num)
{
    const int h = blockIdx.x;                                     // hypothesis = stacked match row (wave-uniform)
    const int p = find_elem(seg, nb, h);
    score_hypothesis(R, t, ss, tt, h, seg[p], seg[p + 1], azi_n, inlier_th, inlier_num);
}

// per pair: argmax of the inlier counts (first maximum) and the winner's inlier mask
__global__ void __launch_bounds__(1024) k_best_mask_b(const int* __restrict__ inlier_num, const float* __restrict__ R, const float* __restrict__ t,
                                                    const float* __restrict__ ss, const float* __restrict__ tt, const int* __restrict__ seg,
                                                    float azi_n, float inlier_th, int* __restrict__ best_out, unsigned char* __restrict__ mask)
{
    const int p = blockIdx.x, lo = seg[p];
    best_and_mask(inlier_num, R, t, ss, tt, lo, seg[p + 1] - lo, azi_n, inlier_th, best_out + p, mask);
}

// per pair: ascending indices (local to the pair) of the winner's inliers, and their count
__global__ void __launch_bounds__(WAVE) k_mask_compact_b(const unsigned char* __restrict__ mask, const int* __restrict__ seg, int* __restrict__ idx,
                                                      int* __restrict__ count)
{
    const int p = blockIdx.x, lo = seg[p];
    mask_compact(mask + lo, seg[p + 1] - lo, idx + lo, count + p);
}

// blockIdx.y = pair: the single-pair sampler / checks / scoring on the pair's own rows
__global__ void __launch_bounds__(WAVE) k_ransac_b(const float* __restrict__ src_all, const float* __restrict__ tgt_all, const int* __restrict__ idx_all,
                                                const int* __restrict__ count, const int* __restrict__ seg, int nhyp, RecoverSeeds seeds,
                                                float max_dist, float edge_sim, unsigned long long* __restrict__ keys_all, float* __restrict__ Ts_all)
{
    const int p = blockIdx.y, lo = seg[p];
    ransac_hypothesis(src_all + 3 * (size_t)lo, tgt_all + 3 * (size_t)lo, idx_all + lo, count[p], nhyp, seeds.s[p], max_dist, edge_sim,
                      blockIdx.x * WAVE + threadIdx.x, keys_all + (size_t)p * nhyp, Ts_all + 12 * (size_t)p * nhyp);
}

__global__ void __launch_bounds__(1024) k_ransac_pick_b(const unsigned long long* __restrict__ keys_all, const float* __restrict__ Ts_all, int nhyp,
                                                      float* __restrict__ T_out_all)
{
    const int p = blockIdx.x;
    ransac_pick(keys_all + (size_t)p * nhyp, Ts_all + 12 * (size_t)p * nhyp, nhyp, T_out_all + 16 * (size_t)p, nullptr);
}

// one workgroup per pair: post_refine on the pair's rows (iters == 0: the RANSAC pose is copied); pairs with fewer than three
// matches get the identity (ThreeDMatch/test.py:242-245)
__global__ void __launch_bounds__(REF_THREADS) k_post_refine_b(const float* __restrict__ T_init_all, const float* __restrict__ src_all,
                                                             const float* __restrict__ tgt_all, const int* __restrict__ seg, float thr, int iters,
                                                             float* __restrict__ T_out_all)
{
    const int p = blockIdx.x, lo = seg[p], m = seg[p + 1] - lo;
    float* T_out = T_out_all + 16 * (size_t)p;
    if (m < 3) {
        if (threadIdx.x < 16) T_out[threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.f : 0.f;
        return;
    }
    post_refine(T_init_all + 16 * (size_t)p, src_all + 3 * (size_t)lo, tgt_all + 3 * (size_t)lo, m, thr, iters, T_out, nullptr);
}

extern "C" size_t buf_recover_poses_ws_bytes(int m_total, int nb, int nhyp)
{
    WsCarver w(nullptr, 0);
    const size_t m = (size_t)(m_total > 0 ? m_total : 1), b = (size_t)(nb > 0 ? nb : 1);
    w.take<int>(b + 1); w.take<float>(9 * m); w.take<float>(3 * m); w.take<int>(m); w.take<int>(b); w.take<unsigned char>(m);
    w.take<int>(m); w.take<int>(b); w.take<unsigned long long>(b * nhyp); w.take<float>(12 * b * nhyp); w.take<float>(16 * b);
    return w.used();
}

// ind f32[M], ss/tt f32[M,3], ss_R/tt_R f32[M,9]: the matches of nb pairs stacked (pair p owns seg_host[p] rows); seeds_host
// u64[nb]; refine_iters 0 = no post-refinement (KITTI) -> poses f32[nb,4,4].
extern "C" int buf_recover_poses_batched(const float* ind, const float* ss_kpts, const float* tt_kpts, const float* ss_R, const float* tt_R,
                                         const int* seg_host, int nb, const unsigned long long* seeds_host, int azi_n, float inlier_th,
                                         int nhyp, float max_dist, float edge_similarity, float refine_threshold, int refine_iters,
                                         float* poses_out, void* ws, size_t ws_bytes, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    BUF_REQUIRE(nb >= 0 && nhyp > 0 && azi_n > 0 && refine_iters >= 0, BUF_EINVAL, "buf_recover_poses_batched: nb=%d nhyp=%d", nb, nhyp);
    if (nb == 0) return BUF_OK;
    BUF_REQUIRE(seg_host && seeds_host && poses_out && ws, BUF_EINVAL, "buf_recover_poses_batched: null argument");
    long long mt = 0;
    for (int p = 0; p < nb; p++) { BUF_REQUIRE(seg_host[p] >= 0, BUF_EINVAL, "buf_recover_poses_batched: negative segment"); mt += seg_host[p]; }
    BUF_REQUIRE(mt < 0x7fffffffLL, BUF_EINVAL, "buf_recover_poses_batched: too many matches");
    const int M = (int)mt;
    BUF_REQUIRE(M == 0 || (ind && ss_kpts && tt_kpts && ss_R && tt_R), BUF_EINVAL, "buf_recover_poses_batched: null argument");
    BUF_REQUIRE(ws_bytes >= buf_recover_poses_ws_bytes(M, nb, nhyp), BUF_EWORKSPACE, "buf_recover_poses_batched: workspace %zu < %zu",
                ws_bytes, buf_recover_poses_ws_bytes(M, nb, nhyp));
    WsCarver w(ws, ws_bytes);
    const size_t m = (size_t)(M > 0 ? M : 1);
    int* seg = w.take<int>((size_t)nb + 1);
    float* R = w.take<float>(9 * m);
    float* t = w.take<float>(3 * m);
    int* num = w.take<int>(m);
    int* best = w.take<int>((size_t)nb);
    unsigned char* mask = w.take<unsigned char>(m);
    int* idx = w.take<int>(m);
    int* count = w.take<int>((size_t)nb);
    unsigned long long* keys = w.take<unsigned long long>((size_t)nb * nhyp);
    float* Ts = w.take<float>(12 * (size_t)nb * nhyp);
    float* Tr = w.take<float>(16 * (size_t)nb);
    int rc = upload_offsets(seg, seg_host, nb, M, "buf_recover_poses_batched", s);
    if (rc) return rc;
    if (M > 0) {
        k_hypotheses<<<cdiv(M, 256), 256, 0, s>>>(ind, ss_kpts, tt_kpts, ss_R, tt_R, M, (float)azi_n, R, t);
        k_score_b<<<M, 256, 0, s>>>(R, t, ss_kpts, tt_kpts, seg, nb, (float)azi_n, inlier_th, num);
    }
    k_best_mask_b<<<nb, 1024, 0, s>>>(num, R, t, ss_kpts, tt_kpts, seg, (float)azi_n, inlier_th, best, mask);
    k_mask_compact_b<<<nb, WAVE, 0, s>>>(mask, seg, idx, count);
    float* ransac_out = refine_iters > 0 ? Tr : poses_out;
    for (int p0 = 0; p0 < nb; p0 += RB_MAXB) {
        RecoverSeeds sd;
        const int cnt = nb - p0 < RB_MAXB ? nb - p0 : RB_MAXB;
        for (int i = 0; i < cnt; i++) sd.s[i] = seeds_host[p0 + i];
        k_ransac_b<<<dim3(cdiv(nhyp, WAVE), cnt), WAVE, 0, s>>>(ss_kpts, tt_kpts, idx, count + p0, seg + p0, nhyp, sd, max_dist, edge_similarity,
                                                            keys + (size_t)p0 * nhyp, Ts + 12 * (size_t)p0 * nhyp);
    }
    k_ransac_pick_b<<<nb, 1024, 0, s>>>(keys, Ts, nhyp, ransac_out);
    k_post_refine_b<<<nb, REF_THREADS, 0, s>>>(ransac_out, ss_kpts, tt_kpts, seg, refine_threshold, refine_iters, poses_out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
