/* TEST INFRASTRUCTURE ONLY -- see buffer_oracle.h.
 *
 * Plain-C restatement of the reference's CPU cores (cpp_wrappers/) and of the
 * external CUDA operators' documented semantics (pointnet2_ops, KNN_CUDA),
 * written from the algorithm descriptions, not copied.  Build with
 * -ffp-contract=off so that every distance is the reference's
 * ((dx*dx + dy*dy) + dz*dz) in IEEE fp32.
 *
 * Pinning: orc_grid_subsample_batch / orc_radius_neighbors are checked against
 * the reference's own compiled cores (oracle/_ref, built from /root/reference
 * in place) by tests/test_oracle_vs_ref.py (3DMatch-size and KITTI-shape pyramids) and against tests/golden/pyramid_*.npz.
 * orc_fps / orc_ball_query / orc_three_nn / orc_knn restate third-party CUDA
 * extensions that are NOT under /root/reference (pointnet2_ops unpinned git,
 * KNN_CUDA 0.2): parity UNPINNED for those four.
 */
#include "buffer_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------ A1 ---- */
typedef struct { uint64_t key; int idx; } key_idx_t;

static int cmp_key_idx(const void* a, const void* b)
{
    const key_idx_t* x = (const key_idx_t*)a;
    const key_idx_t* y = (const key_idx_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* grid_subsampling.cpp:24-31 (origin / grid dims), :51-56 (voxel key),
 * :62-70 + grid_subsampling.h:95-100 (sum in input order), :85-87 (barycentre). */
static int grid_subsample_one(const float* p, int n, float dl, float* out, uint64_t* out_key)
{
    if (n <= 0) return 0;
    float mnx = p[0], mny = p[1], mnz = p[2], mxx = p[0], mxy = p[1], mxz = p[2];
    for (int i = 0; i < n; i++) {          /* cloud.cpp:27-66 */
        float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        if (x < mnx) mnx = x;  if (y < mny) mny = y;  if (z < mnz) mnz = z;
        if (x > mxx) mxx = x;  if (y > mxy) mxy = y;  if (z > mxz) mxz = z;
    }
    float inv = 1 / dl;                     /* (1/sampleDl) is a float division */
    float ox = floorf(mnx * inv) * dl;
    float oy = floorf(mny * inv) * dl;
    float oz = floorf(mnz * inv) * dl;
    uint64_t NX = (uint64_t)(int64_t)floorf((mxx - ox) / dl) + 1;
    uint64_t NY = (uint64_t)(int64_t)floorf((mxy - oy) / dl) + 1;
    (void)mxz;

    key_idx_t* ki = (key_idx_t*)malloc(sizeof(key_idx_t) * (size_t)n);
    for (int i = 0; i < n; i++) {
        uint64_t iX = (uint64_t)(int64_t)floorf((p[3 * i] - ox) / dl);
        uint64_t iY = (uint64_t)(int64_t)floorf((p[3 * i + 1] - oy) / dl);
        uint64_t iZ = (uint64_t)(int64_t)floorf((p[3 * i + 2] - oz) / dl);
        ki[i].key = iX + NX * iY + NX * NY * iZ;
        ki[i].idx = i;
    }
    qsort(ki, (size_t)n, sizeof(key_idx_t), cmp_key_idx);
    int m = 0;
    for (int a = 0; a < n;) {
        int b = a;
        float sx = 0, sy = 0, sz = 0;
        int count = 0;
        while (b < n && ki[b].key == ki[a].key) {   /* ascending idx == input order */
            const float* q = p + 3 * ki[b].idx;
            sx += q[0]; sy += q[1]; sz += q[2];
            count++; b++;
        }
        float w = (float)(1.0 / count);              /* PointXYZ * (double -> float) */
        out[3 * m] = sx * w; out[3 * m + 1] = sy * w; out[3 * m + 2] = sz * w;
        if (out_key) out_key[m] = ki[a].key;
        m++;
        a = b;
    }
    free(ki);
    return m;
}

int orc_grid_subsample_batch(const float* pts, int n, const int* batches, int nb,
                             float dl, int max_p, float* out_pts, int* out_b,
                             uint64_t* out_key)
{
    if (max_p < 1) max_p = n;                        /* grid_subsampling.cpp:134-135 */
    int sum_b = 0, m_tot = 0;
    float* tmp = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    uint64_t* tk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    for (int b = 0; b < nb; b++) {
        if (sum_b + batches[b] > n) { free(tmp); free(tk); return -1; }
        int m = grid_subsample_one(pts + 3 * (size_t)sum_b, batches[b], dl, tmp, tk);
        if (m > max_p) m = max_p;                    /* :186-200, in OUR row order */
        memcpy(out_pts + 3 * (size_t)m_tot, tmp, sizeof(float) * 3 * (size_t)m);
        if (out_key) memcpy(out_key + m_tot, tk, sizeof(uint64_t) * (size_t)m);
        out_b[b] = m;
        m_tot += m;
        sum_b += batches[b];
    }
    free(tmp); free(tk);
    return m_tot;
}

/* ------------------------------------------------------------------ A2 ---- */
typedef struct { float d2; int idx; } dist_idx_t;

static int cmp_dist_idx(const void* a, const void* b)
{
    const dist_idx_t* x = (const dist_idx_t*)a;
    const dist_idx_t* y = (const dist_idx_t*)b;
    if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;   /* nanoflann.hpp:1286-1287 */
    return x->idx < y->idx ? -1 : (x->idx > y->idx);      /* tie order: ours (ref unspecified) */
}

/* Uniform-grid search instead of the reference's KD-tree: the accept test
 * (d2 < r2, nanoflann.hpp:249-253) and d2 itself (nanoflann.hpp:433-441:
 * result += diff*diff for x, y, z in turn, fp32) are what define the output. */
int orc_radius_neighbors(const float* q, int nq, const float* s, int ns,
                         const int* qb, const int* sb, int nb, float radius, int** out)
{
    float r2 = radius * radius;                     /* neighbors.cpp:228 */
    dist_idx_t** rows = (dist_idx_t**)calloc((size_t)(nq > 0 ? nq : 1), sizeof(dist_idx_t*));
    int* cnt = (int*)calloc((size_t)(nq > 0 ? nq : 1), sizeof(int));
    int max_count = 0;
    int sum_q = 0, sum_s = 0;
    for (int b = 0; b < nb; b++) {
        const float* S = s + 3 * (size_t)sum_s;
        int nsb = sb[b], nqb = qb[b];
        if (nsb > 0 && nqb > 0) {
            /* cell grid over this element's supports, cell edge >= radius */
            float mn[3] = { S[0], S[1], S[2] }, mx[3] = { S[0], S[1], S[2] };
            for (int i = 0; i < nsb; i++) for (int c = 0; c < 3; c++) {
                float v = S[3 * i + c];
                if (v < mn[c]) mn[c] = v;
                if (v > mx[c]) mx[c] = v;
            }
            double cell = radius > 0 ? (double)radius : 1.0;
            int dim[3];
            for (;;) {
                double tot = 1;
                for (int c = 0; c < 3; c++) {
                    dim[c] = (int)floor(((double)mx[c] - mn[c]) / cell) + 1;
                    tot *= dim[c];
                }
                if (tot <= 1.6e7) break;
                cell *= 2;
            }
            int ncell = dim[0] * dim[1] * dim[2];
            int* start = (int*)calloc((size_t)ncell + 1, sizeof(int));
            int* cid = (int*)malloc(sizeof(int) * (size_t)nsb);
            for (int i = 0; i < nsb; i++) {
                int c[3];
                for (int k = 0; k < 3; k++) {
                    c[k] = (int)floor(((double)S[3 * i + k] - mn[k]) / cell);
                    if (c[k] >= dim[k]) c[k] = dim[k] - 1;
                }
                cid[i] = c[0] + dim[0] * (c[1] + dim[1] * c[2]);
                start[cid[i] + 1]++;
            }
            for (int c = 0; c < ncell; c++) start[c + 1] += start[c];
            int* order = (int*)malloc(sizeof(int) * (size_t)nsb);
            int* fill = (int*)calloc((size_t)ncell, sizeof(int));
            for (int i = 0; i < nsb; i++) order[start[cid[i]] + fill[cid[i]]++] = i;
            free(fill); free(cid);

            for (int j = 0; j < nqb; j++) {
                const float* Q = q + 3 * (size_t)(sum_q + j);
                int lo[3], hi[3], skip = 0;
                for (int k = 0; k < 3; k++) {
                    double a = floor(((double)Q[k] - mn[k]) / cell) - 1;
                    double z = floor(((double)Q[k] - mn[k]) / cell) + 1;
                    if (z < 0 || a > dim[k] - 1) { skip = 1; break; }
                    lo[k] = a < 0 ? 0 : (int)a;
                    hi[k] = z > dim[k] - 1 ? dim[k] - 1 : (int)z;
                }
                int cap = 0, m = 0;
                dist_idx_t* row = NULL;
                if (!skip)
                for (int cz = lo[2]; cz <= hi[2]; cz++)
                for (int cy = lo[1]; cy <= hi[1]; cy++)
                for (int cx = lo[0]; cx <= hi[0]; cx++) {
                    int c = cx + dim[0] * (cy + dim[1] * cz);
                    for (int t = start[c]; t < start[c + 1]; t++) {
                        int i = order[t];
                        float d2 = 0, diff;
                        diff = Q[0] - S[3 * i];     d2 += diff * diff;
                        diff = Q[1] - S[3 * i + 1]; d2 += diff * diff;
                        diff = Q[2] - S[3 * i + 2]; d2 += diff * diff;
                        if (d2 < r2) {
                            if (m == cap) {
                                cap = cap ? 2 * cap : 32;
                                row = (dist_idx_t*)realloc(row, sizeof(dist_idx_t) * (size_t)cap);
                            }
                            row[m].d2 = d2;
                            row[m].idx = i + sum_s;  /* neighbors.cpp:321 */
                            m++;
                        }
                    }
                }
                if (m > 1) qsort(row, (size_t)m, sizeof(dist_idx_t), cmp_dist_idx);
                rows[sum_q + j] = row;
                cnt[sum_q + j] = m;
                if (m > max_count) max_count = m;
            }
            free(order); free(start);
        }
        sum_q += qb[b];
        sum_s += sb[b];
    }
    size_t tot = (size_t)nq * (size_t)max_count;
    int* o = (int*)malloc(sizeof(int) * (tot ? tot : 1));
    for (int j = 0; j < nq; j++) {
        for (int k = 0; k < max_count; k++)
            o[(size_t)j * max_count + k] = k < cnt[j] ? rows[j][k].idx : ns;  /* :322-324 */
        free(rows[j]);
    }
    free(rows); free(cnt);
    *out = o;
    return max_count;
}

/* ------------------------------------------------------------------ A6 ---- */
/* Upstream kernel shape [recalled]: T = min(512, 2^floor(log2 n)) threads stride
 * over k; per thread strict '>' keeps its first maximum; the pairwise tree
 * keeps the LOWER thread on equality. temp starts at 1e10; points with
 * x*x+y*y+z*z <= 1e-3 (compared in double) never update or compete. */
int orc_fps_threads(int n)
{
    /* upstream opt_n_threads(): pow_2 = log(n) / log(2) truncated, block = clamp(1 << pow_2, 1, 512) */
    if (n < 1) return 1;
    int pow_2 = (int)(log((double)n) / log(2.0));
    int p = pow_2 >= 9 ? 512 : (1 << pow_2);
    return p < 1 ? 1 : p;
}

void orc_fps(const float* xyz, int nbatch, int n, int m, int* idx)
{
    if (m <= 0) return;
    float* temp = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int T = orc_fps_threads(n);
    float* tb = (float*)malloc(sizeof(float) * (size_t)T);
    int* ti = (int*)malloc(sizeof(int) * (size_t)T);
    for (int b = 0; b < nbatch; b++) {
        const float* P = xyz + 3 * (size_t)b * n;
        int* I = idx + (size_t)b * m;
        for (int k = 0; k < n; k++) temp[k] = 1e10f;
        int old = 0;
        I[0] = 0;
        for (int j = 1; j < m; j++) {
            float x1 = P[3 * old], y1 = P[3 * old + 1], z1 = P[3 * old + 2];
            for (int t = 0; t < T; t++) { tb[t] = -1.0f; ti[t] = 0; }
            for (int k = 0; k < n; k++) {
                float x2 = P[3 * k], y2 = P[3 * k + 1], z2 = P[3 * k + 2];
                float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
                if ((double)mag <= 1e-3) continue;
                float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                float d2 = d < temp[k] ? d : temp[k];
                temp[k] = d2;
                int t = k % T;
                if (d2 > tb[t]) { tb[t] = d2; ti[t] = k; }
            }
            /* the upstream shared-memory tree: slot t absorbs slot t+s for s = T/2 .. 1 and keeps its own
             * candidate on equality, so among equal maxima the thread smallest in BIT-REVERSED order wins */
            for (int s = T / 2; s >= 1; s /= 2)
                for (int t = 0; t < s; t++)
                    if (tb[t + s] > tb[t]) { tb[t] = tb[t + s]; ti[t] = ti[t + s]; }
            old = ti[0];
            I[j] = old;
        }
    }
    free(temp); free(tb); free(ti);
}

/* ------------------------------------------------------------- A8 / A10 ---- */
void orc_ball_query(const float* xyz, const float* new_xyz, int nbatch, int n, int m,
                    float radius, int nsample, int* idx)
{
    float r2 = radius * radius;
    memset(idx, 0, sizeof(int) * (size_t)nbatch * m * nsample);
    for (int b = 0; b < nbatch; b++) {
        const float* P = xyz + 3 * (size_t)b * n;
        const float* Q = new_xyz + 3 * (size_t)b * m;
        int* I = idx + (size_t)b * m * nsample;
        for (int j = 0; j < m; j++) {
            float qx = Q[3 * j], qy = Q[3 * j + 1], qz = Q[3 * j + 2];
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; k++) {
                float x = P[3 * k], y = P[3 * k + 1], z = P[3 * k + 2];
                float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
                if (d2 < r2) {
                    if (cnt == 0) for (int l = 0; l < nsample; l++) I[(size_t)j * nsample + l] = k;
                    I[(size_t)j * nsample + cnt] = k;
                    cnt++;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ A18 ---- */
void orc_three_nn(const float* unknown, const float* known, int nbatch, int n, int m,
                  float* dist, int* idx)
{
    for (int b = 0; b < nbatch; b++)
    for (int j = 0; j < n; j++) {
        const float* u = unknown + 3 * ((size_t)b * n + j);
        double best1 = 1e40, best2 = 1e40, best3 = 1e40;
        int i1 = 0, i2 = 0, i3 = 0;
        for (int k = 0; k < m; k++) {
            const float* p = known + 3 * ((size_t)b * m + k);
            float d = (u[0] - p[0]) * (u[0] - p[0]) + (u[1] - p[1]) * (u[1] - p[1]) +
                      (u[2] - p[2]) * (u[2] - p[2]);
            if (d < best1)      { best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = k; }
            else if (d < best2) { best3 = best2; i3 = i2; best2 = d; i2 = k; }
            else if (d < best3) { best3 = d; i3 = k; }
        }
        float* D = dist + 3 * ((size_t)b * n + j);
        int* I = idx + 3 * ((size_t)b * n + j);
        D[0] = sqrtf((float)best1); D[1] = sqrtf((float)best2); D[2] = sqrtf((float)best3);
        I[0] = i1; I[1] = i2; I[2] = i3;
    }
}

/* ------------------------------------------------------------------ A12 ---- */
void orc_knn(const float* ref, const float* query, int nbatch, int n, int nq, int d, int k,
             float* dist, int64_t* idx)
{
    float* bd = (float*)malloc(sizeof(float) * (size_t)k);
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
    for (int b = 0; b < nbatch; b++)
    for (int j = 0; j < nq; j++) {
        const float* Q = query + (size_t)d * ((size_t)b * nq + j);
        int have = 0;
        for (int i = 0; i < n; i++) {
            const float* R = ref + (size_t)d * ((size_t)b * n + i);
            float ssd = 0;
            for (int c = 0; c < d; c++) { float t = R[c] - Q[c]; ssd += t * t; }
            /* ascending insertion, strict '<' keeps the earlier index on ties */
            int pos = have;
            while (pos > 0 && ssd < bd[pos - 1]) pos--;
            if (pos >= k) continue;
            int last = have < k ? have : k - 1;
            for (int t = last; t > pos; t--) { bd[t] = bd[t - 1]; bi[t] = bi[t - 1]; }
            bd[pos] = ssd; bi[pos] = i;
            if (have < k) have++;
        }
        for (int t = 0; t < k; t++) {
            dist[((size_t)b * nq + j) * k + t] = t < have ? sqrtf(bd[t]) : INFINITY;
            idx[((size_t)b * nq + j) * k + t] = t < have ? bi[t] : 0;
        }
    }
    free(bd); free(bi);
}


/* ================================================================================================
 * open3d 0.13.0 pre-processing (PARITY UNPINNED: restated from open3d's published sources)
 * ================================================================================================ */
typedef struct { uint64_t key; int idx; } o3d_kv;
static int o3d_kv_cmp(const void* a, const void* b)
{
    const o3d_kv* x = (const o3d_kv*)a; const o3d_kv* y = (const o3d_kv*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* PointCloud::VoxelDownSample (open3d/geometry/PointCloud.cpp) */
int orc_o3d_voxel_downsample(const double* pts, const double* normals, int n, double voxel, double* out_pts, double* out_normals)
{
    if (voxel <= 0.0) return -1;                       /* "[VoxelDownSample] voxel_size <= 0." */
    if (n <= 0) return 0;
    double mn[3] = { pts[0], pts[1], pts[2] }, mx[3] = { pts[0], pts[1], pts[2] };
    for (int i = 1; i < n; i++)
        for (int c = 0; c < 3; c++) {
            double v = pts[3 * (size_t)i + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
    double o[3];
    uint64_t N[3];
    for (int c = 0; c < 3; c++) {
        o[c] = mn[c] - voxel * 0.5;                    /* voxel_min_bound */
        N[c] = (uint64_t)(int64_t)(floor((mx[c] - o[c]) / voxel) + 1.0);
    }
    o3d_kv* kv = (o3d_kv*)malloc(sizeof(o3d_kv) * (size_t)n);
    if (!kv) return -2;
    for (int i = 0; i < n; i++) {
        uint64_t ix = (uint64_t)(int64_t)floor((pts[3 * (size_t)i] - o[0]) / voxel);
        uint64_t iy = (uint64_t)(int64_t)floor((pts[3 * (size_t)i + 1] - o[1]) / voxel);
        uint64_t iz = (uint64_t)(int64_t)floor((pts[3 * (size_t)i + 2] - o[2]) / voxel);
        kv[i].key = ix + N[0] * iy + N[0] * N[1] * iz;
        kv[i].idx = i;
    }
    qsort(kv, (size_t)n, sizeof(o3d_kv), o3d_kv_cmp);
    int m = 0;
    for (int a = 0; a < n;) {
        int b = a;
        double s[3] = { 0, 0, 0 }, sn[3] = { 0, 0, 0 };
        while (b < n && kv[b].key == kv[a].key) {      /* AccumulatedPoint::AddPoint, input order */
            for (int c = 0; c < 3; c++) {
                s[c] += pts[3 * (size_t)kv[b].idx + c];
                if (normals) sn[c] += normals[3 * (size_t)kv[b].idx + c];
            }
            b++;
        }
        for (int c = 0; c < 3; c++) {
            out_pts[3 * (size_t)m + c] = s[c] / (double)(b - a);
            if (normals) out_normals[3 * (size_t)m + c] = sn[c] / (double)(b - a);
        }
        m++;
        a = b;
    }
    free(kv);
    return m;
}

static void o3d_cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* EstimateNormals.cpp ComputeEigenvector0: A symmetric as (a00,a01,a02,a11,a12,a22) */
static void o3d_evec0(const double* A, double ev, double* out)
{
    double r0[3] = { A[0] - ev, A[1], A[2] }, r1[3] = { A[1], A[3] - ev, A[4] }, r2[3] = { A[2], A[4], A[5] - ev };
    double c[3][3];
    o3d_cross3(r0, r1, c[0]); o3d_cross3(r0, r2, c[1]); o3d_cross3(r1, r2, c[2]);
    int best = 0;
    double dbest = -1.0;
    for (int i = 0; i < 3; i++) {
        double d = c[i][0] * c[i][0] + c[i][1] * c[i][1] + c[i][2] * c[i][2];
        if (d > dbest) { dbest = d; best = i; }
    }
    double inv = 1.0 / sqrt(dbest);
    for (int i = 0; i < 3; i++) out[i] = c[best][i] * inv;
}

/* ComputeEigenvector1 */
static void o3d_evec1(const double* A, const double* w, double ev, double* out)
{
    double U[3], V[3];
    if (fabs(w[0]) > fabs(w[1])) {
        double inv = 1.0 / sqrt(w[0] * w[0] + w[2] * w[2]);
        U[0] = -w[2] * inv; U[1] = 0.0; U[2] = w[0] * inv;
    } else {
        double inv = 1.0 / sqrt(w[1] * w[1] + w[2] * w[2]);
        U[0] = 0.0; U[1] = w[2] * inv; U[2] = -w[1] * inv;
    }
    o3d_cross3(w, U, V);
    double AU[3], AV[3];
    AU[0] = A[0] * U[0] + A[1] * U[1] + A[2] * U[2]; AU[1] = A[1] * U[0] + A[3] * U[1] + A[4] * U[2]; AU[2] = A[2] * U[0] + A[4] * U[1] + A[5] * U[2];
    AV[0] = A[0] * V[0] + A[1] * V[1] + A[2] * V[2]; AV[1] = A[1] * V[0] + A[3] * V[1] + A[4] * V[2]; AV[2] = A[2] * V[0] + A[4] * V[1] + A[5] * V[2];
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if ((a00 > a01 ? a00 : a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            for (int i = 0; i < 3; i++) out[i] = m01 * U[i] - m00 * V[i];
        } else
            for (int i = 0; i < 3; i++) out[i] = U[i];
    } else {
        if ((a11 > a01 ? a11 : a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
            for (int i = 0; i < 3; i++) out[i] = m11 * U[i] - m01 * V[i];
        } else
            for (int i = 0; i < 3; i++) out[i] = U[i];
    }
}

/* FastEigen3x3 (D. Eberly, "A Robust Eigensolver for 3x3 Symmetric Matrices") */
void orc_o3d_fast_eigen3x3(const double* cov, double* nrm)
{
    double A[6], mc = cov[0];
    for (int i = 1; i < 6; i++) if (cov[i] > mc) mc = cov[i];
    if (mc == 0.0) { nrm[0] = nrm[1] = nrm[2] = 0.0; return; }
    for (int i = 0; i < 6; i++) A[i] = cov[i] / mc;
    double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        double q = (A[0] + A[3] + A[5]) / 3.0;
        double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        double c00 = b11 * b22 - A[4] * A[4], c01 = A[1] * b22 - A[4] * A[2], c02 = A[1] * A[4] - b11 * A[2];
        double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double hd = det * 0.5;
        if (hd < -1.0) hd = -1.0;
        if (hd > 1.0) hd = 1.0;
        double angle = acos(hd) / 3.0;
        double beta2 = cos(angle) * 2.0, beta0 = cos(angle + 2.09439510239319549) * 2.0, beta1 = -(beta0 + beta2);
        double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (hd >= 0) {
            o3d_evec0(A, e2, v2);
            if (e2 < e0 && e2 < e1) { memcpy(nrm, v2, sizeof v2); return; }
            o3d_evec1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(nrm, v1, sizeof v1); return; }
            o3d_cross3(v1, v2, nrm);
        } else {
            o3d_evec0(A, e0, v0);
            if (e0 < e1 && e0 < e2) { memcpy(nrm, v0, sizeof v0); return; }
            o3d_evec1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { memcpy(nrm, v1, sizeof v1); return; }
            o3d_cross3(v0, v1, nrm);
        }
    } else {
        nrm[0] = nrm[1] = nrm[2] = 0.0;
        if (cov[0] < cov[3] && cov[0] < cov[5]) nrm[0] = 1.0;
        else if (cov[3] < cov[0] && cov[3] < cov[5]) nrm[1] = 1.0;
        else nrm[2] = 1.0;
    }
}

typedef struct { double d2; int idx; } o3d_dn;
static int o3d_dn_cmp(const void* a, const void* b)
{
    const o3d_dn* x = (const o3d_dn*)a; const o3d_dn* y = (const o3d_dn*)b;
    if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* EstimateNormals + OrientNormalsTowardsCameraLocation */
void orc_o3d_estimate_normals(const float* pts, int n, int knn, const double* camera, int orient, float* normals)
{
    o3d_dn* dn = (o3d_dn*)malloc(sizeof(o3d_dn) * (size_t)(n > 0 ? n : 1));
    int k = knn < n ? knn : n;
    for (int i = 0; i < n; i++) {
        double q[3] = { pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2] };
        for (int j = 0; j < n; j++) {
            double dx = (double)pts[3 * (size_t)j] - q[0], dy = (double)pts[3 * (size_t)j + 1] - q[1], dz = (double)pts[3 * (size_t)j + 2] - q[2];
            dn[j].d2 = dx * dx + dy * dy + dz * dz;
            dn[j].idx = j;
        }
        qsort(dn, (size_t)n, sizeof(o3d_dn), o3d_dn_cmp);
        double nrm[3] = { 0.0, 0.0, 1.0 };
        if (k >= 3) {                                  /* ComputeCovariance: cumulants over the k neighbours */
            double cum[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int t = 0; t < k; t++) {
                const float* p = pts + 3 * (size_t)dn[t].idx;
                double x = p[0], y = p[1], z = p[2];
                cum[0] += x; cum[1] += y; cum[2] += z;
                cum[3] += x * x; cum[4] += x * y; cum[5] += x * z; cum[6] += y * y; cum[7] += y * z; cum[8] += z * z;
            }
            for (int t = 0; t < 9; t++) cum[t] /= (double)k;
            double cov[6] = { cum[3] - cum[0] * cum[0], cum[4] - cum[0] * cum[1], cum[5] - cum[0] * cum[2],
                              cum[6] - cum[1] * cum[1], cum[7] - cum[1] * cum[2], cum[8] - cum[2] * cum[2] };
            orc_o3d_fast_eigen3x3(cov, nrm);
        }
        if (nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2] == 0.0) { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }
        if (orient) {
            double r[3] = { camera[0] - q[0], camera[1] - q[1], camera[2] - q[2] };
            if (nrm[0] * r[0] + nrm[1] * r[1] + nrm[2] * r[2] < 0.0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
        }
        for (int c = 0; c < 3; c++) normals[3 * (size_t)i + c] = (float)nrm[c];
    }
    free(dn);
}
