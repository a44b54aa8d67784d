// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// extern "C" doorway into the reference's own C++ cores, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile, target
// `_ref`).  Nothing from the reference is copied here: this file only declares
// the two entry points it calls and marshals flat arrays into the
// std::vector<PointXYZ> containers those entry points take.
//
//   batch_nanoflann_neighbors  cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332
//   batch_grid_subsampling     cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211
//
// The stock CPython wrappers (wrapper.cpp) do not build against NumPy 2.x, so
// ctypes goes through this shim instead.
#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"
#include <cstring>
#include <cstdlib>

extern "C" {

// Returns max_count (columns); *out is malloc'd int32[Nq*max_count], free with ref_free.
int ref_batch_query(const float* q, int nq, const float* s, int ns,
                    const int* qb, const int* sb, int nb, float radius, int** out)
{
    std::vector<PointXYZ> queries((const PointXYZ*)q, (const PointXYZ*)q + nq);
    std::vector<PointXYZ> supports((const PointXYZ*)s, (const PointXYZ*)s + ns);
    std::vector<int> q_batches(qb, qb + nb), s_batches(sb, sb + nb);
    std::vector<int> nbrs;
    batch_nanoflann_neighbors(queries, supports, q_batches, s_batches, nbrs, radius);
    int maxc = nq > 0 ? (int)(nbrs.size() / (size_t)nq) : 0;
    *out = (int*)malloc(sizeof(int) * (nbrs.size() ? nbrs.size() : 1));
    if (!nbrs.empty()) memcpy(*out, nbrs.data(), sizeof(int) * nbrs.size());
    return maxc;
}

// Returns M (rows); *out_pts is malloc'd float[M*3]; out_b is caller-owned int[nb].
int ref_subsample_batch(const float* p, int n, const int* b, int nb, float dl, int max_p,
                        float** out_pts, int* out_b)
{
    std::vector<PointXYZ> pts((const PointXYZ*)p, (const PointXYZ*)p + n);
    std::vector<int> batches(b, b + nb);
    std::vector<PointXYZ> sub;
    std::vector<float> f0, f1;
    std::vector<int> c0, c1, sb;
    batch_grid_subsampling(pts, sub, f0, f1, c0, c1, batches, sb, dl, max_p);
    for (int i = 0; i < nb; i++) out_b[i] = sb[i];
    *out_pts = (float*)malloc(sizeof(float) * 3 * (sub.size() ? sub.size() : 1));
    if (!sub.empty()) memcpy(*out_pts, sub.data(), sizeof(float) * 3 * sub.size());
    return (int)sub.size();
}

void ref_free(void* p) { free(p); }

}
