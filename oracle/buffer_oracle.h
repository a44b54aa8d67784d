/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference algorithms on
 * BUFFER's registration-inference hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (buffer_amd/, libbuffer_hip.so) never does.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  All index decisions are plain IEEE fp32 in the
 * written evaluation order, FMA contraction off (-ffp-contract=off).
 */
#ifndef BUFFER_ORACLE_H
#define BUFFER_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106,109-211
 * out_pts has room for n rows; out_b for nb counts.  Rows are emitted per batch
 * element in ASCENDING VOXEL KEY order (the reference emits libstdc++
 * unordered_map iteration order; the multiset of rows is identical, bit for bit).
 * out_key (optional, may be NULL) receives the uint64 voxel key per row.
 * Returns number of rows M, or <0 on error. */
int orc_grid_subsample_batch(const float* pts, int n, const int* batches, int nb,
                             float dl, int max_p, float* out_pts, int* out_b,
                             uint64_t* out_key);

/* cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332 (+ nanoflann.hpp:249-253,433-441,1286-1287)
 * Returns max_count; *out is malloc'd int32[nq*max_count] (free with orc_free).
 * Rows sorted ascending by (d2, index); padded with ns (total support count). */
int orc_radius_neighbors(const float* q, int nq, const float* s, int ns,
                         const int* qb, const int* sb, int nb, float radius, int** out);
void orc_free(void* p);

/* pointnet2_ops furthest_point_sample [recalled upstream semantics, SURVEY Appendix C].
 * xyz[b][n][3] -> idx[b][m]. */
void orc_fps(const float* xyz, int b, int n, int m, int* idx);

/* pointnet2_ops ball_query [recalled]: first nsample hits in index order, d2 < r2,
 * first hit replicated into all slots, zero when no hit. idx[b][m][nsample]. */
void orc_ball_query(const float* xyz, const float* new_xyz, int b, int n, int m,
                    float radius, int nsample, int* idx);

/* pointnet2_ops three_nn [recalled]: dist = sqrt of 3 smallest d2, idx int32. */
void orc_three_nn(const float* unknown, const float* known, int b, int n, int m,
                  float* dist, int* idx);

/* knn_cuda.KNN(k, transpose_mode=True) [recalled]: ref[b][n][d], query[b][q][d]
 * -> dist[b][q][k] (Euclidean, ascending), idx int64[b][q][k]. */
void orc_knn(const float* ref, const float* query, int b, int n, int q, int d, int k,
             float* dist, int64_t* idx);

/* open3d 0.13.0 PointCloud::VoxelDownSample (open3d/geometry/PointCloud.cpp; pip dependency of the reference,
 * README.md:28, called at ThreeDMatch/dataset.py:93,104,125,129) -- PARITY UNPINNED (open3d absent from
 * /root/reference and from this image; restated from its published source).
 * voxel_min_bound = min - voxel/2, index = floor((p - voxel_min_bound)/voxel), row = fp64 mean of the voxel's points
 * (and normals, may be NULL) in input order; rows in ascending voxel-key order (open3d: unordered_map order).
 * pts f64[n,3]; out_pts/out_normals have room for n rows.  Returns the number of rows, <0 on error. */
int orc_o3d_voxel_downsample(const double* pts, const double* normals, int n, double voxel, double* out_pts, double* out_normals);

/* open3d 0.13.0 EstimateNormals(KDTreeSearchParamKNN(knn)) + OrientNormalsTowardsCameraLocation
 * (open3d/geometry/EstimateNormals.cpp: ComputeCovariance by cumulants, FastEigen3x3 = Eberly's robust 3x3 symmetric
 * eigen solver, smallest eigenvector; ThreeDMatch/dataset.py:141-150) -- PARITY UNPINNED, as above.
 * Brute-force k nearest by (fp64 d2, index), the query point included.  pts f32[n,3] -> normals f32[n,3]. */
void orc_o3d_estimate_normals(const float* pts, int n, int knn, const double* camera, int orient, float* normals);
/* the eigen solver alone: cov = (c00,c01,c02,c11,c12,c22) -> eigenvector of the smallest eigenvalue */
void orc_o3d_fast_eigen3x3(const double* cov, double* normal);

#ifdef __cplusplus
}
#endif
#endif
