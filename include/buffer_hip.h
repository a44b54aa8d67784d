/* buffer_hip.h -- C ABI of libbuffer_hip.so, the MI355X (gfx950) drop-in for the
 * native operators on BUFFER's registration-inference hot path.
 *
 * Conventions
 *   - every entry point returns 0 on success, a negative BUF_E* code on failure;
 *     buf_last_error() returns a thread-local message for the last failure.
 *   - pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). All work
 *     is enqueued asynchronously on it unless the comment says "synchronises".
 *   - scratch memory is caller-owned: size it with the matching *_ws_bytes().
 *   - no torch types, no C++ types, no hidden global state.
 *
 * Reference interfaces replaced (paths relative to the BUFFER repository):
 *   cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238        radius_neighbors.batch_query
 *   cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333      grid_subsampling.subsample_batch
 *   cpp_wrappers/cpp_subsampling/wrapper.cpp:338-566     grid_subsampling.subsample
 *   pointnet2_ops.pointnet2_utils (README.md:31)         furthest_point_sample, gather_operation,
 *                                                        ball_query, grouping_operation, three_nn
 *   knn_cuda.KNN (README.md:32)                          brute-force k-NN
 *   torch_batch_svd.svd (README.md:35)                   batched 3x3 SVD
 * and the fused device stages the build adds behind the same boundary
 * (models/point_learner.py, models/patch_embedder.py, models/BUFFER.py call sites
 * are cited per function below).
 */
#ifndef BUFFER_HIP_H
#define BUFFER_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BUF_OK            0
#define BUF_EINVAL       -1   /* bad argument (shape, null pointer, negative size) */
#define BUF_EHIP         -2   /* HIP runtime error (message carries hipGetErrorString) */
#define BUF_EWORKSPACE   -3   /* workspace too small */
#define BUF_ECAPACITY    -4   /* data-dependent capacity exceeded (e.g. voxel table) */
#define BUF_ENODEVICE    -5   /* no gfx950 device visible */

const char* buf_last_error(void);
int         buf_version(void);
/* Number of visible HIP devices (<0 on error); does not create a context. */
int         buf_device_count(void);

/* ------------------------------------------------------------------------------------------
 * A2  radius neighbours -- cpp_neighbors.batch_query (neighbors.cpp:211-332).
 *
 * d2 = ((dx*dx + dy*dy) + dz*dz) in fp32 without contraction, accept d2 < r*r,
 * rows ascending by (d2, support index), padded with ns_total.  Support indices are
 * global (stacked).  A cell grid (edge >= radius, auto-coarsened to `cells_per_elem`)
 * replaces the reference's KD-tree.
 *
 * buf_grid_t is a plain host struct filled by buf_grid_build and read by buf_grid_query;
 * the grid itself lives in the caller's workspace.
 */
typedef struct buf_grid {
    void*   ws;            /* device workspace handed to buf_grid_build            */
    size_t  ws_bytes;
    int     ns, nb;
    int64_t cells_per_elem;
    float   radius;
    /* device sub-allocations inside ws */
    void*   desc;          /* per-element grid descriptors                         */
    int*    s_off;         /* int32[nb+1] support offsets                          */
    int*    table;         /* int32[nb*cells_per_elem] inclusive cell ends         */
    void*   sorted;        /* float4[ns]: x,y,z,bitcast(global index), cell order  */
    int*    order;         /* int32[ns]: global support index in cell order        */
    void*   scan_tmp;
} buf_grid_t;

int64_t buf_grid_default_cells(int ns, int nb);
size_t  buf_grid_ws_bytes(int ns, int nb, int64_t cells_per_elem);
int     buf_grid_build(buf_grid_t* g, const float* supports, int ns, const int* s_batches_host, int nb,
                       float radius, int64_t cells_per_elem, void* ws, size_t ws_bytes, void* stream);
/* queries f32[nq,3]; q_order (nullable) int32[nq]: processing order (thread t handles query
 * q_order[t]) -- a spatially coherent order keeps a wavefront inside few cells; the result
 * does not depend on it.  nbr_out int32[nq,k_out] (k_out may be 0: count only);
 * counts_out (nullable) int32[nq] = untruncated neighbour counts; max_count_out (nullable)
 * int32[1], atomically max-ed (caller zeroes it).  radius may differ from the build radius
 * as long as it is <= the grid's cell edge. */
int     buf_grid_query(const buf_grid_t* g, const float* queries, int nq, const int* q_batches_host,
                       const int* q_order, float radius, int k_out, int* nbr_out, int* counts_out,
                       int* max_count_out, void* stream);
/* Build + query in one call (what batch_query does). */
int     buf_radius_neighbors(const float* queries, int nq, const float* supports, int ns,
                             const int* q_batches_host, const int* s_batches_host, int nb, float radius,
                             int k_out, int* nbr_out, int* counts_out, int* max_count_out,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A1  grid subsampling -- cpp_subsampling.subsample_batch (grid_subsampling.cpp:5-106,109-211).
 *
 * Voxel key and barycentre arithmetic are the reference's, in fp32, summed in input order;
 * rows are emitted per batch element in ascending voxel-key order (the reference emits
 * libstdc++ unordered_map order: same multiset of rows, bit for bit).
 * out_pts f32[n,3] (capacity n rows); out_batches_host int32[nb]; returns M via *out_m_host.
 * `max_cells` bounds the dense voxel table (sum over elements); BUF_ECAPACITY if exceeded.
 * Synchronises `stream` (the row count has to reach the host).
 */
size_t  buf_grid_subsample_ws_bytes(int n, int nb, int64_t max_cells);
int     buf_grid_subsample_batch(const float* pts, int n, const int* batches_host, int nb, float dl,
                                 int max_p, float* out_pts, int* out_batches_host, int* out_m_host,
                                 int64_t max_cells, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BUFFER_HIP_H */
