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
 *   - no torch types, no C++ types.  Process state is limited to: the thread-local error message, the per-device
 *     record of raised dynamic-LDS limits (hipFuncSetAttribute is per device; a process may drive several GPUs),
 *     and the optional HIP-event timing registry behind buf_timing_enable (off by default, mutex-guarded).
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
/* Measurement aid (off by default): when enabled, buf_grid_query (kernel id 0), buf_cylindrical_net (1) and
 * buf_cost_volume_net (2) bracket their kernel with HIP events on the launch stream.
 * buf_timing_collect_kernel synchronises on the events of one kernel id, returns the number of launches and
 * their summed duration / algorithmic work (bytes for id 0, flops for ids 1 and 2), and drops them.
 * buf_timing_collect = buf_timing_collect_kernel(0, ...). */
#define BUF_TIMED_GRID_QUERY 0
#define BUF_TIMED_CYL_NET    1
#define BUF_TIMED_COST_NET   2
#define BUF_TIMED_SELECT_PATCHES 3   /* A8:  work = 12*Nf + 12*P + 12*P*nsample bytes */
#define BUF_TIMED_PATCH_VOXELIZE 4   /* A10: work = 12*P*npts + 4*P*16*ncentres bytes */
#define BUF_TIMED_FPS            5   /* A6:  work = rounds (m) of the launch; bytes = 12*N' + 4*m per cloud */
#define BUF_TIMED_NN1            6   /* A12: work = 2*Q*N*D flops */
#define BUF_TIMED_VN_GATHER      7   /* A4:  work = 4*N*K + 12*N + 12*N*Cin + 12*N*Cout bytes */
#define BUF_TIMED_GRID_SUBSAMPLE 8   /* A1:  work = 12*N + 12*M(capacity N) + 4*B bytes; spans the whole kernel sequence */
#define BUF_TIMED_DESC_HEAD      9   /* A11 tail: work = (2*32*140*4 + 128) bytes per patch */
#define BUF_TIMED_CYL_NET_SPLIT  10  /* A11 dense, split-f16 form (buf_cylindrical_net_split): work = dense flops, as id 1 */
#define BUF_TIMED_COST_NET_SPLIT 11  /* A13, split-f16 form */
#define BUF_TIMED_NKERNELS       12
void        buf_timing_enable(int on);
long long   buf_timing_collect(double* total_ms, double* total_bytes);
long long   buf_timing_collect_kernel(int kernel_id, double* total_ms, double* total_work);

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
    const float* supports; /* the support array the grid was built from (must stay alive while queried) */
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
/* queries f32[nq,3]; q_order (nullable) int32[nq]: processing order, any permutation of 0..nq-1 (slot t handles query
 * q_order[t]) -- a spatially coherent order keeps a wavefront inside few cells; the result does not depend on it.  The fast
 * kernel takes a slot's batch element from the slot, so an order that keeps every element's queries inside that element's
 * range [q_off[b], q_off[b+1]) runs entirely on it (the grid's own order, or the order of another grid over the same
 * clouds, do); a query placed in another element's range is handed to the slower lane-per-query pass (same result).
 * Passing the grid's own supports with q_order == g->order (a self query in cell order) lets the kernel take index and
 * coordinates from the cell-ordered array in one load.  nbr_out int32[nq,k_out] (k_out may be 0: count only);
 * counts_out (nullable) int32[nq] = untruncated neighbour counts; max_count_out (nullable)
 * int32[1], atomically max-ed (caller zeroes it).  radius may differ from the build radius
 * as long as it is <= the grid's cell edge. */
int     buf_grid_query(const buf_grid_t* g, const float* queries, int nq, const int* q_batches_host,
                       const int* q_order, float radius, int k_out, int* nbr_out, int* counts_out,
                       int* max_count_out, void* todo_ws, void* stream);
/* todo_ws: int32[nq] scratch (rows longer than 64 neighbours -- and, in the cell-centric self query, cells whose 27-cell
 * candidate set exceeds the LDS stage -- are redone by a second, unbounded pass).  It may be null ONLY when k_out == 0 and
 * either q_order is null or the call is a self query (queries == the grid's supports, nq == ns, q_order == g->order); a
 * self query without it runs on the query-centric kernel.  A self query WITH it may write the list in the count-only
 * (k_out == 0) pass as well (stage-overflow queries), so the workspace must hold nq ints there too.
 * Build + query in one call (what batch_query does); ws >= buf_grid_ws_bytes(ns,nb,0) + 4*nq bytes. */
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
 * When every element has <= 16384 points the call is one workgroup per element with the sort in LDS (k_vox_fused:
 * no table in global memory, `max_cells` unused, no capacity to exceed); otherwise, or with BUF_VOX_FUSED=0 in the
 * environment, the global-table counting sort.  Both forms emit the same rows.
 * Synchronises `stream` (the row count has to reach the host).
 */
size_t  buf_grid_subsample_ws_bytes(int n, int nb, int64_t max_cells, int fdim);
/* feats (nullable) f32[n,fdim] -> out_feats f32[M,fdim]: per-voxel feature means, summed in input order. */
int     buf_grid_subsample_batch(const float* pts, int n, const int* batches_host, int nb, float dl,
                                 int max_p, const float* feats, int fdim, float* out_pts, float* out_feats,
                                 int* out_batches_host, int* out_m_host,
                                 int64_t max_cells, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * pointnet2_ops.pointnet2_utils surface (external CUDA package, README.md:31; call sites
 * models/BUFFER.py:266-271, models/patch_embedder.py:100-104, utils/common.py:442-455).
 * All tensors contiguous fp32 / int32 in device memory.
 */
/* furthest_point_sample: xyz f32[b,n,3] -> idx int32[b,m]; starts at index 0, skips points with
 * |p|^2 <= 1e-3, arg-max tie rule of the upstream 512-thread kernel.  ws only for n > 32768. */
size_t  buf_fps_ws_bytes(int b, int n);
int     buf_fps(const float* xyz, int b, int n, int m, int* idx_out, void* ws, size_t ws_bytes, void* stream);
/* ragged batch (clouds of different size sampled concurrently, one workgroup each): xyz f32[sum(n),3] stacked,
 * lengths_host int[b] -> idx int32[b,m], indices local to each cloud.  ws sized for (1, sum(n)). */
int     buf_fps_ragged(const float* xyz, const int* lengths_host, int b, int m, int* idx_out, void* ws, size_t ws_bytes,
                       void* stream);
/* gather_operation: feat f32[b,c,n], idx int32[b,m] -> f32[b,c,m] */
int     buf_gather(const float* feat, const int* idx, int b, int c, int n, int m, float* out, void* stream);
/* grouping_operation: feat f32[b,c,n], idx int32[b,m,nsample] -> f32[b,c,m,nsample] */
int     buf_group(const float* feat, const int* idx, int b, int c, int n, int m, int nsample, float* out, void* stream);
/* ball_query: first nsample points (index order) with d2 < radius^2; unused slots = first hit; rows
 * without any hit are all zero.  idx int32[b,m,nsample]. */
int     buf_ball_query(const float* xyz, const float* new_xyz, int b, int n, int m, float radius, int nsample,
                       int* idx, void* stream);
/* three_nn: unknown f32[b,n,3], known f32[b,m,3] -> dist f32[b,n,3] (sqrt of the 3 smallest d2), idx int32[b,n,3] */
int     buf_three_nn(const float* unknown, const float* known, int b, int n, int m, float* dist, int* idx, void* stream);
/* MiniSpinNet.select_patches fused (models/patch_embedder.py:93-121): pts f32[n,3] (already permuted),
 * kpts f32[m,3] -> patches f32[m,nsample,3] with the keypoint in every unused slot and in the last slot. */
int     buf_select_patches(const float* pts, const float* kpts, int n, int m, float radius, int nsample,
                           float* patches, void* stream);
/* The same (models/patch_embedder.py:93-121) for every cloud of a step in one launch: pts f32[sum n_c,3] = the (already permuted) support clouds stacked,
 * lengths_host int[nc], kpts f32[nc*m,3] (m keypoints per cloud) -> patches f32[nc*m,nsample,3].  Builds an A2 cell grid
 * over the stacked clouds in `ws` and walks a per-query bitmask of in-ball points in index order (identical results to
 * buf_select_patches cloud by cloud). */
size_t  buf_select_patches_batched_ws_bytes(int n_total, int nc);
int     buf_select_patches_batched(const float* pts, const int* lengths_host, int nc, const float* kpts, int m, float radius,
                                   int nsample, float* patches, void* ws, size_t ws_bytes, void* stream);
/* The random permutation of the support cloud that precedes the ball query (patch_embedder.py:97-98: torch.randperm) for nc
 * clouds in one launch: out[off_c + j] = cloud_c[prp_c(j)], prp_c = a keyed pseudo-random permutation of [0, n_c) (4-round
 * Feistel network, cycle-walked).  clouds_host: nc HOST-side pointers to DEVICE arrays f32[n_c,3]; keys_host: nc keys. */
int     buf_permute_clouds(const float* const* clouds_host, const int* lengths_host, const unsigned long long* keys_host,
                           int nc, float* out, void* stream);

/* A5b  ordered compaction of `x > threshold` (torch.where at models/BUFFER.py:255-259): ascending indices.
 * x f32 read with element stride `stride` (score[:,0] of an [n,1] tensor: stride 1); idx_out int32[n] capacity;
 * count_out int32[1] on the device. */
size_t  buf_compact_ws_bytes(int n);
int     buf_compact_greater(const float* x, int stride, int n, float threshold, int* idx_out, int* count_out,
                            void* ws, size_t ws_bytes, void* stream);

/* knn_cuda.KNN(k, transpose_mode=True) (README.md:32; models/BUFFER.py:347,352):
 * ref f32[b,n,d], query f32[b,q,d] -> dist f32[b,q,k] (Euclidean, ascending), idx int64[b,q,k]. d,k <= 64. */
size_t  buf_knn_ws_bytes(int b, int q, int k);
/* k = 1, d = 32 (the mutual-matching calls): with a workspace of buf_knn1_ws_bytes(b, n, q) bytes buf_knn ranks the pairs on the
 * f16 matrix pipe and forms the reference's fp32 sum only for the pairs it cannot separate from the best one (csrc/pointops.hip,
 * k_nn1f_*): same indices and distances, bit for bit, as the exact scan it falls back to with the smaller workspace. */
size_t  buf_knn1_ws_bytes(int b, int n, int q);
int     buf_knn(const float* ref, const float* query, int b, int n, int q, int d, int k, float* dist,
                long long* idx, void* ws, size_t ws_bytes, void* stream);

/* torch_batch_svd.svd (README.md:35; utils/common.py:715): a f32[n,3,3] -> u,s,v with a = u diag(s) v^T, s descending. */
int     buf_svd3x3_batched(const float* a, int n, float* u, float* s, float* v, void* stream);

/* ------------------------------------------------------------------------------------------
 * A4/A5  Vector-Neuron blocks (models/point_learner.py:315-416,467-582,246-265; models/vn_layers.py:46-75,108-130).
 * Feature rows f32[N,3C] channel-major/xyz-minor.  bn_scale = w/sqrt(var+1e-5), bn_shift = b - mean*bn_scale
 * (both null when Cout == 1: the reference skips VN batch-norm there).
 */
/* gather + VN-linear + VN-BN + VN-leaky + mean over all k slots.  mode 1: [f,delta]; mode 6: [f,delta,f x delta,mean(delta)].
 * idx int32[nq,k] with shadow index >= ns.  wf,wd f32[cout, cin+1 | cin+3]. */
int     buf_vn_gather_block(const float* q_pts, const float* s_pts, const float* feats, const int* idx,
                            int nq, int ns, int k, int cin, int cout, int mode, float scale,
                            const float* wf, const float* wd, const float* bn_scale, const float* bn_shift,
                            float slope, float* out, void* stream);
/* The same for mode '1' with the channel contraction hoisted out of the neighbour loop (csrc/vn.hip: the feature part of both
 * VN-linear maps is formed once per SUPPORT point, a neighbour slot adds the delta column): 4-5 x fewer operations in the resnet
 * blocks, results within half an ulp of a partial sum of buf_vn_gather_block.  ws: buf_vn_gather_pre_ws_bytes(ns, cout) bytes. */
size_t  buf_vn_gather_pre_ws_bytes(int ns, int cout);
int     buf_vn_gather_block_pre(const float* q_pts, const float* s_pts, const float* feats, const int* idx,
                                int nq, int ns, int k, int cin, int cout, float scale,
                                const float* wf, const float* wd, const float* bn_scale, const float* bn_shift,
                                float slope, float* out, void* ws, size_t ws_bytes, void* stream);
/* point-wise VN layer on concat(a[ind_a[i*ind_stride]] , b[i]) (+ residual); ind_a null = identity;
 * rows with index >= na read as zeros (closest_pool shadow); wd null = plain VN linear. */
int     buf_vn_pointwise(const float* a, const int* ind_a, int ind_stride, int na, int ca, const float* b, int cb,
                         int n, int cout, const float* wf, const float* wd, const float* bn_scale,
                         const float* bn_shift, float slope, const float* residual, float* out, void* stream);
/* max_pool (models/KPConv/blocks.py:104-121): out[i,f] = max_k feats_pad[idx[i,k], f], zero shadow row. */
int     buf_gather_max(const float* feats, const int* idx, int nq, int ns, int k, int width, float* out, void* stream);
/* Conv1d(kernel 1) of the score heads (models/point_learner.py:128-136,163-171): out[i] = W x[i] + b, x f32[n,cin],
 * W f32[cout,cin], b f32[cout] (device), cin, cout <= 32; activation 0 none, 1 sigmoid, 2 softplus. */
int     buf_row_linear(const float* x, int n, int cin, int cout, const float* w, const float* b, int activation, float* out,
                       void* stream);
/* InstanceNorm1d (biased variance, no affine) over contiguous row segments: the score heads of
 * models/point_learner.py:128-136,163-171 normalise over the stacked points of ONE pair; lens_host[s] rows per
 * segment (HOST int[nseg], sum = n).  x f32[n,c] -> out f32[n,c].  Deterministic (no atomics). */
size_t  buf_segment_instance_norm_ws_bytes(int nseg, int c);
int     buf_segment_instance_norm(const float* x, int n, int c, const int* lens_host, int nseg, float eps, float* out,
                                  void* ws, size_t ws_bytes, void* stream);
/* VNStdFeature tail (vn_layers.py:213-219): x f32[n,3c], z f32[n,9] -> f32[n,3c] invariant scalars. */
int     buf_vn_std(const float* x, const float* z, int n, int c, float* out, void* stream);
/* One whole score head of models/point_learner.py:128-136 (Ref.inv_layer -> eps) / :163-171 (Keypt.invar_layer -> saliency) in 7
 * launches: VNStdFeature (vn1 -> vn2 -> vn_lin -> x . z, vn_layers.py:169-222) + Conv1d 30 -> 20 | InstanceNorm1d over the pair |
 * Conv1d 20 -> c1 | InstanceNorm1d | Conv1d c1 -> 1 + activation (0 none, 1 sigmoid, 2 softplus).  x f32[n,30] -> out f32[n,1];
 * lens_host: rows per pair (HOST int[nseg], sum = n).  vn*_wf / _wd: map_to_feat / map_to_dir [Cout,Cin]; _bsc / _bsh: folded VN
 * batch-norm (null: none); lin: vn_lin's map_to_feat [3,5]; w* / b*: Conv1d weights [Cout,Cin] / bias.  Only the released
 * widths (10 -> 10 -> 5 -> 3; 30 -> 20 -> c1 <= 10 -> 1); bit-identical to buf_vn_pointwise x 3, buf_vn_std, buf_row_linear and
 * buf_segment_instance_norm called one after the other. */
size_t  buf_score_head_ws_bytes(int n, int nseg);
int     buf_score_head(const float* x, int n, const int* lens_host, int nseg,
                       const float* vn1_wf, const float* vn1_wd, const float* vn1_bsc, const float* vn1_bsh, float vn1_slope,
                       const float* vn2_wf, const float* vn2_wd, const float* vn2_bsc, const float* vn2_bsh, float vn2_slope,
                       const float* lin, const float* w0, const float* b0, const float* w1, const float* b1, int c1,
                       const float* w2, const float* b2, int final_activation, float eps, float* out,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A9+A10 (+ point MLP of A11) fused: axis alignment, normalisation, cylindrical voxelisation
 * (420 ball queries of `nsample` per patch), azimuth de-rotation, Conv1x1(3->16)+BN+ReLU, max.
 * (models/patch_embedder.py:123-171,74-79; utils/common.py:431-498,501-525)
 * patches f32[np,npts,3] (keypoint in the last slot); axis f32[np,3] or null (KITTI/ETH: R = I);
 * centres f32[ncentres,3]; azi_cs f32[azi_n,2] = cos,sin of -i*2pi/azi_n; mlp_* are HOST arrays
 * (w[16,3], b[16], bn_scale[16], bn_shift[16]).  out_x f32[np,16,ncentres]; out_R f32[np,3,3];
 * out_rand f32[np,3]; out_patches (nullable) f32[np,npts,3].
 * ws: device workspace of buf_patch_voxelize_ws_bytes(ncentres) bytes (the split-f16 operand tables of the distance and
 * MLP matrix instructions, rebuilt by every call; BUF_EWORKSPACE when too small).  The hit decisions are the reference's
 * fp32 test `d2 < r2` bit for bit (csrc/voxelize.hip: matrix-pipe filter + fp32 re-test of the pairs it cannot call).
 */
size_t  buf_patch_voxelize_ws_bytes(int ncentres);
int     buf_patch_voxelize(const float* patches, const float* axis, int npatch, int npts, float des_r,
                           const float* centres, int ncentres, int azi_n, const float* azi_cs, float voxel_r,
                           int nsample, const float* mlp_w_host, const float* mlp_b_host, const float* bn_scale_host,
                           const float* bn_shift_host, float* out_x, float* out_R, float* out_rand, float* out_patches,
                           void* ws, size_t ws_bytes, void* stream);

/* A11 (dense)  Cylindrical_Net (models/patchnet.py:15-85): Conv3d(16->64, 3x3x3) + 7 x Conv2d 3x3, BatchNorms folded, circular
 * azimuth / zero elevation padding (utils/common.py:265-310), as ONE kernel in the Winograd F(2x2,3x3) domain, all fp32
 * (csrc/convnet_wg.hip: 40 instead of 75 matrix instructions per 4 input x 16 output channels; a different fp32 summation
 * order, within 1e-6 of the output scale of a direct-form evaluation: tests/native/convnet_direct.hip is that cross-check).
 * x f32[np,48,140] (= [np,16,3,7,20]) -> y f32[np,32,140] (= [np,32,7,20]); bias_host[l]: DEVICE pointers to [Cout].
 * wt_host[l]: DEVICE pointers to U = G g G^T of the BN-folded filters ([Cout,Cin,3,3]; layer 0: Cin = c16*3 + depth) in the
 * tiling [N-group][i][k-step][n2][lk][li][j] = U[i][j][16 (NG g + n2) + li][4 ks + lk] with NG = buf_winograd_group(Cin, Cout)
 * N-tiles per group (the N-tiles one wavefront owns: its k-steps are contiguous in memory), as buf_winograd_tile_weights lays
 * them out (fp64 on the host; the bottom tile row, whose window ends in the elevation padding, runs as a two-tap form in
 * elevation -- 8 instead of 12 matrix instructions -- on U_0 and U_1 - U_2, the latter formed in registers).  Cin a multiple of 16 (of 32 for Cout 128), Cout in {32, 64, 128}, last layer 32. */
int     buf_winograd_group(int cin, int cout);                                                   /* host only: NG of a layer */
int     buf_winograd_tile_weights(const float* w_host, int cout, int cin, float* out_host);   /* host only: [Cout,Cin,3,3] -> 16*Cout*Cin floats */
int     buf_winograd_tile_filters(const float* w_host, int cout, int cin, int ng, int nblk, float* out_host);
                                                   /* host only: the same tiling with N-groups of ng and nblk = 4 | 5 blocks -> 4*nblk*Cout*Cin floats */
int     buf_cylindrical_net_wg(const float* x, int npatch, const float* const* wt_host, const float* const* bias_host,
                               const int* cin_host, const int* cout_host, const int* relu_host, float* y, void* stream);
int     buf_cylindrical_net_wg_supports(const int* cin_host, const int* cout_host);   /* host only: 0 if the kernel is built for these 8 widths */

/* A11 (dense), split-f16 form -- the same stack with fp32-EQUIVALENT arithmetic on the f16 matrix pipe (csrc/convnet_h3.hip; opt-in,
 * the all-fp32 kernel above stays the default): every fp32 operand is split once into hi = f16(x) and lo' = f16((x - hi) 2^11)
 * (x = hi + 2^-11 lo' to one fp32 ulp), a product sum is [sum hi hi] + 2^-11 [sum hi lo' + sum lo' hi] in two fp32 accumulators:
 * three v_mfma_f32_16x16x32_f16 per (16 outputs x 16 positions x 32 channels), direct 9-tap form.  Measured against the float64
 * stack: not worse than the fp32 kernels (tests/test_model_gpu.py, profiles/r04_f16_split.txt).  Requires every activation and
 * weight below 65504 in magnitude (f16 range): weights are checked by the tiler, an activation that leaves the range sets bit 0
 * of *status_dev (DEVICE int32, nullable; the caller clears and reads it) -- the BARE kernel: a caller that does not manage the range
 * itself uses buf_cylindrical_net_split_safe below, which re-runs such patches on the fp32 kernel.
 * x f32[np,Cin0,140] -> y f32[np,32,140]; bias_host[l]: DEVICE f32[Cout]; wt_host[l]: DEVICE u16 planes as buf_split_tile_filters
 * lays them out from the BN-folded filters [Cout,Cin,3,3]:
 *   out[((((g 9 + tap) KS + ks) 2 + n2) 2 + plane) 512 + (kg 16 + row) 8 + i] = plane(w[32 g + 16 n2 + row][32 ks + 8 kg + i][tap]),
 *   tap = 3 ky + kx, KS = ceil(Cin / 32) (channels beyond Cin zero), plane 0 = hi, plane 1 = lo'; buf_split_filter_count u16 values.
 * Cin a multiple of 16, <= 128, not in 65..96; Cout in {32, 64, 128}; last layer 32. */
long long buf_split_filter_count(int cout, int cin);                                             /* host only */
int     buf_split_tile_filters(const float* w_host, int cout, int cin, unsigned short* out_host);   /* host only */
int     buf_cylindrical_net_split(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host,
                                  const int* cin_host, const int* cout_host, const int* relu_host, float* y, int* status_dev, void* stream);
/* The same stack with buf_descriptor_head fused behind its last layer (the [32][140] map stays in LDS): head_params = DEVICE
 * f32[545] as for buf_descriptor_head -> desc f32[np,32], equi f32[np,32,140]; bit-identical to the two calls in sequence. */
int     buf_cylindrical_net_split_head(const float* x, int npatch, const void* const* wt_host, const float* const* bias_host,
                                       const int* cin_host, const int* cout_host, const int* relu_host, const float* head_params,
                                       float* desc, float* equi, int* status_dev, void* stream);

/* The split form made SAFE BY CONSTRUCTION (csrc/split_safe.hip; what buffer_amd.ops.CylindricalNetSplit calls): flags_ws (DEVICE
 * int32[np], scratch) is cleared, the split kernel sets flags_ws[p] for every patch whose input or hidden activation left the f16 range
 * (NaN included), then buf_cylindrical_net_wg's kernel runs over the same grid with the flags as a mask -- workgroups of clear patches
 * return at once, flagged patches are recomputed in fp32 and overwrite their result (bit-identical to buf_cylindrical_net_wg [+
 * buf_descriptor_head] for those patches).  Same stream, no host round trip; nothing is returned from behind an overflow.
 * head_params null: y_or_equi = y f32[np,32,140]; else desc f32[np,32] and y_or_equi = equi f32[np,32,140].
 * wt_split_host as buf_cylindrical_net_split, wt_wg_host as buf_cylindrical_net_wg, bias_host shared.  Replaces the model's
 * torch layers of models/patchnet.py:15-85 exactly like the two entry points it combines. */
int     buf_cylindrical_net_split_safe(const float* x, int npatch, const void* const* wt_split_host, const float* const* wt_wg_host,
                                       const float* const* bias_host, const int* cin_host, const int* cout_host, const int* relu_host,
                                       const float* head_params, float* y_or_equi, float* desc, int* status_dev, int* flags_ws, void* stream);

/* A11 (head)  attention pooling + normalisation (models/patch_embedder.py:66-72,81-84): pool_layer
 * (Conv2d 1x1 32->16 + BN + ReLU, Conv2d 1x1 16->1 + BN + ReLU), desc = normalize(mean(y * w)),
 * equi = normalize(y, channel).  y f32[np,32,140] -> desc f32[np,32], equi f32[np,32,140].
 * params: DEVICE f32[545] = w0 [16][32], b0 [16], w3 [16], b3 [1] with the BatchNorms folded. */
int     buf_descriptor_head(const float* y, int npatch, const float* params, float* desc, float* equi, void* stream);

/* ------------------------------------------------------------------------------------------
 * A13  CostVolume + CostNet (models/BUFFER.py:37-66, models/patchnet.py:88-147) fused on fp32 MFMA: the
 * [m,32,20,5,20] cost tensor is never built.  s_eq,t_eq f32[m,32,5,20] (elevation rows 1..ele_n-2 of the
 * equivariant maps) -> ind f32[m] (expected azimuth shift).  wt_host/bias_host: HOST arrays of 10 DEVICE
 * pointers, BN folded; layers 6..9: weights W[K][Cout] with K = ((dn*KH + dk)*KW + dl)*Cin + c, the last layer (20
 * outputs) zero-padded to 32 columns / biases.  Layers 1..5 run in the Winograd F(2x2,3x3) domain as 3x3 correlations over
 * (n, l): buf_winograd_tile_filters(w2d, Cout, Cin2d, ng, 4, out) with ng = buf_cost_winograd_group(layer) N-tiles per group
 * (16*Cout*Cin2d floats each; a layer whose group is 0 takes the [K][Cout] form), w2d = the (3,1,3) filters [Cout,Cin,3(dn),3(dl)]
 * for layers 2..5 and, for layer 1 (which collapses k: 3 -> 1), [64][dk*32 + c][dn][dl] = W1[o][c][dn][dk][dl], Cin2d = 96.  Layer 0 is linear in cost = S(shifted) - T and is passed SEPARATED
 * (exact up to fp32 re-association): wt_host[0] = Ws[480][32] followed by Wt[288][32],
 *   Ws[(dk*5 + e+2)*32 + c][o] = sum over dl-dn=e of W0[o][c][dn][dk][dl],  Wt[(dk*3 + dl)*32 + c][o] = sum over dn of W0
 * (buffer_amd.ops.separate_cost_layer0).  Every [K][Cout] matrix is stored in the MFMA B-operand tiling: blocks
 * [K/16][Cout/16] of 256 floats, block (g, n) laid out [lk 0..3][li 0..15][p 0..3] = W[16g + 4lk + p][16n + li]
 * (buffer_amd.ops.mfma_tile_weights(w, lk_major=True) is the host-side re-layout). */
int     buf_cost_winograd_group(int layer);           /* host only: ng of layers 1..5 (0 for the others) */
int     buf_cost_volume_net(const float* s_eq, const float* t_eq, int m, const float* const* wt_host,
                            const float* const* bias_host, float* ind_out, void* stream);
/* The same with the row gather of models/BUFFER.py:285-292 (ss_equi = src_equi[s_mids], [:, :, 1:ele_n-1]) fused in:
 * equi f32[rows,32,ele_n,20] = the full equivariant maps of all keypoints (ele_n = 7),
 * match i pairs row s_rows[i] with row t_rows[i] (DEVICE int64[m]); elevation rows 1..ele_n-2 are read inside the kernel. */
int     buf_cost_volume_net_gather(const float* equi, int ele_n, const long long* s_rows, const long long* t_rows, int m,
                                   const float* const* wt_host, const float* const* bias_host, float* ind_out, void* stream);

/* A13, split-f16 form -- the same network with fp32-EQUIVALENT arithmetic on the f16 matrix pipe (csrc/costnet_h3.hip; opt-in, see
 * buf_cylindrical_net_split for the operand split): layer 0 separated as above, every layer a direct-form correlation.
 * wt_host: HOST array of 11 DEVICE pointers to u16 planes laid out by buf_split_tile_gemm(w [Cout][Cin][ntaps], Cout, Cin, ntaps, nt, out):
 *   [0] layer 0 S-term [32][32][15 taps = dk*5 + e+2] = Ws, nt 2      [1] layer 0 T-term [32][32][9 taps = dk*3 + dl] = Wt, nt 2
 *   [2] layer 1 [64][96 = dk*32 + c][9 taps = dn*3 + dl], nt 2         [3..8] layers 2..7 [Cout][Cin][9 taps = dn*3 + dl], nt 2 (layer 7: nt 1)
 *   [9] layer 8, nt 1          [10] layer 9 [20][32][4 taps = dn*2 + dl], nt 1 (20 outputs padded to 32)
 * out[((((g ntaps + tap) KS + ks) nt + n2) 2 + plane) 512 + (kg 16 + row) 8 + i] = plane(w[16 nt g + 16 n2 + row][32 ks + 8 kg + i][tap]).
 * bias_host: 10 DEVICE f32 pointers (layer 9: 32 values, 20 used).  status_dev as in buf_cylindrical_net_split. */
long long buf_split_gemm_count(int cout, int cin, int ntaps, int nt);                               /* host only */
int     buf_split_tile_gemm(const float* w_host, int cout, int cin, int ntaps, int nt, unsigned short* out_host);   /* host only */
int     buf_cost_volume_net_split(const float* s_eq, const float* t_eq, int m, const void* const* wt_host,
                                  const float* const* bias_host, float* ind_out, int* status_dev, void* stream);
int     buf_cost_volume_net_split_gather(const float* equi, int ele_n, const long long* s_rows, const long long* t_rows, int m,
                                         const void* const* wt_host, const float* const* bias_host, float* ind_out,
                                         int* status_dev, void* stream);
/* Safe by construction, as buf_cylindrical_net_split_safe (models/patchnet.py:88-147): per-match flags in flags_ws (DEVICE int32[m]),
 * the fp32 kernel of buf_cost_volume_net[_gather] re-runs the flagged matches in the same stream.  s_rows null: dense s_eq / t_eq
 * f32[m,32,5,20]; else s_eq = t_eq = equi f32[rows,32,7,20] with DEVICE int64 row ids.  wt_split_host / bias_split_host as
 * buf_cost_volume_net_split, wt_f32_host / bias_f32_host as buf_cost_volume_net. */
int     buf_cost_volume_net_split_safe(const float* s_eq, const float* t_eq, int ele_n, const long long* s_rows, const long long* t_rows,
                                       int m, const void* const* wt_split_host, const float* const* bias_split_host,
                                       const float* const* wt_f32_host, const float* const* bias_f32_host, float* ind_out,
                                       int* status_dev, int* flags_ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * A14  hypotheses + all-vs-all scoring (models/BUFFER.py:295-311): ind f32[m] -> R f32[m,3,3], t f32[m,3],
 * inlier_num int32[m], best_out int32[1] (first arg-max), best_mask uint8[m]. */
int     buf_hypotheses_score(const float* ind, const float* ss_kpts, const float* tt_kpts, const float* ss_R,
                             const float* tt_R, int m, int azi_n, float inlier_th, float* R_out, float* t_out,
                             int* inlier_num, int* best_out, unsigned char* best_mask, void* stream);
/* A15  deterministic 3-point RANSAC over `corr` (int32[ncorr] indices into src/tgt, both f32[*,3]);
 * replaces open3d registration_ransac_based_on_correspondence (models/BUFFER.py:314-326).
 * T_out f32[4,4]; info_out (nullable) int32[2] = {inliers of the winner, winner id or -1}. */
size_t  buf_ransac_ws_bytes(int nhyp);
int     buf_ransac_kabsch(const float* src, const float* tgt, const int* corr, int ncorr, int nhyp,
                          unsigned long long seed, float max_dist, float edge_similarity, float* T_out,
                          int* info_out, void* ws, size_t ws_bytes, void* stream);
/* The same on the correspondences selected by mask uint8[m] (best_mask of buf_hypotheses_score): index list and count
 * stay on the device, so a pose recovery is enqueued without a host round trip.  ws: buf_ransac_masked_ws_bytes(m, nhyp). */
size_t  buf_ransac_masked_ws_bytes(int m, int nhyp);
int     buf_ransac_kabsch_masked(const float* src, const float* tgt, const unsigned char* mask, int m, int nhyp,
                                 unsigned long long seed, float max_dist, float edge_similarity, float* T_out,
                                 int* info_out, void* ws, size_t ws_bytes, void* stream);
/* A16  post_refinement (models/BUFFER.py:382-418,424-464): <= iters rounds of weighted Kabsch in one launch.
 * T_init,T_out f32[4,4]; src,tgt f32[m,3]; info_out (nullable) int32[2] = {last inlier count, rounds run}. */
int     buf_post_refine(const float* T_init, const float* src, const float* tgt, int m, float inlier_threshold,
                        int iters, float* T_out, int* info_out, void* stream);
/* A14 + A15 + A16 (models/BUFFER.py:295-311 hypotheses and scoring, :314-326 RANSAC, :382-464 post_refinement) for every pair of
 * a step in one set of launches: the matches of nb pairs stacked (pair p owns seg_host[p]
 * consecutive rows of ind f32[M], ss/tt_kpts f32[M,3], ss/tt_R f32[M,9]); seeds_host u64[nb] (RANSAC sampler seed per pair);
 * refine_iters = 0 skips the post-refinement (KITTI) -> poses f32[nb,4,4] (identity for pairs with fewer than 3 matches).
 * Pair by pair bit-identical to buf_hypotheses_score + buf_ransac_kabsch_masked + buf_post_refine. */
size_t  buf_recover_poses_ws_bytes(int m_total, int nb, int nhyp);
int     buf_recover_poses_batched(const float* ind, const float* ss_kpts, const float* tt_kpts, const float* ss_R, const float* tt_R,
                                  const int* seg_host, int nb, const unsigned long long* seeds_host, int azi_n, float inlier_th,
                                  int nhyp, float max_dist, float edge_similarity, float refine_threshold, int refine_iters,
                                  float* poses_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * N1  pre-processing that feeds the path (ThreeDMatch/dataset.py:93,104,125-153; KITTI/dataset.py): the open3d
 * 0.13.0 calls of the reference's datasets, restated from open3d's published algorithms (parity unpinned).
 *
 * buf_voxel_downsample = PointCloud.voxel_down_sample: voxel index floor((p - (min - voxel/2)) / voxel), one row per
 * occupied voxel = fp64 mean of its points (and normals, not re-normalised), rows in ascending voxel-key order.
 * pts / normals: f32[n,3] (is_f64 = 0) or f64[n,3] (is_f64 = 1), normals nullable; out_pts / out_normals f64[>=n,3];
 * *out_m_host = rows written.  max_cells as in buf_grid_subsample_batch.
 *
 * buf_knn_normals = estimate_normals(KDTreeSearchParamKNN(knn)) + orient_normals_towards_camera_location on the
 * candidate lists of a radius search: cand int32[nq,ncand] = buf_grid_query output (sorted by distance, >= ns = empty),
 * knn <= ncand <= 48.  The knn nearest are re-ranked in fp64; covariance by cumulants, normal = eigenvector of the
 * smallest eigenvalue (open3d FastEigen3x3).  qidx (nullable) maps row -> point (for retries on a subset);
 * deficient[row] = 1 when the row holds fewer than min(knn, ns) points (its search ball was too small: retry with a
 * larger radius), else the normal of point qidx[row] is written.  camera_host: HOST double[3]. */
size_t  buf_voxel_downsample_ws_bytes(int n, int64_t max_cells);
int     buf_voxel_downsample(const void* pts, const void* normals, int is_f64, int n, double voxel_size, double* out_pts,
                             double* out_normals, int* out_m_host, int64_t max_cells, void* ws, size_t ws_bytes, void* stream);
/* The same for nb clouds stacked in pts (lengths_host int[nb]) in ONE set of launches and one host round trip: out_pts holds the
 * voxel means of cloud 0, then cloud 1, ... (out_lengths_host int[nb]); every cloud has its own bounding box / voxel grid, so the
 * rows equal those of nb separate calls. */
size_t  buf_voxel_downsample_batch_ws_bytes(int n, int nb, int64_t max_cells);
int     buf_voxel_downsample_batch(const void* pts, int is_f64, int n, const int* lengths_host, int nb, double voxel_size,
                                   double* out_pts, int* out_lengths_host, int64_t max_cells, void* ws, size_t ws_bytes, void* stream);
int     buf_knn_normals(const float* pts, int ns, const int* qidx, int nq, const int* cand, int ncand, int knn,
                        const double* camera_host, int orient, float* normals, unsigned char* deficient, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BUFFER_HIP_H */
