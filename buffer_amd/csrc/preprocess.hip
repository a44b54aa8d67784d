// N1 -- the pre-processing that feeds the path (ThreeDMatch/dataset.py:93,104,125-153; KITTI/dataset.py):
// open3d 0.13.0 `PointCloud.voxel_down_sample`, `estimate_normals()` (30-NN covariance, smallest eigenvector) and
// `orient_normals_towards_camera_location()`.  open3d is a pip dependency of the reference (README.md:28), not part
// of /root/reference: the algorithms restated here are the published ones (open3d/geometry/PointCloud.cpp
// VoxelDownSample, EstimateNormals.cpp ComputeCovariance/FastEigen3x3 = D. Eberly, "A Robust Eigensolver for 3x3
// Symmetric Matrices"), all arithmetic in fp64 like open3d's Eigen::Vector3d clouds.  PARITY UNPINNED.
#include "common.h"
#include <vector>

// ---------------------------------------------------------------------------------- voxel_down_sample
// voxel_min_bound = min - voxel/2; index = floor((p - voxel_min_bound) / voxel) per axis; one output row per
// occupied voxel = mean of its points (and of their normals, not re-normalised) in fp64.  Rows come out in ascending
// voxel-key order (open3d: unordered_map order; the reference shuffles the rows right after, dataset.py:95,112).
// Reuses the bucketed counting sort of subsample.hip: every voxel's run is contiguous and in INPUT order.
struct O3dGrid { double o[3]; double voxel; };

#define O3D_BBOX_BLOCKS 64
// bounding box of every batch element in two steps: O3D_BBOX_BLOCKS partial boxes per element (grid-stride inside its point
// range), then one workgroup per element folds them and derives the element's grid
template <typename T>
__global__ void __launch_bounds__(256) k_o3d_bbox_partial(const T* __restrict__ pts, const int* __restrict__ off, double* __restrict__ part)
{
    const int b = blockIdx.y, lo = off[b], hi = off[b + 1];
    double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
    for (int i = lo + blockIdx.x * 256 + threadIdx.x; i < hi; i += 256 * O3D_BBOX_BLOCKS) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double v = (double)pts[3 * (size_t)i + c];
            mn[c] = v < mn[c] ? v : mn[c];
            mx[c] = v > mx[c] ? v : mx[c];
        }
    }
    __shared__ double smn[3][4], smx[3][4];
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double a = mn[c], z = mx[c];
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            a = fmin(a, __shfl_xor(a, d, WAVE));
            z = fmax(z, __shfl_xor(z, d, WAVE));
        }
        if (lane == 0) { smn[c][w] = a; smx[c][w] = z; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        double* dst = part + 6 * ((size_t)b * O3D_BBOX_BLOCKS + blockIdx.x);
        dst[c] = fmin(fmin(smn[c][0], smn[c][1]), fmin(smn[c][2], smn[c][3]));
        dst[3 + c] = fmax(fmax(smx[c][0], smx[c][1]), fmax(smx[c][2], smx[c][3]));
    }
}

__global__ void __launch_bounds__(O3D_BBOX_BLOCKS) k_o3d_bbox(const double* __restrict__ part, const int* __restrict__ off, double voxel,
                                                            VoxGrid* __restrict__ grid, O3dGrid* __restrict__ og)
{
    const int b = blockIdx.x, lo = off[b], hi = off[b + 1];
    const double* src = part + 6 * ((size_t)b * O3D_BBOX_BLOCKS + threadIdx.x);
    double bb[6];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double a = src[c], z = src[3 + c];
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            a = fmin(a, __shfl_xor(a, d, WAVE));
            z = fmax(z, __shfl_xor(z, d, WAVE));
        }
        bb[c] = a; bb[3 + c] = z;
    }
    if (threadIdx.x == 0) {                                          // O3D_BBOX_BLOCKS == WAVE: one wavefront holds all partials
        VoxGrid g;
        O3dGrid o;
        o.voxel = voxel;
        double N[3];
        for (int c = 0; c < 3; c++) {
            double a = bb[c], z = bb[3 + c];
            if (hi <= lo) { a = 0.0; z = 0.0; }
            o.o[c] = a - voxel * 0.5;                               // voxel_min_bound
            N[c] = floor((z - o.o[c]) / voxel) + 1.0;
            g.o[c] = (float)o.o[c];
        }
        g.dl = (float)voxel;
        g.lo = lo; g.hi = hi;
        g.NX = (unsigned long long)(long long)N[0];
        g.NY = (unsigned long long)(long long)N[1];
        g.cells = fmax(N[0], 1.0) * fmax(N[1], 1.0) * fmax(N[2], 1.0);
        g.nbuckets = 1; g.table_off = 0;
        grid[b] = g;
        og[b] = o;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_o3d_count(const T* __restrict__ pts, int n, const int* __restrict__ off, int nb,
                                                 const VoxGrid* __restrict__ grid, const O3dGrid* __restrict__ og,
                                                 const VoxStatus* __restrict__ st, int* __restrict__ table, int* __restrict__ cell_of,
                                                 unsigned long long* __restrict__ keys)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || st->error) return;
    const int b = nb > 1 ? find_elem(off, nb, i) : 0;
    const VoxGrid g = grid[b];
    const O3dGrid o = og[b];
    unsigned long long ix = (unsigned long long)(long long)floor(((double)pts[3 * (size_t)i] - o.o[0]) / o.voxel);
    unsigned long long iy = (unsigned long long)(long long)floor(((double)pts[3 * (size_t)i + 1] - o.o[1]) / o.voxel);
    unsigned long long iz = (unsigned long long)(long long)floor(((double)pts[3 * (size_t)i + 2] - o.o[2]) / o.voxel);
    unsigned long long key = ix + g.NX * iy + g.NX * g.NY * iz;
    unsigned long long bk = key / st->B;
    if (bk > (unsigned long long)(g.nbuckets - 1)) bk = (unsigned long long)(g.nbuckets - 1);
    int c = (int)(g.table_off + (long long)bk);
    keys[i] = key;
    cell_of[i] = c;
    atomicAdd(&table[c], 1);
}

template <typename T>
__global__ void __launch_bounds__(256) k_o3d_emit(const T* __restrict__ pts, const T* __restrict__ normals, const float4* __restrict__ sorted,
                                                const unsigned long long* __restrict__ key_sorted, const int* __restrict__ cell_sorted,
                                                const int* __restrict__ rowidx, int n, const VoxStatus* __restrict__ st,
                                                double* __restrict__ out, double* __restrict__ out_normals)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || st->error) return;
    if (!vox_is_head(key_sorted, cell_sorted, p)) return;
    const unsigned long long k = key_sorted[p];
    const int c = cell_sorted[p];
    int e = p + 1;
    while (e < n && cell_sorted[e] == c && key_sorted[e] == k) e++;
    double s[3] = { 0.0, 0.0, 0.0 }, sn[3] = { 0.0, 0.0, 0.0 };   // AccumulatedPoint::AddPoint, in input order
    for (int t = p; t < e; t++) {
        const size_t i = (size_t)__float_as_int(sorted[t].w);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            s[d] += (double)pts[3 * i + d];
            if (normals) sn[d] += (double)normals[3 * i + d];
        }
    }
    const double cnt = (double)(e - p);
    const int r = rowidx[p];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        out[3 * (size_t)r + d] = s[d] / cnt;                         // GetAveragePoint
        if (normals) out_normals[3 * (size_t)r + d] = sn[d] / cnt;   // GetAverageNormal (not re-normalised)
    }
}

static VoxWs carve_o3d(WsCarver& w, int n, int nb, int64_t max_cells, O3dGrid** og, double** part)
{
    VoxWs v = carve_vox(w, n, nb, max_cells, 0);
    *og = w.take<O3dGrid>((size_t)nb);
    *part = w.take<double>(6 * (size_t)O3D_BBOX_BLOCKS * nb);
    return v;
}

extern "C" size_t buf_voxel_downsample_batch_ws_bytes(int n, int nb, int64_t max_cells)
{
    WsCarver w(nullptr, 0);
    O3dGrid* og; double* part;
    carve_o3d(w, n, nb > 0 ? nb : 1, max_cells, &og, &part);
    return w.used();
}

extern "C" size_t buf_voxel_downsample_ws_bytes(int n, int64_t max_cells) { return buf_voxel_downsample_batch_ws_bytes(n, 1, max_cells); }

// nb clouds stacked in pts (lengths_host); output rows of cloud b start at the sum of the earlier clouds' counts
template <typename T>
static int voxel_downsample_impl(const T* pts, const T* normals, int n, const int* lengths_host, int nb, double voxel, double* out_pts,
                                 double* out_normals, int* out_lengths_host, int64_t max_cells, void* ws, size_t ws_bytes, hipStream_t s)
{
    WsCarver w(ws, ws_bytes);
    O3dGrid* og; double* part;
    VoxWs v = carve_o3d(w, n, nb, max_cells, &og, &part);
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_voxel_downsample: workspace %zu < %zu bytes", ws_bytes, w.used());
    int rc = upload_offsets(v.off, lengths_host, nb, n, "buf_voxel_downsample", s);
    if (rc) return rc;
    BUF_CHECK_HIP(hipMemsetAsync(v.table, 0, sizeof(int) * (size_t)max_cells, s));
    k_o3d_bbox_partial<T><<<dim3(O3D_BBOX_BLOCKS, nb), 256, 0, s>>>(pts, v.off, part);
    k_o3d_bbox<<<nb, O3D_BBOX_BLOCKS, 0, s>>>(part, v.off, voxel, v.grids, og);
    k_vox_offsets<<<1, 1024, 0, s>>>(v.grids, nb, (long long)max_cells, v.st);
    int blocks = cdiv(n, 256);
    k_o3d_count<T><<<blocks, 256, 0, s>>>(pts, n, v.off, nb, v.grids, og, v.st, v.table, v.cell_of, v.keys);
    rc = exclusive_scan_i32(v.table, (long long)max_cells, v.scan_tmp, nullptr, s);
    if (rc) return rc;
    // only the input index (.w) of the scattered rows is used afterwards: the coordinates are re-read in fp64
    k_cell_scatter<<<blocks, 256, 0, s>>>((const float*)pts, n, v.cell_of, &v.st->error, v.table, v.sorted_tmp);
    k_vox_rank<<<blocks, 256, 0, s>>>(v.cell_of, v.keys, v.table, v.sorted_tmp, n, v.st, v.sorted, v.key_sorted, v.cell_sorted, v.head);   // + head flags
    rc = exclusive_scan_i32(v.head, n, v.scan_tmp, v.total, s);
    if (rc) return rc;
    k_o3d_emit<T><<<blocks, 256, 0, s>>>(pts, normals, v.sorted, v.key_sorted, v.cell_sorted, v.head, n, v.st, out_pts, out_normals);
    k_vox_counts<<<cdiv(nb + 1, 64), 64, 0, s>>>(v.head, v.off, nb, n, 0, v.total, v.st, v.counts);
    BUF_LAUNCH_CHECK();
    std::vector<int> hc((size_t)nb + 2);
    hipError_t e = hipMemcpyAsync(hc.data(), v.counts, sizeof(int) * hc.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    BUF_CHECK_HIP(e);
    BUF_REQUIRE(!hc[nb], BUF_ECAPACITY, "buf_voxel_downsample: bucket table does not fit max_cells=%lld", (long long)max_cells);
    for (int b2 = 0; b2 < nb; b2++) out_lengths_host[b2] = hc[b2];
    return BUF_OK;
}

extern "C" int buf_voxel_downsample_batch(const void* pts, int is_f64, int n, const int* lengths_host, int nb, double voxel_size,
                                          double* out_pts, int* out_lengths_host, int64_t max_cells, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(lengths_host && out_lengths_host && ws, BUF_EINVAL, "buf_voxel_downsample_batch: null argument");
    BUF_REQUIRE(n >= 0 && nb > 0, BUF_EINVAL, "buf_voxel_downsample_batch: n=%d nb=%d", n, nb);
    BUF_REQUIRE(voxel_size > 0.0, BUF_EINVAL, "buf_voxel_downsample_batch: voxel_size=%g must be > 0", voxel_size);
    BUF_REQUIRE(max_cells >= 2LL * nb && max_cells < 0x7fffffffLL, BUF_EINVAL, "buf_voxel_downsample_batch: max_cells=%lld", (long long)max_cells);
    if (n == 0) { for (int b = 0; b < nb; b++) out_lengths_host[b] = 0; return BUF_OK; }
    BUF_REQUIRE(pts && out_pts, BUF_EINVAL, "buf_voxel_downsample_batch: null points");
    if (is_f64)
        return voxel_downsample_impl<double>((const double*)pts, nullptr, n, lengths_host, nb, voxel_size, out_pts, nullptr, out_lengths_host,
                                             max_cells, ws, ws_bytes, (hipStream_t)stream);
    return voxel_downsample_impl<float>((const float*)pts, nullptr, n, lengths_host, nb, voxel_size, out_pts, nullptr, out_lengths_host,
                                        max_cells, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int buf_voxel_downsample(const void* pts, const void* normals, int is_f64, int n, double voxel_size, double* out_pts,
                                    double* out_normals, int* out_m_host, int64_t max_cells, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(out_m_host && ws, BUF_EINVAL, "buf_voxel_downsample: null argument");
    BUF_REQUIRE(n >= 0, BUF_EINVAL, "buf_voxel_downsample: n=%d", n);
    BUF_REQUIRE(voxel_size > 0.0, BUF_EINVAL, "buf_voxel_downsample: voxel_size=%g must be > 0", voxel_size);   // open3d: "voxel_size <= 0"
    BUF_REQUIRE(max_cells >= 2 && max_cells < 0x7fffffffLL, BUF_EINVAL, "buf_voxel_downsample: max_cells=%lld", (long long)max_cells);
    if (n == 0) { *out_m_host = 0; return BUF_OK; }
    BUF_REQUIRE(pts && out_pts && (!normals || out_normals), BUF_EINVAL, "buf_voxel_downsample: null points");
    // k_cell_scatter reads 3 floats per point from the input viewed as float: fine for both element types
    if (is_f64)
        return voxel_downsample_impl<double>((const double*)pts, (const double*)normals, n, &n, 1, voxel_size, out_pts, out_normals, out_m_host,
                                             max_cells, ws, ws_bytes, (hipStream_t)stream);
    return voxel_downsample_impl<float>((const float*)pts, (const float*)normals, n, &n, 1, voxel_size, out_pts, out_normals, out_m_host,
                                        max_cells, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- estimate_normals
// D. Eberly's non-iterative symmetric 3x3 eigen solver as open3d uses it (EstimateNormals.cpp: FastEigen3x3,
// ComputeEigenvector0/1): returns the eigenvector of the SMALLEST eigenvalue.
__host__ __device__ static inline void o3d_cross(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

__host__ __device__ static inline void o3d_eigenvector0(const double A[6], double eval0, double* ev)
{   // A = (a00, a01, a02, a11, a12, a22)
    double r0[3] = { A[0] - eval0, A[1], A[2] }, r1[3] = { A[1], A[3] - eval0, A[4] }, r2[3] = { A[2], A[4], A[5] - eval0 };
    double c01[3], c02[3], c12[3];
    o3d_cross(r0, r1, c01); o3d_cross(r0, r2, c02); o3d_cross(r1, r2, c12);
    double d0 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
    double d1 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
    double d2 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
    double dmax = d0;
    const double* best = c01;
    if (d1 > dmax) { dmax = d1; best = c02; }
    if (d2 > dmax) { dmax = d2; best = c12; }
    double inv = 1.0 / sqrt(dmax);
    ev[0] = best[0] * inv; ev[1] = best[1] * inv; ev[2] = best[2] * inv;
}

__host__ __device__ static inline void o3d_eigenvector1(const double A[6], const double* ev0, double eval1, double* ev)
{
    double U[3], V[3];
    if (fabs(ev0[0]) > fabs(ev0[1])) {
        double inv = 1.0 / sqrt(ev0[0] * ev0[0] + ev0[2] * ev0[2]);
        U[0] = -ev0[2] * inv; U[1] = 0.0; U[2] = ev0[0] * inv;
    } else {
        double inv = 1.0 / sqrt(ev0[1] * ev0[1] + ev0[2] * ev0[2]);
        U[0] = 0.0; U[1] = ev0[2] * inv; U[2] = -ev0[1] * inv;
    }
    o3d_cross(ev0, U, V);
    double AU[3] = { A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2], A[2] * U[0] + A[4] * U[1] + A[5] * U[2] };
    double AV[3] = { A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2], A[2] * V[0] + A[4] * V[1] + A[5] * V[2] };
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - eval1;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - eval1;
    double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        double mx = a00 > a01 ? a00 : a01;
        if (mx > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else            { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            for (int d = 0; d < 3; d++) ev[d] = m01 * U[d] - m00 * V[d];
        } else {
            for (int d = 0; d < 3; d++) ev[d] = U[d];
        }
    } else {
        double mx = a11 > a01 ? a11 : a01;
        if (mx > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
            else            { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
            for (int d = 0; d < 3; d++) ev[d] = m11 * U[d] - m01 * V[d];
        } else {
            for (int d = 0; d < 3; d++) ev[d] = U[d];
        }
    }
}

// cov = (c00, c01, c02, c11, c12, c22) -> normal (zero vector when the matrix is zero)
__host__ __device__ static inline void o3d_fast_eigen3x3(const double cov[6], double* nrm)
{
    double A[6];
    double mc = cov[0];
    for (int i = 1; i < 6; i++) mc = cov[i] > mc ? cov[i] : mc;       // A.maxCoeff()
    if (mc == 0.0) { nrm[0] = nrm[1] = nrm[2] = 0.0; return; }
    for (int i = 0; i < 6; i++) A[i] = cov[i] / mc;
    double norm = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (norm > 0) {
        double q = (A[0] + A[3] + A[5]) / 3.0;
        double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        double c00 = b11 * b22 - A[4] * A[4];
        double c01 = A[1] * b22 - A[4] * A[2];
        double c02 = A[1] * A[4] - b11 * A[2];
        double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        double half_det = det * 0.5;
        half_det = half_det < -1.0 ? -1.0 : (half_det > 1.0 ? 1.0 : half_det);
        double angle = acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        double beta2 = cos(angle) * 2.0, beta0 = cos(angle + two_thirds_pi) * 2.0, beta1 = -(beta0 + beta2);
        double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
        double v0[3], v1[3], v2[3];
        if (half_det >= 0) {
            o3d_eigenvector0(A, e2, v2);
            if (e2 < e0 && e2 < e1) { nrm[0] = v2[0]; nrm[1] = v2[1]; nrm[2] = v2[2]; return; }
            o3d_eigenvector1(A, v2, e1, v1);
            if (e1 < e0 && e1 < e2) { nrm[0] = v1[0]; nrm[1] = v1[1]; nrm[2] = v1[2]; return; }
            o3d_cross(v1, v2, v0);
            nrm[0] = v0[0]; nrm[1] = v0[1]; nrm[2] = v0[2];
        } else {
            o3d_eigenvector0(A, e0, v0);
            if (e0 < e1 && e0 < e2) { nrm[0] = v0[0]; nrm[1] = v0[1]; nrm[2] = v0[2]; return; }
            o3d_eigenvector1(A, v0, e1, v1);
            if (e1 < e0 && e1 < e2) { nrm[0] = v1[0]; nrm[1] = v1[1]; nrm[2] = v1[2]; return; }
            o3d_cross(v0, v1, v2);
            nrm[0] = v2[0]; nrm[1] = v2[1]; nrm[2] = v2[2];
        }
    } else {
        // diagonal matrix: the axis of the smallest diagonal entry, z on ties
        if (cov[0] < cov[3] && cov[0] < cov[5]) { nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; }
        else if (cov[3] < cov[0] && cov[3] < cov[5]) { nrm[0] = 0.0; nrm[1] = 1.0; nrm[2] = 0.0; }
        else { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }
    }
}

#define NRM_MAXC 48      // candidates per point handed over by the radius search

// One lane per query row.  cand[row][0..ncand) = support indices sorted by fp32 distance (>= ns: empty slot), as
// produced by buf_grid_query.  The knn nearest are re-selected in fp64 by (d2, index) rank among the candidates
// (exact whenever the row holds every point of its search ball or >= knn candidates nearer than the ball's rim);
// rows that cannot be decided are flagged in `deficient` for a retry with a larger radius.
__global__ void __launch_bounds__(64) k_knn_normals(const float* __restrict__ pts, int ns, const int* __restrict__ qidx, int nq,
                                                  const int* __restrict__ cand, int ncand, int knn, double cam_x,
                                                  double cam_y, double cam_z, int orient, float* __restrict__ normals,
                                                  unsigned char* __restrict__ deficient)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= nq) return;
    const int qi = qidx ? qidx[row] : row;
    const double qx = pts[3 * (size_t)qi], qy = pts[3 * (size_t)qi + 1], qz = pts[3 * (size_t)qi + 2];
    // candidate distances / ids live in LDS columns (one per lane): 48 fp64 + 48 int per lane would not fit registers
    __shared__ double d2s[NRM_MAXC][64];
    __shared__ int ids[NRM_MAXC][64];
    const int ln = threadIdx.x;
    int nvalid = 0;
    for (int j = 0; j < ncand; j++) {
        const int c = cand[(size_t)row * ncand + j];
        const bool ok = c >= 0 && c < ns;
        double dx = 0.0, dy = 0.0, dz = 0.0;
        if (ok) { dx = (double)pts[3 * (size_t)c] - qx; dy = (double)pts[3 * (size_t)c + 1] - qy; dz = (double)pts[3 * (size_t)c + 2] - qz; }
        ids[j][ln] = ok ? c : 0x7fffffff;
        d2s[j][ln] = ok ? dx * dx + dy * dy + dz * dz : 1e300;
        nvalid += ok ? 1 : 0;
    }
    // a short list holds the whole search ball and must reach min(knn, ns) points; a full one holds the ncand nearest
    const int need = knn < ns ? knn : ns;
    if (nvalid < need) { deficient[row] = 1; return; }
    deficient[row] = 0;
    double cum[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int taken = 0;
    for (int j = 0; j < ncand; j++) {
        const double dj = d2s[j][ln];
        const int ij = ids[j][ln];
        if (ij == 0x7fffffff) continue;
        int rank = 0;
        for (int t = 0; t < ncand; t++) {
            const double dt = d2s[t][ln];
            rank += (dt < dj || (dt == dj && ids[t][ln] < ij)) ? 1 : 0;
        }
        if (rank < need) {
            const size_t c = (size_t)ij;
            const double x = pts[3 * c], y = pts[3 * c + 1], z = pts[3 * c + 2];
            cum[0] += x; cum[1] += y; cum[2] += z;
            cum[3] += x * x; cum[4] += x * y; cum[5] += x * z; cum[6] += y * y; cum[7] += y * z; cum[8] += z * z;
            taken++;
        }
    }
    double nrm[3];
    if (taken >= 3) {                                                  // ComputeCovariance (cumulants)
        for (int i = 0; i < 9; i++) cum[i] /= (double)taken;
        double cov[6] = { cum[3] - cum[0] * cum[0], cum[4] - cum[0] * cum[1], cum[5] - cum[0] * cum[2],
                          cum[6] - cum[1] * cum[1], cum[7] - cum[1] * cum[2], cum[8] - cum[2] * cum[2] };
        o3d_fast_eigen3x3(cov, nrm);
    } else {                                                           // covariance = Identity
        nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0;
    }
    if (nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2] == 0.0) { nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0; }   // EstimateNormals
    if (orient) {                                                      // OrientNormalsTowardsCameraLocation
        const double rx = cam_x - qx, ry = cam_y - qy, rz = cam_z - qz;
        if (nrm[0] * rx + nrm[1] * ry + nrm[2] * rz < 0.0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    }
    normals[3 * (size_t)qi] = (float)nrm[0]; normals[3 * (size_t)qi + 1] = (float)nrm[1]; normals[3 * (size_t)qi + 2] = (float)nrm[2];
}

extern "C" int buf_knn_normals(const float* pts, int ns, const int* qidx, int nq, const int* cand, int ncand, int knn,
                               const double* camera_host, int orient, float* normals, unsigned char* deficient, void* stream)
{
    BUF_REQUIRE(ns >= 0 && nq >= 0 && knn >= 1, BUF_EINVAL, "buf_knn_normals: ns=%d nq=%d knn=%d", ns, nq, knn);
    BUF_REQUIRE(ncand >= 1 && ncand <= NRM_MAXC && knn <= ncand, BUF_EINVAL, "buf_knn_normals: ncand=%d (1..%d, >= knn=%d)", ncand, NRM_MAXC, knn);
    if (nq == 0) return BUF_OK;
    BUF_REQUIRE(pts && cand && normals && deficient && (!orient || camera_host), BUF_EINVAL, "buf_knn_normals: null argument");
    double cx = orient ? camera_host[0] : 0.0, cy = orient ? camera_host[1] : 0.0, cz = orient ? camera_host[2] : 0.0;
    k_knn_normals<<<cdiv(nq, 64), 64, 0, (hipStream_t)stream>>>(pts, ns, qidx, nq, cand, ncand, knn, cx, cy, cz, orient, normals, deficient);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
