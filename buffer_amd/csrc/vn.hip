// A4/A5 -- Vector-Neuron blocks of the point-wise learner (models/point_learner.py, models/vn_layers.py).
//
// Feature rows are f32[N, 3C], channel-major / xyz-minor (point_learner.py:196,261,407).
// One lane per (point, output channel): the lanes of one point read the same gathered rows (one
// transaction), weights sit in LDS.  Nothing of size [N,K,C] is ever materialised: the reference
// moves ~0.5 GB per block through [1,C,3,N,K] temporaries, here the traffic is the index table,
// the gathered rows and the [N,3Cout] result.
#include "common.h"

#define VN_EPS 1e-6f     // models/vn_layers.py:10

struct VnParams {
    const float* wf;        // [Cout, Cin']  map_to_feat
    const float* wd;        // [Cout, Cin']  map_to_dir   (null: linear only)
    const float* bn_scale;  // [Cout] w / sqrt(var + 1e-5) (null: no VN batch-norm, i.e. Cout == 1)
    const float* bn_shift;  // [Cout] b - mean * scale
    float slope;            // negative_slope
};

// VNBatchNorm (vn_layers.py:108-130) + VN leaky ReLU (:69-75) on one output channel
__device__ __forceinline__ void vn_epilogue(float& px, float& py, float& pz, float dx, float dy, float dz,
                                            bool has_bn, float bsc, float bsh, float slope)
{
    if (has_bn) {
        float norm = sqrtf(px * px + py * py + pz * pz) + VN_EPS;
        float nbn = norm * bsc + bsh;
        px = px / norm * nbn; py = py / norm * nbn; pz = pz / norm * nbn;
    }
    float dot = px * dx + py * dy + pz * dz;
    if (!(dot >= 0.f)) {
        float dsq = dx * dx + dy * dy + dz * dz;
        float f = dot / (dsq + VN_EPS);
        float rx = px - f * dx, ry = py - f * dy, rz = pz - f * dz;
        px = slope * px + (1.f - slope) * rx;
        py = slope * py + (1.f - slope) * ry;
        pz = slope * pz + (1.f - slope) * rz;
    } else {
        px = slope * px + (1.f - slope) * px;
        py = slope * py + (1.f - slope) * py;
        pz = slope * pz + (1.f - slope) * pz;
    }
}

// VNNBlock / the conv half of VNNResnetBlock (point_learner.py:315-416, 467-552):
// gather neighbours -> [f, delta] (mode '1') or [f, delta, f x delta, mean_K(delta)] (mode '6', Cin == 1)
// -> VN-linear -> VN-BN -> VN-leaky -> mean over ALL K slots (shadows included).
__global__ void __launch_bounds__(256) k_vn_gather(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                 const float* __restrict__ feats, const int* __restrict__ idx,
                                                 int nq, int ns, int K, int cin, int cout, int mode6, float scale,
                                                 VnParams P, float* __restrict__ out)
{
    extern __shared__ float lds[];
    const int cinp = cin + (mode6 ? 3 : 1);
    float* wf = lds;
    float* wd = lds + cout * cinp;
    for (int t = threadIdx.x; t < cout * cinp; t += 256) { wf[t] = P.wf[t]; wd[t] = P.wd[t]; }
    __syncthreads();
    // blocks of one XCD take neighbouring points: the rows they gather (shared between neighbouring queries) then live in ONE L2
    long long t = (long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= (long long)nq * cout) return;
    int i = (int)(t / cout), o = (int)(t % cout);
    const float* wfo = wf + o * cinp;
    const float* wdo = wd + o * cinp;
    bool has_bn = P.bn_scale != nullptr;
    float bsc = has_bn ? P.bn_scale[o] : 0.f, bsh = has_bn ? P.bn_shift[o] : 0.f;
    float qx = q_pts[3 * (size_t)i], qy = q_pts[3 * (size_t)i + 1], qz = q_pts[3 * (size_t)i + 2];
    const int* row = idx + (size_t)i * K;
    // Dot products over the input channels and the sums over the K slots run in fp64 and are rounded once (round 3): an fp32
    // sum is one summation order among many, and on the KITTI branch (80 m coordinates, features near zero behind VN-BN) this
    // kernel's order put the HIP path at twice the reference's own distance from the float64 network (eps 9.0e-4 vs 4.3e-4 of
    // scale); with fp64 sums it sits at the reference's level (6e-4: what is left is the fp32 rounding of the stored
    // features, which every fp32 run has).
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (mode6) {                                            // mean over K of delta (point_learner.py:392)
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int k = 0; k < K; k++) {
            int j = row[k];
            if (j < ns) {
                sx += (double)((s_pts[3 * (size_t)j] - qx) / scale);
                sy += (double)((s_pts[3 * (size_t)j + 1] - qy) / scale);
                sz += (double)((s_pts[3 * (size_t)j + 2] - qz) / scale);
            }
        }
        mx = (float)(sx / (double)K); my = (float)(sy / (double)K); mz = (float)(sz / (double)K);
    }
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int k = 0; k < K; k++) {
        int j = row[k];
        bool real = j < ns;                                  // shadow: delta = 0, features = 0 (:329-349)
        float ex = 0.f, ey = 0.f, ez = 0.f;
        if (real) {
            ex = (s_pts[3 * (size_t)j] - qx) / scale;
            ey = (s_pts[3 * (size_t)j + 1] - qy) / scale;
            ez = (s_pts[3 * (size_t)j + 2] - qz) / scale;
        }
        float px = 0.f, py = 0.f, pz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
        double Px = 0, Py = 0, Pz = 0, Dx = 0, Dy = 0, Dz = 0;
        if (mode6) {
            float fx = 0.f, fy = 0.f, fz = 0.f;
            if (real) { fx = feats[3 * (size_t)j]; fy = feats[3 * (size_t)j + 1]; fz = feats[3 * (size_t)j + 2]; }
            float cx = fy * ez - fz * ey, cy = fz * ex - fx * ez, cz = fx * ey - fy * ex;
            px = wfo[0] * fx + wfo[1] * ex + wfo[2] * cx + wfo[3] * mx;
            py = wfo[0] * fy + wfo[1] * ey + wfo[2] * cy + wfo[3] * my;
            pz = wfo[0] * fz + wfo[1] * ez + wfo[2] * cz + wfo[3] * mz;
            dx = wdo[0] * fx + wdo[1] * ex + wdo[2] * cx + wdo[3] * mx;
            dy = wdo[0] * fy + wdo[1] * ey + wdo[2] * cy + wdo[3] * my;
            dz = wdo[0] * fz + wdo[1] * ez + wdo[2] * cz + wdo[3] * mz;
        } else {
            if (real) {
                const float* f = feats + (size_t)j * 3 * cin;
                for (int c = 0; c < cin; c++) {
                    double a = wfo[c], b = wdo[c];
                    double fx = f[3 * c], fy = f[3 * c + 1], fz = f[3 * c + 2];
                    Px += a * fx; Py += a * fy; Pz += a * fz; Dx += b * fx; Dy += b * fy; Dz += b * fz;
                }
                { double a = wfo[cin], b = wdo[cin];
                  Px += a * ex; Py += a * ey; Pz += a * ez; Dx += b * ex; Dy += b * ey; Dz += b * ez; }
                px = (float)Px; py = (float)Py; pz = (float)Pz; dx = (float)Dx; dy = (float)Dy; dz = (float)Dz;
            }
        }
        vn_epilogue(px, py, pz, dx, dy, dz, has_bn, bsc, bsh, P.slope);
        ax += (double)px; ay += (double)py; az += (double)pz;
    }
    float* dst = out + (size_t)i * 3 * cout + 3 * o;
    dst[0] = (float)(ax / (double)K); dst[1] = (float)(ay / (double)K); dst[2] = (float)(az / (double)K);      // mean_pool, vn_layers.py:165-166
}

// ---- mode '6' (block 0: one input vector channel) with the gathered rows staged in LDS (round 4) -------------------------------
// k_vn_gather walks a neighbour row as a chain of dependent loads (index -> support point + feature row, ~1000 cycles per slot and
// K slots per lane, every output lane of a point repeating the chain).  Here a workgroup owns VG6_PTS points: their index rows
// are read once (coalesced), every (point, slot)'s support coordinates and feature vector -- 24 bytes -- are fetched with all loads
// in flight at once into LDS as delta = (s - q) / scale and f, the mean of the deltas is formed once per point, and the lanes
// (point, output channel) then run the slots from LDS.  Same arithmetic, same order of the fp64 sums: bit-identical to k_vn_gather.
#define VG6_PTS 32
__global__ void __launch_bounds__(256) k_vn_gather6_lds(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                      const float* __restrict__ feats, const int* __restrict__ idx,
                                                      int nq, int ns, int K, int cout, float scale, VnParams P, float* __restrict__ out)
{
    extern __shared__ float lds[];
    float* wf = lds;                                   // [cout][4]
    float* wd = wf + cout * 4;
    float* ef = wd + cout * 4;                         // [VG6_PTS][K][8]: ex, ey, ez, real, fx, fy, fz, -
    float* mean = ef + (size_t)VG6_PTS * K * 8;        // [VG6_PTS][4]
    for (int t = threadIdx.x; t < cout * 4; t += 256) { wf[t] = P.wf[t]; wd[t] = P.wd[t]; }
    const int p0 = xcd_contiguous_block(blockIdx.x, gridDim.x) * VG6_PTS;
    const int np = min(VG6_PTS, nq - p0);
    for (int t = threadIdx.x; t < np * K; t += 256) {
        const int pl = t / K, i = p0 + pl;
        const int j = idx[(size_t)p0 * K + t];
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < ns) {
            const float qx = q_pts[3 * (size_t)i], qy = q_pts[3 * (size_t)i + 1], qz = q_pts[3 * (size_t)i + 2];
            a = make_float4((s_pts[3 * (size_t)j] - qx) / scale, (s_pts[3 * (size_t)j + 1] - qy) / scale, (s_pts[3 * (size_t)j + 2] - qz) / scale, 1.f);
            b = make_float4(feats[3 * (size_t)j], feats[3 * (size_t)j + 1], feats[3 * (size_t)j + 2], 0.f);
        }
        reinterpret_cast<float4*>(ef)[2 * t] = a;
        reinterpret_cast<float4*>(ef)[2 * t + 1] = b;
    }
    __syncthreads();
    if ((int)threadIdx.x < np) {                       // mean over K of delta (point_learner.py:392), sequential fp64 sum as k_vn_gather
        double sx = 0.0, sy = 0.0, sz = 0.0;
        const float4* e4 = reinterpret_cast<const float4*>(ef) + (size_t)threadIdx.x * K * 2;
        for (int k = 0; k < K; k++) {
            const float4 a = e4[2 * k];
            if (a.w != 0.f) { sx += (double)a.x; sy += (double)a.y; sz += (double)a.z; }
        }
        mean[4 * threadIdx.x] = (float)(sx / (double)K); mean[4 * threadIdx.x + 1] = (float)(sy / (double)K); mean[4 * threadIdx.x + 2] = (float)(sz / (double)K);
    }
    __syncthreads();
    const bool has_bn = P.bn_scale != nullptr;
    for (int t = threadIdx.x; t < np * cout; t += 256) {
        const int pl = t / cout, o = t - pl * cout;
        const float* wfo = wf + o * 4;
        const float* wdo = wd + o * 4;
        const float bsc = has_bn ? P.bn_scale[o] : 0.f, bsh = has_bn ? P.bn_shift[o] : 0.f;
        const float mx = mean[4 * pl], my = mean[4 * pl + 1], mz = mean[4 * pl + 2];
        const float4* e4 = reinterpret_cast<const float4*>(ef) + (size_t)pl * K * 2;
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (int k = 0; k < K; k++) {
            const float4 a = e4[2 * k], b = e4[2 * k + 1];
            const float ex = a.x, ey = a.y, ez = a.z, fx = b.x, fy = b.y, fz = b.z;
            const float cx = fy * ez - fz * ey, cy = fz * ex - fx * ez, cz = fx * ey - fy * ex;
            float px = wfo[0] * fx + wfo[1] * ex + wfo[2] * cx + wfo[3] * mx;
            float py = wfo[0] * fy + wfo[1] * ey + wfo[2] * cy + wfo[3] * my;
            float pz = wfo[0] * fz + wfo[1] * ez + wfo[2] * cz + wfo[3] * mz;
            float dx = wdo[0] * fx + wdo[1] * ex + wdo[2] * cx + wdo[3] * mx;
            float dy = wdo[0] * fy + wdo[1] * ey + wdo[2] * cy + wdo[3] * my;
            float dz = wdo[0] * fz + wdo[1] * ez + wdo[2] * cz + wdo[3] * mz;
            vn_epilogue(px, py, pz, dx, dy, dz, has_bn, bsc, bsh, P.slope);
            ax += (double)px; ay += (double)py; az += (double)pz;
        }
        float* dst = out + (size_t)(p0 + pl) * 3 * cout + 3 * o;
        dst[0] = (float)(ax / (double)K); dst[1] = (float)(ay / (double)K); dst[2] = (float)(az / (double)K);
    }
}

// ---- mode '1' with the channel contraction hoisted out of the neighbour loop (round 4) ----------------------------------------
// VN-linear is linear and its input is [f_j (Cin channels), delta_ij]: the feature part of both maps depends on the SUPPORT point
// only, PF[j] = [Wf[:, :Cin] f_j | Wd[:, :Cin] f_j] (k_vn_linear_pre: N_s x 2 Cout dot products over Cin, fp64, rounded once),
// and a neighbour slot costs 6 multiply-adds for the delta column instead of 6 (Cin + 1): the K-fold repetition of the
// contraction -- 12 Cin of the ~12 Cin + 40 operations per slot -- is gone (Cin = 10 .. 40 in the four resnet blocks).
// The sum [PF + w_delta * delta] is formed in fp64 from the fp32 PF: one rounding more than the all-fp64 dot product of
// k_vn_gather, of half an ulp of the partial sum (tests/test_model_gpu.py bounds the result against the float64 network as before).
__global__ void __launch_bounds__(256) k_vn_linear_pre(const float* __restrict__ feats, int ns, int cin, int cout, VnParams P,
                                                     float* __restrict__ pf)
{
    extern __shared__ float lds[];
    const int cinp = cin + 1;
    float* wf = lds;
    float* wd = lds + cout * cinp;
    for (int t = threadIdx.x; t < cout * cinp; t += 256) { wf[t] = P.wf[t]; wd[t] = P.wd[t]; }
    __syncthreads();
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)ns * 2 * cout) return;
    const int j = (int)(t / (2 * cout)), o2 = (int)(t % (2 * cout));
    const float* w = (o2 < cout ? wf + o2 * cinp : wd + (o2 - cout) * cinp);
    const float* f = feats + (size_t)j * 3 * cin;
    double x = 0.0, y = 0.0, z = 0.0;
    for (int c = 0; c < cin; c++) {
        const double a = w[c];
        x += a * (double)f[3 * c]; y += a * (double)f[3 * c + 1]; z += a * (double)f[3 * c + 2];
    }
    float* d = pf + (size_t)t * 3;
    d[0] = (float)x; d[1] = (float)y; d[2] = (float)z;
}

// A workgroup owns VG6_PTS points: their index rows and deltas are staged once in LDS (16 bytes per (point, slot): ex, ey, ez,
// support index), so the slot loop of a lane (point, output channel) has no dependent index -> row chain left: the PF reads of
// several slots are in flight together (unrolled by four).
__global__ void __launch_bounds__(256) k_vn_gather_pre(const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                     const float* __restrict__ pf, const int* __restrict__ idx,
                                                     int nq, int ns, int K, int cin, int cout, float scale,
                                                     VnParams P, float* __restrict__ out)
{
    extern __shared__ float lds[];
    float4* ej = reinterpret_cast<float4*>(lds);        // [VG6_PTS][K]: ex, ey, ez, bitcast support index (-1: shadow)
    const int p0 = xcd_contiguous_block(blockIdx.x, gridDim.x) * VG6_PTS;
    const int np = min(VG6_PTS, nq - p0);
    for (int t = threadIdx.x; t < np * K; t += 256) {
        const int pl = t / K, i = p0 + pl;
        const int j = idx[(size_t)p0 * K + t];
        float4 a = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (j < ns) {
            const float qx = q_pts[3 * (size_t)i], qy = q_pts[3 * (size_t)i + 1], qz = q_pts[3 * (size_t)i + 2];
            a = make_float4((s_pts[3 * (size_t)j] - qx) / scale, (s_pts[3 * (size_t)j + 1] - qy) / scale, (s_pts[3 * (size_t)j + 2] - qz) / scale,
                            __int_as_float(j));
        }
        ej[t] = a;
    }
    __syncthreads();
    const int cinp = cin + 1;
    const bool has_bn = P.bn_scale != nullptr;
    for (int t = threadIdx.x; t < np * cout; t += 256) {
        const int pl = t / cout, o = t - pl * cout;
        const double wfe = P.wf[o * cinp + cin], wde = P.wd[o * cinp + cin];
        const float bsc = has_bn ? P.bn_scale[o] : 0.f, bsh = has_bn ? P.bn_shift[o] : 0.f;
        const float4* e4 = ej + (size_t)pl * K;
        double ax = 0.0, ay = 0.0, az = 0.0;
#pragma unroll 4
        for (int k = 0; k < K; k++) {
            const float4 e = e4[k];
            const int j = __float_as_int(e.w);
            float px = 0.f, py = 0.f, pz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
            if (j >= 0) {                                    // shadow: delta = 0, features = 0 (:329-349)
                const float* a = pf + ((size_t)j * 2 * cout + o) * 3;
                const float* b = a + (size_t)cout * 3;
                px = (float)((double)a[0] + wfe * (double)e.x); py = (float)((double)a[1] + wfe * (double)e.y); pz = (float)((double)a[2] + wfe * (double)e.z);
                dx = (float)((double)b[0] + wde * (double)e.x); dy = (float)((double)b[1] + wde * (double)e.y); dz = (float)((double)b[2] + wde * (double)e.z);
            }
            vn_epilogue(px, py, pz, dx, dy, dz, has_bn, bsc, bsh, P.slope);
            ax += (double)px; ay += (double)py; az += (double)pz;
        }
        float* dst = out + (size_t)(p0 + pl) * 3 * cout + 3 * o;
        dst[0] = (float)(ax / (double)K); dst[1] = (float)(ay / (double)K); dst[2] = (float)(az / (double)K);      // mean_pool, vn_layers.py:165-166
    }
}

// Point-wise VN layer (VNLinearLeakyReLU with dim=4; VNBlock, unary, shortcut, fc_layer, VNStdFeature):
//   in_i = concat( A[ia(i)] (ca channels; row through ind_a[i*ind_stride], >= na -> zeros) , B[i] (cb channels) )
//   out_i = VN(in_i) (+ residual_i)
// ind_a implements closest_pool (models/KPConv/blocks.py:88-101), B the skip concat (point_learner.py:189-191),
// residual the resnet sum (:577).  wd == null: plain VNLinear (VNStdFeature.vn_lin).
__global__ void __launch_bounds__(256) k_vn_pointwise(const float* __restrict__ A, const int* __restrict__ ind_a, int ind_stride,
                                                    int na, int ca, const float* __restrict__ B, int cb, int n, int cout,
                                                    VnParams P, const float* __restrict__ residual, float* __restrict__ out)
{
    extern __shared__ float lds[];
    const int cin = ca + cb;
    float* wf = lds;
    float* wd = lds + cout * cin;
    bool has_dir = P.wd != nullptr;
    for (int t = threadIdx.x; t < cout * cin; t += 256) { wf[t] = P.wf[t]; if (has_dir) wd[t] = P.wd[t]; }
    __syncthreads();
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * cout) return;
    int i = (int)(t / cout), o = (int)(t % cout);
    const float* wfo = wf + o * cin;
    const float* wdo = wd + o * cin;
    double Px = 0, Py = 0, Pz = 0, Dx = 0, Dy = 0, Dz = 0;
    int ia = ind_a ? ind_a[(size_t)i * ind_stride] : i;
    if (ca > 0 && ia < na) {
        const float* f = A + (size_t)ia * 3 * ca;
        for (int c = 0; c < ca; c++) {
            double a = wfo[c], b = has_dir ? wdo[c] : 0.f;
            double fx = f[3 * c], fy = f[3 * c + 1], fz = f[3 * c + 2];
            Px += a * fx; Py += a * fy; Pz += a * fz; Dx += b * fx; Dy += b * fy; Dz += b * fz;
        }
    }
    if (cb > 0) {
        const float* f = B + (size_t)i * 3 * cb;
        for (int c = 0; c < cb; c++) {
            double a = wfo[ca + c], b = has_dir ? wdo[ca + c] : 0.f;
            double fx = f[3 * c], fy = f[3 * c + 1], fz = f[3 * c + 2];
            Px += a * fx; Py += a * fy; Pz += a * fz; Dx += b * fx; Dy += b * fy; Dz += b * fz;
        }
    }
    float px = (float)Px, py = (float)Py, pz = (float)Pz, dx = (float)Dx, dy = (float)Dy, dz = (float)Dz;
    if (has_dir) {
        bool has_bn = P.bn_scale != nullptr;
        vn_epilogue(px, py, pz, dx, dy, dz, has_bn, has_bn ? P.bn_scale[o] : 0.f, has_bn ? P.bn_shift[o] : 0.f, P.slope);
    }
    size_t off = (size_t)i * 3 * cout + 3 * o;
    if (residual) { px += residual[off]; py += residual[off + 1]; pz += residual[off + 2]; }
    out[off] = px; out[off + 1] = py; out[off + 2] = pz;
}

// max_pool (models/KPConv/blocks.py:104-121): element-wise max over the gathered rows, zero shadow row.
__global__ void __launch_bounds__(256) k_gather_max(const float* __restrict__ feats, const int* __restrict__ idx, int nq, int ns,
                                                  int K, int width, float* __restrict__ out)
{
    long long t = (long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= (long long)nq * width) return;
    int i = (int)(t / width), f = (int)(t % width);
    const int* row = idx + (size_t)i * K;
    float m = -3.4e38f;
    for (int k = 0; k < K; k++) {
        int j = row[k];
        float v = j < ns ? feats[(size_t)j * width + f] : 0.f;
        m = fmaxf(m, v);
    }
    out[t] = m;
}

// VNStdFeature tail (vn_layers.py:213-219): x_std[i, c*3+k] = sum_j x[i,c,j] * z[i,k,j]
__global__ void __launch_bounds__(256) k_vn_std(const float* __restrict__ x, const float* __restrict__ z, int n, int c,
                                              float* __restrict__ out)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * c * 3) return;
    int i = (int)(t / (c * 3)), r = (int)(t % (c * 3));
    int ch = r / 3, k = r % 3;
    const float* xv = x + (size_t)i * 3 * c + 3 * ch;
    const float* zv = z + (size_t)i * 9 + 3 * k;
    out[t] = xv[0] * zv[0] + xv[1] * zv[1] + xv[2] * zv[2];
}

// ------------------------------------------------------------------------------------------
static VnParams make_params(const float* wf, const float* wd, const float* bn_scale, const float* bn_shift, float slope)
{
    VnParams p;
    p.wf = wf; p.wd = wd; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.slope = slope;
    return p;
}

extern "C" int buf_vn_gather_block(const float* q_pts, const float* s_pts, const float* feats, const int* idx,
                                   int nq, int ns, int k, int cin, int cout, int mode, float scale,
                                   const float* wf, const float* wd, const float* bn_scale, const float* bn_shift,
                                   float slope, float* out, void* stream)
{
    BUF_REQUIRE(nq >= 0 && ns >= 0 && k > 0 && cin > 0 && cout > 0, BUF_EINVAL, "buf_vn_gather_block: bad sizes");
    BUF_REQUIRE(mode == 1 || mode == 6, BUF_EINVAL, "buf_vn_gather_block: mode %d (only '1' and '6' are reachable)", mode);
    BUF_REQUIRE(mode != 6 || cin == 1, BUF_EINVAL, "buf_vn_gather_block: mode '6' needs one input vector channel");
    BUF_REQUIRE(scale != 0.f, BUF_EINVAL, "buf_vn_gather_block: scale == 0");
    if (nq == 0) return BUF_OK;
    BUF_REQUIRE(q_pts && s_pts && feats && idx && wf && wd && out, BUF_EINVAL, "buf_vn_gather_block: null argument");
    int cinp = cin + (mode == 6 ? 3 : 1);
    size_t lds = sizeof(float) * 2 * (size_t)cout * cinp;
    long long total = (long long)nq * cout;
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 4.0 * nq * k + 12.0 * nq + 12.0 * nq * cin + 12.0 * nq * cout, BUF_TIMED_VN_GATHER);
    static const bool direct6 = getenv("BUF_VN_GATHER_DIRECT") != nullptr;        // development switch: the round-1..3 kernel for mode '6' too
    const size_t lds6 = sizeof(float) * (8 * (size_t)cout + (size_t)VG6_PTS * k * 8 + VG6_PTS * 4);
    if (mode == 6 && !direct6 && lds6 <= 48 * 1024)
        k_vn_gather6_lds<<<cdiv(nq, VG6_PTS), 256, lds6, (hipStream_t)stream>>>(q_pts, s_pts, feats, idx, nq, ns, k, cout, scale,
                                                                              make_params(wf, wd, bn_scale, bn_shift, slope), out);
    else
        k_vn_gather<<<cdiv(total, 256), 256, lds, (hipStream_t)stream>>>(q_pts, s_pts, feats, idx, nq, ns, k, cin, cout,
                                                                       mode == 6 ? 1 : 0, scale,
                                                                       make_params(wf, wd, bn_scale, bn_shift, slope), out);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// mode '1' through the hoisted contraction; ws: f32[ns * 6 * cout] (buf_vn_gather_pre_ws_bytes)
extern "C" size_t buf_vn_gather_pre_ws_bytes(int ns, int cout) { return sizeof(float) * 6 * (size_t)(ns > 0 ? ns : 1) * (size_t)(cout > 0 ? cout : 1); }

extern "C" int buf_vn_gather_block_pre(const float* q_pts, const float* s_pts, const float* feats, const int* idx,
                                       int nq, int ns, int k, int cin, int cout, float scale,
                                       const float* wf, const float* wd, const float* bn_scale, const float* bn_shift,
                                       float slope, float* out, void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(nq >= 0 && ns >= 0 && k > 0 && cin > 0 && cout > 0, BUF_EINVAL, "buf_vn_gather_block_pre: bad sizes");
    BUF_REQUIRE(scale != 0.f, BUF_EINVAL, "buf_vn_gather_block_pre: scale == 0");
    if (nq == 0) return BUF_OK;
    BUF_REQUIRE(q_pts && s_pts && feats && idx && wf && wd && out, BUF_EINVAL, "buf_vn_gather_block_pre: null argument");
    BUF_REQUIRE(ws && ws_bytes >= buf_vn_gather_pre_ws_bytes(ns, cout), BUF_EWORKSPACE, "buf_vn_gather_block_pre: workspace of %zu bytes, need %zu",
                ws_bytes, buf_vn_gather_pre_ws_bytes(ns, cout));
    const size_t lds = sizeof(float) * 2 * (size_t)cout * (cin + 1);
    BUF_REQUIRE(lds <= 48 * 1024, BUF_EINVAL, "buf_vn_gather_block_pre: weights of %zu bytes exceed the 48 KiB of LDS a launch gets without opting in", lds);
    const VnParams P = make_params(wf, wd, bn_scale, bn_shift, slope);
    TimedSpan span;
    bool timed = timing_begin((hipStream_t)stream, &span, 4.0 * nq * k + 12.0 * nq + 12.0 * nq * cin + 12.0 * nq * cout, BUF_TIMED_VN_GATHER);
    if (ns > 0)
        k_vn_linear_pre<<<cdiv((long long)ns * 2 * cout, 256), 256, lds, (hipStream_t)stream>>>(feats, ns, cin, cout, P, (float*)ws);
    const size_t lds_g = sizeof(float4) * (size_t)VG6_PTS * k;
    BUF_REQUIRE(lds_g <= 48 * 1024, BUF_EINVAL, "buf_vn_gather_block_pre: k=%d too large for the LDS stage", k);
    k_vn_gather_pre<<<cdiv(nq, VG6_PTS), 256, lds_g, (hipStream_t)stream>>>(q_pts, s_pts, (const float*)ws, idx, nq, ns, k, cin, cout, scale, P, out);
    if (timed) timing_end((hipStream_t)stream, &span);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_vn_pointwise(const float* a, const int* ind_a, int ind_stride, int na, int ca, const float* b, int cb,
                                int n, int cout, const float* wf, const float* wd, const float* bn_scale,
                                const float* bn_shift, float slope, const float* residual, float* out, void* stream)
{
    BUF_REQUIRE(n >= 0 && ca >= 0 && cb >= 0 && ca + cb > 0 && cout > 0, BUF_EINVAL, "buf_vn_pointwise: bad sizes");
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(wf && out && (ca == 0 || a) && (cb == 0 || b), BUF_EINVAL, "buf_vn_pointwise: null argument");
    size_t lds = sizeof(float) * 2 * (size_t)cout * (ca + cb);
    BUF_REQUIRE(lds <= 64 * 1024, BUF_EINVAL, "buf_vn_pointwise: weights exceed 64 KiB of LDS");
    long long total = (long long)n * cout;
    k_vn_pointwise<<<cdiv(total, 256), 256, lds, (hipStream_t)stream>>>(a, ind_a, ind_stride, na, ca, b, cb, n, cout,
                                                                      make_params(wf, wd, bn_scale, bn_shift, slope),
                                                                      residual, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_gather_max(const float* feats, const int* idx, int nq, int ns, int k, int width, float* out, void* stream)
{
    BUF_REQUIRE(nq >= 0 && ns >= 0 && k > 0 && width > 0, BUF_EINVAL, "buf_gather_max: bad sizes");
    if (nq == 0) return BUF_OK;
    BUF_REQUIRE(feats && idx && out, BUF_EINVAL, "buf_gather_max: null argument");
    long long total = (long long)nq * width;
    k_gather_max<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(feats, idx, nq, ns, k, width, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

extern "C" int buf_vn_std(const float* x, const float* z, int n, int c, float* out, void* stream)
{
    BUF_REQUIRE(n >= 0 && c > 0, BUF_EINVAL, "buf_vn_std: bad sizes");
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(x && z && out, BUF_EINVAL, "buf_vn_std: null argument");
    long long total = (long long)n * c * 3;
    k_vn_std<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x, z, n, c, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------
// InstanceNorm1d over contiguous row segments (models/point_learner.py:131,133: the score heads normalise over
// the stacked src+tgt points OF ONE PAIR; a batch of pairs = one segment per pair).  Two-pass statistics
// (mean, then biased variance of the deviations) without atomics: every segment is cut into SEG_SPLITS chunks,
// per-chunk partial sums are combined in a fixed order, so the result does not depend on scheduling.
#define SEG_SPLITS 64

// partial[seg][split][ch] = sum over the chunk's rows of (x - mean)^(SQ ? 2 : 1); blockDim = c * rows_per_iter
template <bool SQ>
__global__ void __launch_bounds__(256) k_seg_partial(const float* __restrict__ x, const int* __restrict__ seg_off, int c,
                                                   const float* __restrict__ mean, double* __restrict__ partial)
{
    __shared__ double sh[256];
    const int seg = blockIdx.x, split = blockIdx.y;
    const int lo = seg_off[seg], hi = seg_off[seg + 1];
    const int chunk = (hi - lo + SEG_SPLITS - 1) / SEG_SPLITS;
    const int r0 = lo + split * chunk, r1 = min(r0 + chunk, hi);
    const int rpi = blockDim.x / c, ch = threadIdx.x % c, rr = threadIdx.x / c;
    const float mu = SQ ? mean[seg * c + ch] : 0.f;
    double acc = 0.0;                                        // fp64 sums (tens of thousands of rows per segment), rounded once in k_seg_final
    for (int r = r0 + rr; r < r1; r += rpi) {
        float v = x[(size_t)r * c + ch] - mu;
        acc += SQ ? (double)v * (double)v : (double)v;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < c) {
        double t = 0.0;
        for (int k = 0; k < rpi; k++) t += sh[k * c + threadIdx.x];
        partial[((size_t)seg * SEG_SPLITS + split) * c + threadIdx.x] = t;
    }
}

__global__ void k_seg_final(const double* __restrict__ partial, const int* __restrict__ seg_off, int nseg, int c, float* __restrict__ out)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg * c) return;
    int seg = t / c, ch = t % c;
    double s = 0.0;
    for (int k = 0; k < SEG_SPLITS; k++) s += partial[((size_t)seg * SEG_SPLITS + k) * c + ch];
    int cnt = seg_off[seg + 1] - seg_off[seg];
    out[t] = cnt > 0 ? (float)(s / (double)cnt) : 0.f;
}

__global__ void __launch_bounds__(256) k_seg_apply(const float* __restrict__ x, const int* __restrict__ seg_off, int nseg, int c, long long total,
                                                 const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                 float* __restrict__ out)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int row = (int)(t / c), ch = (int)(t % c);
    int seg = find_elem(seg_off, nseg, row);
    out[t] = (x[t] - mean[seg * c + ch]) / sqrtf(var[seg * c + ch] + eps);
}

extern "C" size_t buf_segment_instance_norm_ws_bytes(int nseg, int c)
{
    if (nseg <= 0 || c <= 0) return 256;
    return 512 + sizeof(int) * ((size_t)nseg + 1) + sizeof(double) * (size_t)nseg * c * SEG_SPLITS + sizeof(float) * (size_t)nseg * c * 2 + 256;
}

// x f32[n,c] (rows of segment s contiguous, lens_host[s] rows each) -> out f32[n,c] = (x - mean_s) / sqrt(var_s + eps)
extern "C" int buf_segment_instance_norm(const float* x, int n, int c, const int* lens_host, int nseg, float eps, float* out,
                                         void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(n >= 0 && c > 0 && c <= 256 && nseg > 0, BUF_EINVAL, "buf_segment_instance_norm: n=%d c=%d nseg=%d", n, c, nseg);
    BUF_REQUIRE(lens_host && ws, BUF_EINVAL, "buf_segment_instance_norm: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver w(ws, ws_bytes);
    int* off = w.take<int>((size_t)nseg + 1);
    double* partial = w.take<double>((size_t)nseg * c * SEG_SPLITS);
    float* mean = w.take<float>((size_t)nseg * c);
    float* var = w.take<float>((size_t)nseg * c);
    BUF_REQUIRE(w.ok, BUF_EWORKSPACE, "buf_segment_instance_norm: workspace %zu < %zu", ws_bytes, w.used());
    int rc = upload_offsets(off, lens_host, nseg, n, "buf_segment_instance_norm", s);
    if (rc) return rc;
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(x && out, BUF_EINVAL, "buf_segment_instance_norm: null argument");
    const int threads = c * (256 / c);
    dim3 grid(nseg, SEG_SPLITS);
    k_seg_partial<false><<<grid, threads, 0, s>>>(x, off, c, nullptr, partial);
    k_seg_final<<<cdiv(nseg * c, 64), 64, 0, s>>>(partial, off, nseg, c, mean);
    k_seg_partial<true><<<grid, threads, 0, s>>>(x, off, c, mean, partial);
    k_seg_final<<<cdiv(nseg * c, 64), 64, 0, s>>>(partial, off, nseg, c, var);
    long long total = (long long)n * c;
    k_seg_apply<<<(unsigned)cdiv(total, 256), 256, 0, s>>>(x, off, nseg, c, total, mean, var, eps, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}

// ------------------------------------------------------------------------------------------
// Conv1d(kernel 1) of the score heads (models/point_learner.py:128-136,163-171): out[i] = W x[i] + b with
// cin <= 32, cout <= 32.  One lane per row, weights broadcast from LDS; activation: 0 none, 1 sigmoid, 2 softplus.
#define ROWLIN_MAX 32
__global__ void __launch_bounds__(256) k_row_linear(const float* __restrict__ x, int n, int cin, int cout, const float* __restrict__ w,
                                                  const float* __restrict__ b, int act, float* __restrict__ out)
{
    __shared__ float ws[ROWLIN_MAX * ROWLIN_MAX + ROWLIN_MAX];
    for (int i = threadIdx.x; i < cout * cin; i += 256) ws[i] = w[i];
    for (int i = threadIdx.x; i < cout; i += 256) ws[ROWLIN_MAX * ROWLIN_MAX + i] = b[i];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float xv[ROWLIN_MAX];
#pragma unroll
    for (int c = 0; c < ROWLIN_MAX; c++) xv[c] = c < cin ? x[(size_t)row * cin + c] : 0.f;
    for (int o = 0; o < cout; o++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < ROWLIN_MAX; c++) acc += c < cin ? ws[o * cin + c] * xv[c] : 0.f;   // ascending c, like a dot product
        acc += ws[ROWLIN_MAX * ROWLIN_MAX + o];
        if (act == 1) acc = 1.f / (1.f + expf(-acc));
        else if (act == 2) acc = acc > 20.f ? acc : log1pf(expf(acc));                        // F.softplus (threshold 20)
        out[(size_t)row * cout + o] = acc;
    }
}

extern "C" int buf_row_linear(const float* x, int n, int cin, int cout, const float* w, const float* b, int activation, float* out,
                              void* stream)
{
    BUF_REQUIRE(n >= 0 && cin > 0 && cin <= ROWLIN_MAX && cout > 0 && cout <= ROWLIN_MAX, BUF_EINVAL,
                "buf_row_linear: n=%d cin=%d cout=%d (widths up to %d)", n, cin, cout, ROWLIN_MAX);
    BUF_REQUIRE(activation >= 0 && activation <= 2, BUF_EINVAL, "buf_row_linear: activation %d", activation);
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(x && w && b && out, BUF_EINVAL, "buf_row_linear: null argument");
    k_row_linear<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, cin, cout, w, b, activation, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
