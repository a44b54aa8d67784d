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
    // (Round 5: the fp64 multiply-adds are written as fma(): the product of two fp32 values is EXACT in fp64 (48 <= 53 bits), so
    // a * f + P rounds once either way -- bit-identical to the separate multiply and add of -ffp-contract=off, half the fp64
    // instructions.)
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
                    Px = fma(a, fx, Px); Py = fma(a, fy, Py); Pz = fma(a, fz, Pz); Dx = fma(b, fx, Dx); Dy = fma(b, fy, Dy); Dz = fma(b, fz, Dz);
                }
                { double a = wfo[cin], b = wdo[cin];
                  Px = fma(a, ex, Px); Py = fma(a, ey, Py); Pz = fma(a, ez, Pz); Dx = fma(b, ex, Dx); Dy = fma(b, ey, Dy); Dz = fma(b, ez, Dz); }
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
#ifdef VG6_PROBE_NOP     // development builds of tools/pk_bisect.sh only: drain the LDS reads and idle VG6_PROBE_NOP + 1 cycles before the arithmetic
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop %0" :: "n"(VG6_PROBE_NOP) : "memory");
#endif
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
        x = fma(a, (double)f[3 * c], x); y = fma(a, (double)f[3 * c + 1], y); z = fma(a, (double)f[3 * c + 2], z);
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
                px = (float)fma(wfe, (double)e.x, (double)a[0]); py = (float)fma(wfe, (double)e.y, (double)a[1]); pz = (float)fma(wfe, (double)e.z, (double)a[2]);
                dx = (float)fma(wde, (double)e.x, (double)b[0]); dy = (float)fma(wde, (double)e.y, (double)b[1]); dz = (float)fma(wde, (double)e.z, (double)b[2]);
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
            Px = fma(a, fx, Px); Py = fma(a, fy, Py); Pz = fma(a, fz, Pz); Dx = fma(b, fx, Dx); Dy = fma(b, fy, Dy); Dz = fma(b, fz, Dz);
        }
    }
    if (cb > 0) {
        const float* f = B + (size_t)i * 3 * cb;
        for (int c = 0; c < cb; c++) {
            double a = wfo[ca + c], b = has_dir ? wdo[ca + c] : 0.f;
            double fx = f[3 * c], fy = f[3 * c + 1], fz = f[3 * c + 2];
            Px = fma(a, fx, Px); Py = fma(a, fy, Py); Pz = fma(a, fz, Pz); Dx = fma(b, fx, Dx); Dy = fma(b, fy, Dy); Dz = fma(b, fz, Dz);
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
        acc = SQ ? fma((double)v, (double)v, acc) : acc + (double)v;       // (fp32 x fp32 is exact in fp64: the fma rounds like mul + add)
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

// ------------------------------------------------------------------------------------------
// One score head (models/point_learner.py:128-136,163-171; vn_layers.py:169-222) in 7 launches instead of 17 (round 5):
//   k_score_head_a      VNStdFeature (vn1 -> vn2 -> vn_lin -> x . z) + Conv1d 30 -> 20           (was 3 x k_vn_pointwise, k_vn_std, k_row_linear)
//   k_seg_sum / k_seg_sq   the InstanceNorm statistics (fixed 64-way split per segment, fp64)       (was 4 kernels per norm)
//   k_row_linear_norm   normalise on load + Conv1d 20 -> 10, then again for 10 -> 1 + activation   (was k_seg_apply + k_row_linear)
// Every value is formed by the operations of the separate kernels in their order (k_vn_pointwise's fp64 dot products rounded
// once, vn_epilogue, k_vn_std's three-term sums, k_row_linear's ascending-channel sums, k_seg_partial / k_seg_final's chunked
// fp64 sums, k_seg_apply's (x - mean) / sqrtf(var + eps)): bit-identical to the 17-launch path (tests/test_model_gpu.py).
#define SH_C 10        // input vector channels (x f32[n, 30])
#define SH_C1 10       // vn1
#define SH_C2 5        // vn2
#define SH_CZ 3        // vn_lin
#define SH_H1 20       // Conv1d 30 -> 20
struct ScoreHeadParams {
    VnParams vn1, vn2;
    const float* lin;       // [3][5] map_to_feat of vn_lin (linear only)
    const float* w0;        // [20][30]
    const float* b0;        // [20]
};

template <int CIN, int COUT>
__device__ __forceinline__ void sh_vn_layer(const float (&in)[CIN * 3], const float* __restrict__ wf, const float* __restrict__ wd,
                                            const float* __restrict__ bsc, const float* __restrict__ bsh, bool has_bn, float slope,
                                            float (&out)[COUT * 3])
{
#pragma unroll
    for (int o = 0; o < COUT; o++) {
        double Px = 0, Py = 0, Pz = 0, Dx = 0, Dy = 0, Dz = 0;
#pragma unroll
        for (int c = 0; c < CIN; c++) {
            double a = wf[o * CIN + c], b = wd[o * CIN + c];
            double fx = in[3 * c], fy = in[3 * c + 1], fz = in[3 * c + 2];
            Px = fma(a, fx, Px); Py = fma(a, fy, Py); Pz = fma(a, fz, Pz); Dx = fma(b, fx, Dx); Dy = fma(b, fy, Dy); Dz = fma(b, fz, Dz);
        }
        float px = (float)Px, py = (float)Py, pz = (float)Pz, dx = (float)Dx, dy = (float)Dy, dz = (float)Dz;
        vn_epilogue(px, py, pz, dx, dy, dz, has_bn, has_bn ? bsc[o] : 0.f, has_bn ? bsh[o] : 0.f, slope);
        out[3 * o] = px; out[3 * o + 1] = py; out[3 * o + 2] = pz;
    }
}

__global__ void __launch_bounds__(128) k_score_head_a(const float* __restrict__ x, int n, ScoreHeadParams H, float* __restrict__ h1)
{
    __shared__ float w1f[SH_C1 * SH_C], w1d[SH_C1 * SH_C], w2f[SH_C2 * SH_C1], w2d[SH_C2 * SH_C1], wl[SH_CZ * SH_C2];
    __shared__ float bn[2 * SH_C1 + 2 * SH_C2], w0[SH_H1 * 3 * SH_C], b0[SH_H1];
    const bool bn1 = H.vn1.bn_scale != nullptr, bn2 = H.vn2.bn_scale != nullptr;
    for (int t = threadIdx.x; t < SH_C1 * SH_C; t += 128) { w1f[t] = H.vn1.wf[t]; w1d[t] = H.vn1.wd[t]; }
    for (int t = threadIdx.x; t < SH_C2 * SH_C1; t += 128) { w2f[t] = H.vn2.wf[t]; w2d[t] = H.vn2.wd[t]; }
    for (int t = threadIdx.x; t < SH_CZ * SH_C2; t += 128) wl[t] = H.lin[t];
    for (int t = threadIdx.x; t < SH_C1; t += 128) { bn[t] = bn1 ? H.vn1.bn_scale[t] : 0.f; bn[SH_C1 + t] = bn1 ? H.vn1.bn_shift[t] : 0.f; }
    for (int t = threadIdx.x; t < SH_C2; t += 128) { bn[2 * SH_C1 + t] = bn2 ? H.vn2.bn_scale[t] : 0.f; bn[2 * SH_C1 + SH_C2 + t] = bn2 ? H.vn2.bn_shift[t] : 0.f; }
    for (int t = threadIdx.x; t < SH_H1 * 3 * SH_C; t += 128) w0[t] = H.w0[t];
    for (int t = threadIdx.x; t < SH_H1; t += 128) b0[t] = H.b0[t];
    __syncthreads();
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    float xv[3 * SH_C], y1[3 * SH_C1], y2[3 * SH_C2], z[3 * SH_CZ];
#pragma unroll
    for (int c = 0; c < 3 * SH_C; c++) xv[c] = x[(size_t)i * 3 * SH_C + c];
    sh_vn_layer<SH_C, SH_C1>(xv, w1f, w1d, bn, bn + SH_C1, bn1, H.vn1.slope, y1);
    sh_vn_layer<SH_C1, SH_C2>(y1, w2f, w2d, bn + 2 * SH_C1, bn + 2 * SH_C1 + SH_C2, bn2, H.vn2.slope, y2);
#pragma unroll
    for (int o = 0; o < SH_CZ; o++) {                           // vn_lin: plain VNLinear (k_vn_pointwise with wd == null)
        double Px = 0, Py = 0, Pz = 0;
#pragma unroll
        for (int c = 0; c < SH_C2; c++) {
            double a = wl[o * SH_C2 + c];
            Px = fma(a, (double)y2[3 * c], Px); Py = fma(a, (double)y2[3 * c + 1], Py); Pz = fma(a, (double)y2[3 * c + 2], Pz);
        }
        z[3 * o] = (float)Px; z[3 * o + 1] = (float)Py; z[3 * o + 2] = (float)Pz;
    }
    float hs[3 * SH_C];                                          // k_vn_std: x_std[c*3+k] = sum_j x[c,j] z[k,j]
#pragma unroll
    for (int c = 0; c < SH_C; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) hs[3 * c + k] = xv[3 * c] * z[3 * k] + xv[3 * c + 1] * z[3 * k + 1] + xv[3 * c + 2] * z[3 * k + 2];
#pragma unroll 4
    for (int o = 0; o < SH_H1; o++) {                           // k_row_linear: ascending channel, then the bias
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3 * SH_C; c++) acc += w0[o * 3 * SH_C + c] * hs[c];
        acc += 0.f; acc += 0.f;                                  // (k_row_linear runs its loop to 32 channels with zeros past cin)
        acc += b0[o];
        h1[(size_t)i * SH_H1 + o] = acc;
    }
}

// pass 1 / pass 2 of the InstanceNorm statistics: k_seg_partial<false> as it is; the second pass forms the mean from the first pass's
// partial sums itself (k_seg_final's sum in k_seg_final's order) instead of waiting for a kernel that does
__global__ void __launch_bounds__(256) k_seg_sq(const float* __restrict__ x, const int* __restrict__ seg_off, int c,
                                              const double* __restrict__ partial1, double* __restrict__ partial2)
{
    __shared__ double sh[256];
    __shared__ float mu_s[ROWLIN_MAX];
    const int seg = blockIdx.x, split = blockIdx.y;
    const int lo = seg_off[seg], hi = seg_off[seg + 1];
    if (threadIdx.x < c) {
        double s = 0.0;
        for (int k = 0; k < SEG_SPLITS; k++) s += partial1[((size_t)seg * SEG_SPLITS + k) * c + threadIdx.x];
        mu_s[threadIdx.x] = hi > lo ? (float)(s / (double)(hi - lo)) : 0.f;
    }
    __syncthreads();
    const int chunk = (hi - lo + SEG_SPLITS - 1) / SEG_SPLITS;
    const int r0 = lo + split * chunk, r1 = min(r0 + chunk, hi);
    const int rpi = blockDim.x / c, ch = threadIdx.x % c, rr = threadIdx.x / c;
    const float mu = mu_s[ch];
    double acc = 0.0;
    for (int r = r0 + rr; r < r1; r += rpi) {
        float v = x[(size_t)r * c + ch] - mu;
        acc = fma((double)v, (double)v, acc);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < c) {
        double t = 0.0;
        for (int k = 0; k < rpi; k++) t += sh[k * c + threadIdx.x];
        partial2[((size_t)seg * SEG_SPLITS + split) * c + threadIdx.x] = t;
    }
}

// k_seg_apply + k_row_linear in one: a block's 256 rows touch a few segments; their mean / variance come from the partial sums
#define RLN_SEGS 8
__global__ void __launch_bounds__(256) k_row_linear_norm(const float* __restrict__ x, int n, int cin, int cout, const float* __restrict__ w,
                                                       const float* __restrict__ b, int act, const int* __restrict__ seg_off, int nseg,
                                                       const double* __restrict__ partial1, const double* __restrict__ partial2, float eps,
                                                       float* __restrict__ out)
{
    __shared__ float ws[ROWLIN_MAX * ROWLIN_MAX + ROWLIN_MAX];
    __shared__ float mu_s[RLN_SEGS * ROWLIN_MAX], var_s[RLN_SEGS * ROWLIN_MAX];
    for (int i = threadIdx.x; i < cout * cin; i += 256) ws[i] = w[i];
    for (int i = threadIdx.x; i < cout; i += 256) ws[ROWLIN_MAX * ROWLIN_MAX + i] = b[i];
    const int row0 = blockIdx.x * 256, rowl = min(row0 + 255, n - 1);
    const int seg0 = find_elem(seg_off, nseg, row0), seg1 = find_elem(seg_off, nseg, rowl);
    for (int base = seg0; base <= seg1; base += RLN_SEGS) {          // (more than RLN_SEGS segments inside 256 rows: in rounds)
        __syncthreads();
        for (int t = threadIdx.x; t < RLN_SEGS * cin; t += 256) {
            const int seg = base + t / cin, ch = t % cin;
            if (seg > seg1) continue;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < SEG_SPLITS; k++) s1 += partial1[((size_t)seg * SEG_SPLITS + k) * cin + ch];
            for (int k = 0; k < SEG_SPLITS; k++) s2 += partial2[((size_t)seg * SEG_SPLITS + k) * cin + ch];
            const int cnt = seg_off[seg + 1] - seg_off[seg];
            mu_s[t] = cnt > 0 ? (float)(s1 / (double)cnt) : 0.f;
            var_s[t] = cnt > 0 ? (float)(s2 / (double)cnt) : 0.f;
        }
        __syncthreads();
        const int row = row0 + threadIdx.x;
        if (row >= n) continue;
        const int seg = find_elem(seg_off, nseg, row);
        if (seg < base || seg >= base + RLN_SEGS) continue;
        const float* mu = mu_s + (seg - base) * cin;
        const float* var = var_s + (seg - base) * cin;
        float xv[ROWLIN_MAX];
#pragma unroll
        for (int c = 0; c < ROWLIN_MAX; c++) xv[c] = c < cin ? (x[(size_t)row * cin + c] - mu[c]) / sqrtf(var[c] + eps) : 0.f;
        for (int o = 0; o < cout; o++) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < ROWLIN_MAX; c++) acc += c < cin ? ws[o * cin + c] * xv[c] : 0.f;
            acc += ws[ROWLIN_MAX * ROWLIN_MAX + o];
            if (act == 1) acc = 1.f / (1.f + expf(-acc));
            else if (act == 2) acc = acc > 20.f ? acc : log1pf(expf(acc));
            out[(size_t)row * cout + o] = acc;
        }
    }
}

extern "C" size_t buf_score_head_ws_bytes(int n, int nseg)
{
    if (n < 0 || nseg <= 0) return 256;
    return 1024 + sizeof(int) * ((size_t)nseg + 1) + 2 * (sizeof(double) * (size_t)nseg * SH_H1 * SEG_SPLITS + 256) +
           sizeof(float) * (size_t)(n > 0 ? n : 1) * (SH_H1 + 10) + 512;
}

// x f32[n,30] (10 vector channels) -> score f32[n,1]; segments = pairs (lens_host[nseg], rows contiguous); widths of the released heads
// only (VNStdFeature 10 -> 10 -> 5 -> 3, Conv1d 30 -> 20 -> 10 -> 1): anything else is BUF_EINVAL and the caller keeps the per-layer path.
// vn1_* / vn2_*: map_to_feat, map_to_dir, folded VN-BatchNorm scale / shift (null: none); lin: vn_lin's map_to_feat;
// w / b: the three Conv1d layers; final_activation: 0 none, 1 sigmoid, 2 softplus.
extern "C" int buf_score_head(const float* x, int n, const int* lens_host, int nseg,
                              const float* vn1_wf, const float* vn1_wd, const float* vn1_bsc, const float* vn1_bsh, float vn1_slope,
                              const float* vn2_wf, const float* vn2_wd, const float* vn2_bsc, const float* vn2_bsh, float vn2_slope,
                              const float* lin, const float* w0, const float* b0, const float* w1, const float* b1, int c1,
                              const float* w2, const float* b2, int final_activation, float eps, float* out,
                              void* ws, size_t ws_bytes, void* stream)
{
    BUF_REQUIRE(n >= 0 && nseg > 0 && c1 > 0 && c1 <= ROWLIN_MAX, BUF_EINVAL, "buf_score_head: n=%d nseg=%d c1=%d", n, nseg, c1);
    BUF_REQUIRE(final_activation >= 0 && final_activation <= 2, BUF_EINVAL, "buf_score_head: activation %d", final_activation);
    BUF_REQUIRE(lens_host && ws, BUF_EINVAL, "buf_score_head: null argument");
    hipStream_t s = (hipStream_t)stream;
    WsCarver wc(ws, ws_bytes);
    int* off = wc.take<int>((size_t)nseg + 1);
    double* p1 = wc.take<double>((size_t)nseg * SH_H1 * SEG_SPLITS);
    double* p2 = wc.take<double>((size_t)nseg * SH_H1 * SEG_SPLITS);
    float* h1 = wc.take<float>((size_t)(n > 0 ? n : 1) * SH_H1);
    float* h2 = wc.take<float>((size_t)(n > 0 ? n : 1) * c1);
    BUF_REQUIRE(wc.ok && c1 <= 10, BUF_EWORKSPACE, "buf_score_head: workspace %zu < %zu (or c1 > 10)", ws_bytes, wc.used());
    int rc = upload_offsets(off, lens_host, nseg, n, "buf_score_head", s);
    if (rc) return rc;
    if (n == 0) return BUF_OK;
    BUF_REQUIRE(x && vn1_wf && vn1_wd && vn2_wf && vn2_wd && lin && w0 && b0 && w1 && b1 && w2 && b2 && out, BUF_EINVAL, "buf_score_head: null argument");
    ScoreHeadParams H;
    H.vn1 = make_params(vn1_wf, vn1_wd, vn1_bsc, vn1_bsh, vn1_slope);
    H.vn2 = make_params(vn2_wf, vn2_wd, vn2_bsc, vn2_bsh, vn2_slope);
    H.lin = lin; H.w0 = w0; H.b0 = b0;
    k_score_head_a<<<cdiv(n, 128), 128, 0, s>>>(x, n, H, h1);
    dim3 grid(nseg, SEG_SPLITS);
    int threads = SH_H1 * (256 / SH_H1);
    k_seg_partial<false><<<grid, threads, 0, s>>>(h1, off, SH_H1, nullptr, p1);
    k_seg_sq<<<grid, threads, 0, s>>>(h1, off, SH_H1, p1, p2);
    k_row_linear_norm<<<cdiv(n, 256), 256, 0, s>>>(h1, n, SH_H1, c1, w1, b1, 0, off, nseg, p1, p2, eps, h2);
    threads = c1 * (256 / c1);
    k_seg_partial<false><<<grid, threads, 0, s>>>(h2, off, c1, nullptr, p1);
    k_seg_sq<<<grid, threads, 0, s>>>(h2, off, c1, p1, p2);
    k_row_linear_norm<<<cdiv(n, 256), 256, 0, s>>>(h2, n, c1, 1, w2, b2, final_activation, off, nseg, p1, p2, eps, out);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
