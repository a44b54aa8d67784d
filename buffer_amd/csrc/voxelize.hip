// A9 + A10 + the point MLP of A11, fused (models/patch_embedder.py:123-171,74-79;
// utils/common.py:431-498,501-525).
//
// Reference: axis_align -> normalize -> sphere_query (420 ball queries of 10 samples per patch,
// materialising [P,420,10,3]) -> var_to_invar -> Conv2d1x1(3->16)+BN+ReLU -> max over the 10 samples.
// Here: one workgroup per patch; the aligned, normalised 512-point patch sits in LDS (8 KB); one lane
// per cylindrical voxel centre scans it in index order (LDS broadcast reads) and notes its first 10 hits;
// then all centre lanes run the azimuth de-rotation, the 3->16 MLP, BN, ReLU and the running max slot by slot.
// HBM traffic: 6 KB in, 26.9 KB out per patch; [P,420,10,3] never exists.
#include "common.h"

#define VOX_THREADS 448     // 7 wavefronts >= 420 centres
#define VOX_MAXPTS 1024
#define VOX_CH 16
#define VOX_MAXS 16           // max samples per voxel kept in the hit list

struct VoxMlp {
    float w[VOX_CH][3];     // Desc.pnt_layer.0.weight
    float b[VOX_CH];        // Desc.pnt_layer.0.bias
    float s[VOX_CH];        // BN folded: gamma / sqrt(var + 1e-5)
    float t[VOX_CH];        //            beta - mean * s
};

__global__ void __launch_bounds__(VOX_THREADS) k_patch_voxelize(const float* __restrict__ patches, const float* __restrict__ axis,
                                                             int npts, float des_r, const float* __restrict__ centres,
                                                             int ncentres, int azi_n, const float* __restrict__ azi_cs,
                                                             float voxel_r2, int nsample, VoxMlp M,
                                                             float* __restrict__ out_x, float* __restrict__ out_R,
                                                             float* __restrict__ out_rand, float* __restrict__ out_patches)
{
    __shared__ float4 pts[VOX_MAXPTS];
    __shared__ float Rs[9];
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* src = patches + (size_t)p * npts * 3;
    if (tid == 0) {
        float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        float rx = 1.f, ry = 0.f, rz = 0.f;                      // KITTI/ETH: rand_axis = e_x, R = I (:143-147)
        if (axis) {
            // RodsRotatFormula(z_axis, e_z) (utils/common.py:501-525), returned transposed
            float ax = axis[3 * (size_t)p], ay = axis[3 * (size_t)p + 1], az = axis[3 * (size_t)p + 2];
            float cx = ay, cy = -ax, cz = 0.f;                   // a x e_z
            float na = sqrtf(ax * ax + ay * ay + az * az);
            float cs = az / (fmaxf(na, 1e-8f) * 1.0f);           // cosine_similarity
            cs = fminf(fmaxf(cs, -1.f), 1.f);
            float th = acosf(cs);
            float nc = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);   // F.normalize
            cx /= nc; cy /= nc; cz /= nc;
            rx = cx; ry = cy; rz = cz;                           // rand_axis = normalize(z_axis x e_z) (:138-141)
            float K[9] = { 0, -cz, cy, cz, 0, -cx, -cy, cx, 0 };
            float K2[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
            float sn = sinf(th), oc = 1.f - cosf(th);
            float Rm[9];
            for (int i = 0; i < 9; i++) Rm[i] = (i % 4 == 0 ? 1.f : 0.f) + sn * K[i] + oc * K2[i];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R[3 * i + j] = Rm[3 * j + i];
        }
        for (int i = 0; i < 9; i++) { Rs[i] = R[i]; out_R[9 * (size_t)p + i] = R[i]; }
        out_rand[3 * (size_t)p] = rx; out_rand[3 * (size_t)p + 1] = ry; out_rand[3 * (size_t)p + 2] = rz;
    }
    __syncthreads();
    // centre on the keypoint (= last slot, :124-125), rotate (delta @ R), divide by des_r (:168-171)
    float kx = src[3 * (size_t)(npts - 1)], ky = src[3 * (size_t)(npts - 1) + 1], kz = src[3 * (size_t)(npts - 1) + 2];
    for (int k = tid; k < npts; k += VOX_THREADS) {
        float dx = src[3 * (size_t)k] - kx, dy = src[3 * (size_t)k + 1] - ky, dz = src[3 * (size_t)k + 2] - kz;
        float x = dx * Rs[0] + dy * Rs[3] + dz * Rs[6];
        float y = dx * Rs[1] + dy * Rs[4] + dz * Rs[7];
        float z = dx * Rs[2] + dy * Rs[5] + dz * Rs[8];
        x = x / des_r; y = y / des_r; z = z / des_r;
        pts[k] = make_float4(x, y, z, 0.f);
        if (out_patches) {
            float* d = out_patches + ((size_t)p * npts + k) * 3;
            d[0] = x; d[1] = y; d[2] = z;
        }
    }
    __syncthreads();

    const int c = tid;
    const bool active = c < ncentres;
    float cx = 0.f, cy = 0.f, cz = 0.f, ca = 1.f, sa = 0.f;
    if (active) {
        cx = centres[3 * c]; cy = centres[3 * c + 1]; cz = centres[3 * c + 2];
        int az = c % azi_n;                                      // ordering rad -> ele -> azi (utils/common.py:422-428)
        ca = azi_cs[2 * az]; sa = azi_cs[2 * az + 1];            // cos/sin of -az * 2pi/azi_n (:485-491)
    }
    // Phase 1: every centre lane records the indices of its first `nsample` hits (index order).  Only a tiny
    // store sits in the divergent branch: running the 16-channel MLP right here would execute it once per point
    // for one or two active lanes (a point is inside ~8 of the 420 balls).
    __shared__ unsigned short hits[VOX_MAXS][VOX_THREADS];
    int cnt = 0, nreal = 0;                                      // accepted samples / samples kept in the list
    bool zero_slot = false;
    // branch-light scan, 8 points per step so that the LDS reads of a step are in flight together
    for (int k0 = 0; k0 < npts; k0 += 8) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = pts[min(k0 + j, npts - 1)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = k0 + j;
            const bool hit = active && k < npts && cnt < nsample && sqdist3(cx, cy, cz, q[j].x, q[j].y, q[j].z) < voxel_r2;
            const bool keep = hit && k != 0;                     // utils/common.py:447-449: a hit on point 0 is zeroed
            if (keep) hits[nreal][tid] = (unsigned short)k;
            nreal += keep ? 1 : 0;
            cnt += hit ? 1 : 0;
            zero_slot = zero_slot || (hit && k == 0);
        }
        if ((k0 & 31) == 24 && !__any(active && cnt < nsample)) break;
    }
    // Phase 2: slot by slot, all centre lanes evaluate de-rotation -> 3->16 MLP -> BN -> ReLU -> running max together.
    float acc[VOX_CH];
#pragma unroll
    for (int ch = 0; ch < VOX_CH; ch++) acc[ch] = -3.4e38f;
    for (int sidx = 0; __any(sidx < nreal); sidx++) {
        if (sidx < nreal) {
            float4 q = pts[hits[sidx][tid]];
            float nx = q.x * ca - q.y * sa, ny = q.x * sa + q.y * ca, nz = q.z;
#pragma unroll
            for (int ch = 0; ch < VOX_CH; ch++) {
                float h = M.w[ch][0] * nx + M.w[ch][1] * ny + M.w[ch][2] * nz + M.b[ch];
                acc[ch] = fmaxf(acc[ch], fmaxf(h * M.s[ch] + M.t[ch], 0.f));
            }
        }
    }
    if (active) {
        bool padded = cnt < nsample || zero_slot;                // zeroed slots go through the MLP as the origin
#pragma unroll
        for (int ch = 0; ch < VOX_CH; ch++) {
            float v = acc[ch];
            if (padded) v = fmaxf(v, fmaxf(M.b[ch] * M.s[ch] + M.t[ch], 0.f));
            out_x[((size_t)p * VOX_CH + ch) * ncentres + c] = v;
        }
    }
}

extern "C" int buf_patch_voxelize(const float* patches, const float* axis, int npatch, int npts, float des_r,
                                  const float* centres, int ncentres, int azi_n, const float* azi_cs, float voxel_r,
                                  int nsample, const float* mlp_w, const float* mlp_b, const float* bn_scale,
                                  const float* bn_shift, float* out_x, float* out_R, float* out_rand, float* out_patches,
                                  void* stream)
{
    BUF_REQUIRE(npatch >= 0 && npts > 0 && npts <= VOX_MAXPTS, BUF_EINVAL, "buf_patch_voxelize: npts=%d (max %d)", npts, VOX_MAXPTS);
    BUF_REQUIRE(nsample <= VOX_MAXS, BUF_EINVAL, "buf_patch_voxelize: nsample=%d (max %d)", nsample, VOX_MAXS);
    BUF_REQUIRE(ncentres > 0 && ncentres <= VOX_THREADS && azi_n > 0 && nsample > 0, BUF_EINVAL,
                "buf_patch_voxelize: ncentres=%d (max %d)", ncentres, VOX_THREADS);
    if (npatch == 0) return BUF_OK;
    BUF_REQUIRE(patches && centres && azi_cs && mlp_w && mlp_b && bn_scale && bn_shift && out_x && out_R && out_rand,
                BUF_EINVAL, "buf_patch_voxelize: null argument");
    VoxMlp M;   // host copies of the 16x3 MLP (HOST pointers: tiny, passed by value to the kernel)
    memcpy(M.w, mlp_w, sizeof(M.w)); memcpy(M.b, mlp_b, sizeof(M.b));
    memcpy(M.s, bn_scale, sizeof(M.s)); memcpy(M.t, bn_shift, sizeof(M.t));
    k_patch_voxelize<<<npatch, VOX_THREADS, 0, (hipStream_t)stream>>>(patches, axis, npts, des_r, centres, ncentres, azi_n,
                                                                    azi_cs, voxel_r * voxel_r, nsample, M, out_x, out_R,
                                                                    out_rand, out_patches);
    BUF_LAUNCH_CHECK();
    return BUF_OK;
}
