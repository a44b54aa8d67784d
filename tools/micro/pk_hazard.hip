// Stand-alone reproducer attempt for the round-5 finding (profiles/r05_packed_fp32_hazard.txt): packed-fp32 vector instructions of one
// wavefront returning wrong values in lanes 0..15 while ANOTHER wavefront of the same SIMD issues v_mfma_f32_16x16x32_f16.
//
// Victims (each compared bitwise against a twin that computes the same IEEE values without packed instructions):
//   A  register-only inline asm: v_pk_mul_f32 (op_sel swizzle) -> v_pk_add_f32 (op_sel + neg) -> v_pk_mul_f32 with an SGPR-pair
//      operand -> v_pk_add_f32, 256 rounds, an integer checksum of every intermediate; twin = v_mul_f32 / v_sub_f32 / v_add_f32.
//   B  compiled C++ mirroring the inner loop of k_vn_gather6_lds (csrc/vn.hip): two 16-byte LDS reads per slot, cross product, two
//      4-term contractions per axis, norm, IEEE divisions, fp64 accumulation.  This file is compiled TWICE: -DPK_VARIANT=1 with the
//      compiler free to form v_pk_*_f32 (it does: check with llvm-objdump) and -DPK_VARIANT=0 with -target-feature -packed-fp32-ops;
//      the two objects are linked into one program, outputs compared bitwise.
//   C  = B without the LDS stage (operands from registers, the same arithmetic): separates "packed ops behind ds_read_b96" from
//      "packed ops" (VERDICT r5 item 5's bisect).
// Aggressors on a second stream, sized to leave wavefront slots free on every SIMD: v_mfma_f32_16x16x32_f16 loop, v_mfma_f32_16x16x4_f32
// loop, none.  N launches of every victim beside each; a launch counts as bad if ANY lane differs from its twin / from the golden
// run made with no aggressor.
//
// build + run: tools/micro/pk_hazard.sh  (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, two objects)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#ifndef PK_VARIANT
#define PK_VARIANT 1
#endif

#define VB_K 24            // neighbour slots per point (the layer-0 limit of the pyramid is 15..25)
#define VB_PTS 32
#define VB_COUT 8          // lanes of a point: 256 threads = 32 points x 8 output channels

// ---- victim B / C: the arithmetic of k_vn_gather6_lds's slot loop -------------------------------------------------------------
// FLAGS (the second bisect, round 6): bit 0 = no IEEE division (a multiplication in its place), bit 1 = fp32 instead of fp64 accumulation,
// bit 2 = no square root.  The twin is compiled from the same source, so every variant still has its bitwise reference.
template <int FLAGS>
__device__ __forceinline__ void vb_epilogue(float& px, float& py, float& pz, float dx, float dy, float dz, float bsc, float bsh, float slope)
{
    const float n2 = px * px + py * py + pz * pz;
    float norm = ((FLAGS & 4) ? n2 : sqrtf(n2)) + 1e-6f;
    float nbn = norm * bsc + bsh;
    if (FLAGS & 1) { const float s = 0.37f * nbn; px = px * s; py = py * s; pz = pz * s; }
    else { px = px / norm * nbn; py = py / norm * nbn; pz = pz / norm * nbn; }
    float dot = px * dx + py * dy + pz * dz;
    if (!(dot >= 0.f)) {
        float dsq = dx * dx + dy * dy + dz * dz;
        float f = (FLAGS & 1) ? dot * 0.61f : dot / (dsq + 1e-6f);
        float rx = px - f * dx, ry = py - f * dy, rz = pz - f * dz;
        px = slope * px + (1.f - slope) * rx; py = slope * py + (1.f - slope) * ry; pz = slope * pz + (1.f - slope) * rz;
    } else {
        px = slope * px + (1.f - slope) * px; py = slope * py + (1.f - slope) * py; pz = slope * pz + (1.f - slope) * pz;
    }
}

template <bool LDS_STAGE, int FLAGS>
__device__ __forceinline__ void vb_body(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ out, int seed)
{
    __shared__ __attribute__((aligned(16))) float ef[VB_PTS * VB_K * 8];
    __shared__ float wf[VB_COUT * 4], wd[VB_COUT * 4], mean[VB_PTS * 4];
    const int tid = threadIdx.x;
    if (tid < VB_COUT * 4) { wf[tid] = w[tid]; wd[tid] = w[VB_COUT * 4 + tid]; }
    const float* base = src + ((size_t)(blockIdx.x + seed) % 64) * VB_PTS * VB_K * 8;
    for (int t = tid; t < VB_PTS * VB_K * 2; t += 256) reinterpret_cast<float4*>(ef)[t] = reinterpret_cast<const float4*>(base)[t];
    if (tid < VB_PTS) { mean[4 * tid] = 0.01f * tid; mean[4 * tid + 1] = -0.02f * tid; mean[4 * tid + 2] = 0.005f * tid; }
    __syncthreads();
    const int pl = tid / VB_COUT, o = tid - pl * VB_COUT;
    const float* wfo = wf + o * 4;
    const float* wdo = wd + o * 4;
    const float bsc = 0.9f + 0.01f * o, bsh = 0.05f * o, slope = 0.2f;
    const float mx = mean[4 * pl], my = mean[4 * pl + 1], mz = mean[4 * pl + 2];
    const float4* e4 = reinterpret_cast<const float4*>(ef) + (size_t)pl * VB_K * 2;
    const float4* g4 = reinterpret_cast<const float4*>(base) + (size_t)pl * VB_K * 2;
    double ax = 0.0, ay = 0.0, az = 0.0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    float4 ra[VB_K], rb[VB_K];
    if (!LDS_STAGE) {
#pragma unroll
        for (int k = 0; k < VB_K; k++) { ra[k] = g4[2 * k]; rb[k] = g4[2 * k + 1]; }
    }
#pragma unroll 1
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
        for (int k = 0; k < VB_K; k++) {
            const float4 a = LDS_STAGE ? e4[2 * k] : ra[k], b = LDS_STAGE ? e4[2 * k + 1] : rb[k];
            const float ex = a.x, ey = a.y, ez = a.z, fx = b.x, fy = b.y, fz = b.z;
            const float cx = fy * ez - fz * ey, cy = fz * ex - fx * ez, cz = fx * ey - fy * ex;
            float px = wfo[0] * fx + wfo[1] * ex + wfo[2] * cx + wfo[3] * mx;
            float py = wfo[0] * fy + wfo[1] * ey + wfo[2] * cy + wfo[3] * my;
            float pz = wfo[0] * fz + wfo[1] * ez + wfo[2] * cz + wfo[3] * mz;
            float dx = wdo[0] * fx + wdo[1] * ex + wdo[2] * cx + wdo[3] * mx;
            float dy = wdo[0] * fy + wdo[1] * ey + wdo[2] * cy + wdo[3] * my;
            float dz = wdo[0] * fz + wdo[1] * ez + wdo[2] * cz + wdo[3] * mz;
            vb_epilogue<FLAGS>(px, py, pz, dx, dy, dz, bsc, bsh, slope);
            if (FLAGS & 2) { sx += px; sy += py; sz += pz; }
            else { ax += (double)px; ay += (double)py; az += (double)pz; }
        }
    }
    float* dst = out + ((size_t)blockIdx.x * 256 + tid) * 3;
    if (FLAGS & 2) { dst[0] = sx * (1.f / VB_K); dst[1] = sy * (1.f / VB_K); dst[2] = sz * (1.f / VB_K); }
    else { dst[0] = (float)(ax / (double)VB_K); dst[1] = (float)(ay / (double)VB_K); dst[2] = (float)(az / (double)VB_K); }
}

#if PK_VARIANT == 1
#define VICTIM(NAME, LDS, FLAGS) extern "C" __global__ void __launch_bounds__(256) NAME##_pk(const float* src, const float* w, float* out, int seed) { vb_body<LDS, FLAGS>(src, w, out, seed); }
#else
#define VICTIM(NAME, LDS, FLAGS) extern "C" __global__ void __launch_bounds__(256) NAME##_sc(const float* src, const float* w, float* out, int seed) { vb_body<LDS, FLAGS>(src, w, out, seed); }
#endif
VICTIM(victim_b, true, 0)          // LDS-fed, full arithmetic
VICTIM(victim_c, false, 0)         // register-fed, full arithmetic
VICTIM(victim_c1, false, 1)        // ... without IEEE divisions
VICTIM(victim_c2, false, 2)        // ... with fp32 accumulation (no v_cvt_f64_f32 / v_add_f64)
VICTIM(victim_c4, false, 4)        // ... without the square root
VICTIM(victim_c7, false, 7)        // ... none of the three: multiplications and additions only

#if PK_VARIANT == 1
#define DECL_SC(NAME) extern "C" __global__ void NAME##_sc(const float* src, const float* w, float* out, int seed);
DECL_SC(victim_b) DECL_SC(victim_c) DECL_SC(victim_c1) DECL_SC(victim_c2) DECL_SC(victim_c4) DECL_SC(victim_c7)

// ---- victim A: register-only inline asm -----------------------------------------------------------------------------------------
// out[wave] = 64-bit mask of lanes whose packed checksum differs from the scalar twin's; out2 = the packed checksum per lane
// MODE bit 0: the operand pairs of the packed instructions are assembled by 32-bit v_mov_b32 writes right before them (the compiled victims do
// that all the time: v_mov_b32 v49, v46 ; v_pk_mul_f32 v[44:45], v[48:49], ...); bit 1: 160 further VGPRs are kept live (2 waves per SIMD, the
// packed operands sit in high registers), as in the compiled victims.
template <int MODE>
__global__ void __launch_bounds__(256) victim_a(const f2* __restrict__ sarg, int* __restrict__ flag, unsigned long long* __restrict__ lanes, unsigned* __restrict__ sums, int rounds)
{
    const f2 s = sarg[0];                                        // wave-uniform: an SGPR pair
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const float t0 = (float)(gid % 977) * (1.f / 977.f);
    f4 pad[40];
    if (MODE & 2) {
#pragma unroll
        for (int i = 0; i < 40; i++) { pad[i] = (f4){ t0 + i, t0 - i, t0 * i, 1.f }; asm volatile("" : "+v"(pad[i])); }
    }
    f2 a = { 0.25f + t0, 1.5f - t0 }, b = { 1.f - 0.5f * t0, 0.75f + 0.25f * t0 };
    float a0 = a.x, a1 = a.y;
    const float b0 = b.x, b1 = b.y;
    unsigned ck = 0u, cs = 0u;
    for (int it = 0; it < rounds; it++) {
        f2 t, u;
        if (MODE & 1) {
            f2 a2, b2, u2;
            asm volatile("v_mov_b32 %0, %1" : "=v"(a2.x) : "v"(a.x)); asm volatile("v_mov_b32 %0, %1" : "=v"(a2.y) : "v"(a.y));
            asm volatile("v_mov_b32 %0, %1" : "=v"(b2.x) : "v"(b.x)); asm volatile("v_mov_b32 %0, %1" : "=v"(b2.y) : "v"(b.y));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a2), "v"(b2));
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(u) : "v"(t));
            asm volatile("v_mov_b32 %0, %1" : "=v"(u2.x) : "v"(u.x)); asm volatile("v_mov_b32 %0, %1" : "=v"(u2.y) : "v"(u.y));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "s"(s), "v"(u2));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b2));
        } else if (MODE & 4) {
            // write-after-read: the 32-bit instructions right behind a packed one overwrite ITS source registers (the compiled victims
            // do: v_pk_mul_f32 v[4:5], v[10:11], v[8:9] follows v_mov_b32 v8, v5 follows v_pk_mul_f32 .., v[4:5]); `junk` must never
            // reach the products
            const float junk = 1000.f + t0;
            asm volatile("v_mov_b32 v200, %[bx]\n\tv_mov_b32 v201, %[by]\n\t"
                         "v_pk_mul_f32 %[t], %[a], v[200:201] op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_mov_b32 v200, %[j]\n\tv_mov_b32 v201, %[j]"
                         : [t] "=&v"(t) : [a] "v"(a), [bx] "v"(b.x), [by] "v"(b.y), [j] "v"(junk) : "v200", "v201");
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(u) : "v"(t));
            asm volatile("v_mov_b32 v202, %[ux]\n\tv_mov_b32 v203, %[uy]\n\t"
                         "v_pk_mul_f32 %[a], %[s], v[202:203]\n\t"
                         "v_mov_b32 v202, %[j]\n\tv_mov_b32 v203, %[j]"
                         : [a] "=&v"(a) : [s] "s"(s), [ux] "v"(u.x), [uy] "v"(u.y), [j] "v"(junk) : "v202", "v203");
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                 // (a1 b0, a0 b1)
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(u) : "v"(t));  // (t0 - t1, t1 - t0)
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "s"(s), "v"(u));                                                // (s0 u0, s1 u1): SGPR pair
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        }
        ck = ck * 31u + (__float_as_uint(a.x) ^ (__float_as_uint(a.y) >> 3) ^ __float_as_uint(t.x) ^ __float_as_uint(u.y));
    }
    for (int it = 0; it < rounds; it++) {
        float p0, p1, u0, u1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(a1), "v"(b0));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(a0), "v"(b1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u0) : "v"(p0), "v"(p1));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u1) : "v"(p1), "v"(p0));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a0) : "s"(s.x), "v"(u0));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a1) : "s"(s.y), "v"(u1));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b1));
        cs = cs * 31u + (__float_as_uint(a0) ^ (__float_as_uint(a1) >> 3) ^ __float_as_uint(p0) ^ __float_as_uint(u1));
    }
    if (MODE & 2) {
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < 40; i++) { asm volatile("" : "+v"(pad[i])); z += pad[i][0] + pad[i][1] + pad[i][2] + pad[i][3]; }
        if (z == 12345.678f) ck ^= 1u;                           // keeps the pad registers live across both loops
    }
    const unsigned long long m = __ballot(ck != cs);
    if ((threadIdx.x & 63) == 0 && m) {
        atomicOr(flag, 1);
        atomicAdd(&lanes[0], (unsigned long long)__builtin_popcountll(m & 0xffffull));
        atomicAdd(&lanes[1], (unsigned long long)__builtin_popcountll(m >> 16));
    }
    sums[gid] = ck;
}

// per = words each victim thread wrote (1: victim A's checksum, 3: an output vector); lanes[0] / lanes[1] count the differing victim
// THREADS in lanes 0..15 / 16..63 of their wavefront
__global__ void __launch_bounds__(256) cmp_u32(const unsigned* __restrict__ a, const unsigned* __restrict__ g, size_t n, int per, int* __restrict__ flag,
                                             unsigned long long* __restrict__ lanes)
{
    bool d = false;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n / per; t += (size_t)gridDim.x * 256) {
        bool dt = false;
        for (int j = 0; j < per; j++) dt |= a[t * per + j] != g[t * per + j];
        if (dt && lanes) atomicAdd(&lanes[(t & 63) < 16 ? 0 : 1], 1ull);
        d |= dt;
    }
    if (__ballot(d) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---- aggressors -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) aggr_f16(float* out, int iters)
{
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    f4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(64) aggr_f32(float* out, int iters)
{
    float a = 0.01f * threadIdx.x, b = 0.5f - 0.01f * threadIdx.x;
    f4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef void (*victim_fn)(const float*, const float*, float*, int);
struct Victim { const char* name; victim_fn pk, sc; float* gold; };

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 10000;
    const int vwg = 1024;                                  // victim workgroups of 256 threads: 4096 wavefronts, 4 per SIMD
    const int awaves = 256 * 4 * 2;                        // aggressor: 2 single-wave workgroups per SIMD
    hipStream_t sv, sa;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, hi));      // the keypoint stream of the product is the high-priority one
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));
    std::vector<float> hsrc((size_t)64 * VB_PTS * VB_K * 8), hw(2 * VB_COUT * 4);
    unsigned r = 12345u;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (float)((r >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; };
    for (auto& v : hsrc) v = rnd();
    for (size_t i = 3; i < hsrc.size(); i += 8) hsrc[i] = 1.f;
    for (auto& v : hw) v = rnd() * 2.f;
    Victim V[] = { { "B  LDS-fed, full arithmetic", victim_b_pk, victim_b_sc, nullptr }, { "C  register-fed, full arithmetic", victim_c_pk, victim_c_sc, nullptr },
                   { "C1 register-fed, no IEEE division", victim_c1_pk, victim_c1_sc, nullptr }, { "C2 register-fed, fp32 accumulation (no fp64)", victim_c2_pk, victim_c2_sc, nullptr },
                   { "C4 register-fed, no square root", victim_c4_pk, victim_c4_sc, nullptr }, { "C7 register-fed, mul / add only", victim_c7_pk, victim_c7_sc, nullptr } };
    const int NV = sizeof(V) / sizeof(V[0]), NA = 6, NF = NA + 2 * NV;
    float *dsrc, *dw, *ob, *aout;
    f2* dsarg;
    unsigned *dsums, *gsums;
    unsigned long long* dlanes;
    int* dflags;
    const size_t on = (size_t)vwg * 256 * 3, na = (size_t)vwg * 256;
    CK(hipMalloc(&dsrc, hsrc.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&ob, on * 4));
    for (int v = 0; v < NV; v++) CK(hipMalloc(&V[v].gold, on * 4));
    CK(hipMalloc(&aout, (size_t)awaves * 64 * 4)); CK(hipMalloc(&dsarg, sizeof(f2)));
    CK(hipMalloc(&dsums, na * 4)); CK(hipMalloc(&gsums, na * 4)); CK(hipMalloc(&dlanes, 16 * (NA + NV))); CK(hipMalloc(&dflags, (size_t)NF * launches * 4));
    CK(hipMemcpy(dsrc, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const f2 hs = { 0.3125f, -0.28125f };
    CK(hipMemcpy(dsarg, &hs, sizeof(hs), hipMemcpyHostToDevice));
    // goldens: the twins WITHOUT packed instructions, alone on the chip
    CK(hipMemset(dflags, 0, (size_t)NF * launches * 4)); CK(hipMemset(dlanes, 0, 16 * (NA + NV)));
    victim_a<0><<<vwg, 256, 0, sv>>>(dsarg, dflags, dlanes, gsums, 256);
    for (int v = 0; v < NV; v++) V[v].sc<<<vwg, 256, 0, sv>>>(dsrc, dw, V[v].gold, 0);
    CK(hipStreamSynchronize(sv));
    {   // the goldens are not degenerate
        std::vector<float> t(on); CK(hipMemcpy(t.data(), V[0].gold, on * 4, hipMemcpyDeviceToHost));
        double sum = 0; int nan = 0; for (float v : t) { if (v != v) nan++; else sum += v < 0 ? -v : v; }
        std::vector<unsigned> u(na); CK(hipMemcpy(u.data(), gsums, na * 4, hipMemcpyDeviceToHost));
        printf("golden: victim B mean |out| %.4f, NaNs %d; victim A checksum[0..2] %08x %08x %08x\n", sum / on, nan, u[0], u[1], u[2]);
    }
    int it16 = 20000, it32 = 20000;
    for (int which = 0; which < 2; which++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, sa));
        if (which == 0) aggr_f16<<<awaves, 64, 0, sa>>>(aout, 20000); else aggr_f32<<<awaves, 64, 0, sa>>>(aout, 20000);
        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        (which == 0 ? it16 : it32) = (int)(20000 * 2.0f / ms) + 1;
        printf("aggressor %s: 20000 iterations x 8 MFMAs on %d wavefronts = %.2f ms -> %d iterations per launch (2 ms)\n", which == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x4_f32", awaves, ms, which == 0 ? it16 : it32);
    }
    const char* anames[3] = { "none", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32" };
    const int batch = 25;
    std::vector<int> hf((size_t)NF * launches);
    for (int ag = 0; ag < 3; ag++) {
        CK(hipMemset(dflags, 0, (size_t)NF * launches * 4)); CK(hipMemset(dlanes, 0, 16 * (NA + NV)));
        CK(hipDeviceSynchronize());
        double aggr_ms = 0, victim_ms = 0;
        for (int l0 = 0; l0 < launches; l0 += batch) {
            hipEvent_t a0, a1, v0, v1;
            CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
            CK(hipEventRecord(a0, sa));
            for (int rep = 0; rep < 12 && ag; rep++) {           // ~24 ms of aggressor per victim batch; both streams drain before the next batch
                if (ag == 1) aggr_f16<<<awaves, 64, 0, sa>>>(aout, it16); else aggr_f32<<<awaves, 64, 0, sa>>>(aout, it32);
            }
            CK(hipEventRecord(a1, sa));
            CK(hipEventRecord(v0, sv));
            for (int l = l0; l < l0 + batch && l < launches; l++) {
                int* f = dflags + (size_t)NF * l;
                victim_a<0><<<vwg, 256, 0, sv>>>(dsarg, f + 0, dlanes + 0, dsums, 256);
                victim_a<1><<<vwg, 256, 0, sv>>>(dsarg, f + 1, dlanes + 2, dsums, 256);
                victim_a<2><<<vwg, 256, 0, sv>>>(dsarg, f + 2, dlanes + 4, dsums, 256);
                victim_a<3><<<vwg, 256, 0, sv>>>(dsarg, f + 3, dlanes + 6, dsums, 256);
                victim_a<4><<<vwg, 256, 0, sv>>>(dsarg, f + 4, dlanes + 8, dsums, 256);
                victim_a<6><<<vwg, 256, 0, sv>>>(dsarg, f + 5, dlanes + 10, dsums, 256);
                for (int v = 0; v < NV; v++) {
                    V[v].pk<<<vwg, 256, 0, sv>>>(dsrc, dw, ob, 0);
                    cmp_u32<<<256, 256, 0, sv>>>((const unsigned*)ob, (const unsigned*)V[v].gold, on, 3, f + NA + 2 * v, dlanes + 2 * (NA + v));
                    V[v].sc<<<vwg, 256, 0, sv>>>(dsrc, dw, ob, 0);
                    cmp_u32<<<256, 256, 0, sv>>>((const unsigned*)ob, (const unsigned*)V[v].gold, on, 3, f + NA + 1 + 2 * v, nullptr);
                }
            }
            CK(hipEventRecord(v1, sv));
            CK(hipEventSynchronize(v1)); CK(hipEventSynchronize(a1));
            float va, aa;
            CK(hipEventElapsedTime(&va, v0, v1)); CK(hipEventElapsedTime(&aa, a0, a1));
            victim_ms += va; aggr_ms += aa;
            CK(hipEventDestroy(a0)); CK(hipEventDestroy(a1)); CK(hipEventDestroy(v0)); CK(hipEventDestroy(v1));
        }
        CK(hipMemcpy(hf.data(), dflags, hf.size() * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> hl(2 * (NA + NV)); CK(hipMemcpy(hl.data(), dlanes, 16 * (NA + NV), hipMemcpyDeviceToHost));
        std::vector<long> c(NF, 0);
        for (int l = 0; l < launches; l++) for (int k = 0; k < NF; k++) c[k] += hf[(size_t)NF * l + k] != 0;
        printf("aggressor %-24s %d launches of every victim (victim stream %.0f ms, aggressor stream busy %.0f ms)\n", anames[ag], launches, victim_ms, aggr_ms);
        static const char* an[6] = { "A0 asm v_pk_mul/add_f32 chain, register-only", "A1 ... operand pairs written by v_mov_b32 halves", "A2 ... A0 with 160 more live VGPRs",
                                     "A3 ... A1 with 160 more live VGPRs", "A4 ... sources overwritten right behind the pk op", "A6 ... A4 with 160 more live VGPRs" };
        for (int m = 0; m < NA; m++)
            printf("   %-48s: bad vs in-kernel scalar twin %ld (differing lanes 0-15: %llu, lanes 16-63: %llu)\n", an[m], c[m], hl[2 * m], hl[2 * m + 1]);
        for (int v = 0; v < NV; v++)
            printf("   %-48s: packed build bad %ld (differing threads in lanes 0-15: %llu, lanes 16-63: %llu) | build without packed ops bad %ld\n", V[v].name, c[NA + 2 * v], hl[2 * (NA + v)],
                   hl[2 * (NA + v) + 1], c[NA + 1 + 2 * v]);
        fflush(stdout);
    }
    return 0;
}
#endif
