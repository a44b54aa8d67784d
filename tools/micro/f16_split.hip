// Development aid (round 4): numerics of the split-f16 form of an fp32 dot product on the f16 matrix pipe.
//   x = hi + 2^-11 lo',  hi = f16(x) (RNE),  lo' = f16((x - hi) * 2^11)   (|x - hi - 2^-11 lo'| <= 2^-23 |x|)
//   sum x w  ~  [hi_x hi_w]  +  2^-11 [hi_x lo'_w + lo'_x hi_w]           (two fp32 accumulators, three MFMAs per k-step)
// Questions answered on the hardware: (1) does v_mfma_f32_16x16x32_f16 keep f16 subnormal INPUTS (it does or it does not:
// printed), (2) how far is the three-product form from the float64 dot product over K = 1152 (a 128-channel 3x3 layer), next
// to v_mfma_f32_16x16x4_f32 and to a plain fp32 FMA chain, (3) the bf16 x 6 form of the same for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o f16_split f16_split.hip ; run: ./f16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define K 1152

// A[16][K] row-major, B[K][16] given as Bt[16][K]; D[16][16]
__global__ void k_dot(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ Dsplit, float* __restrict__ Df32,
                      float* __restrict__ Dbf6, float* __restrict__ Dsplit1acc)
{
    const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
    f4 accm = { 0, 0, 0, 0 }, accc = { 0, 0, 0, 0 }, acc32 = { 0, 0, 0, 0 }, accb = { 0, 0, 0, 0 }, acc1 = { 0, 0, 0, 0 };
    for (int k0 = 0; k0 < K; k0 += 32) {
        h8 ah, al, bh, bl, alu, blu;
        b8 a0, a1, a2, b0, b1, b2;
        for (int i = 0; i < 8; i++) {
            const float a = A[r * K + k0 + kg * 8 + i], b = Bt[r * K + k0 + kg * 8 + i];
            ah[i] = (_Float16)a; al[i] = (_Float16)((a - (float)ah[i]) * 2048.f); alu[i] = (_Float16)(a - (float)ah[i]);
            bh[i] = (_Float16)b; bl[i] = (_Float16)((b - (float)bh[i]) * 2048.f); blu[i] = (_Float16)(b - (float)bh[i]);
            a0[i] = (__bf16)a; float ra = a - (float)a0[i]; a1[i] = (__bf16)ra; ra -= (float)a1[i]; a2[i] = (__bf16)ra;
            b0[i] = (__bf16)b; float rb = b - (float)b0[i]; b1[i] = (__bf16)rb; rb -= (float)b1[i]; b2[i] = (__bf16)rb;
        }
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accm, 0, 0, 0);
        accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accc, 0, 0, 0);
        accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accc, 0, 0, 0);
        // unscaled low parts into ONE accumulator (what a single-accumulator form would give)
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(alu, bh, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, blu, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc1, 0, 0, 0);
        // bf16 x 6: smallest terms first
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, accb, 0, 0, 0);
        for (int s = 0; s < 8; s++) {
            const float a = A[r * K + k0 + s * 4 + kg], b = Bt[r * K + k0 + s * 4 + kg];
            acc32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc32, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; i++) {
        const int m = kg * 4 + i, n = r;     // D[m][n]
        Dsplit[m * 16 + n] = accm[i] + accc[i] * (1.f / 2048.f);
        Df32[m * 16 + n] = acc32[i];
        Dbf6[m * 16 + n] = accb[i];
        Dsplit1acc[m * 16 + n] = acc1[i];
    }
}

__global__ void k_denorm(float* out)
{
    const int lane = threadIdx.x;
    h8 a, b, c;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)9.5367431640625e-07f /* 2^-20: subnormal */; b[i] = (_Float16)1.0f; c[i] = (_Float16)6.103515625e-05f /* 2^-14 */; }
    f4 z = { 0, 0, 0, 0 };
    f4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);    // 32 * 2^-20 = 2^-15 if subnormal inputs are kept
    f4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c, c, z, 0, 0, 0);    // 32 * 2^-28 = 2^-23
    f4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, z, 0, 0, 0);    // 32 * 2^-40 = 2^-35
    if (lane == 0) { out[0] = d1[0]; out[1] = d2[0]; out[2] = d3[0]; }
    // conversion: f32 -> f16 of a value in the subnormal range
    volatile float tiny = 3.0e-6f;
    out[3] = (float)(_Float16)tiny;
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main()
{
    float* dout; hipMalloc(&dout, 64);
    k_denorm<<<1, 64>>>(dout);
    float h[4]; hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
    printf("subnormal f16 inputs: 32 x (2^-20 x 1) = %.6e (kept: %.6e)   32 x (2^-14)^2 = %.6e (%.6e)   32 x (2^-20)^2 = %.6e (%.6e)   f16(3e-6) = %.6e\n",
           h[0], ldexp(1.0, -15), h[1], ldexp(1.0, -23), h[2], ldexp(1.0, -35), h[3]);
    for (int variant = 0; variant < 3; variant++) {
        srand(1 + variant);
        std::vector<float> A(16 * K), Bt(16 * K);
        for (auto& v : A) { double x = nrand(); v = variant == 2 ? (float)x : (float)(x > 0 ? x : 0); }          // post-ReLU-like | signed
        for (auto& v : Bt) v = (float)(nrand() * (variant == 1 ? 3e-4 : 0.05));                                // weights | tiny weights
        float *dA, *dB, *d0, *d1, *d2, *d3;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024); hipMalloc(&d3, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        k_dot<<<1, 64>>>(dA, dB, d0, d1, d2, d3);
        float s[256], f[256], b6[256], s1[256];
        hipMemcpy(s, d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(f, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(b6, d2, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(s1, d3, 1024, hipMemcpyDeviceToHost);
        double es = 0, ef = 0, ec = 0, eb = 0, e1 = 0, scale = 0, rs = 0, rf = 0;
        for (int m = 0; m < 16; m++)
            for (int n = 0; n < 16; n++) {
                double ref = 0; float chain = 0;
                for (int k = 0; k < K; k++) { ref += (double)A[m * K + k] * (double)Bt[n * K + k]; chain = fmaf(A[m * K + k], Bt[n * K + k], chain); }
                scale = fmax(scale, fabs(ref));
                es = fmax(es, fabs(s[m * 16 + n] - ref)); ef = fmax(ef, fabs(f[m * 16 + n] - ref)); ec = fmax(ec, fabs(chain - ref));
                eb = fmax(eb, fabs(b6[m * 16 + n] - ref)); e1 = fmax(e1, fabs(s1[m * 16 + n] - ref));
                rs += (s[m * 16 + n] - ref) * (s[m * 16 + n] - ref); rf += (f[m * 16 + n] - ref) * (f[m * 16 + n] - ref);
            }
        printf("variant %d (K = %d, scale %.3g): max err / scale   f16x3 (2 acc, scaled lo) %.3e   f16x3 (1 acc, unscaled lo) %.3e   bf16x6 %.3e   "
               "mfma f32 %.3e   fmaf chain %.3e   | rms f16x3 %.3e  f32 %.3e\n",
               variant, K, scale, es / scale, e1 / scale, eb / scale, ef / scale, ec / scale, sqrt(rs / 256) / scale, sqrt(rf / 256) / scale);
    }
    return 0;
}
