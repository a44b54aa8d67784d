// Development aid: does VALU / LDS issue hide behind v_mfma_f32_16x16x4_f32 on gfx950?
// Each loop iteration issues 16 independent MFMAs; after every MFMA come K VALU adds (independent registers) and
// L ds_read_b64.  Reports cycles of wall time per MFMA per SIMD for 1 and 2 wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip ; run: ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int K, int L, int PK>
__global__ void __launch_bounds__(256, 2) k_mix(float* out, int iters)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-3f;
    __syncthreads();
    float a = 1.f + threadIdx.x * 1e-3f, b = 2.f - threadIdx.x * 1e-3f;
    f4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    float v[4] = { a, b, a + b, a - b };
    f2 p[2] = { { a, b }, { b, a } };
    f2 d[2] = { { 0, 0 }, { 0, 0 } };
    unsigned addr = (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 1]) : "v"(p[(k + 1) & 1]));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k & 3]) : "v"(v[(k + 1) & 3]));
            }
#pragma unroll
            for (int l = 0; l < L; l++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[l & 1]) : "v"(addr), "i"(l * 512));
        }
        if (L) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = v[0] + v[1] + v[2] + v[3] + p[0].x + p[1].y + d[0].x + d[1].y;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VALU issue rate alone: 64 independent v_add_f32 / v_fma_f32 / v_pk_add_f32 per loop iteration, W wavefronts per SIMD
template <int KIND>
__global__ void __launch_bounds__(256, 2) k_valu(float* out, int iters)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
    f2 p[4] = { { v[0], v[1] }, { v[2], v[3] }, { v[4], v[5] }, { v[6], v[7] } };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(v[(k + 3) & 7]));
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 7]) : "v"(v[(k + 3) & 7]));
            else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 3]) : "v"(p[(k + 1) & 3]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run_valu(float* out, int wgs_per_cu)
{
    const int iters = 4000, ncu = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_valu<KIND><<<ncu * wgs_per_cu, 256>>>(out, 10);
    (void)hipEventRecord(e0);
    k_valu<KIND><<<ncu * wgs_per_cu, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)iters * 64 * wgs_per_cu;
    printf("%s alone, waves/SIMD %d : %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
           KIND == 0 ? "v_add_f32" : KIND == 1 ? "v_fma_f32" : "v_pk_add_f32", wgs_per_cu, ms * 1e-3 * 2.4e9 / per_simd);
}

template <int K, int L, int PK>
static void run(float* out, int wgs_per_cu)
{
    const int iters = 4000, ncu = 256;
    size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 32 * 1024;
    (void)hipFuncSetAttribute((const void*)k_mix<K, L, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_mix<K, L, PK><<<ncu * wgs_per_cu, 256, lds>>>(out, 10);
    (void)hipEventRecord(e0);
    k_mix<K, L, PK><<<ncu * wgs_per_cu, 256, lds>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 16 * wgs_per_cu;
    double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    printf("VALU/MFMA %d (%s)  ds_read_b64/MFMA %d  waves/SIMD %d : %.1f cycles per MFMA at 2.4 GHz (%.0f %% of the matrix pipe)\n", K,
           PK ? "pk" : "f32", L, wgs_per_cu, cyc, 100.0 * 32 / cyc);
}

int main()
{
    float* out; (void)hipMalloc(&out, 1 << 22);
    for (int w = 1; w <= 4; w *= 2) { run_valu<0>(out, w); run_valu<1>(out, w); run_valu<2>(out, w); }
    for (int w = 1; w <= 2; w++) {
        run<0, 0, 0>(out, w); run<1, 0, 0>(out, w); run<2, 0, 0>(out, w); run<3, 0, 0>(out, w); run<4, 0, 0>(out, w); run<6, 0, 0>(out, w);
        run<2, 0, 1>(out, w); run<4, 0, 1>(out, w);
        run<0, 1, 0>(out, w); run<0, 2, 0>(out, w); run<2, 1, 0>(out, w); run<2, 1, 1>(out, w);
    }
    return 0;
}
