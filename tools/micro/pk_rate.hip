// Development aid (round 3): issue cost of packed fp32 VALU instructions against plain ones, no MFMAs around.
// One loop iteration = 32 independent instructions of one kind; 8 wavefronts per SIMD so that latency does not show.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters)
{
    f2 v[16];
    for (int i = 0; i < 16; i++) v[i] = (f2){ 1.f + threadIdx.x * 1e-3f + i, 2.f - i * 1e-2f };
    f2 a = { 1.0001f, 0.9999f };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(a.x));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i].x) : "v"(a.x));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(a.x));
            }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, float* out)
{
    const int iters = 20000, blocks = 256 * 8;       // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_rate<KIND><<<blocks, 256>>>(out, 10);
    (void)hipEventRecord(e0);
    k_rate<KIND><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 32 * 8;       // 8 wavefronts per SIMD
    printf("%-14s %.2f cycles per wavefront instruction at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 / inst_per_simd);
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_add_f32", out); run<5>("v_mul_f32", out); run<4>("v_fma_f32", out);
    run<1>("v_pk_add_f32", out); run<2>("v_pk_mul_f32", out); run<3>("v_pk_fma_f32", out);
    return 0;
}
