// Development aid (round 3): what does a block of VALU work cost beside fp32 MFMAs when it is GROUPED instead of interleaved?
// v_mfma_f32_16x16x4_f32 runs on the f32 vector lanes (MI355X_MICROARCH: "runs at the f32 VECTOR rate"), so VALU work does
// not hide behind it; tools/micro/mfma_coissue showed a fixed cost per MFMA -> VALU -> MFMA alternation on top of the VALU
// issue slots.  Here one loop iteration = G MFMAs back to back, then G*K/2 VALU adds (K2 = VALU per two MFMAs), then L
// ds_read_b64, for 1 wavefront per SIMD (two 128-thread workgroups per CU, 512 registers each) and 2 per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_group mfma_group.hip ; run: ./mfma_group
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// THREADS = 128: two waves per workgroup, two workgroups per CU (LDS 80 KB each) -> one wave per SIMD
// THREADS = 256: four waves per workgroup, two workgroups per CU -> two waves per SIMD
template <int THREADS, int G, int K2, int L, int INDEP>
__global__ void __launch_bounds__(THREADS, THREADS == 128 ? 1 : 2) k_group(float* out, int iters, int* simd_ids)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = i * 1e-3f;
    __syncthreads();
    // more than 256 registers (VGPR + AGPR): a SIMD then holds ONE such wavefront, so the dispatcher spreads the four waves of
    // the two co-resident workgroups over the four SIMDs (with a small register count it stacks two of them on one SIMD and
    // leaves another idle: measured, HW_REG_HW_ID)
    if (THREADS == 128) asm volatile("" ::: "v255", "a15");
    float a = 1.f + threadIdx.x * 1e-3f, b = 2.f - threadIdx.x * 1e-3f;
    f4 acc[24];
    for (int i = 0; i < 24; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = a + i;
    f2 d[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
    unsigned addr = (threadIdx.x & 63) * 8;
    constexpr int NV = G * K2 / 2;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < G; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 24]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < NV; k++) {
            if (INDEP) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[k & 15]) : "v"(a), "v"(b));          // no dependence at all
            else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k & 15]) : "v"(v[(k + 5) & 15]));
        }
#pragma unroll
        for (int l = 0; l < L; l++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[l & 3]) : "v"(addr), "i"((l & 7) * 512));
        if (L) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += d[i].x + d[i].y;
    for (int i = 0; i < 24; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (simd_ids && (threadIdx.x & 63) == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        simd_ids[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = (int)hwid;
    }
}

template <int THREADS, int G, int K2, int L, int INDEP>
static void run(float* out, int* ids)
{
    const int ncu = 256;
    size_t lds = 80 * 1024;
    (void)hipFuncSetAttribute((const void*)k_group<THREADS, G, K2, L, INDEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 20000 * 24 / G;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_group<THREADS, G, K2, L, INDEP><<<ncu * 2, THREADS, lds>>>(out, 10, nullptr);
    (void)hipEventRecord(e0);
    k_group<THREADS, G, K2, L, INDEP><<<ncu * 2, THREADS, lds>>>(out, iters, ids);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int waves_per_simd = THREADS / 128;
    double mfma_per_simd = (double)iters * G * waves_per_simd;
    double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    printf("waves/SIMD %d  group %2d MFMA + %2d VALU (%s, %.1f per MFMA) + %2d ds_read_b64 : %.1f cycles per MFMA at 2.4 GHz (%.0f %% of the matrix pipe)\n",
           waves_per_simd, G, G * K2 / 2, INDEP ? "indep" : "chain", K2 / 2.0, L, cyc, 100.0 * 32 / cyc);
    if (ids && THREADS == 128) {   // SIMD placement of the four waves of the two workgroups that share CU 0 (first dispatched)
        int h[8]; (void)hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
        printf("    HW_ID of the first 8 waves (wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]):");
        for (int i = 0; i < 8; i++) printf(" simd%d/cu%d/se%d", (h[i] >> 4) & 3, (h[i] >> 8) & 15, (h[i] >> 13) & 7);
        printf("\n");
    }
}

int main()
{
    float* out; (void)hipMalloc(&out, 1 << 22);
    int* ids; (void)hipMalloc(&ids, 1 << 16);
    // one wave per SIMD
    run<128, 24, 0, 0, 1>(out, ids);
    run<128, 1, 2, 0, 1>(out, nullptr); run<128, 4, 2, 0, 1>(out, nullptr); run<128, 24, 1, 0, 1>(out, nullptr);
    run<128, 24, 2, 0, 1>(out, nullptr); run<128, 24, 4, 0, 1>(out, nullptr);
    // two waves per SIMD
    run<256, 24, 0, 0, 1>(out, nullptr);
    run<256, 12, 1, 0, 1>(out, nullptr); run<256, 12, 2, 0, 1>(out, nullptr); run<256, 12, 3, 0, 1>(out, nullptr); run<256, 12, 4, 0, 1>(out, nullptr);
    run<256, 12, 5, 0, 1>(out, nullptr); run<256, 12, 6, 0, 1>(out, nullptr); run<256, 12, 8, 0, 1>(out, nullptr);
    run<256, 4, 2, 0, 1>(out, nullptr); run<256, 4, 3, 0, 1>(out, nullptr); run<256, 4, 4, 0, 1>(out, nullptr); run<256, 4, 6, 0, 1>(out, nullptr);
    run<256, 1, 2, 0, 1>(out, nullptr); run<256, 1, 4, 0, 1>(out, nullptr); run<256, 1, 6, 0, 1>(out, nullptr);
    run<256, 24, 2, 0, 1>(out, nullptr); run<256, 24, 4, 0, 1>(out, nullptr);
    run<256, 12, 2, 12, 1>(out, nullptr); run<256, 12, 4, 12, 1>(out, nullptr);
    return 0;
}
