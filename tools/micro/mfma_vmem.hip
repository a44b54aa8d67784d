// Development aid (round 3): what does a global_load_dwordx4 cost beside fp32 MFMAs?  Per loop iteration: 48 MFMAs in four
// groups of 12, 12 v_add_f32 after each group, and NV 1-KB wave loads (L1/L2-resident 64 KB table) spread over the groups;
// the data of an iteration is waited for one iteration later (vmcnt(NV)).  Two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_vmem mfma_vmem.hip ; run: ./mfma_vmem
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int SADDR, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS == 128 ? 1 : 2) k_vmem(float* out, const float* tab, int iters)
{
    extern __shared__ float lds[];
    if (THREADS == 128) asm volatile("" ::: "v255", "a15");
    float a = 1.f + threadIdx.x * 1e-3f + lds[0] * 0.f, b = 2.f - threadIdx.x * 1e-3f;
    f4 acc[12];
    for (int i = 0; i < 12; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = a + i;
    f4 w[8];
    for (int i = 0; i < 8; i++) w[i] = (f4){ 0, 0, 0, 0 };
    const float* p = tab + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048;
    unsigned voff = ((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048) * 4;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 rsrc = { (unsigned)(size_t)tab, (unsigned)((size_t)tab >> 32) & 0xffff, 0x7fffffffu, 0x00027000u };
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    unsigned soff = __builtin_amdgcn_readfirstlane(iters & 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
#pragma unroll
            for (int i = 0; i < 12; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < 12; k++) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[k & 15]) : "v"(a), "v"(b));
#pragma unroll
            for (int l = 0; l < NV / 4; l++) {
                if (SADDR == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(w[g * (NV / 4) + l]) : "v"(voff), "s"(tab), "i"(l * 1024));
                else if (SADDR == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(w[g * (NV / 4) + l]) : "v"(voff), "s"(rsrc), "s"(soff), "i"(l * 1024));
                else if (SADDR == 3) asm volatile("s_mov_b32 %1, 0\n\tbuffer_load_dwordx4 %0, %2, %3, %1 offen offset:%4" : "=v"(w[g * (NV / 4) + l]), "+s"(soff) : "v"(voff), "s"(rsrc), "i"(l * 1024));
                else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(w[g * (NV / 4) + l]) : "v"(p), "i"(l * 1024));
            }
        }
        if (NV) asm volatile("s_waitcnt vmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    for (int i = 0; i < 8; i++) s += w[i][0] + w[i][3];
    for (int i = 0; i < 12; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NV, int SADDR, int THREADS>
static void run(float* out, const float* tab)
{
    const int iters = 10000, ncu = 256;
    (void)hipFuncSetAttribute((const void*)k_vmem<NV, SADDR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_vmem<NV, SADDR, THREADS><<<ncu * 2, THREADS, 80 * 1024>>>(out, tab, 10);
    (void)hipEventRecord(e0);
    k_vmem<NV, SADDR, THREADS><<<ncu * 2, THREADS, 80 * 1024>>>(out, tab, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 48 * (THREADS / 128));
    static const char* form[4] = { "global, 64-bit VGPR address", "global, SGPR base + VGPR offset", "buffer, offen + SGPR soffset", "buffer, offen + soffset set by s_mov just before" };
    printf("48 MFMA + 48 VALU + %d dwordx4 loads (%s) per iteration, %d waves/SIMD: %.2f cycles per MFMA (%.0f %%)\n", NV, form[SADDR], THREADS / 128, cyc, 100.0 * 32 / cyc);
}

int main()
{
    float* out; (void)hipMalloc(&out, 1 << 22);
    float* tab; (void)hipMalloc(&tab, 1 << 20); (void)hipMemset(tab, 0, 1 << 20);
    run<0, 0, 256>(out, tab); run<8, 0, 256>(out, tab); run<8, 1, 256>(out, tab); run<8, 2, 256>(out, tab); run<8, 3, 256>(out, tab);
    run<0, 0, 128>(out, tab); run<8, 0, 128>(out, tab); run<8, 1, 128>(out, tab); run<8, 2, 128>(out, tab); run<8, 3, 128>(out, tab);
    return 0;
}
