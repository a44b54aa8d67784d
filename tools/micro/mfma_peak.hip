// Development aid: sustained fp32 MFMA rate of the two tile shapes (no memory traffic).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) k_peak(float* out, int iters, float a0, float b0)
{
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    if (SHAPE == 16) {
        f4 acc[16];
        for (int i = 0; i < 16; i++) acc[i] = (f4){ 0, 0, 0, 0 };
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f16 acc[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

// dependent-issue rate: NACC independent accumulators used round-robin (NACC = 1: every MFMA waits for the previous one)
template <int NACC>
__global__ void __launch_bounds__(256, 1) k_chain(float* out, int iters, float a0, float b0)
{
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 16x16x4 MFMAs with the operand traffic of the conv kernel woven in: per 2 MFMAs one ds_read2st64_b32 whose
// results feed later MFMAs, plus address VALU; 72 MFMAs per "group" like <MT=9,NT=2>.
__global__ void __launch_bounds__(256, 2) k_mix(float* out, int iters, int stride, int rnd)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 18432; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = rnd ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 15) * 0.01f; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f4 acc[18];
    for (int i = 0; i < 18; i++) acc[i] = (f4){ 0, 0, 0, 0 };
    float a[2][36], b[8];
    for (int i = 0; i < 8; i++) b[i] = rnd ? lds[(lane * 37 + i * 101) % 18432] : 0.5f + i;
    const float* base = lds + (lane >> 4) * 144 + (lane & 15);
    for (int j = 0; j < 36; j++) a[0][j] = base[(j % 9) * 16 + (j / 9) * 576];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float* src = base + ((it * 2 + h) & 7) * stride;
#pragma unroll
            for (int j = 0; j < 36; j++) a[h ^ 1][j] = src[(j % 9) * 16 + (j / 9) * 576];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int t = 0; t < 9; t++)
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        acc[t * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][p * 9 + t], b[p * 2 + u], acc[t * 2 + u], 0, 0, 0);
#pragma unroll
            for (int i_ = 0; i_ < 36; i_++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x120, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 18; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, iters = 20000;
    for (int shape = 16; shape <= 32; shape += 16) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (shape == 16) k_peak<16><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
            else k_peak<32><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: 16 x 2048 flop (16x16x4) or 8 x 4096 flop (32x32x2) = 32768 flop
            double flop = (double)blocks * 4 * iters * 32768.0;
            printf("mfma_f32_%s: %.2f ms  %.1f TFLOP/s\n", shape == 16 ? "16x16x4" : "32x32x2", ms, flop / ms / 1e9);
        }
    }
    for (int nacc = 1; nacc <= 8; nacc *= 2) {
        const int it3 = 20000, blocks1 = 256;        // one workgroup of 4 wavefronts per CU: one wavefront per SIMD
        hipEventRecord(e0);
        if (nacc == 1) k_chain<1><<<blocks1, 256>>>(out, it3, 1.0f, 0.5f);
        else if (nacc == 2) k_chain<2><<<blocks1, 256>>>(out, it3, 1.0f, 0.5f);
        else if (nacc == 4) k_chain<4><<<blocks1, 256>>>(out, it3, 1.0f, 0.5f);
        else k_chain<8><<<blocks1, 256>>>(out, it3, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks1 * 4 * it3 * 16 * 2048.0;
        printf("one wavefront per SIMD, %d independent accumulators: %.2f ms  %.1f TFLOP/s\n", nacc, ms, flop / ms / 1e9);
    }
    hipFuncSetAttribute((const void*)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int rep = 0; rep < 6; rep++) {
        const int it2 = 2000;
        hipEventRecord(e0);
        k_mix<<<blocks, 256, 73728>>>(out, it2, 4, rep >= 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * it2 * 2 * 72 * 2048.0;
        printf("16x16x4 + woven LDS operand reads (36 per 72 MFMAs), %s data: %.2f ms  %.1f TFLOP/s\n", rep >= 3 ? "random" : "regular", ms, flop / ms / 1e9);
    }
    return 0;
}
