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
    return 0;
}
