// issue rate of the f32 matrix instructions on gfx950: cycles per instruction with 1 / 4 waves per SIMD
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_rate scripts/dev/mfma_rate.hip && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND> __global__ void rate(float *out, long long *ticks, int iters)
{
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 acc[6];
    f32x16 big = {0};
    for (int i = 0; i < 6; i++) acc[i] = (f32x4){0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 24; r++) big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 6; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    s += big[0] + big[5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {       // whole workgroup: first start to last end
        atomicMin((unsigned long long *)&ticks[0], (unsigned long long)t0);
        atomicMax((unsigned long long *)&ticks[1], (unsigned long long)t1);
    }
}

int main()
{
    float *out; long long *ticks, h[2];
    hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&ticks, 16);
    const int iters = 1000;
    const char *names[3] = {"4x4x1 (16 blocks)", "16x16x4", "32x32x2"};
    for (int kind = 0; kind < 3; kind++)
        for (int threads = 256; threads <= 1024; threads *= 2) {
            for (int rep = 0; rep < 2; rep++) {
                h[0] = 0x7fffffffffffffffLL; h[1] = 0;
                hipMemcpy(ticks, h, 16, hipMemcpyHostToDevice);
                if (kind == 0) rate<0><<<1, threads>>>(out, ticks, iters);
                else if (kind == 1) rate<1><<<1, threads>>>(out, ticks, iters);
                else rate<2><<<1, threads>>>(out, ticks, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            const double dt = (double)(h[1] - h[0]);
            printf("%-18s %d waves/SIMD: %.2f ticks per instruction per wave, %.2f per SIMD instruction\n", names[kind], threads / 256,
                   dt / (iters * 24.0), dt / (iters * 24.0) / (threads / 256));
        }
    return 0;
}
