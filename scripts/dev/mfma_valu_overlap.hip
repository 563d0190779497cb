// Do vector and matrix instructions of DIFFERENT waves on one SIMD overlap on gfx950?  One workgroup of 8 waves (2 per
// SIMD): waves 0 - 3 run a matrix-instruction loop, waves 4 - 7 a packed-fma loop; each alone, then together.
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_valu_overlap scripts/dev/mfma_valu_overlap.hip && gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// mode bit 0: waves 0-3 multiply; bit 1: waves 4-7 run vector fma
template <int KIND> __global__ void overlap(float *out, long long *ticks, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 acc[6];
    f32x16 big = {0};
    v2f v[8];
    for (int i = 0; i < 6; i++) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 8; i++) v[i] = (v2f){a + i, b - i};
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; it++) {
                if constexpr (KIND == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);      // 24 x 8 cycles
                } else if constexpr (KIND == 1) {
#pragma unroll
                    for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);          // 6 x 32 cycles
                } else {
#pragma unroll
                    for (int r = 0; r < 3; r++) big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);              // 3 x 64 cycles
                }
            }
    } else if (mode & 2) {
        const v2f m = {1.0001f, 0.9999f}, c = {0.001f, -0.001f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_elementwise_fma(v[i], m, c);      // 48 x 4 cycles
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 6; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; i++) s += v[i].x + v[i].y;
    s += big[0] + big[5];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { ticks[2 * wave] = t0; ticks[2 * wave + 1] = t1; }
}

int main()
{
    float *out; long long *ticks, h[16];
    hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&ticks, 16 * 8);
    const int iters = 2000;
    const char *names[3] = {"4x4x1 (16 blocks)", "16x16x4", "32x32x2"};
    for (int kind = 0; kind < 3; kind++)
        for (int mode = 1; mode <= 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) overlap<0><<<1, 512>>>(out, ticks, iters, mode);
                else if (kind == 1) overlap<1><<<1, 512>>>(out, ticks, iters, mode);
                else overlap<2><<<1, 512>>>(out, ticks, iters, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, 16 * 8, hipMemcpyDeviceToHost);
            double mm = 0, vv = 0;
            for (int w = 0; w < 4; w++) { mm += (double)(h[2 * w + 1] - h[2 * w]) / 4; vv += (double)(h[2 * w + 9] - h[2 * w + 8]) / 4; }
            printf("%-18s %-28s matrix waves %8.1f ticks/iteration (192 cycles of matrix work), vector waves %8.1f (192 cycles of packed fma)\n",
                   names[kind], mode == 1 ? "matrix alone" : mode == 2 ? "vector alone" : "matrix + vector together", mm / iters, vv / iters);
        }
    return 0;
}
