// do operand loads overlap with f32 matrix instructions on gfx950?  One workgroup of 16 waves on one CU; a wave either
// multiplies (24 v_mfma_f32_4x4x1 per iteration), or loads (3 KB per iteration: global dwordx4 from an L1-resident window,
// global dword x 12, or ds_read_b128 from LDS), or does both (3 loads, 24 multiplies, wait for the loads).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl scripts/dev/mfma_vmem_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mm(f32x4 (&acc)[6], float a, float b)
{
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
}

// lmask / mmask: bit w set = wave w loads / multiplies;  lkind 0 global dwordx4, 1 global dword, 2 LDS b128,
// 3 global dwordx4 with the moving part of the address in scalar registers (no vector ALU instruction on the load path)
__global__ __launch_bounds__(1024) void ovl(const float *src, float *out, long long *ticks, int iters, int lmask, int mmask, int lkind)
{
    __shared__ float lds[16 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) lds[i] = i;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 acc[6];
    for (int i = 0; i < 6; i++) acc[i] = (f32x4){0, 0, 0, 0};
    const float *p = src + lane * 4;            // 16 KB window: L1 hits after the first touch
    const float *p1 = src + lane;
    f32x4 x0 = {0, 0, 0, 0}, x1 = x0, x2 = x0;
    float y[12] = {0};
    const bool do_load = (lmask >> wave) & 1, do_mm = (mmask >> wave) & 1;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (do_load) {
            if (lkind == 0) {
                const float *q = p + ((it * 3) & 15) * 256;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x0) : "v"(q));
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x1) : "v"(q + 256));
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x2) : "v"(q + 512));
            } else if (lkind == 1) {
                const float *q = p1 + ((it * 12) & 63) * 64;
#pragma unroll
                for (int k = 0; k < 12; k++) asm volatile("global_load_dword %0, %1, off" : "=v"(y[k]) : "v"(q + 64 * k));
            } else if (lkind == 3) {
                const float *q = src + ((it * 3) & 15) * 256;
                const unsigned lo = lane * 16;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x0) : "v"(lo), "s"(q));
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(x1) : "v"(lo), "s"(q));
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(x2) : "v"(lo), "s"(q));
            } else {
                const unsigned q = (unsigned)(size_t)(lds + lane * 4 + ((it * 3) & 15) * 256) ;
                asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(q));
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(x1) : "v"(q));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(x2) : "v"(q));
            }
        }
        if (do_mm) mm(acc, a, b);
        if (do_load) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2));
#pragma unroll
            for (int k = 0; k < 12; k++) asm volatile("" :: "v"(y[k]));
        }
    }
    const long long t1 = clock64();
    float r = x0.x + x1.y + x2.z + y[0] + y[11];
    for (int i = 0; i < 6; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[threadIdx.x] = r;
    if (lane == 0) {
        atomicMin((unsigned long long *)&ticks[0], (unsigned long long)t0);
        atomicMax((unsigned long long *)&ticks[1], (unsigned long long)t1);
    }
}

int main()
{
    float *src, *out; long long *ticks, h[2];
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&ticks, 16);
    const int iters = 2000;
    struct Case { const char *name; int lmask, mmask; } cases[] = {
        {"all 16 waves load", 0xffff, 0},
        {"all 16 waves multiply", 0, 0xffff},
        {"all 16 waves load + multiply", 0xffff, 0xffff},
        {"waves 8-15 load (2 per SIMD)", 0xff00, 0},
        {"waves 0-7 multiply (2 per SIMD)", 0, 0x00ff},
        {"waves 0-7 multiply, 8-15 load", 0xff00, 0x00ff},
        {"waves 0-7 load", 0x00ff, 0},
        {"waves 0-7 load, 8-15 multiply", 0x00ff, 0xff00},
        {"waves 0-3 multiply, 4-7 load", 0x00f0, 0x000f},
        {"waves 0-3 load, 4-7 multiply", 0x000f, 0x00f0},
        {"SIMDs 0,1 multiply (8 waves)", 0, 0x3333},
        {"SIMDs 2,3 load (8 waves)", 0xcccc, 0},
        {"SIMDs 0,1 multiply, SIMDs 2,3 load", 0xcccc, 0x3333},
    };
    const char *ln[4] = {"global dwordx4", "global dword x 12", "LDS b128", "global x4, saddr"};
    for (int lkind = 0; lkind < 4; lkind++)
        for (const Case &c : cases) {
            if (lkind > 0 && c.lmask == 0) continue;
            for (int rep = 0; rep < 2; rep++) {
                h[0] = 0x7fffffffffffffffLL; h[1] = 0;
                hipMemcpy(ticks, h, 16, hipMemcpyHostToDevice);
                ovl<<<1, 1024>>>(src, out, ticks, iters, c.lmask, c.mmask, lkind);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("%-18s %-38s %8.1f ticks per iteration\n", ln[lkind], c.name, (double)(h[1] - h[0]) / iters);
        }
    return 0;
}
