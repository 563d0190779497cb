// micro test: lane layout of v_mfma_f32_4x4x1_16b_f32 (asymmetric data)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)
{
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}
int main()
{
    float ha[64], hb[64], hd[256];
    // hypothesis: lane = 4*block + i (A row i) / 4*block + j (B col j); D[block][i][j] in lane 4*block+j, reg i
    for (int l = 0; l < 64; l++) { ha[l] = 1.0f + (l % 4) + 10.0f * (l / 4); hb[l] = 0.5f + 2.0f * (l % 4) + 100.0f * (l / 4); }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; blk++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float want = ha[4 * blk + i] * hb[4 * blk + j];
                float got = hd[(4 * blk + j) * 4 + i];
                if (want != got) { if (bad < 8) printf("blk %d i %d j %d want %g got %g\n", blk, i, j, want, got); bad++; }
            }
    printf("layout hypothesis %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    return 0;
}
