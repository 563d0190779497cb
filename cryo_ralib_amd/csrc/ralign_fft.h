// In-register small DFTs for gfx950 (64-wide waves): radix 2/4/8/16 on float2 arrays with
// compile-time indices, natural-order output.  SIGN = -1 forward (e^{-i..}), +1 inverse.
#pragma once
#include <hip/hip_runtime.h>

namespace ralign {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * (SIGN * i)
template <int SIGN> __device__ __forceinline__ float2 cmuli(float2 a)
{
    return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// a * e^{SIGN * 2 pi i * NUM / DEN} with compile-time angle
template <int SIGN, int NUM, int DEN> __device__ __forceinline__ float2 ctw(float2 a)
{
    constexpr int n = ((NUM % DEN) + DEN) % DEN;
    if constexpr (n == 0) return a;
    else if constexpr (4 * n == DEN) return cmuli<SIGN>(a);
    else if constexpr (2 * n == DEN) return make_float2(-a.x, -a.y);
    else if constexpr (4 * n == 3 * DEN) return cmuli<-SIGN>(a);
    else {
        constexpr double ang = 6.283185307179586476925286766559 * (double)n / (double)DEN;
        const float c = (float)__builtin_cos(ang);
        const float s = (float)(SIGN * __builtin_sin(ang));
        return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

template <int SIGN> __device__ __forceinline__ void dft2(float2 &a, float2 &b)
{
    float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

template <int SIGN> __device__ __forceinline__ void dft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3)
{
    float2 a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = cmuli<SIGN>(csub(v1, v3));
    v0 = cadd(a, c); v1 = cadd(b, d); v2 = csub(a, c); v3 = csub(b, d);
}

template <int SIGN, int R> struct Dft;
template <int SIGN> struct Dft<SIGN, 1> { static __device__ __forceinline__ void run(float2 *) {} };
template <int SIGN> struct Dft<SIGN, 2> {
    static __device__ __forceinline__ void run(float2 *v) { dft2<SIGN>(v[0], v[1]); }
};
template <int SIGN> struct Dft<SIGN, 4> {
    static __device__ __forceinline__ void run(float2 *v) { dft4<SIGN>(v[0], v[1], v[2], v[3]); }
};
template <int SIGN> struct Dft<SIGN, 8> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        // radix-2 DIT: evens / odds through dft4
        float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4<SIGN>(e0, e1, e2, e3);
        dft4<SIGN>(o0, o1, o2, o3);
        o1 = ctw<SIGN, 1, 8>(o1); o2 = ctw<SIGN, 2, 8>(o2); o3 = ctw<SIGN, 3, 8>(o3);
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
    }
};
template <int SIGN> struct Dft<SIGN, 16> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        // 4 x 4: y_b[c] = sum_a v[4a+b] W4^{ac};  X[c + 4e] = sum_b (y_b[c] W16^{bc}) W4^{be}
        float2 y[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            y[b][0] = v[b]; y[b][1] = v[4 + b]; y[b][2] = v[8 + b]; y[b][3] = v[12 + b];
            dft4<SIGN>(y[b][0], y[b][1], y[b][2], y[b][3]);
        }
        y[1][1] = ctw<SIGN, 1, 16>(y[1][1]); y[1][2] = ctw<SIGN, 2, 16>(y[1][2]); y[1][3] = ctw<SIGN, 3, 16>(y[1][3]);
        y[2][1] = ctw<SIGN, 2, 16>(y[2][1]); y[2][2] = ctw<SIGN, 4, 16>(y[2][2]); y[2][3] = ctw<SIGN, 6, 16>(y[2][3]);
        y[3][1] = ctw<SIGN, 3, 16>(y[3][1]); y[3][2] = ctw<SIGN, 6, 16>(y[3][2]); y[3][3] = ctw<SIGN, 9, 16>(y[3][3]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float2 t0 = y[0][c], t1 = y[1][c], t2 = y[2][c], t3 = y[3][c];
            dft4<SIGN>(t0, t1, t2, t3);
            v[c] = t0; v[c + 4] = t1; v[c + 8] = t2; v[c + 12] = t3;
        }
    }
};

}  // namespace ralign
