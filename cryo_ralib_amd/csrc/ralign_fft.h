// In-register small DFTs for gfx950 (64-wide waves): radix 2/4/8/16 on float2 arrays with
// compile-time indices, natural-order output.  SIGN = -1 forward (e^{-i..}), +1 inverse.
// Complex values are carried as 2-wide vectors so that the adds, multiplies and fused
// multiply-adds map onto the packed-f32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: one instruction per complex add, two per complex multiply).
#pragma once
#include <hip/hip_runtime.h>

namespace ralign {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f to_v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }

// The swaps and sign flips of complex arithmetic ride on the operand modifiers of the packed instructions (op_sel picks
// the half of each source that feeds the low / high result, neg_lo / neg_hi negate it): hipcc only folds broadcasts
// into op_sel and spends a v_mov + v_xor on every swap, hence the three one-instruction forms below.

// a + (SIGN * i) * b
template <int SIGN> __device__ __forceinline__ v2f vadd_rot(v2f a, v2f b)
{
    v2f r;
    if constexpr (SIGN > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// t + a.y * (i * b) = (t.x - a.y b.y, t.y + a.y b.x): the second half of a complex product (fused multiply-add)
__device__ __forceinline__ v2f vfma_rot_y(v2f a, v2f b, v2f t)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// t + (i * a) * s.x = (t.x - a.y s.x, t.y + a.x s.x); s is a wave-uniform constant (scalar register pair)
__device__ __forceinline__ v2f vfma_irot_s(v2f a, v2f s, v2f t)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(s), "v"(t));
    return r;
}

// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f vadd_conj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f vsub_conj(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a * b (complex): (a.x b.x, a.x b.y) + a.y * (-b.y, b.x)
__device__ __forceinline__ v2f vmul(v2f a, v2f b)
{
    return vfma_rot_y(a, b, a.xx * b);
}
// Real-FFT split step of one pair: X_k = E + W O and X_{H-k} = conj(E - W O) with E = (Z_k + conj Z_{H-k}) / 2,
// O = -i (Z_k - conj Z_{H-k}) / 2, W = w = e^{-2 pi i k / n}: seven packed instructions (the halves and the -i ride on the
// twiddle: wh = -i w / 2)
__device__ __forceinline__ void vsplit_pair(v2f zk, v2f zm, v2f w, v2f &xk, v2f &xm)
{
    const v2f e2 = vadd_conj(zk, zm), o2 = vsub_conj(zk, zm);
    const v2f hp = {0.5f, -0.5f};
    v2f wh, t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(wh) : "v"(w), "v"(hp));      // (w.y / 2, -w.x / 2)
    t = vmul(o2, wh);
    xk = __builtin_elementwise_fma(e2, (v2f){0.5f, 0.5f}, t);
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1]" : "=v"(xm) : "v"(e2), "v"(hp), "v"(t));          // (e2.x / 2 - t.x, t.y - e2.y / 2)
}

// a * (SIGN * i)
template <int SIGN> __device__ __forceinline__ v2f vmuli(v2f a)
{
    return SIGN > 0 ? (v2f){-a.y, a.x} : (v2f){a.y, -a.x};
}
// a * e^{SIGN * 2 pi i * NUM / DEN} with compile-time angle
template <int SIGN, int NUM, int DEN> __device__ __forceinline__ v2f vtw(v2f a)
{
    constexpr int n = ((NUM % DEN) + DEN) % DEN;
    if constexpr (n == 0) return a;
    else if constexpr (4 * n == DEN) return vmuli<SIGN>(a);
    else if constexpr (2 * n == DEN) return -a;
    else if constexpr (4 * n == 3 * DEN) return vmuli<-SIGN>(a);
    else {
        constexpr double ang = 6.283185307179586476925286766559 * (double)n / (double)DEN;
        const float c = (float)__builtin_cos(ang);
        const float s = (float)(SIGN * __builtin_sin(ang));
        return vfma_irot_s(a, (v2f){s, s}, a * (v2f){c, c});
    }
}

template <int SIGN> __device__ __forceinline__ void vdft4(v2f &v0, v2f &v1, v2f &v2, v2f &v3)
{
    const v2f a = v0 + v2, b = v0 - v2, c = v1 + v3, e = v1 - v3;
    v0 = a + c; v1 = vadd_rot<SIGN>(b, e); v2 = a - c; v3 = vadd_rot<-SIGN>(b, e);
}

// float2 front end used by the kernels
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return to_f2(to_v(a) + to_v(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return to_f2(to_v(a) - to_v(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return to_f2(vmul(to_v(a), to_v(b))); }
template <int SIGN> __device__ __forceinline__ float2 cmuli(float2 a) { return to_f2(vmuli<SIGN>(to_v(a))); }
template <int SIGN, int NUM, int DEN> __device__ __forceinline__ float2 ctw(float2 a) { return to_f2(vtw<SIGN, NUM, DEN>(to_v(a))); }

template <int SIGN> __device__ __forceinline__ void dft2(float2 &a, float2 &b)
{
    const v2f x = to_v(a), y = to_v(b);
    a = to_f2(x + y);
    b = to_f2(x - y);
}

template <int SIGN> __device__ __forceinline__ void dft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3)
{
    v2f a = to_v(v0), b = to_v(v1), c = to_v(v2), d = to_v(v3);
    vdft4<SIGN>(a, b, c, d);
    v0 = to_f2(a); v1 = to_f2(b); v2 = to_f2(c); v3 = to_f2(d);
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
        v2f e0 = to_v(v[0]), e1 = to_v(v[2]), e2 = to_v(v[4]), e3 = to_v(v[6]);
        v2f o0 = to_v(v[1]), o1 = to_v(v[3]), o2 = to_v(v[5]), o3 = to_v(v[7]);
        vdft4<SIGN>(e0, e1, e2, e3);
        vdft4<SIGN>(o0, o1, o2, o3);
        o1 = vtw<SIGN, 1, 8>(o1); o2 = vtw<SIGN, 2, 8>(o2); o3 = vtw<SIGN, 3, 8>(o3);
        v[0] = to_f2(e0 + o0); v[4] = to_f2(e0 - o0);
        v[1] = to_f2(e1 + o1); v[5] = to_f2(e1 - o1);
        v[2] = to_f2(e2 + o2); v[6] = to_f2(e2 - o2);
        v[3] = to_f2(e3 + o3); v[7] = to_f2(e3 - o3);
    }
};
template <int SIGN> struct Dft<SIGN, 16> {
    static __device__ __forceinline__ void run(float2 *v)
    {
        // 4 x 4: y_b[c] = sum_a v[4a+b] W4^{ac};  X[c + 4e] = sum_b (y_b[c] W16^{bc}) W4^{be}
        v2f y[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            y[b][0] = to_v(v[b]); y[b][1] = to_v(v[4 + b]); y[b][2] = to_v(v[8 + b]); y[b][3] = to_v(v[12 + b]);
            vdft4<SIGN>(y[b][0], y[b][1], y[b][2], y[b][3]);
        }
        y[1][1] = vtw<SIGN, 1, 16>(y[1][1]); y[1][2] = vtw<SIGN, 2, 16>(y[1][2]); y[1][3] = vtw<SIGN, 3, 16>(y[1][3]);
        y[2][1] = vtw<SIGN, 2, 16>(y[2][1]); y[2][2] = vtw<SIGN, 4, 16>(y[2][2]); y[2][3] = vtw<SIGN, 6, 16>(y[2][3]);
        y[3][1] = vtw<SIGN, 3, 16>(y[3][1]); y[3][2] = vtw<SIGN, 6, 16>(y[3][2]); y[3][3] = vtw<SIGN, 9, 16>(y[3][3]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            v2f t0 = y[0][c], t1 = y[1][c], t2 = y[2][c], t3 = y[3][c];
            vdft4<SIGN>(t0, t1, t2, t3);
            v[c] = to_f2(t0); v[c + 4] = to_f2(t1); v[c + 8] = to_f2(t2); v[c + 12] = to_f2(t3);
        }
    }
};

}  // namespace ralign
