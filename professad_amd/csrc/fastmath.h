// Lean fp64 transcendentals for the fused pointwise physics (gfx950).
//
// The fused z kernels (PBE mid stage, WGC99 powers and combine) are bound by fp64 vector issue, not by HBM: the
// correctly rounded library functions cost 111 (log), 55 (exp), 35 (cbrt), 24 (sqrt), 11 (1/x) and ~250 (pow)
// instructions each, and one PBE point needs two logs, an exp, a cube root, a square root and about ten quotients.
// The versions here are accurate to a few ulp (relative error <= ~1e-15, far inside the 1e-10 parity bar of the
// tests and the 1e-8 Ha/atom bar of the north star) for POSITIVE, FINITE, NORMAL arguments -- which is what a density,
// a Wigner-Seitz radius or a PBE enhancement argument is -- and cost 5 (1/x), ~32 (log), ~20 (exp), ~25 (n^(-1/6) with
// every root of n the GGA formulas need derived from it by multiplications).
// The fp32 build (real = float) uses the hardware's transcendental instructions directly (v_rcp_f32, v_log_f32, v_exp_f32:
// 1 ulp, quarter rate) instead of the library's IEEE wrappers -- logf / expf / cbrtf / the division sequence cost 8-25
// instructions each, which made the fp32 GGA mid stage as long an instruction stream as the fp64 one (3 677 vs 3 695 VALU
// instructions per wave, profiles/r03_sq_counters_f32_before.md).  Arguments are positive normal numbers as above; relative
// errors stay at a few fp32 ulp (the fp32 parity tolerances are 5e-6 / 5e-4, tests/test_gpu_f32.py).
#pragma once
#include <hip/hip_runtime.h>

namespace ofdft {
namespace fm {

// (Round 4, measured negative: a Horner step p = fma(p, z, C) compiles to v_mov_b32 x 2 (C into a VGPR pair) + v_fmac_f64, and 204 of
// the 1 453 vector instructions of one point pair of the GGA mid stage are such moves.  Forcing C into an SGPR operand of v_fma_f64
// by inline asm removed 72 of them -- zpbe2 0.338 -> 0.338 ms, it is not bound by its instruction count -- and made the
// register-heavy WGC99 combine kernel 0.25 -> 0.36-0.39 ms slower (the asm operands lengthen its live ranges).  Not kept.)
// ---- reciprocal: hardware seed (~2^-23 relative) + ONE third-order step r (1 + e + e^2), e = 1 - x r: the error after it is
// e^3 ~ 2^-69, below the rounding of the last fma (round 4; was two Newton steps = one fma more); no scaling / fix-up
// (normal-range arguments).  OFDFT_RCP_NEWTON2=1 restores the two-step form.
#ifndef OFDFT_RCP_NEWTON2
#define OFDFT_RCP_NEWTON2 0
#endif
__device__ __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
#if OFDFT_RCP_NEWTON2
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
#endif
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- natural logarithm, x > 0 finite normal:  x = m 2^e, m in [sqrt(1/2), sqrt(2)),  log m = 2 atanh(s), s = (m-1)/(m+1)
__device__ __forceinline__ double log(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;                                   // exact
    const double s = f * rcp(2.0 + f);
    const double z = s * s;                                     // <= 0.02944
    // 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ...): the term z^10/21 is 2e-17 of the sum
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    const double s2 = s + s;
    const double lm = __builtin_fma(s2 * z, p, s2);             // log m
    const double de = (double)e;
    // ln 2 split so that e * hi is exact for |e| < 2^11 (hi has 41 significant bits)
    const double ln2_hi = 0x1.62e42fefa38p-1, ln2_lo = 0x1.ef35793c7673p-45;
    return __builtin_fma(de, ln2_hi, __builtin_fma(de, ln2_lo, lm));
}
__device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }     // v_log_f32 = log2

// ---- exponential, |x| < ~700:  x = k ln 2 + r, |r| <= ln(2)/2, Taylor to r^13/13! (next term < 5e-18)
__device__ __forceinline__ double exp(double x) {
    const double k = __builtin_rint(x * 1.44269504088896340736);
    const double ln2_hi = 0x1.62e42fefa38p-1, ln2_lo = 0x1.ef35793c7673p-45;
    const double r = __builtin_fma(-k, ln2_lo, __builtin_fma(-k, ln2_hi, x));
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);          // (inline constants: no operand to place)
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}
__device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// x^y for x > 0
__device__ __forceinline__ double pow_pos(double x, double y) { return exp(y * log(x)); }
__device__ __forceinline__ float pow_pos(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

// ---- n^(-1/6) for n > 0 finite normal, any magnitude: n = m 2^(6q + r) with m 2^r in [0.5, 32); fp32 seed
// (v_log_f32 / v_exp_f32, ~2^-21) + two Newton steps on y^-6 = n (error -> 3.5 error^2 per step)
__device__ __forceinline__ double rsixth(double n) {
    const int e = __builtin_amdgcn_frexp_exp(n);
    const double m = __builtin_amdgcn_frexp_mant(n);
    // q = floor(e / 6) for |e| <= 1100 by a multiply-shift; r = e - 6 q in 0..5
    const int q = ((e + 1200) * 10923 >> 16) - 200;
    const int r = e - 6 * q;
    const double a = __builtin_amdgcn_ldexp(m, r);              // [0.5, 32)
    double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)a) * (-1.0f / 6.0f));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y3 = y2 * y;
        const double h = __builtin_fma(-a, y3 * y3, 1.0);       // 1 - a y^6
        y = __builtin_fma(y * h, 1.0 / 6.0, y);
    }
    return __builtin_amdgcn_ldexp(y, -q);
}

// every root of the density the local / semilocal formulas use, from ONE n^(-1/6)
template <class T> struct Roots {
    T y;        // n^(-1/6)
    T inv13;    // n^(-1/3)
    T n13;      // n^(1/3)
    T inv_n;    // 1 / n
};
// ... from an n^(-1/6) the caller already holds (kernels that need the roots in several places form y once per point)
template <class T> __device__ __forceinline__ Roots<T> roots_from_y(T n, T y) {
    Roots<T> r;
    r.y = y;
    r.inv13 = y * y;
    const T y4 = r.inv13 * r.inv13;
    r.n13 = n * y4;
    r.inv_n = y4 * r.inv13;
    return r;
}
__device__ __forceinline__ Roots<double> roots(double n) {
    Roots<double> r;
    r.y = rsixth(n);
    r.inv13 = r.y * r.y;
    const double y4 = r.inv13 * r.inv13;
    r.n13 = n * y4;
    r.inv_n = y4 * r.inv13;
    return r;
}
// fp32: n^(-1/6) = 2^(-log2(n) / 6) from the hardware log2 / exp2 (error ~ |log2 n| / 6 ulp) + ONE Newton step on y^-6 = n
__device__ __forceinline__ float rsixth(float n) {
    float y = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(n) * (-1.0f / 6.0f));
    const float y2 = y * y, y3 = y2 * y;
    const float h = __builtin_fmaf(-n, y3 * y3, 1.0f);          // 1 - n y^6
    return __builtin_fmaf(y * h, 1.0f / 6.0f, y);
}
__device__ __forceinline__ Roots<float> roots(float n) {
    Roots<float> r;
    r.y = rsixth(n);
    r.inv13 = r.y * r.y;
    const float y4 = r.inv13 * r.inv13;
    r.n13 = n * y4;
    r.inv_n = y4 * r.inv13;
    return r;
}

}  // namespace fm
}  // namespace ofdft
