// In-register radix butterflies and the multi-stage Stockham line FFT for gfx950 (fp64; fp32 with -DOFDFT_REAL_F32).
//
// A line of LEN complex points is owned by P = LEN/E threads, E points per thread.  On entry and
// on exit thread j holds v[q] = x[j + P*q], q = 0..E-1 (so global loads/stores with consecutive j
// are coalesced and no bit-reversal pass exists).  Stages are radix-R_s Stockham steps done in
// registers; between stages the line is exchanged through an LDS line buffer (real and imaginary
// parts separately, so the buffer is LEN doubles + padding).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace ofdft {

// Every C-ABI entry point works on its context's device and leaves the calling thread's current device as it found it
// (torch keeps its own idea of the current device; an engine on another GPU must not change it behind torch's back).
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
        }
    }
    ~DeviceScope() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// The arithmetic type of the grid data.  The library is built once per precision (FFTW-style): the default build is
// fp64 (the reference's precision, parity 1e-8 Ha/atom), -DOFDFT_REAL_F32 builds the fp32 variant of the same kernels
// (BASELINE config 5).  Energy sums, their partials and every scalar derived from them are fp64 in both (acc_t).
#ifdef OFDFT_REAL_F32
typedef float real;
typedef float2 cplx;
#else
typedef double real;
typedef double2 cplx;
#endif
typedef double acc_t;
// fp32 build: the LDS exchange of a line transform moves whole complex numbers (8-byte accesses, one write + one read per
// element and stage) instead of real and imaginary parts separately (4-byte accesses, two of each, and twice the wave
// synchronisations): ds_write_b32 sustains 64 B/clk/CU where ds_write_b64 sustains ~85, and a b32 read half a b64 read's
// bytes per LDS cycle (MI355X_MICROARCH.md, LDS).  Kernels that opt in (StageP<..., CX = true>) give every line buffer
// twice the reals; element positions and bank behaviour are then those of the fp64 build's 8-byte accesses.
#ifndef OFDFT_F32_CX
#define OFDFT_F32_CX 1
#endif
constexpr bool kCX = sizeof(real) == 4 && OFDFT_F32_CX != 0;
constexpr int kCXMul = kCX ? 2 : 1;          // reals per element of a CX line buffer
__host__ __device__ __forceinline__ cplx mkc(real x, real y) {
    cplx c;
    c.x = x;
    c.y = y;
    return c;
}

// fp32 build (round 5): complex arithmetic on the register PAIR -- v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 do both components of
// a complex number in one instruction at the rate of a scalar one (the hardware's 2 x fp32 rate exists only in this form), so a
// complex product is 2 instructions instead of 4 and a butterfly's add / subtract pair 2 instead of 4.  Written component-wise the
// compiler found the packed form for about half of the sums and almost none of the products: the fp32 z kernels at 1024-point rows
// (config 5) spent 60-65 % of their time issuing vector instructions -- the same count as the fp64 build for half the bytes.
// Measured (profiles/r05_ab_f32_packed.jsonl, one box, two alternations): static vector instructions of zf_density<512, 8> 2 194 ->
// 1 864, of zi_combine 12 059 -> 10 566; on the device zf_density 3.3 -> 3.0 ps per point and zi_combine 7.6 -> 7.25 at 1024-point rows,
// zpbe 9.95 -> 9.35 at 256^3 -- but zi_wgc 8.55 -> 9.25, and the chirp-z kernels 67 -> 118 ps per point and pass (255^3 fp32 3.98 ->
// 6.0 ms): the evaluation gains ~1 % on plan-served grids and loses 50 % on the reference's own odd grids.  OFF by default;
// OFDFT_F32_PK=1 selects the packed forms.
#ifndef OFDFT_F32_PK
#define OFDFT_F32_PK 0
#endif
#if defined(OFDFT_REAL_F32) && OFDFT_F32_PK
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t c2v(cplx a) { v2f_t r = {a.x, a.y}; return r; }
__device__ __forceinline__ cplx v2c(v2f_t v) { return mkc(v.x, v.y); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return v2c(c2v(a) + c2v(b)); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return v2c(c2v(a) - c2v(b)); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {       // (a.x b.x - a.y b.y, a.x b.y + a.y b.x) = a.y (-b.y, b.x) + a.x b
    const v2f_t t = (v2f_t){a.x, a.x} * c2v(b);
    return v2c(__builtin_elementwise_fma((v2f_t){a.y, a.y}, (v2f_t){-b.y, b.x}, t));
}
#else
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return mkc(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return mkc(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return mkc(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
#endif
__device__ __forceinline__ cplx cconj(cplx a) { return mkc(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cplx mul_mi(cplx a) {
    return INV ? mkc(-a.y, a.x) : mkc(a.y, -a.x);
}

// cos(2 pi k / 16), sin(2 pi k / 16)
__device__ constexpr real kCos16[16] = {
    1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
    0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613,
    -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173,
    0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613};
__device__ constexpr real kSin16[16] = {
    0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613,
    1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
    0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613,
    -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173};

// a * W_R^K  (W = exp(-2 pi i / R) forward, conjugate for inverse), K, R compile-time
template <int R, int K, bool INV> __device__ __forceinline__ cplx mul_w(cplx a) {
    constexpr int idx = (K * (16 / R)) & 15;
    if constexpr (idx == 0) return a;
    else if constexpr (idx == 4) return mul_mi<INV>(a);
    else if constexpr (idx == 8) return mkc(-a.x, -a.y);
    else if constexpr (idx == 12) return mul_mi<!INV>(a);
    else {
        constexpr real c = kCos16[idx];
        constexpr real s = INV ? kSin16[idx] : -kSin16[idx];
#if defined(OFDFT_REAL_F32) && OFDFT_F32_PK
        return cmul(a, mkc(c, s));
#else
        return mkc(a.x * c - a.y * s, a.x * s + a.y * c);
#endif
    }
}

// ---- buffer (SRD) addressing: wave-uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset.
// All uniform strides are folded into the base pointer, so a thread keeps ONE offset VGPR for a whole
// line instead of a 64-bit address per element (that alone was ~60 VGPRs in the 16-point-per-thread
// kernels).  The base must be built from kernel arguments / blockIdx only (provably uniform).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// tell the compiler a 64-bit value is wave-uniform (it is, by construction, wherever this is used)
__device__ __forceinline__ long long uniform64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000);
}
// one complex element = one 16-byte (fp64) or 8-byte (fp32) buffer access
constexpr unsigned kCB = (unsigned)sizeof(cplx);          // bytes per complex element (byte offsets are built from it)
template <int AUX> __device__ __forceinline__ cplx buf_load_c_aux(const cplx* ubase, unsigned voff_bytes) {
    if constexpr (sizeof(cplx) == 16) {
        u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ubase), (int)voff_bytes, 0, AUX);
        return *reinterpret_cast<cplx*>(&t);
    } else {
        u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(ubase), (int)voff_bytes, 0, AUX);
        return *reinterpret_cast<cplx*>(&t);
    }
}
template <int AUX> __device__ __forceinline__ void buf_store_c_aux(cplx* ubase, unsigned voff_bytes, cplx v) {
    if constexpr (sizeof(cplx) == 16)
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<u32x4*>(&v), make_rsrc(ubase), (int)voff_bytes, 0, AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<u32x2*>(&v), make_rsrc(ubase), (int)voff_bytes, 0, AUX);
}
// streaming variants: AUX bit 1 = nt, for data that is touched once per pass and not re-read before it leaves the caches
__device__ __forceinline__ cplx buf_load_c(const cplx* ubase, unsigned voff_bytes) { return buf_load_c_aux<0>(ubase, voff_bytes); }
__device__ __forceinline__ void buf_store_c(cplx* ubase, unsigned voff_bytes, cplx v) { buf_store_c_aux<0>(ubase, voff_bytes, v); }
// plain-pointer nontemporal accesses (per-lane 64-bit addresses: the exchange-buffer side of the slab-decomposed y pass)
typedef real real2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx nt_load_c(const cplx* p) {
    const real2_t t = __builtin_nontemporal_load(reinterpret_cast<const real2_t*>(p));
    return mkc(t.x, t.y);
}
__device__ __forceinline__ void nt_store_c(cplx* p, cplx v) {
    real2_t t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<real2_t*>(p));
}
__device__ __forceinline__ real buf_load_d(const real* ubase, unsigned voff_bytes) {
    if constexpr (sizeof(real) == 8) {
        u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(ubase), (int)voff_bytes, 0, 0);
        return *reinterpret_cast<real*>(&t);
    } else {
        unsigned t = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(ubase), (int)voff_bytes, 0, 0);
        return *reinterpret_cast<real*>(&t);
    }
}

// natural-order in-place DFT of R points held in registers
template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<1, INV> {
    static __device__ __forceinline__ void run(cplx*) {}
};
template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cplx* a) {
        cplx t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};
template <bool INV> struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cplx* a) {
        cplx s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
        cplx s13 = cadd(a[1], a[3]), d13 = mul_mi<INV>(csub(a[1], a[3]));
        a[0] = cadd(s02, s13);
        a[1] = cadd(d02, d13);
        a[2] = csub(s02, s13);
        a[3] = csub(d02, d13);
    }
};

template <int R, int K, bool INV> struct Combine {
    static __device__ __forceinline__ void run(cplx* a, const cplx* e, const cplx* o) {
        cplx t = mul_w<R, K, INV>(o[K]);
        a[K] = cadd(e[K], t);
        a[K + R / 2] = csub(e[K], t);
        if constexpr (K + 1 < R / 2) Combine<R, K + 1, INV>::run(a, e, o);
    }
};

template <int R, bool INV> struct Dft {
    static_assert(R == 8 || R == 16, "radix");
    static __device__ __forceinline__ void run(cplx* a) {
        cplx e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
        Dft<R / 2, INV>::run(e);
        Dft<R / 2, INV>::run(o);
        Combine<R, 0, INV>::run(a, e, o);
    }
};

// ---- odd and mixed radices (grids whose extents are 2^a 3^b 5^c): cos / sin of 2 pi m / N as literals
template <int N> struct UnitRoots;
template <> struct UnitRoots<3> {
    static constexpr double c[3] = {1.0, -0.5000000000000000000000000, -0.5000000000000000000000000};
    static constexpr double s[3] = {0.0, 0.8660254037844385965883021, -0.8660254037844385965883021};
};
template <> struct UnitRoots<5> {
    static constexpr double c[5] = {1.0, 0.3090169943749474512628694, -0.8090169943749474512628694, -0.8090169943749474512628694, 0.3090169943749474512628694};
    static constexpr double s[5] = {0.0, 0.9510565162951535311819384, 0.5877852522924731371034568, -0.5877852522924731371034568, -0.9510565162951535311819384};
};
template <> struct UnitRoots<6> {
    static constexpr double c[6] = {1.0, 0.5000000000000000000000000, -0.5000000000000000000000000, -1.0, -0.5000000000000000000000000, 0.5000000000000000000000000};
    static constexpr double s[6] = {0.0, 0.8660254037844385965883021, 0.8660254037844385965883021, 0.0, -0.8660254037844385965883021, -0.8660254037844385965883021};
};
template <> struct UnitRoots<9> {
    static constexpr double c[9] = {1.0, 0.7660444431189780134516809, 0.1736481776669303589422100, -0.5000000000000000000000000, -0.9396926207859084279050421, -0.9396926207859084279050421, -0.5000000000000000000000000, 0.1736481776669303589422100, 0.7660444431189780134516809};
    static constexpr double s[9] = {0.0, 0.6427876096865393629187224, 0.9848077530122080203156543, 0.8660254037844385965883021, 0.3420201433256687129080831, -0.3420201433256687129080831, -0.8660254037844385965883021, -0.9848077530122080203156543, -0.6427876096865393629187224};
};
template <> struct UnitRoots<10> {
    static constexpr double c[10] = {1.0, 0.8090169943749474512628694, 0.3090169943749474512628694, -0.3090169943749474512628694, -0.8090169943749474512628694, -1.0, -0.8090169943749474512628694, -0.3090169943749474512628694, 0.3090169943749474512628694, 0.8090169943749474512628694};
    static constexpr double s[10] = {0.0, 0.5877852522924731371034568, 0.9510565162951535311819384, 0.9510565162951535311819384, 0.5877852522924731371034568, 0.0, -0.5877852522924731371034568, -0.9510565162951535311819384, -0.9510565162951535311819384, -0.5877852522924731371034568};
};
template <> struct UnitRoots<12> {
    static constexpr double c[12] = {1.0, 0.8660254037844385965883021, 0.5000000000000000000000000, 0.0, -0.5000000000000000000000000, -0.8660254037844385965883021, -1.0, -0.8660254037844385965883021, -0.5000000000000000000000000, 0.0, 0.5000000000000000000000000, 0.8660254037844385965883021};
    static constexpr double s[12] = {0.0, 0.5000000000000000000000000, 0.8660254037844385965883021, 1.0, 0.8660254037844385965883021, 0.5000000000000000000000000, 0.0, -0.5000000000000000000000000, -0.8660254037844385965883021, -1.0, -0.8660254037844385965883021, -0.5000000000000000000000000};
};
template <> struct UnitRoots<15> {
    static constexpr double c[15] = {1.0, 0.9135454576426008665990253, 0.6691306063588582375700753, 0.3090169943749474512628694, -0.1045284632676534708473071, -0.5000000000000000000000000, -0.8090169943749474512628694, -0.9781476007338056888329447, -0.9781476007338056888329447, -0.8090169943749474512628694, -0.5000000000000000000000000, -0.1045284632676534708473071, 0.3090169943749474512628694, 0.6691306063588582375700753, 0.9135454576426008665990253};
    static constexpr double s[15] = {0.0, 0.4067366430758002082690439, 0.7431448254773942441175905, 0.9510565162951535311819384, 0.9945218953682732898613494, 0.8660254037844385965883021, 0.5877852522924731371034568, 0.2079116908177593425754992, -0.2079116908177593425754992, -0.5877852522924731371034568, -0.8660254037844385965883021, -0.9945218953682732898613494, -0.9510565162951535311819384, -0.7431448254773942441175905, -0.4067366430758002082690439};
};

// a * W_N^K, W_N = exp(-2 pi i / N) (forward; conjugate for the inverse), K compile-time
template <int N, int K, bool INV> __device__ __forceinline__ cplx mul_root(cplx a) {
    constexpr int k = ((K % N) + N) % N;
    if constexpr (k == 0) {
        return a;
    } else if constexpr (2 * k == N) {
        return mkc(-a.x, -a.y);
    } else if constexpr (4 * k == N) {
        return mul_mi<INV>(a);
    } else if constexpr (4 * k == 3 * N) {
        return mul_mi<!INV>(a);
    } else {
        constexpr real c = (real)UnitRoots<N>::c[k];
        constexpr real sn = (real)(INV ? UnitRoots<N>::s[k] : -UnitRoots<N>::s[k]);
#if defined(OFDFT_REAL_F32) && OFDFT_F32_PK
        return cmul(a, mkc(c, sn));
#else
        return mkc(a.x * c - a.y * sn, a.x * sn + a.y * c);
#endif
    }
}
template <bool INV> struct Dft<3, INV> {
    static __device__ __forceinline__ void run(cplx* a) {
        constexpr real s = (real)UnitRoots<3>::s[1];
        const cplx t1 = cadd(a[1], a[2]);
        const cplx t2 = mkc(a[0].x - 0.5 * t1.x, a[0].y - 0.5 * t1.y);
        const cplx d = csub(a[1], a[2]);
        const cplx t3 = INV ? mkc(-s * d.y, s * d.x) : mkc(s * d.y, -s * d.x);      // -+ i s (a1 - a2)
        a[0] = cadd(a[0], t1);
        a[1] = cadd(t2, t3);
        a[2] = csub(t2, t3);
    }
};
template <bool INV> struct Dft<5, INV> {
    static __device__ __forceinline__ void run(cplx* a) {
        constexpr real c1 = (real)UnitRoots<5>::c[1], c2 = (real)UnitRoots<5>::c[2];
        constexpr real s1 = (real)UnitRoots<5>::s[1], s2 = (real)UnitRoots<5>::s[2];
        const cplx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        const cplx m1 = mkc(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
        const cplx m2 = mkc(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
        const cplx n1 = mkc(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        const cplx n2 = mkc(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        // forward: X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2; inverse: signs swapped
        const cplx in1 = INV ? mkc(-n1.y, n1.x) : mkc(n1.y, -n1.x);
        const cplx in2 = INV ? mkc(-n2.y, n2.x) : mkc(n2.y, -n2.x);
        a[0] = mkc(a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y);
        a[1] = cadd(m1, in1);
        a[4] = csub(m1, in1);
        a[2] = cadd(m2, in2);
        a[3] = csub(m2, in2);
    }
};
// R = R1 R2 by one Cooley-Tukey step in registers: sub-sequences x[n2 + R2 n1] -> DFT_R1 -> times W_R^(n2 k1) -> DFT_R2 over n2
template <int R1, int R2, bool INV> struct DftCT {
    static constexpr int R = R1 * R2;
    template <int N2, int K1> static __device__ __forceinline__ void twiddle_row(cplx (&y)[R2][R1]) {
        if constexpr (K1 < R1) {
            y[N2][K1] = mul_root<R, N2 * K1, INV>(y[N2][K1]);
            twiddle_row<N2, K1 + 1>(y);
        }
    }
    template <int N2> static __device__ __forceinline__ void twiddle_all(cplx (&y)[R2][R1]) {
        if constexpr (N2 < R2) {
            twiddle_row<N2, 1>(y);
            twiddle_all<N2 + 1>(y);
        }
    }
    static __device__ __forceinline__ void run(cplx* a) {
        cplx y[R2][R1];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) y[n2][n1] = a[n2 + R2 * n1];
            Dft<R1, INV>::run(y[n2]);
        }
        twiddle_all<1>(y);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cplx z[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) z[n2] = y[n2][k1];
            Dft<R2, INV>::run(z);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) a[k1 + R1 * k2] = z[k2];
        }
    }
};
template <bool INV> struct Dft<6, INV> { static __device__ __forceinline__ void run(cplx* a) { DftCT<3, 2, INV>::run(a); } };
template <bool INV> struct Dft<9, INV> { static __device__ __forceinline__ void run(cplx* a) { DftCT<3, 3, INV>::run(a); } };
template <bool INV> struct Dft<10, INV> { static __device__ __forceinline__ void run(cplx* a) { DftCT<5, 2, INV>::run(a); } };
template <bool INV> struct Dft<12, INV> { static __device__ __forceinline__ void run(cplx* a) { DftCT<3, 4, INV>::run(a); } };
template <bool INV> struct Dft<15, INV> { static __device__ __forceinline__ void run(cplx* a) { DftCT<3, 5, INV>::run(a); } };

// ---- per-length plan: stage radices, threads per line P, register slots per thread E.
// Stage s has LEN / R_s butterflies, dealt round-robin over the P threads of the line: thread j takes butterflies
// j, j + P, ... (NB_s = ceil(LEN / (R_s P)) of them; when LEN / R_s is not a multiple of P some lanes idle in the last
// round -- that is how extents with factors 3 and 5 keep P a power of two, i.e. tiles of whole wavefronts).  Butterfly
// beta of the FIRST stage takes x[beta + t LEN/R_0], t < R_0; butterfly beta of the LAST stage leaves
// X[beta + u LEN/R_last]: a thread's q-th register slot therefore holds element j + cin(q) on entry and j + cout(q) on
// exit (for the power-of-two plans both are P q and every slot is used: loads and stores with consecutive j coalesce
// and no reordering pass exists).
template <int LEN_, int P_, int NST_, int R0_, int R1_, int R2_, int R3_> struct PlanBase {
    static constexpr int LEN = LEN_;
    static constexpr int NST = NST_;
    static constexpr int P = P_;
    static __host__ __device__ constexpr int radix(int s) { return s == 0 ? R0_ : (s == 1 ? R1_ : (s == 2 ? R2_ : R3_)); }
    static __host__ __device__ constexpr int nbf(int s) { return LEN_ / radix(s); }                  // butterflies of stage s
    static __host__ __device__ constexpr int nb(int s) { return (nbf(s) + P_ - 1) / P_; }            // ... per thread
    static __host__ __device__ constexpr int slots(int s) { return radix(s) * nb(s); }
    static __host__ __device__ constexpr int emax(int s) { return s >= NST_ ? 0 : (slots(s) > emax(s + 1) ? slots(s) : emax(s + 1)); }
    static constexpr int E = emax(0);
    static_assert(R0_ * (NST_ > 1 ? R1_ : 1) * (NST_ > 2 ? R2_ : 1) * (NST_ > 3 ? R3_ : 1) == LEN_, "radices must multiply to LEN");
    // element offsets of register slot q relative to the thread index j, on entry / on exit, and whether the slot is used
    static __host__ __device__ constexpr int cin(int q) { return (q % nb(0)) * P_ + (q / nb(0)) * nbf(0); }
    static __host__ __device__ constexpr int cout(int q) { return (q % nb(NST_ - 1)) * P_ + (q / nb(NST_ - 1)) * nbf(NST_ - 1); }
    static __host__ __device__ constexpr bool slot_in(int q) { return q < slots(0); }
    static __host__ __device__ constexpr bool slot_out(int q) { return q < slots(NST_ - 1); }
    static constexpr bool FULL_IN = nbf(0) % P_ == 0, FULL_OUT = nbf(NST_ - 1) % P_ == 0;
    static __device__ __forceinline__ bool lane_in(int j, int q) { return FULL_IN || j + (q % nb(0)) * P_ < nbf(0); }
    static __device__ __forceinline__ bool lane_out(int j, int q) { return FULL_OUT || j + (q % nb(NST_ - 1)) * P_ < nbf(NST_ - 1); }
    // every slot used on both sides, element j + P q in slot q: the layout the power-of-two kernels were written for
    static constexpr bool EXACT = FULL_IN && FULL_OUT && slots(0) == E && slots(NST_ - 1) == E && P_ * E == LEN_;
};
template <int LEN> struct Plan;
#define OFDFT_PLAN(LEN_, NST_, R0_, R1_, R2_, E_)                                                                     \
    template <> struct Plan<LEN_> : PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, 1> {                              \
        static_assert(PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, 1>::E == E_ && PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, 1>::EXACT, "plan"); \
    };
OFDFT_PLAN(8, 1, 8, 1, 1, 8)
OFDFT_PLAN(16, 1, 16, 1, 1, 16)
OFDFT_PLAN(32, 2, 8, 4, 1, 8)
OFDFT_PLAN(64, 2, 8, 8, 1, 8)
OFDFT_PLAN(128, 2, 16, 8, 1, 16)
OFDFT_PLAN(256, 2, 16, 16, 1, 16)
OFDFT_PLAN(512, 3, 8, 8, 8, 8)
OFDFT_PLAN(1024, 3, 16, 8, 8, 16)
#undef OFDFT_PLAN
// extents with factors 3 and 5 (OFDFT_MIXED_SIZES lists them for the dispatch switches): LEN, P, stages, radices
#define OFDFT_GPLAN(LEN_, P_, NST_, R0_, R1_, R2_) template <> struct Plan<LEN_> : PlanBase<LEN_, P_, NST_, R0_, R1_, R2_, 1> {};
OFDFT_GPLAN(48, 8, 2, 6, 8, 1)          // 8 + 6 butterflies over 8 threads: E = 8
OFDFT_GPLAN(96, 8, 2, 8, 12, 1)         // E = 16
OFDFT_GPLAN(120, 16, 2, 8, 15, 1)       // 15 / 8 butterflies: E = 15
OFDFT_GPLAN(144, 16, 2, 16, 9, 1)       // 9 / 16: E = 16
OFDFT_GPLAN(160, 16, 2, 16, 10, 1)      // 10 / 16: E = 16
OFDFT_GPLAN(192, 16, 2, 12, 16, 1)      // 16 / 12: E = 16
OFDFT_GPLAN(240, 16, 2, 16, 15, 1)      // 15 / 16: E = 16
OFDFT_GPLAN(250, 32, 3, 5, 5, 10)       // 50 / 50 / 25: E = 10
OFDFT_GPLAN(270, 32, 3, 6, 9, 5)        // 45 / 30 / 54: E = 12
OFDFT_GPLAN(288, 32, 3, 6, 6, 8)        // 48 / 48 / 36: E = 16
OFDFT_GPLAN(320, 32, 3, 8, 8, 5)        // 40 / 40 / 64: E = 16
OFDFT_GPLAN(384, 32, 3, 8, 8, 6)        // 48 / 48 / 64: E = 16
OFDFT_GPLAN(480, 32, 3, 16, 6, 5)       // 30 / 80 / 96: E = 18
#undef OFDFT_GPLAN

// padded position inside an LDS line buffer (breaks the power-of-two strides of the exchange)
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
template <int LEN> struct LineBuf {
    // doubles per line buffer; +2 staggers consecutive lines over the banks
    static constexpr int STRIDE = LEN + (LEN >> 4) + 2;
};
// Layout of a line buffer per PLAN.  The padded layout keeps the strided exchange writes of every radix on distinct bank
// pairs, but a unit-stride read by the 32 lanes of one ds_read_b64 group crosses a pad and wraps onto its own first
// bank pair: 2 LDS cycles instead of 1 (SQ_LDS_BANK_CONFLICT 40-49 % of the LDS cycles of the wave-local z kernels,
// whose rows are read by 32 lanes).  The radix-4 plans of those kernels (ZPlan<LEN, 4>) therefore swizzle inside each
// 16-element block instead: position i ^ 5 b with b = bits 4-5 of i -- aligned unit-stride runs of 32 stay a permutation
// of the 32 bank pairs, and the stride-4 / stride-16 writes of radix-4 stages land on distinct pairs too (b reaches bits
// 0-1 and 2-3).  Radix-8 / -16 stages need the padded form (their 16 write lanes differ in more than two block bits).
#ifndef OFDFT_LDS_SWIZZLE_Z
#define OFDFT_LDS_SWIZZLE_Z 1
#endif
// A third form, for plans whose lines are interleaved over the lanes of a wave (the wave-local fused x pass, xwave.h):
// position i ^ ((XMUL * ((i >> XS) & XM)) & 31), parameters per length found by enumerating the exchange pattern against
// the bank rules of MI355X_MICROARCH.md (ds_read_b64: two 32-lane groups, bank (a/4) mod 64; ds_write_b64: four 16-lane
// groups, bank (a/4) mod 32): conflict-free reads AND writes where the padded form costs 2 cycles per LDS access.
struct LdsLayoutDefault { static constexpr bool SWIZZLE = false; static constexpr int XS = 0, XM = 0, XMUL = 0; };
template <class PL> struct LdsLayout : LdsLayoutDefault {};
template <class PL> __device__ __forceinline__ int lpos(int i) {
    if constexpr (LdsLayout<PL>::XM != 0) return i ^ ((LdsLayout<PL>::XMUL * ((i >> LdsLayout<PL>::XS) & LdsLayout<PL>::XM)) & 31);
    else if constexpr (LdsLayout<PL>::SWIZZLE) return i ^ (5 * ((i >> 4) & 3));
    else return lpad(i);
}
template <class PL> constexpr int line_stride() {
    // swizzled: whole 16-element blocks, then 16 more so that two 16-lane rows of one read group sit on opposite halves of
    // the bank row
    return LdsLayout<PL>::SWIZZLE ? ((PL::LEN + 15) / 16) * 16 + 16 : LineBuf<PL::LEN>::STRIDE;
}

// workgroup-wide or wave-local synchronisation of the LDS exchange.  WAVE = true is legal when all P
// threads of a line are lanes of ONE wavefront: a wave's LDS instructions execute in program order, so a
// ds_write followed by a ds_read of the same wave needs no s_barrier -- only a compiler-level fence.
template <bool WAVE> __device__ __forceinline__ void exchange_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// TWTAB: `tw` is a copy of the table in LDS and every power W^(t k) is READ from it (R - 1 ds_read_b128) instead of being formed
// from W^k by the product tree (24 fp64 instructions per radix-8 butterfly): for kernels bound by their instruction count
template <class PL, int S, int NS, bool INV, bool WAVE, bool TWTAB = false, bool CX = false> struct StageP {
    static constexpr int LEN = PL::LEN;
    static constexpr int R = PL::radix(S);
    static constexpr int E = PL::E;
    static constexpr int P = PL::P;
    static constexpr int NBF = PL::nbf(S);     // butterflies of this stage (LEN / R)
    static constexpr int NB = PL::nb(S);       // ... per thread (also the register stride of a butterfly's inputs)
    static constexpr bool FULL = NBF % P == 0; // every lane busy in every round

    // lx: a per-line constant (< 32) XORed into every position -- lets kernels whose lanes interleave several lines place
    // neighbouring lines on complementary banks (xwave.h); 0 for everybody else
    static __device__ __forceinline__ void run(cplx (&v)[E], int j, real* line, const cplx* __restrict__ tw, int lx = 0) {
        // ---- twiddle + butterflies
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (FULL || j + b * P < NBF) {
                cplx a[R];
#pragma unroll
                for (int t = 0; t < R; ++t) a[t] = v[b + t * NB];
                if constexpr (NS > 1) {
                    // twiddles W^(t k), t = 1..R-1: ONE table load (W^k) and a depth-<=4 product tree for the
                    // powers (<= ~5 ulp), instead of R-1 loads whose prefetch would pin 4(R-1) VGPRs
                    const int k = (j + b * P) % NS;
                    constexpr int TSTEP = LEN / (NS * R);
                    cplx w[R];
                    if constexpr (TWTAB) {
#pragma unroll
                        for (int t = 1; t < R; ++t) {
                            w[t] = tw[k * (TSTEP * t)];
                            if (INV) w[t].y = -w[t].y;
                        }
                    } else {
                        w[1] = tw[k * TSTEP];
                        if (INV) w[1].y = -w[1].y;
#pragma unroll
                        for (int t = 2; t < R; ++t) {
                            const int hi = (t >= 8) ? 8 : ((t >= 4) ? 4 : 2);   // largest power of two <= t
                            w[t] = (t == hi) ? cmul(w[t / 2], w[t / 2]) : cmul(w[hi], w[t - hi]);
                        }
                    }
#pragma unroll
                    for (int t = 1; t < R; ++t) a[t] = cmul(a[t], w[t]);
                }
                Dft<R, INV>::run(a);
#pragma unroll
                for (int t = 0; t < R; ++t) v[b + t * NB] = a[t];
            }
        }
        // ---- exchange through LDS (not after the last stage): this stage's outputs at their Stockham positions, the next
        // stage's butterfly inputs (element beta + t LEN/R') back into its register slots
        if constexpr (S + 1 < PL::NST) {
            constexpr int R2 = PL::radix(S + 1), NBF2 = PL::nbf(S + 1), NB2 = PL::nb(S + 1);
            constexpr bool FULL2 = NBF2 % P == 0;
            int base[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int jb = j + b * P;
                base[b] = (jb / NS) * (NS * R) + (jb % NS);
            }
            if constexpr (CX) {          // whole complex numbers through a buffer of complex elements (fp32 build)
                cplx* cl = reinterpret_cast<cplx*>(line);
                exchange_sync<WAVE>();
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (FULL || j + b * P < NBF) {
#pragma unroll
                        for (int u = 0; u < R; ++u) cl[lpos<PL>(base[b] + u * NS) ^ lx] = v[b + u * NB];
                    }
                exchange_sync<WAVE>();
#pragma unroll
                for (int b = 0; b < NB2; ++b)
                    if (FULL2 || j + b * P < NBF2) {
#pragma unroll
                        for (int t = 0; t < R2; ++t) v[b + t * NB2] = cl[lpos<PL>(j + b * P + t * NBF2) ^ lx];
                    }
            } else {
                exchange_sync<WAVE>();
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (FULL || j + b * P < NBF) {
#pragma unroll
                        for (int u = 0; u < R; ++u) line[lpos<PL>(base[b] + u * NS) ^ lx] = v[b + u * NB].x;
                    }
                exchange_sync<WAVE>();
                real re[E];
#pragma unroll
                for (int b = 0; b < NB2; ++b)
                    if (FULL2 || j + b * P < NBF2) {
#pragma unroll
                        for (int t = 0; t < R2; ++t) re[b + t * NB2] = line[lpos<PL>(j + b * P + t * NBF2) ^ lx];
                    }
                exchange_sync<WAVE>();
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (FULL || j + b * P < NBF) {
#pragma unroll
                        for (int u = 0; u < R; ++u) line[lpos<PL>(base[b] + u * NS) ^ lx] = v[b + u * NB].y;
                    }
                exchange_sync<WAVE>();
#pragma unroll
                for (int b = 0; b < NB2; ++b)
                    if (FULL2 || j + b * P < NBF2) {
#pragma unroll
                        for (int t = 0; t < R2; ++t) v[b + t * NB2] = mkc(re[b + t * NB2], line[lpos<PL>(j + b * P + t * NBF2) ^ lx]);
                    }
            }
            StageP<PL, S + 1, NS * R, INV, WAVE, TWTAB, CX>::run(v, j, line, tw, lx);
        }
    }
};

// Full line transform.  `line` = this line's LDS buffer (LineBuf<LEN>::STRIDE doubles; unused when
// the plan has one stage).  `tw` = forward table W_LEN^m, m = 0..LEN-1 (global memory).
// Every thread of the workgroup must call this (it contains __syncthreads()).
template <int LEN, bool INV, bool CX = false>
__device__ __forceinline__ void line_fft(cplx (&v)[Plan<LEN>::E], int j, real* line, const cplx* __restrict__ tw) {
    StageP<Plan<LEN>, 0, 1, INV, false, false, CX>::run(v, j, line, tw);
}

// ---- small-footprint plans for the z kernels: E = 8 or 4 complex points per lane.  A line's P = LEN/E
// threads are lanes of one wave, so the exchanges need no barrier (extra stages only cost LDS traffic), and
// the small register footprint leaves room for fused pointwise math.
template <int LEN_, int E_> struct ZPlan;
// (fp32, rows of 128 and 256 points: the b32 bank rules -- two 32-lane groups over 32 banks for reads AND writes -- leave the
// i ^ 5b swizzle with 2-way write conflicts; i ^ ((i >> 2) & 31) is conflict-free in the enumeration (tools/lds_conflicts.py).
// Measured (fp32, 256^3, two alternations): zi_combine 0.154 -> 0.141 ms, zi_wgc 0.140 -> 0.135, zf_powers 0.108 -> 0.103,
// evaluation 1.69 -> 1.65 ms.  OFDFT_Z4_XOR_F32=0: the fp64 swizzle in the fp32 build too)
#ifndef OFDFT_Z4_XOR_F32
#define OFDFT_Z4_XOR_F32 1
#endif
template <int LEN_> struct LdsLayout<ZPlan<LEN_, 4>> : LdsLayoutDefault {
    static constexpr bool XORF = OFDFT_Z4_XOR_F32 && sizeof(real) == 4 && LEN_ >= 128;
    static constexpr bool SWIZZLE = OFDFT_LDS_SWIZZLE_Z != 0;
    static constexpr int XS = XORF ? 2 : 0, XM = XORF ? 31 : 0, XMUL = XORF ? 1 : 0;
};
#define OFDFT_ZPLAN(LEN_, E_, NST_, R0_, R1_, R2_, R3_)                                              \
    template <> struct ZPlan<LEN_, E_> : PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, R3_> {     \
        static_assert(PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, R3_>::E == E_ && PlanBase<LEN_, LEN_ / E_, NST_, R0_, R1_, R2_, R3_>::EXACT, "plan"); \
    };
OFDFT_ZPLAN(8, 8, 1, 8, 1, 1, 1)
OFDFT_ZPLAN(16, 8, 2, 8, 2, 1, 1)
OFDFT_ZPLAN(32, 8, 2, 8, 4, 1, 1)
OFDFT_ZPLAN(64, 8, 2, 8, 8, 1, 1)
OFDFT_ZPLAN(128, 8, 3, 8, 4, 4, 1)
OFDFT_ZPLAN(256, 8, 3, 8, 8, 4, 1)
OFDFT_ZPLAN(512, 8, 3, 8, 8, 8, 1)
OFDFT_ZPLAN(8, 4, 2, 4, 2, 1, 1)
OFDFT_ZPLAN(16, 4, 2, 4, 4, 1, 1)
OFDFT_ZPLAN(32, 4, 3, 4, 4, 2, 1)
OFDFT_ZPLAN(64, 4, 3, 4, 4, 4, 1)
OFDFT_ZPLAN(128, 4, 4, 4, 4, 4, 2)
OFDFT_ZPLAN(256, 4, 4, 4, 4, 4, 4)
#undef OFDFT_ZPLAN

// E to use for a z kernel that wants `want` points per lane: lines longer than 64*want need more
template <int M, int WANT> struct ZPick { static constexpr int E = (M / WANT <= 64) ? WANT : 8; };
// rows whose half length M = n2 / 2 has factors 3 and 5: one plan per M (P a power of two <= 64, E as small as the
// factorisation allows), whatever lane width the kernel asked for
#ifndef OFDFT_Z120_STAGES4
#define OFDFT_Z120_STAGES4 1       // (240^3: zpbe2 40.2 -> 30.4 ps per point, zi_combine 22.6 -> 18.3, evaluation 3.12 -> 2.91 ms)
#endif
// Round 3, second half: stage ORDER chosen for the entry / exit patterns -- the fused z kernels do their pointwise physics (pow, PBE,
// the combine) in the register pattern of the first stage (rows going INTO a forward transform) or of the last one (rows coming OUT
// of an inverse), so a stage that leaves 16-20 butterflies over 32 lanes makes that part run with half the lanes masked.  A/B per
// extent (tools/shape_ab.sh, ms per evaluation of the bench's term set): 120^3 0.521 -> 0.50, 160^3 1.065 -> 1.04, 192^3 1.53 -> 1.50,
// 240^3 3.12 -> 2.91, 270^3 5.00 -> 4.92, 480^3 26.6 -> 25.4; 288^3 (3 x 4 x 4 x 3) measured 1.5 % slower and keeps its order.
#ifndef OFDFT_ZALT
#define OFDFT_ZALT 1
#endif
#ifndef OFDFT_Z60_ALT
#define OFDFT_Z60_ALT OFDFT_ZALT
#endif
#ifndef OFDFT_Z80_ALT
#define OFDFT_Z80_ALT OFDFT_ZALT
#endif
#ifndef OFDFT_Z96_ALT
#define OFDFT_Z96_ALT OFDFT_ZALT
#endif
#ifndef OFDFT_Z135_ALT
#define OFDFT_Z135_ALT OFDFT_ZALT
#endif
#ifndef OFDFT_Z144_ALT
#define OFDFT_Z144_ALT 0
#endif
#ifndef OFDFT_Z240_ALT
#define OFDFT_Z240_ALT OFDFT_ZALT
#endif
#define OFDFT_ZGPLAN(M_, P_, NST_, R0_, R1_, R2_, R3_)                                                               \
    template <> struct ZPlan<M_, PlanBase<M_, P_, NST_, R0_, R1_, R2_, R3_>::E> : PlanBase<M_, P_, NST_, R0_, R1_, R2_, R3_> {}; \
    template <> struct ZPick<M_, 4> { static constexpr int E = PlanBase<M_, P_, NST_, R0_, R1_, R2_, R3_>::E; };   \
    template <> struct ZPick<M_, 8> { static constexpr int E = PlanBase<M_, P_, NST_, R0_, R1_, R2_, R3_>::E; };
OFDFT_ZGPLAN(24, 8, 3, 4, 3, 2, 1)        // E = 4
OFDFT_ZGPLAN(48, 16, 3, 4, 4, 3, 1)       // E = 4
#if OFDFT_Z60_ALT
OFDFT_ZGPLAN(60, 32, 4, 2, 5, 3, 2)       // E = 5: entry and exit stages leave 30 butterflies over 32 lanes (2 slots)
#else
OFDFT_ZGPLAN(60, 16, 3, 4, 5, 3, 1)       // E = 6
#endif
OFDFT_ZGPLAN(72, 32, 3, 4, 6, 3, 1)       // E = 6 (4 x 3 x 3 x 2 with E = 4 measured 4 % slower at 144^3)
#if OFDFT_Z80_ALT
OFDFT_ZGPLAN(80, 32, 3, 4, 5, 4, 1)       // E = 5: exit stage 20 butterflies x 4 slots (62 % of the lanes; 4 x 4 x 5: 16 x 5, 50 %)
#else
OFDFT_ZGPLAN(80, 32, 3, 4, 4, 5, 1)       // E = 5
#endif
#if OFDFT_Z96_ALT
OFDFT_ZGPLAN(96, 32, 4, 4, 4, 2, 3)       // E = 4: exit stage 32 butterflies x 3 slots (every lane)
#else
OFDFT_ZGPLAN(96, 32, 4, 4, 4, 3, 2)       // E = 4 (was 4 x 4 x 6, E = 6: 192^3 1.81 -> 1.51 ms)
#endif
#if OFDFT_Z120_STAGES4
// 4 x 5 x 3 x 2: one more exchange, but the LAST stage leaves 60 butterflies over the 32 lanes (4 slots, 94 % of the lanes) where
// the radix-6 stage leaves 20 (6 slots, 62 %): the fused z kernels do their pointwise physics in the exit pattern of the inverses
OFDFT_ZGPLAN(120, 32, 4, 4, 5, 3, 2)      // E = 6
#else
OFDFT_ZGPLAN(120, 32, 3, 4, 5, 6, 1)      // E = 6
#endif
OFDFT_ZGPLAN(125, 32, 3, 5, 5, 5, 1)      // E = 5
#if OFDFT_Z135_ALT
OFDFT_ZGPLAN(135, 32, 4, 3, 3, 3, 5)      // E = 6 over 32 lanes: two waves per SIMD for every fused z kernel
#else
OFDFT_ZGPLAN(135, 32, 3, 3, 9, 5, 1)      // E = 9 (3 x 3 x 3 x 5 over 64 lanes, E = 5: 270^3 11 % slower)
#endif
#if OFDFT_Z144_ALT
OFDFT_ZGPLAN(144, 64, 4, 3, 4, 4, 3)      // E = 4: entry stage 48 butterflies x 3 slots (75 % of the lanes; 4 first: 36 x 4, 56 %)
#else
OFDFT_ZGPLAN(144, 64, 4, 4, 4, 3, 3)      // E = 4 (was 4 x 6 x 6, E = 6: 288^3 8.34 -> 6.14 ms)
#endif
OFDFT_ZGPLAN(160, 32, 3, 4, 8, 5, 1)      // E = 8 (4 x 4 x 2 x 5 over 64 lanes, E = 5: 320^3 12 % slower)
OFDFT_ZGPLAN(192, 64, 4, 4, 4, 4, 3)      // E = 4
#if OFDFT_Z240_ALT
OFDFT_ZGPLAN(240, 64, 4, 4, 5, 3, 4)      // E = 6: exit stage 60 butterflies x 4 slots (94 % of the lanes; 5 last: 48 x 5, 75 %)
#else
OFDFT_ZGPLAN(240, 64, 4, 4, 4, 3, 5)      // E = 6
#endif
#undef OFDFT_ZGPLAN

// (OFDFT_Z_TWTAB=1: every twiddle power read from the table -- the z kernels keep it in LDS -- instead of formed by the product
// tree: fewer fp64 instructions, more LDS reads; A/B knob for the instruction-bound GGA mid stage)
#ifndef OFDFT_Z_TWTAB
#define OFDFT_Z_TWTAB 0
#endif
template <int LEN, int E, bool INV, bool TT = (OFDFT_Z_TWTAB != 0), bool CX = false>
__device__ __forceinline__ void wave_line_fft(cplx (&v)[E], int j, real* line, const cplx* __restrict__ tw) {
    static_assert(ZPlan<LEN, E>::P <= 64, "a line must fit one wavefront");
    StageP<ZPlan<LEN, E>, 0, 1, INV, true, TT, CX>::run(v, j, line, tw);
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0..N-1
template <int I, int N, class F> __device__ __forceinline__ void static_for_impl(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F> __device__ __forceinline__ void static_for(F f) { static_for_impl<0, N>(f); }

}  // namespace ofdft
