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
__host__ __device__ __forceinline__ cplx mkc(real x, real y) {
    cplx c;
    c.x = x;
    c.y = y;
    return c;
}

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return mkc(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return mkc(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return mkc(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
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
        return mkc(a.x * c - a.y * s, a.x * s + a.y * c);
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

// ---- per-length plan: stage radices, points per thread E, threads per line P = LEN / E
template <int LEN> struct Plan;
#define OFDFT_PLAN(LEN_, NST_, R0_, R1_, R2_, E_)                                   \
    template <> struct Plan<LEN_> {                                                \
        static constexpr int LEN = LEN_;                                           \
        static constexpr int NST = NST_;                                           \
        static constexpr int E = E_;                                               \
        static constexpr int P = LEN_ / E_;                                        \
        static __host__ __device__ constexpr int radix(int s) { return s == 0 ? R0_ : (s == 1 ? R1_ : R2_); } \
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

// padded position inside an LDS line buffer (breaks the power-of-two strides of the exchange)
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
template <int LEN> struct LineBuf {
    // doubles per line buffer; +2 staggers consecutive lines over the banks
    static constexpr int STRIDE = LEN + (LEN >> 4) + 2;
};

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

template <class PL, int S, int NS, bool INV, bool WAVE> struct StageP {
    static constexpr int LEN = PL::LEN;
    static constexpr int R = PL::radix(S);
    static constexpr int E = PL::E;
    static constexpr int P = PL::P;
    static constexpr int NB = E / R;       // butterflies per thread in this stage (also register stride)

    static __device__ __forceinline__ void run(cplx (&v)[E], int j, real* line, const cplx* __restrict__ tw) {
        // ---- twiddle + butterflies
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            cplx a[R];
#pragma unroll
            for (int t = 0; t < R; ++t) a[t] = v[b + t * NB];
            if constexpr (NS > 1) {
                // twiddles W^(t k), t = 1..R-1: ONE table load (W^k) and a depth-<=4 product tree for the
                // powers (<= ~5 ulp), instead of R-1 loads whose prefetch would pin 4(R-1) VGPRs
                const int k = (j + b * P) % NS;
                constexpr int TSTEP = LEN / (NS * R);
                cplx w[R];
                w[1] = tw[k * TSTEP];
                if (INV) w[1].y = -w[1].y;
#pragma unroll
                for (int t = 2; t < R; ++t) {
                    const int hi = (t >= 8) ? 8 : ((t >= 4) ? 4 : 2);   // largest power of two <= t
                    w[t] = (t == hi) ? cmul(w[t / 2], w[t / 2]) : cmul(w[hi], w[t - hi]);
                }
#pragma unroll
                for (int t = 1; t < R; ++t) a[t] = cmul(a[t], w[t]);
            }
            Dft<R, INV>::run(a);
#pragma unroll
            for (int t = 0; t < R; ++t) v[b + t * NB] = a[t];
        }
        // ---- exchange through LDS (not after the last stage)
        if constexpr (S + 1 < PL::NST) {
            int base[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int jb = j + b * P;
                base[b] = (jb / NS) * (NS * R) + (jb % NS);
            }
            exchange_sync<WAVE>();
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < R; ++u) line[lpad(base[b] + u * NS)] = v[b + u * NB].x;
            exchange_sync<WAVE>();
            real re[E];
#pragma unroll
            for (int q = 0; q < E; ++q) re[q] = line[lpad(j + P * q)];
            exchange_sync<WAVE>();
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < R; ++u) line[lpad(base[b] + u * NS)] = v[b + u * NB].y;
            exchange_sync<WAVE>();
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = mkc(re[q], line[lpad(j + P * q)]);
            StageP<PL, S + 1, NS * R, INV, WAVE>::run(v, j, line, tw);
        }
    }
};

// Full line transform.  `line` = this line's LDS buffer (LineBuf<LEN>::STRIDE doubles; unused when
// the plan has one stage).  `tw` = forward table W_LEN^m, m = 0..LEN-1 (global memory).
// Every thread of the workgroup must call this (it contains __syncthreads()).
template <int LEN, bool INV>
__device__ __forceinline__ void line_fft(cplx (&v)[Plan<LEN>::E], int j, real* line, const cplx* __restrict__ tw) {
    StageP<Plan<LEN>, 0, 1, INV, false>::run(v, j, line, tw);
}

// ---- small-footprint plans for the z kernels: E = 8 or 4 complex points per lane.  A line's P = LEN/E
// threads are lanes of one wave, so the exchanges need no barrier (extra stages only cost LDS traffic), and
// the small register footprint leaves room for fused pointwise math.
template <int LEN_, int E_> struct ZPlan;
#define OFDFT_ZPLAN(LEN_, E_, NST_, R0_, R1_, R2_, R3_)                             \
    template <> struct ZPlan<LEN_, E_> {                                           \
        static constexpr int LEN = LEN_;                                           \
        static constexpr int NST = NST_;                                           \
        static constexpr int E = E_;                                               \
        static constexpr int P = LEN_ / E_;                                        \
        static __host__ __device__ constexpr int radix(int s) {                    \
            return s == 0 ? R0_ : (s == 1 ? R1_ : (s == 2 ? R2_ : R3_));           \
        }                                                                          \
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

template <int LEN, int E, bool INV>
__device__ __forceinline__ void wave_line_fft(cplx (&v)[E], int j, real* line, const cplx* __restrict__ tw) {
    static_assert(ZPlan<LEN, E>::P <= 64, "a line must fit one wavefront");
    StageP<ZPlan<LEN, E>, 0, 1, INV, true>::run(v, j, line, tw);
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
