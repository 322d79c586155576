// Real-space and reciprocal-space pointwise kernels + wavefront-shuffle reductions (fp64, gfx950).
// The math follows the closed forms listed in SURVEY.md §8a (reference: src/professad/functionals.py
// and tests/tools_for_tests.py; exact lines cited at each functor).
#pragma once
#include <type_traits>

#include "fastmath.h"
#include "fft_kernels.h"

namespace ofdft {

// host-side fp64 constants are spelled so that they stay exact when the fp32 build compiles with single-precision
// floating literals (-cl-single-precision-constant): long-double literal / integer operands
constexpr double kPi = (double)3.14159265358979323846264338327950288L;
constexpr double kFiveThirds = (double)5 / 3, kFiveSixths = (double)5 / 6;
constexpr real kPiR = (real)kPi;
      // pi in the grid precision (device math on `real` operands)
constexpr real kThird = (real)(1.0L / 3.0L);      // (x / 3 as a product: the quotient by a literal is a full IEEE division sequence on the device)
// Roots and logarithms of constants, spelled out: the device compiler does NOT fold cbrt() / log() / sqrt() of a literal
// (they are library routines there), so `cbrt(3.0 / kPiR)` inside a pointwise function was evaluated at every grid point
// -- 4 cube roots, a logarithm, a square root and 2 quotients per PBE point, about 300 of its 650 instructions.
constexpr real kCbrt9Pi4 = (real)9.570780000627306053141655531512398907L;      // (9 pi^4)^(1/3) = (3 pi^2)^(2/3)
constexpr real kCbrt3OverPi = (real)0.984745021842696541178973376907781690L;   // (3 / pi)^(1/3)
constexpr real kCrs = (real)0.620350490899400016668006812047778167L;           // (3 / (4 pi))^(1/3): rs = kCrs n^(-1/3)
constexpr real kSqrtCrs = (real)0.787623317899743250528011536241166179L;
constexpr real kInvSqrtCrs = (real)1.269642451250142171583074982042715431L;
constexpr real kCbrtPiOver3 = (real)1.015491297563259267239386001491459489L;   // (pi / 3)^(1/3)
constexpr real kPbeGamma = (real)0.031090690869654895034940863712730629L;      // (1 - ln 2) / pi^2
constexpr real kPbeInvGamma = (real)32.16396844291482112072009037084516073L;
constexpr real kLn2 = (real)0.693147180559945309417232121458176568L;
constexpr real kCtf = (real)(0.3L * 9.570780000627306053141655531512398907L);  // C_TF = 0.3 (3 pi^2)^(2/3)
constexpr real kCs2 = (real)(0.25L / 9.570780000627306053141655531512398907L); // s^2 = kCs2 |grad n|^2 / n^(8/3)
constexpr real kCx = (real)(-0.75L * 0.984745021842696541178973376907781690L); // LDA exchange: e_x = kCx n^(4/3)
constexpr int kRedBlocks = 1024;   // grid cap for reducing kernels (partials buffer rows)
constexpr int kRedThreads = 256;
constexpr int kMaxScalars = 28;    // scalars reduced by one kernel (27: the real-space stress sums)

// ---- block reduction of NS scalars; thread 0 writes partial[blockIdx.x * NS + s]
template <int NS>
__device__ __forceinline__ void block_reduce_store(acc_t (&acc)[NS], acc_t* __restrict__ partial) {
    __shared__ acc_t red[kRedThreads / 64][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc_t v = acc[s];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[s] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) red[w][s] = acc[s];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            acc_t t = 0.0;
#pragma unroll
            for (int ww = 0; ww < kRedThreads / 64; ++ww) t += red[ww][s];
            partial[(long long)blockIdx.x * NS + s] = t;
        }
    }
}

// second level: partial[rows][ns] -> out[ns], fixed summation order (bitwise reproducible)
// (host_out: optional pinned, device-visible mirror -- the numbers reach the host without a copy command behind the kernel)
static __global__ __launch_bounds__(kRedThreads) void reduce_partials_kernel(const acc_t* __restrict__ partial, int rows, int ns,
                                                                      acc_t* __restrict__ out, acc_t* __restrict__ host_out = nullptr) {
    const int s = blockIdx.x;
    acc_t acc[1] = {0.0};
    for (int r = threadIdx.x; r < rows; r += kRedThreads) acc[0] += partial[(long long)r * ns + s];
    __shared__ acc_t red[kRedThreads];
    red[threadIdx.x] = acc[0];
    __syncthreads();
    for (int off = kRedThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[s] = red[0];
        if (host_out) host_out[s] = red[0];
    }
}
// Many rows (round 5): the kernel above gives scalar s ONE block that walks its column with a stride of ns doubles -- 80-byte
// strides for the ten combine sums, 17 us at 256^3 (8 192 rows), 129 us at 512^3, 0.5 ms at 1024^3.  First level instead: block b
// sums the rows [b per, (b + 1) per), a CONTIGUOUS range read with unit stride (thread t keeps element t of every run of
// T = ns floor(256 / ns) doubles, i.e. always scalar t mod ns), into mid[b][ns]; the kernel above then reduces the <= 256 rows of
// `mid`.  Fixed order for a given (rows, ns): bitwise reproducible.
constexpr int kRedTwoLevelRows = 2048;     // more rows than this take the two-level form
constexpr int kRedMidRows = 256;           // most first-level blocks (rows of `mid`)
static __global__ __launch_bounds__(kRedThreads) void reduce_rows_kernel(const acc_t* __restrict__ partial, int rows, int ns, int per,
                                                                  acc_t* __restrict__ mid) {
    __shared__ acc_t red[kRedThreads];
    const int rpt = kRedThreads / ns, T = rpt * ns;
    const int r0 = blockIdx.x * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    acc_t a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if ((int)threadIdx.x < T && r1 > r0) {
        const acc_t* p = partial + (long long)r0 * ns;
        const long long n = (long long)(r1 - r0) * ns;
        long long i = threadIdx.x;
        for (; i + 3LL * T < n; i += 4LL * T) {        // four independent loads in flight per thread
            a0 += p[i];
            a1 += p[i + T];
            a2 += p[i + 2LL * T];
            a3 += p[i + 3LL * T];
        }
        for (; i < n; i += T) a0 += p[i];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if ((int)threadIdx.x < ns) {
        acc_t t = 0.0;
        for (int k = 0; k < rpt; ++k) t += red[threadIdx.x + k * ns];
        mid[(long long)blockIdx.x * ns + threadIdx.x] = t;
    }
}

// sum(a) or sum(a^2).  Round 5: 16-byte accesses, four of them in flight per thread -- with one 8-byte (fp32: a pair) load per
// thread and iteration a CU of the 1024-block grid had 8 KB in flight and the kernel streamed 4.0 TB/s at 1024^3 fp32 (0.49 of
// the peak), 5.3 at 256^3 fp64.  (arrays that are not 16-byte aligned -- a view of a larger tensor -- take pairs as before)
template <bool SQUARE>
__global__ __launch_bounds__(kRedThreads) void sum_kernel(const real* __restrict__ a, long long n,
                                                          acc_t* __restrict__ partial) {
    acc_t acc[1] = {0.0};
    constexpr int NV = 16 / (int)sizeof(real);                // reals per 16-byte access
    typedef real vec_t __attribute__((ext_vector_type(NV)));
    const long long stride = (long long)gridDim.x * blockDim.x, gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    auto fold = [&](const vec_t& v) {
        acc_t t = 0.0;
#pragma unroll
        for (int e = 0; e < NV; ++e) t += SQUARE ? (acc_t)v[e] * (acc_t)v[e] : (acc_t)v[e];
        return t;
    };
    long long done = 0;
    if ((reinterpret_cast<unsigned long long>(a) & 15ull) == 0) {
        const vec_t* av = reinterpret_cast<const vec_t*>(a);
        const long long nv = n / NV;
        acc_t b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        long long i = gid;
        for (; i + 3 * stride < nv; i += 4 * stride) {
            const vec_t v0 = av[i], v1 = av[i + stride], v2 = av[i + 2 * stride], v3 = av[i + 3 * stride];
            b0 += fold(v0);
            b1 += fold(v1);
            b2 += fold(v2);
            b3 += fold(v3);
        }
        for (; i < nv; i += stride) b0 += fold(av[i]);
        acc[0] = (b0 + b1) + (b2 + b3);
        done = nv * NV;
    }
    for (long long i = done + gid; i < n; i += stride) acc[0] += SQUARE ? (acc_t)a[i] * (acc_t)a[i] : (acc_t)a[i];
    block_reduce_store<1>(acc, partial);
}

// elementwise unary maps (prep of FFT inputs); 16-byte accesses, scalar tail for odd sizes
enum { MAP_SQRT = 0, MAP_POW = 1, MAP_SCALE_SQ = 2 };
template <int OP> __device__ __forceinline__ real map_op(real x, real p) {
    if (OP == MAP_SQRT) return (x != 0.0) ? sqrt(x) : 0.0;        // functionals.py:242-243
    if (OP == MAP_POW) return pow(x, p);
    return p * x * x;                                               // n = c chi^2, system.py:834
}
// (p_dev: the parameter read from device memory -- the closure scale c of n = c chi^2 left there by closure_scale_reduce_kernel)
template <int OP>
__global__ void map_kernel(const real* __restrict__ a, real* __restrict__ out, long long n, real p, const acc_t* __restrict__ p_dev = nullptr) {
    if (p_dev) p = (real)p_dev[0];
    const long long n2 = n >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const cplx x = reinterpret_cast<const cplx*>(a)[i];
        reinterpret_cast<cplx*>(out)[i] = mkc(map_op<OP>(x.x, p), map_op<OP>(x.y, p));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = map_op<OP>(a[n - 1], p);
}

// WGC99 real-space inputs: A = n^e, B = A theta, C = A theta^2 / 2 (functionals.py:974-981)
static __global__ void wgc_prep_kernel(const real* __restrict__ n, real* __restrict__ A, real* __restrict__ B,
                                real* __restrict__ C, long long npts, real expo, real nref) {
    const long long n2 = npts >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const cplx d = reinterpret_cast<const cplx*>(n)[i];
        const real t0 = d.x - nref, t1 = d.y - nref, a0 = pow(d.x, expo), a1 = pow(d.y, expo);
        reinterpret_cast<cplx*>(A)[i] = mkc(a0, a1);
        reinterpret_cast<cplx*>(B)[i] = mkc(a0 * t0, a1 * t1);
        reinterpret_cast<cplx*>(C)[i] = mkc(0.5 * a0 * t0 * t0, 0.5 * a1 * t1 * t1);
    }
    if ((npts & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long i = npts - 1;
        const real d = n[i], th = d - nref, a = pow(d, expo);
        A[i] = a;
        B[i] = a * th;
        C[i] = 0.5 * a * th * th;
    }
}

// ---- reciprocal space -------------------------------------------------------------------------
struct KGeom {
    SpecGeom g;
    real b[9];   // b = 2 pi inv(box^T), row-major (functional_tools.py:149)
    int y0;        // first global y index held by this rank in the x-pass (y-slab) geometry; 0 on one GPU
    int n1g;       // global extent of axis 1 (= g.n1 on one GPU)
};

__device__ __forceinline__ real ifreq(int i, int n) { return (real)(i <= n / 2 ? i : i - n); }   // :152-154

__device__ __forceinline__ void kvec(const KGeom& kg, long long i, real& kx, real& ky, real& kz, real& k2) {
    int x, y, z;
    spec_decode(kg.g, i, x, y, z);
    const real fa = ifreq(x, kg.g.n0), fb = ifreq(y + kg.y0, kg.n1g), fc = (real)z;     // :155 rfftfreq
    kx = fa * kg.b[0] + fb * kg.b[3] + fc * kg.b[6];                                  // :158-160
    ky = fa * kg.b[1] + fb * kg.b[4] + fc * kg.b[7];
    kz = fa * kg.b[2] + fb * kg.b[5] + fc * kg.b[8];
    k2 = kx * kx + ky * ky + kz * kz;
}

// 1/G^-1(eta) - 3 eta^2 - 1 (functionals.py:617-628,648), fp64: the reference's own form, with the lean logarithm and
// reciprocal of fastmath.h (the library log + two IEEE quotients were ~200 fp64 instructions per k-point of the
// Wang-Teter x pass; this is ~60)
__device__ __forceinline__ double lindhard_shape(double eta) {
    double ginv;
    if (eta == 0.0) ginv = 1.0;
    else if (eta == 1.0) ginv = 0.5;
    else ginv = 0.5 + ((1.0 - eta * eta) * 0.25 * fm::rcp(eta)) * fm::log(fabs((1.0 + eta) * fm::rcp(1.0 - eta)));
    return fm::rcp(ginv) - 3.0 * eta * eta - 1.0;
}
// fp32 (round 5): the direct form cancels twice -- G = 0.5 + (-0.5 + O(eta^-2)) for large eta, then 1/G against 3 eta^2; for
// small eta 1/G - 1 against O(eta^2) -- which left 5e-4 of the Wang-Teter energy of a rough density, so rounds 3-4 evaluated
// the fp64 form at every k-point of the fp32 build: 2.3 of the 5.8 ms of the 1024^3 x pass (538 M k-points) were its issue
// time.  Cancellation-free instead: with c_j = 1 / (4 j^2 - 1) the logarithm's series gives
//   eta < 1:  G = 1 - S,  S = sum_{j>=1} c_j eta^(2j)              f = S / (1 - S) - 3 eta^2
//   eta > 1:  G = sum_{j>=1} c_j eta^(-2j) = (z / 3) T(z), z = eta^-2  f = -3 V / R - 1,  R = sum_j c_(j+1) z^j, V = sum_j c_(j+2) z^j
// (S = z R with z = eta^2: ONE Horner recursion serves both sides).  12 terms are exact to fp32 rounding for z <= 0.36; in
// between, 0.6 < eta < 5/3, the direct form loses at most a digit (|G| >= 0.13): max 2e-6, rms 3e-7 relative over the band,
// 1.5e-7 outside (numpy float32 emulation against 80-bit arithmetic; tests/test_gpu_f32.py pins it on the device).
__device__ __forceinline__ float lindhard_shape(float eta) {
    const float e2 = eta * eta;
    const float z = eta < 1.0f ? e2 : __builtin_amdgcn_rcpf(e2);
    constexpr float c[14] = {0.0f, 1.0f / 3.0f, 1.0f / 15.0f, 1.0f / 35.0f, 1.0f / 63.0f, 1.0f / 99.0f, 1.0f / 143.0f, 1.0f / 195.0f,
                             1.0f / 255.0f, 1.0f / 323.0f, 1.0f / 399.0f, 1.0f / 483.0f, 1.0f / 575.0f, 1.0f / 675.0f};
    float r = c[12], v = c[13];
#pragma unroll
    for (int j = 11; j >= 1; --j) {
        r = __builtin_fmaf(r, z, c[j]);           // R = c_1 + c_2 z + ... + c_12 z^11
        v = __builtin_fmaf(v, z, c[j + 1]);       // V = c_2 + c_3 z + ... + c_13 z^11
    }
    float f;
    if (eta < 1.0f) {
        const float s = z * r;
        f = __builtin_fmaf(s, __builtin_amdgcn_rcpf(1.0f - s), -3.0f * e2);
    } else {
        f = __builtin_fmaf(-3.0f * v, __builtin_amdgcn_rcpf(r), -1.0f);
    }
    if (eta > 0.6f && eta < (5.0f / 3.0f)) {
        const float lg = 0.69314718055994530942f * __builtin_amdgcn_logf(fabsf((1.0f + eta) * __builtin_amdgcn_rcpf(1.0f - eta)));
        const float g = __builtin_fmaf((1.0f - e2) * 0.25f * __builtin_amdgcn_rcpf(eta), lg, 0.5f);
        f = eta == 1.0f ? -2.0f : __builtin_amdgcn_rcpf(g) - 3.0f * e2 - 1.0f;
    }
    return f;
}

enum { SPEC_HARTREE = 0, SPEC_LAPLACE = 1, SPEC_LINDHARD = 2 };
// out = in * f(k); p0,p1 parameters (LINDHARD: p0 = prefactor, p1 = 1/(2 kF))
template <int OP>
__global__ void spec_scale_kernel(const cplx* __restrict__ in, cplx* __restrict__ out, KGeom kg, real p0, real p1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        real kx, ky, kz, k2;
        kvec(kg, i, kx, ky, kz, k2);
        real f;
        if (OP == SPEC_HARTREE) f = (k2 != 0.0) ? 4.0 * kPiR / k2 : 0.0;              // functionals.py:67-70
        else if (OP == SPEC_LAPLACE) f = -k2;                                       // functional_tools.py:227
        else f = p0 * lindhard_shape((k2 != 0.0) ? sqrt(k2) * p1 : 0.0);            // functionals.py:637-638,648
        const cplx a = in[i];
        out[i] = mkc(a.x * f, a.y * f);
    }
}

// acc += f * k^2 * in   (adds -f * Laplacian in reciprocal space)
static __global__ void spec_add_lap_kernel(const cplx* __restrict__ in, cplx* __restrict__ acc, KGeom kg, real f) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        real kx, ky, kz, k2;
        kvec(kg, i, kx, ky, kz, k2);
        const cplx a = in[i], b = acc[i];
        acc[i] = mkc(b.x + f * k2 * a.x, b.y + f * k2 * a.y);
    }
}

// g_j = i k_j * in (functional_tools.py:183)
static __global__ void spec_grad_kernel(const cplx* __restrict__ in, cplx* __restrict__ gx, cplx* __restrict__ gy,
                                 cplx* __restrict__ gz, KGeom kg) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        real kx, ky, kz, k2;
        kvec(kg, i, kx, ky, kz, k2);
        const cplx a = in[i];
        gx[i] = mkc(-kx * a.y, kx * a.x);
        gy[i] = mkc(-ky * a.y, ky * a.x);
        gz[i] = mkc(-kz * a.y, kz * a.x);
    }
}

// out = sum_j i k_j f_j
static __global__ void spec_div_kernel(const cplx* __restrict__ fx, const cplx* __restrict__ fy, const cplx* __restrict__ fz,
                                cplx* __restrict__ out, KGeom kg) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        real kx, ky, kz, k2;
        kvec(kg, i, kx, ky, kz, k2);
        const cplx a = fx[i], b = fy[i], c = fz[i];
        out[i] = mkc(-(kx * a.y + ky * b.y + kz * c.y), kx * a.x + ky * b.x + kz * c.x);
    }
}

// WGC99 spectral mixing, in place: (A,B,C) -> (w0 A + K1 B + K2 C, K1 A + K3 B, K2 A)   SURVEY §8a-8
static __global__ void spec_wgc_mix_kernel(cplx* __restrict__ A, cplx* __restrict__ B, cplx* __restrict__ C,
                                    const cplx* __restrict__ t01, const real* __restrict__ t2v, real ck, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const cplx a = A[i], b = B[i], c = C[i];
        const cplx p = t01[i];                                    // (w0, K1) per k-point; K2 in its own array; K3 = K2 + ck K1
        const real t0 = p.x, t1 = p.y, t2 = t2v[i], t3 = t2 + ck * t1;
        A[i] = mkc(t0 * a.x + t1 * b.x + t2 * c.x, t0 * a.y + t1 * b.y + t2 * c.y);
        B[i] = mkc(t1 * a.x + t3 * b.x, t1 * a.y + t3 * b.y);
        C[i] = mkc(t2 * a.x, t2 * a.y);
    }
}

// WGC99 kernel tables on the k grid (functionals.py:845-939 for w,w',w''; :968-972 for T,K1,K2,K3).  The series is
// summed in fp64 in both builds (once per cell); only the stored tables take the grid precision.
struct WgcSeries {
    double u, v, c1, c2;      // homogeneous-solution constants
    double gamma, nref, pref; // pref = 20 nref^(5/3-alpha-beta)
    double inv2kf;
    const double* ca;         // [nt] A_i / ((u+2i)^2 - v)
    const double* cb;         // [nt] B_i / ((u-2i)^2 - v)
    int nt;
};

// where the k-point tables live when the x pass works on exchange buffers (slab-decomposed path): x-major records
// [x][ main (b, yl, kin) | planes (plane, yl) ] of arr_sz elements, the one-array form of the exchange layout
// -- chunk-major like the buffers (engine_ctx.h: XchgChunks): chunk k holds the kz blocks [kb[k], kb[k + 1]) of every x, the
// remainder planes ride with the last chunk; n0g = x extent (records per chunk)
struct TabMap { int on, nyl, nzm; long long arr_sz; int nch = 1, n0g = 0, nrem = 0; int kb[17] = {0}; };
__device__ __forceinline__ long long tabmap_index(const TabMap& tm, int x, int y, int z) {
    const int b = z < tm.nzm ? (z >> 3) : (tm.nzm >> 3);            // planes: behind the last block
    int k = tm.nch - 1;
    if (z < tm.nzm)
        for (k = 0; k < tm.nch - 1 && b >= tm.kb[k + 1]; ++k) {}
    const int kb0 = tm.kb[k], nbk = tm.kb[k + 1] - kb0;
    const long long arr = ((long long)nbk * 8 + (k == tm.nch - 1 ? tm.nrem : 0)) * tm.nyl;      // record size of chunk k
    const long long base = (long long)tm.n0g * tm.nyl * 8 * kb0;
    const long long in = z < tm.nzm ? (((long long)(b - kb0) * tm.nyl + y) * 8 + (z & 7))
                                    : ((long long)nbk * 8 * tm.nyl + (long long)(z - tm.nzm) * tm.nyl + y);
    return base + x * arr + in;
}

// w, w', w'' (and w''' when THIRD) of the WGC99 kernel at eta != 0, before the prefactor (functionals.py:845-939):
// homogeneous solution + particular series by Horner in eta^2 (inside) or eta^-2 (outside)
template <bool THIRD>
__device__ __forceinline__ void wgc_series(double eta, const WgcSeries& s, double& w0, double& w1, double& w2, double& w3) {
    const bool inner = eta <= 1.0;
    const bool on = (s.u >= 0.0) ? inner : !inner;
    const double C1 = on ? s.c1 : 0.0, C2 = on ? s.c2 : 0.0;
    const double le = log(eta);
    double H0, H1, H2, H3 = 0.0;
    if (s.v > 0.0) {
        const double rv = sqrt(s.v), x = s.u + rv, y = s.u - rv;
        const double px = pow(eta, x - 2.0), py = pow(eta, y - 2.0);
        H0 = (C1 * px + C2 * py) * eta * eta;
        H1 = (C1 * x * px + C2 * y * py) * eta;
        H2 = C1 * x * (x - 1.0) * px + C2 * y * (y - 1.0) * py;
        if (THIRD) H3 = (C1 * x * (x - 1.0) * (x - 2.0) * px + C2 * y * (y - 1.0) * (y - 2.0) * py) / eta;
    } else if (s.v == 0.0) {
        const double pu2 = pow(eta, s.u - 2.0), pu1 = pu2 * eta, pu = pu1 * eta;
        H0 = pu * (C2 * le + C1);
        H1 = C2 * pu1 * (1.0 + s.u * le) + C1 * s.u * pu1;
        H2 = C2 * ((s.u - 1.0) * pu2 * (1.0 + s.u * le) + pu2) + C1 * s.u * (s.u - 1.0) * pu2;
        // third derivative of eta^u (C2 ln eta + C1)
        if (THIRD) {
            const double a3 = s.u * (s.u - 1.0) * (s.u - 2.0), b3 = 3.0 * s.u * s.u - 6.0 * s.u + 2.0;
            H3 = pu2 / eta * (C2 * (a3 * le + b3) + C1 * a3);
        }
    } else {
        const double rv = sqrt(-s.v);
        const double tc = cos(rv * le), ts = sin(rv * le);
        const double p = s.u * tc - rv * ts, q = s.u * ts + rv * tc;
        const double pu2 = pow(eta, s.u - 2.0), pu1 = pu2 * eta, pu = pu1 * eta;
        H0 = pu * (C1 * tc + C2 * ts);
        H1 = pu1 * (C1 * p + C2 * q);
        H2 = pu2 * ((s.u - 1.0) * (C1 * p + C2 * q) + rv * (C2 * p - C1 * q));
        if (THIRD) {      // H = Re[(C1 - i C2) eta^z], z = u + i rv:  H''' = Re[(C1 - i C2) z (z-1) (z-2) eta^(z-3)]
            const double zr = s.u, zi = rv;
            double ar = zr * (zr - 1.0) - zi * zi, ai = zi * (2.0 * zr - 1.0);           // z (z-1)
            const double br = ar * (zr - 2.0) - ai * zi, bi = ar * zi + ai * (zr - 2.0);  // ... (z-2)
            const double dr = C1 * br + C2 * bi, di = C1 * bi - C2 * br;                  // (C1 - i C2) * that
            H3 = pu2 / eta * (dr * tc - di * ts);
        }
    }
    const double x = inner ? eta * eta : 1.0 / (eta * eta);
    double P0 = 0.0, P1 = 0.0, P2 = 0.0, P3 = 0.0;
    for (int t = s.nt - 1; t >= 0; --t) {
        const double ti = 2.0 * t;
        const double c = inner ? s.cb[t] : s.ca[t];
        const double d1 = inner ? ti * c : -ti * c;
        const double d2 = inner ? ti * (ti - 1.0) * c : ti * (ti + 1.0) * c;
        P0 = P0 * x + c;
        P1 = P1 * x + d1;
        P2 = P2 * x + d2;
        if (THIRD) P3 = P3 * x + (inner ? ti * (ti - 1.0) * (ti - 2.0) * c : -ti * (ti + 1.0) * (ti + 2.0) * c);
    }
    P1 /= eta;
    P2 /= eta * eta;
    w0 = H0 + P0;
    w1 = H1 + P1;
    w2 = H2 + P2;
    w3 = THIRD ? H3 + P3 / (eta * eta * eta) : 0.0;
}

// Table layout (round 4): (w0, K1) as one 2-real entry per k-point + K2 in a second array -- 3 reals per k-point.  The
// fourth coefficient of functionals.py:968-972 is not independent: K3 = K2 + ((3 - gamma) / (3 n_ref)) K1 exactly (from the
// definitions below), so the mixes form it in registers (-25 % of the table bytes every fused x pass of the nonlocal chain reads).
static __global__ void wgc_table_kernel(cplx* __restrict__ t01, real* __restrict__ t2, KGeom kg, WgcSeries s, TabMap tm) {
    for (long long ii = (long long)blockIdx.x * blockDim.x + threadIdx.x; ii < kg.g.total; ii += (long long)gridDim.x * blockDim.x) {
        real kx, ky, kz, k2;
        kvec(kg, ii, kx, ky, kz, k2);
        long long i = ii;
        if (tm.on) {
            int x, y, z;
            spec_decode(kg.g, ii, x, y, z);
            i = tabmap_index(tm, x, y, z);
        }
        const double eta = (k2 != 0.0) ? sqrt((double)k2) * s.inv2kf : 0.0;
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3;
        if (eta != 0.0) wgc_series<false>(eta, s, w0, w1, w2, w3);
        w0 *= s.pref;
        w1 *= s.pref;
        w2 *= s.pref;
        // K1 = -eta w' / (6 n_ref), K2 = (eta^2 w'' + (7 - gamma) eta w') / (36 n_ref^2);
        // (K3 = (eta^2 w'' + (1 + gamma) eta w') / (36 n_ref^2) = K2 + (3 - gamma) K1 / (3 n_ref))
        t01[i] = mkc((real)w0, (real)(-eta * w1 / (6.0 * s.nref)));
        t2[i] = (real)((eta * eta * w2 + (7.0 - s.gamma) * eta * w1) / (36.0 * s.nref * s.nref));
    }
}

// ---- mixing functors of the fused x pass (see xfused_kernel in fft_kernels.h) ----------------------
__device__ __forceinline__ void kvec_xyz(const KGeom& kg, int x, int y, int z, real& kx, real& ky, real& kz,
                                         real& k2) {
    const real fa = ifreq(x, kg.g.n0), fb = ifreq(y + kg.y0, kg.n1g), fc = (real)z;
    kx = fa * kg.b[0] + fb * kg.b[3] + fc * kg.b[6];
    ky = fa * kg.b[1] + fb * kg.b[4] + fc * kg.b[7];
    kz = fa * kg.b[2] + fb * kg.b[5] + fc * kg.b[8];
    k2 = kx * kx + ky * ky + kz * kz;
}

// n^ -> [v_H^ = 4 pi/k^2 n^] [, i k_x n^, i k_y n^, i k_z n^]
template <bool HAS_H, bool HAS_G> struct MixDensity {
    KGeom kg;
    static __device__ __forceinline__ constexpr bool imag(int o) { return HAS_H ? o >= 1 : true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return imag(O); }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int y, int z, long long, unsigned) const {
        real kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        if (HAS_H && O == 0) return (k2 != 0.0) ? 4.0 * kPiR / k2 : 0.0;
        constexpr int c = O - (HAS_H ? 1 : 0);
        return c == 0 ? kx : (c == 1 ? ky : kz);
    }
};

// split-derivative form: n^ -> [v_H^], i f_a n^  with the INTEGER frequency f_a along x (the Cartesian gradient is assembled
// from the three index derivatives in zpbe2_kernel)
template <bool HAS_H, bool HAS_L = false> struct MixDensityA {
    KGeom kg;
    // outputs: [v_H^ (real coefficient)], i f_a n^ (imaginary), [-k^2 n^ (real): the Laplacian of n for the q-dependent
    // Pauli-Gaussian members, functional_tools.py:271-287]
    static __device__ __forceinline__ constexpr bool imag(int o) { return o == (HAS_H ? 1 : 0); }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return imag(O); }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int y, int z, long long, unsigned) const {
        if constexpr (O == (HAS_H ? 1 : 0)) {
            return ifreq(x, kg.g.n0);
        } else {
            real kx, ky, kz, k2;
            kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
            if (HAS_H && O == 0) return (k2 != 0.0) ? 4.0 * kPiR / k2 : 0.0;
            return -k2;
        }
    }
};
// G^ -> i f_a G^
struct MixDerivA {
    KGeom kg;
    static __device__ __forceinline__ constexpr bool imag(int) { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return true; }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int, int, long long, unsigned) const { return ifreq(x, kg.g.n0); }
};
// (G_a^, L^) -> i f_a G_a^ + (k^2 / 2) L^ : the x part of the divergence plus the Laplacian of df/d(lap n), folded into the
// quantity the combine kernel subtracts twice (v += df/dn - 2 div + lap(df/dL), tools_for_tests.py:86-118)
struct MixDerivAL {
    KGeom kg;
    static __device__ __forceinline__ constexpr bool imag(int) { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return I == 0; }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int y, int z, long long, unsigned) const {
        if constexpr (I == 0) {
            return ifreq(x, kg.g.n0);
        } else {
            real kx, ky, kz, k2;
            kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
            return 0.5 * k2;
        }
    }
};

// one spectrum times a real f(k): OP as spec_scale_kernel
template <int OP> struct MixScale {
    KGeom kg;
    real p0, p1;
    static __device__ __forceinline__ constexpr bool imag(int) { return false; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return imag(O); }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int y, int z, long long, unsigned) const {
        real kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        if (OP == SPEC_HARTREE) return (k2 != 0.0) ? 4.0 * kPiR / k2 : 0.0;
        if (OP == SPEC_LAPLACE) return -k2;
        return p0 * lindhard_shape((k2 != 0.0) ? sqrt(k2) * p1 : 0.0);
    }
};

// (F_x^, F_y^, F_z^) -> sum_j i k_j F_j^
struct MixDiv {
    KGeom kg;
    static __device__ __forceinline__ constexpr bool imag(int) { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return true; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return imag(O); }
    template <int O, int I>
    __device__ __forceinline__ real coef(int x, int y, int z, long long, unsigned) const {
        real kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        return I == 0 ? kx : (I == 1 ? ky : kz);
    }
};

// WGC99: (A^,B^,C^) -> (w0 A + K1 B + K2 C, K1 A + K3 B, K2 A) with tables in the spectrum layout
struct MixWgc {
    static constexpr bool kTables = true;     // coef() is a load: the fused x pass requests a half's worth in one batch
    const cplx* t01;     // (w0, K1) per k-point, spectrum order
    const real* t2;      // K2 per k-point
    real ck;             // K3 = K2 + ck K1, ck = (3 - gamma) / (3 n_ref)
    static __device__ __forceinline__ constexpr bool imag(int) { return false; }
    // symmetric pattern: (0,0) w0; O+I=1 K1; (0,2),(2,0) K2; (1,1) K3; the rest absent
    template <int O, int I> static __device__ __forceinline__ constexpr bool present() { return O + I <= 2; }
    template <int O, int I> static __device__ __forceinline__ constexpr bool imag_oi() { return false; }
    template <int O, int I>
    __device__ __forceinline__ real coef(int, int, int, long long uoff, unsigned loff) const {
        // identical loads of one k-point are merged by the compiler
        if constexpr (O + I <= 1) {
            const cplx pr = buf_load_c(t01 + uoff, loff * (unsigned)sizeof(cplx));
            return (O + I == 0) ? pr.x : pr.y;
        } else {
            const real k2 = buf_load_d(t2 + uoff, loff * (unsigned)sizeof(real));
            if constexpr (O == 1) {
                const cplx pr = buf_load_c(t01 + uoff, loff * (unsigned)sizeof(cplx));
                return k2 + ck * pr.y;
            } else {
                return k2;
            }
        }
    }
    // wave-local / cross-wave x pass (xwave.h, xcross.h): the three entries of a k-point by one 2-real and one 1-real load,
    // requested with the data
    static constexpr int kTableReals = 3;
    __device__ __forceinline__ void fetch(real (&cf)[3], long long uoff, unsigned loff, bool valid) const {
#ifdef OFDFT_WGC_TABLE_PROBE
        // TIMING PROBE ONLY (tools: ab_generic.sh, profiles/r05_ab_wgc_table_probe.jsonl): no table traffic at all -- the upper bound
        // of what any cheaper source of the coefficients (an L2-resident |k|^2-class table, an interleaved layout) could buy.  Wrong numbers.
        cf[0] = (real)1.0; cf[1] = (real)0.5; cf[2] = valid ? (real)0.25 : (real)0.0;
        (void)uoff; (void)loff;
#else
        const cplx p0 = valid ? buf_load_c(t01 + uoff, loff * (unsigned)sizeof(cplx)) : mkc(0.0, 0.0);
        const real k2 = valid ? buf_load_d(t2 + uoff, loff * (unsigned)sizeof(real)) : (real)0.0;
        cf[0] = p0.x; cf[1] = p0.y; cf[2] = k2;       // w0, K1, K2
#endif
    }
    __device__ __forceinline__ void apply(cplx (&o)[3], const cplx (&in)[3], const real (&cf)[3]) const {
        const real k3 = cf[2] + ck * cf[1];
        o[0] = mkc(cf[0] * in[0].x + cf[1] * in[1].x + cf[2] * in[2].x, cf[0] * in[0].y + cf[1] * in[1].y + cf[2] * in[2].y);
        o[1] = mkc(cf[1] * in[0].x + k3 * in[1].x, cf[1] * in[0].y + k3 * in[1].y);
        o[2] = mkc(cf[2] * in[0].x, cf[2] * in[0].y);
    }
};

// Round 5: cells with orthogonal axes have |k|(x) = |k|(n0 - x) along a line, so the entry of x > n0 / 2 is READ at n0 - x (the
// table keeps its spectrum-order layout; the upper half is simply never touched): both uses of an entry fall into the same
// tile of the cross-wave x pass, moments apart -- the second is a cache hit, and the pass' table traffic from HBM halves
// (traffic / algorithmic bytes of xfused_wgc 1.33 -> 1.17; the pass 23.5 -> 22.5 ps per point at 256^3 fp64, 29.5 -> 26.7 at
// 512-point lines, 17.1 -> 16.4 fp32; upper bound with NO table loads at all: 20.5 / 22.7 / 12.1, profiles/r05_ab_wgc_table_probe.jsonl).
// Its own type: the kernels that do not fold -- and cells with skewed axes -- keep the plain form and its wave-uniform addressing.
struct MixWgcFold : MixWgc {
    int fold_n0;         // n0
};

// ---- XC pointwise math -------------------------------------------------------------------------
// PW92 eps_c(rs) and d eps_c / d rs (functionals.py:1524-1530; tests/tools_for_tests.py:136-144)
__device__ __forceinline__ void pw92(real rs, real& eps, real& deps_drs) {
    const real A = 0.0310907, a1 = 0.2137, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    const real sr = sqrt(rs);
    const real zeta = 2.0 * A * (b1 * sr + b2 * rs + b3 * rs * sr + b4 * rs * rs);
    const real izeta = 1.0 / zeta;
    const real lg = log(1.0 + izeta);
    eps = -2.0 * A * (1.0 + a1 * rs) * lg;
    const real dzeta = 2.0 * A * (0.5 * b1 / sr + b2 + 1.5 * b3 * sr + 2.0 * b4 * rs);
    deps_drs = -2.0 * A * a1 * lg + 2.0 * A * (1.0 + a1 * rs) * dzeta * izeta / (zeta + 1.0);
}
// the same from sr = sqrt(rs) and isr = 1 / sr, with the lean transcendentals of fastmath.h: one reciprocal (of
// zeta (zeta + 1)) serves both quotients, log(1 + 1/zeta) = log((zeta + 1) / zeta)
__device__ __forceinline__ void pw92_roots(real sr, real isr, real& eps, real& deps_drs) {
    const real A = 0.0310907, a1 = 0.2137, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    const real rs = sr * sr;
    const real zeta = 2.0 * A * (sr * (b1 + sr * (b2 + sr * (b3 + sr * b4))));
    const real zp1 = zeta + 1.0;
    const real iz2 = fm::rcp(zeta * zp1);            // 1 / (zeta (zeta + 1))
    const real lg = fm::log(zp1 * zp1 * iz2);        // log((zeta + 1) / zeta)
    const real pre = 2.0 * A * (1.0 + a1 * rs);
    eps = -pre * lg;
    const real dzeta = 2.0 * A * (0.5 * b1 * isr + b2 + 1.5 * b3 * sr + 2.0 * b4 * rs);
    deps_drs = -2.0 * A * a1 * lg + pre * dzeta * iz2;
}

struct XcLocal { real ex, vx, ec, vc; };   // energy densities (per volume) and potentials

// LDA exchange + one of PZ / PW / Chachiyo correlation (functionals.py:1510-1537; tools_for_tests.py:121-152)
__device__ __forceinline__ XcLocal lda_point(real n, unsigned mask, const fm::Roots<real>& q) {
    XcLocal r = {0.0, 0.0, 0.0, 0.0};
    const real n13 = q.n13;
    if (mask & (1u << 6)) {
        r.ex = kCx * n13 * n;
        r.vx = (4.0 / 3.0) * kCx * n13;
    }
    if (mask & ((1u << 7) | (1u << 8) | (1u << 9))) {
        const real rs = kCrs * q.inv13;                         // (3 / (4 pi n))^(1/3)
        if (mask & (1u << 7)) {
            const real gm = -0.1423, b1 = 1.0529, b2 = 0.3334, A = 0.0311, B = -0.048, C = 0.002, D = -0.0116;
            real eps, v;
            if (rs < 1.0) {
                const real lr = fm::log(rs);
                eps = A * lr + B + C * rs * lr + D * rs;
                v = lr * (A + (2.0 / 3.0) * C * rs) + (B - A / 3.0) + rs * ((2.0 * D - C) / 3.0);
            } else {
                const real sr = kSqrtCrs * q.y, iden = fm::rcp(1.0 + b1 * sr + b2 * rs);
                eps = gm * iden;
                v = gm * (1.0 + (7.0 / 6.0) * b1 * sr + (4.0 / 3.0) * b2 * rs) * iden * iden;
            }
            r.ec += eps * n;
            r.vc += v;
        }
        if (mask & (1u << 8)) {
            real eps, d;
            pw92_roots(kSqrtCrs * q.y, kInvSqrtCrs * (n * q.inv13 * q.inv13 * q.y), eps, d);
            r.ec += eps * n;
            r.vc += eps - rs * kThird * d;
        }
        if (mask & (1u << 9)) {
            const real a = (kLn2 - 1.0) / (2.0 * kPiR * kPiR), b = 20.4562557;
            const real irs = fm::rcp(rs);
            const real arg = 1.0 + b * irs + b * irs * irs;
            const real eps = a * fm::log(arg);
            const real d = a * fm::rcp(arg) * (-b * irs * irs - 2.0 * b * irs * irs * irs);
            r.ec += eps * n;
            r.vc += eps - rs * kThird * d;
        }
    }
    return r;
}

__device__ __forceinline__ XcLocal lda_point(real n, unsigned mask) { return lda_point(n, mask, fm::roots(n)); }

struct PbePoint { real fx, fc, fk, dfdn, dfdg; };
// which GGA pieces a pass evaluates: PBE exchange / correlation, and the Pauli part of a GGA kinetic functional
// (kkind 0: LuoKarasievTrickey F = 1/cosh(1.3 s), functionals.py:309-333; 1: Pauli-Gaussian F = exp(-mu s^2), :336-403)
struct GgaSel { int x, c, k, kkind; real kmu, kbeta, klambda, ksigma; };
constexpr int kPbeScalars = 3;     // energy sums of a GGA pass: exchange, correlation, kinetic

// PBE x and c: energy density f, df/dn, df/d|grad n|^2 (functionals.py:1597-1618; tools_for_tests.py:155-207)
// This is the hot pointwise function of the evaluation (the GGA mid-stage kernel is bound by fp64 vector issue): every
// root of n comes from ONE n^(-1/6) (fm::roots), every quotient from a shared 5-instruction reciprocal, log / exp
// from fastmath.h -- about 230 fp64 instructions per point instead of about 600 with the library functions.
__device__ __forceinline__ PbePoint pbe_point(real n, real gn2, const GgaSel& sel) {
    PbePoint r = {0.0, 0.0, 0.0, 0.0, 0.0};
    const fm::Roots<real> q = fm::roots(n);
    const real n13 = q.n13;
    const real inv_n = q.inv_n;
    const bool do_x = sel.x != 0, do_c = sel.c != 0;
    const real n83i = inv_n * inv_n * q.inv13 * q.inv13;                // n^(-8/3)
    if (sel.k) {
        // f = tau_TF F(s^2), tau_TF = C_TF n^(5/3), s^2 = |grad n|^2 / (4 (3 pi^2)^(2/3) n^(8/3))  (functional_tools.py:230-268)
        const real ctf = kCtf, cs = kCs2;
        const real s2 = cs * gn2 * n83i;
        const real tau = ctf * n13 * n13 * n;
        real F, dF;                       // F and dF / d(s^2)
        if (sel.kkind == 0) {
            const real a = 1.3, s = fmin(sqrt(s2), 100.0);                  // clamp as functionals.py:330
            const real ch = cosh(a * s);
            F = 1.0 / ch;
            dF = (s > 1e-8 && s < 100.0) ? -a * tanh(a * s) * F / (2.0 * s) : (s < 100.0 ? -0.5 * a * a : 0.0);
        } else {
            F = fm::exp(-sel.kmu * s2);
            dF = -sel.kmu * F;
        }
        r.fk = tau * F;
        r.dfdn += (5.0 / 3.0) * tau * inv_n * F + tau * dF * (-(8.0 / 3.0) * s2 * inv_n);
        r.dfdg += tau * dF * cs * n83i;
    }
    if (do_x) {
        const real kappa = 0.804, mu = 0.066725 * kPiR * kPiR / 3.0;
        const real ex = kCx * n13;
        const real cs = kCs2;                                             // 0.25 (3 pi^2)^(-2/3)
        const real s2 = cs * gn2 * n83i;
        const real iden = fm::rcp(1.0 + (mu / kappa) * s2);
        const real Fx = 1.0 + kappa - kappa * iden;
        const real dF = mu * iden * iden;
        r.fx = Fx * ex * n;
        r.dfdn += Fx * (4.0 / 3.0) * ex + dF * (-(8.0 / 3.0) * s2 * inv_n) * ex * n;
        r.dfdg += dF * cs * n83i * ex * n;
    }
    if (do_c) {
        const real beta = 0.066725, gam = kPbeGamma, igam = kPbeInvGamma;
        const real rs = kCrs * q.inv13;                                  // rs = (3 / (4 pi n))^(1/3)
        const real sr = kSqrtCrs * q.y;                                  // sqrt(rs) = sqrt(crs) n^(-1/6)
        const real isr = kInvSqrtCrs * (n * q.inv13 * q.inv13 * q.y);    // 1 / sqrt(rs) = n^(1/6) / sqrt(crs) = n y^5 / sqrt(crs)
        real eps, deps_drs;
        pw92_roots(sr, isr, eps, deps_drs);
        const real deps_dn = -rs * (1.0 / 3.0) * inv_n * deps_drs;
        const real ee = fm::exp(-eps * igam);
        const real A = beta * igam * fm::rcp(ee - 1.0 + 1e-30);
        const real dAdn = A * A * (1.0 / beta) * ee * deps_dn;
        const real ct = (1.0 / 16.0) * kCbrtPiOver3;
        const real n43 = n13 * n;
        // n^(-7/3) from the roots already at hand (was a reciprocal of n^(7/3) + 1e-30: the guard's cap kept, one min instead of 5 fmas)
        const real in73 = fmin(inv_n * inv_n * q.inv13, (real)1e30);
        const real t2 = ct * gn2 * in73;
        const real dt2dn = -(7.0 / 3.0) * ct * gn2 * n43 * in73 * in73;
        const real dt2dg = ct * in73;
        const real At2 = A * t2;
        const real num = 1.0 + At2, num2 = 1.0 + 2.0 * At2;
        const real iden = fm::rcp(1.0 + At2 + At2 * At2);
        const real arg = 1.0 + beta * igam * t2 * num * iden;
        const real H = gam * fm::log(arg);
        const real common = t2 * num * iden * iden * num2;
        const real dQn = (dt2dn * num2 + dAdn * t2 * t2) * iden - common * (dt2dn * A + dAdn * t2);
        const real dQg = dt2dg * num2 * iden - common * (dt2dg * A);
        const real boa = beta * fm::rcp(arg);
        r.fc = (eps + H) * n;
        r.dfdn += eps + H + n * (deps_dn + boa * dQn);
        r.dfdg += n * boa * dQg;
    }
    return r;
}

// Pauli-Gaussian with the Laplacian-dependent terms (functionals.py:336-403; tools_for_tests.py:86-118):
//   f = tau_TF (exp(-mu s^2) + beta q^2 - lambda q s^2 + sigma s^4),  q = lap n / (4 (3 pi^2)^(2/3) n^(5/3))
// adds f, df/dn, df/d|grad n|^2 to p and returns df/d(lap n)
__device__ __forceinline__ void pg_laplacian_point(real n, real gn2, real lap, const GgaSel& sel, PbePoint& p,
                                                   real& dfdl) {
    const real ctf = kCtf, cs = kCs2;
    const fm::Roots<real> r = fm::roots(n);
    const real n13 = r.n13, inv_n = r.inv_n;
    const real n53i = inv_n * r.inv13 * r.inv13, n83i = n53i * inv_n;      // n^(-5/3), n^(-8/3)
    const real s2 = cs * gn2 * n83i, q = cs * lap * n53i;
    const real tau = ctf * n13 * n13 * n;
    const real ex = fm::exp(-sel.kmu * s2);
    const real F = ex + sel.kbeta * q * q - sel.klambda * q * s2 + sel.ksigma * s2 * s2;
    const real Fs = -sel.kmu * ex - sel.klambda * q + 2.0 * sel.ksigma * s2;       // dF / d(s^2)
    const real Fq = 2.0 * sel.kbeta * q - sel.klambda * s2;                          // dF / dq
    p.fk = tau * F;
    p.dfdn += (5.0 / 3.0) * tau * inv_n * F + tau * (Fs * (-(8.0 / 3.0) * s2 * inv_n) + Fq * (-(5.0 / 3.0) * q * inv_n));
    p.dfdg += tau * Fs * cs * n83i;
    dfdl = tau * Fq * cs * n53i;
}

// PBE mid stage: grad n -> energy partials (x, c), df/dn, flux_j = df/dg * grad_j n (in place)
static __global__ __launch_bounds__(kRedThreads) void pbe_kernel(const real* __restrict__ n, real* __restrict__ gx,
                                                          real* __restrict__ gy, real* __restrict__ gz,
                                                          real* __restrict__ dfdn, long long npts, GgaSel sel,
                                                          acc_t* __restrict__ partial, real* __restrict__ lapn = nullptr) {
    acc_t acc[kPbeScalars] = {0.0, 0.0, 0.0};
    if (lapn) {      // Pauli-Gaussian members with q-dependence: scalar loop (not a hot path), lapn is overwritten by df/dL
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
            const real a = gx[i], b = gy[i], c = gz[i];
            GgaSel nk = sel;
            nk.k = 0;
            PbePoint p = pbe_point(n[i], a * a + b * b + c * c, nk);
            real dfdl;
            pg_laplacian_point(n[i], a * a + b * b + c * c, lapn[i], sel, p, dfdl);
            acc[0] += p.fx;
            acc[1] += p.fc;
            acc[2] += p.fk;
            dfdn[i] = p.dfdn;
            gx[i] = p.dfdg * a;
            gy[i] = p.dfdg * b;
            gz[i] = p.dfdg * c;
            lapn[i] = dfdl;
        }
        block_reduce_store<kPbeScalars>(acc, partial);
        return;
    }
    const long long n2 = npts >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const cplx d = reinterpret_cast<const cplx*>(n)[i];
        const cplx a = reinterpret_cast<cplx*>(gx)[i], b = reinterpret_cast<cplx*>(gy)[i],
                      c = reinterpret_cast<cplx*>(gz)[i];
        const PbePoint p0 = pbe_point(d.x, a.x * a.x + b.x * b.x + c.x * c.x, sel);
        const PbePoint p1 = pbe_point(d.y, a.y * a.y + b.y * b.y + c.y * c.y, sel);
        acc[0] += p0.fx + p1.fx;
        acc[1] += p0.fc + p1.fc;
        acc[2] += p0.fk + p1.fk;
        reinterpret_cast<cplx*>(dfdn)[i] = mkc(p0.dfdn, p1.dfdn);
        reinterpret_cast<cplx*>(gx)[i] = mkc(p0.dfdg * a.x, p1.dfdg * a.y);
        reinterpret_cast<cplx*>(gy)[i] = mkc(p0.dfdg * b.x, p1.dfdg * b.y);
        reinterpret_cast<cplx*>(gz)[i] = mkc(p0.dfdg * c.x, p1.dfdg * c.y);
    }
    if ((npts & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long i = npts - 1;
        const real a = gx[i], b = gy[i], c = gz[i];
        const PbePoint p = pbe_point(n[i], a * a + b * b + c * c, sel);
        acc[0] += p.fx;
        acc[1] += p.fc;
        acc[2] += p.fk;
        dfdn[i] = p.dfdn;
        gx[i] = p.dfdg * a;
        gy[i] = p.dfdg * b;
        gz[i] = p.dfdg * c;
    }
    block_reduce_store<kPbeScalars>(acc, partial);
}

// ---- final combine: potential + all energy integrands -------------------------------------------
struct CombineArgs {
    const real* n;
    const real* vext;
    const real* vh;
    const real* lap_s;
    const real* conv_b;
    const real* conv_a;
    const real* u0; const real* u1; const real* u2;
    const real* gA; const real* gB; const real* gC;
    const real* dfdn;
    const real* div;
    real* v_out;
    long long npts;
    unsigned mask;
    real wt_alpha, wt_beta, wt_nbar_pa;     // nbar^alpha
    real wgc_alpha, wgc_beta, nref;
    real gtf_inv_n0;   // vWGTF: 1 / n0, n0 = round(N_e) / vol (functionals.py:268-270)
    int gtf_kind;        // 1 = vWGTF1, 2 = vWGTF2
    int wt_is_56;        // alpha = beta = 5/6: n^(-1/6) = 1/sqrt(cbrt n), no pow
    int wgc_sum_53;      // alpha + beta = 5/3: n^(alpha-1) = 1/(cbrt(n) n^(beta-1)), one pow instead of two
    real w_tf = 1.0, w_nl = 1.0;   // weights of the TF / Wang-Teter potentials (stabilised WT-style functional, OFDFT_P_WTS_KIND)
};
// partial scalars: 0 ion-electron, 1 hartree, 2 tf, 3 vw, 4 wt-nl, 5 wgc-nl, 6 lda-x, 7 local-c, 8 sum(v*n), 9 vWGTF
constexpr int kCombineScalars = 10;

// Pauli enhancement factor of vWGTF1 / vWGTF2 and its derivative with respect to d = n / n0 (functionals.py:251-306)
__device__ __forceinline__ void vwgtf_factor(real d, int kind, real& G, real& dG) {
    if (kind == 1) {
        G = 0.9892 * pow(d, -1.2994);
        dG = -1.2994 * G / d;
    } else {
        const real a = 5.7001, b = 0.2563;
        const real db = pow(d, b), th = tanh(a * db - a);
        const real elf = 0.5 * (1.0 + th);
        const real delf = 0.5 * (1.0 - th * th) * a * b * db / d;
        G = sqrt(1.0 / elf - 1.0);
        dG = -delf / (2.0 * G * elf * elf);
    }
}
// e = G tau_TF and its potential d e / d n at one point
__device__ __forceinline__ void vwgtf_point(real n, real n13, real ctf, real inv_n0, int kind, real& e, real& v) {
    real G, dG;
    vwgtf_factor(n * inv_n0, kind, G, dG);
    const real tau = ctf * n13 * n13 * n;
    e = G * tau;
    v = (5.0 / 3.0) * ctf * n13 * n13 * G + tau * dG * inv_n0;
}

// one grid point of the combine: all inputs already in registers
struct CombinePoint {
    real n, vext, vh, lap, cb, cva, u0, u1, u2, gA, gB, gC, dfdn, div;
};
__device__ __forceinline__ real combine_point(const CombineArgs& a, const CombinePoint& p, real ctf,
                                                acc_t (&acc)[kCombineScalars]) {
    const real n = p.n;
    real v = 0.0;
    if (a.mask & 1u) {                                  // ion-electron  functionals.py:46
        acc[0] += n * p.vext;
        v += p.vext;
    }
    if (a.mask & 2u) {                                  // Hartree  functionals.py:72
        acc[1] += 0.5 * n * p.vh;
        v += p.vh;
    }
    const real n13 = cbrt(n);
    if (a.mask & 4u) {                                  // TF  functionals.py:223; tools_for_tests.py:19-20
        const real n23 = n13 * n13;
        acc[2] += ctf * n23 * n;
        v += a.w_tf * (5.0 / 3.0) * ctf * n23;
    }
    if (a.mask & 8u) {                                  // vW  functionals.py:245; tools_for_tests.py:23-26
        const real s = (n != 0.0) ? sqrt(n) : 0.0;
        acc[3] += -0.5 * s * p.lap;
        if (n != 0.0) v += -0.5 * p.lap / s;
    }
    if (a.mask & 16u) {                                 // WT-family NL  functionals.py:650-651; tools_for_tests.py:29-39
        const real pa1 = a.wt_is_56 ? 1.0 / sqrt(n13) : pow(n, a.wt_alpha - 1.0);
        acc[4] += ctf * (pa1 * n - a.wt_nbar_pa) * p.cb;
        if (a.conv_a) {
            const real pb1 = pow(n, a.wt_beta - 1.0);
            v += a.w_nl * ctf * (a.wt_alpha * pa1 * p.cb + a.wt_beta * pb1 * p.cva);
        } else {
            v += a.w_nl * ctf * 2.0 * a.wt_alpha * pa1 * p.cb;
        }
    }
    if (a.mask & 32u) {                                 // WGC99 NL  SURVEY §8a-8
        const real th = n - a.nref;
        const real pb1 = pow(n, a.wgc_beta - 1.0);
        const real pa1 = a.wgc_sum_53 ? 1.0 / (n13 * pb1) : pow(n, a.wgc_alpha - 1.0);
        const real P = pa1 * n, A = pb1 * n, dA = a.wgc_beta * pb1;
        const real conv = p.u0 + th * p.u1 + 0.5 * th * th * p.u2;
        acc[5] += ctf * P * conv;
        v += ctf * (a.wgc_alpha * pa1 * conv + P * (p.u1 + th * p.u2) + p.gA * dA + p.gB * (dA * th + A)
                    + p.gC * (0.5 * dA * th * th + A * th));
    }
    if (a.mask & (0xFu << 6)) {                         // local XC
        const XcLocal x = lda_point(n, a.mask);
        acc[6] += x.ex;
        acc[7] += x.ec;
        v += x.vx + x.vc;
    }
    if (a.mask & (7u << 10)) v += p.dfdn - 2.0 * p.div;   // PBE / GGA kinetic  tools_for_tests.py:168-170
    if (a.mask & (1u << 13)) {                          // vWGTF1 / 2  functionals.py:251-306
        real e, ve;
        vwgtf_point(n, n13, ctf, a.gtf_inv_n0, a.gtf_kind, e, ve);
        acc[9] += e;
        v += ve;
    }
    acc[8] += v * n;
    return v;
}

// Every array pointer in CombineArgs is valid (the host points unused ones at `n`), so all loads of an
// iteration are issued together as 16-byte loads ahead of the arithmetic instead of one dependent load
// per term behind a branch.
static __global__ __launch_bounds__(kRedThreads) void combine_kernel(CombineArgs a, acc_t* __restrict__ partial) {
    const real ctf = kCtf;   // 0.3 (3 pi^2)^(2/3)
    acc_t acc[kCombineScalars];
#pragma unroll
    for (int s = 0; s < kCombineScalars; ++s) acc[s] = 0.0;
    const long long n2 = a.npts >> 1;
#define LD2(ptr) reinterpret_cast<const cplx*>(ptr)[i]
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const cplx n = LD2(a.n), ve = LD2(a.vext), vh = LD2(a.vh), lp = LD2(a.lap_s), cb = LD2(a.conv_b),
                      cva = LD2(a.conv_a ? a.conv_a : a.n), u0 = LD2(a.u0), u1 = LD2(a.u1), u2 = LD2(a.u2),
                      gA = LD2(a.gA), gB = LD2(a.gB), gC = LD2(a.gC), df = LD2(a.dfdn), dv = LD2(a.div);
        const CombinePoint p0{n.x, ve.x, vh.x, lp.x, cb.x, cva.x, u0.x, u1.x, u2.x, gA.x, gB.x, gC.x, df.x, dv.x};
        const CombinePoint p1{n.y, ve.y, vh.y, lp.y, cb.y, cva.y, u0.y, u1.y, u2.y, gA.y, gB.y, gC.y, df.y, dv.y};
        const real v0 = combine_point(a, p0, ctf, acc);
        const real v1 = combine_point(a, p1, ctf, acc);
        if (a.v_out) reinterpret_cast<cplx*>(a.v_out)[i] = mkc(v0, v1);
    }
#undef LD2
    if ((a.npts & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long i = a.npts - 1;
        const CombinePoint p{a.n[i], a.vext[i], a.vh[i], a.lap_s[i], a.conv_b[i], (a.conv_a ? a.conv_a : a.n)[i], a.u0[i],
                             a.u1[i], a.u2[i], a.gA[i], a.gB[i], a.gC[i], a.dfdn[i], a.div[i]};
        const real v = combine_point(a, p, ctf, acc);
        if (a.v_out) a.v_out[i] = v;
    }
    block_reduce_store<kCombineScalars>(acc, partial);
}

// y = x or y += x (grid arrays: T = real; device-resident energy sums: T = acc_t)
template <class T>
__global__ void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = accumulate ? y[i] + x[i] : x[i];
}

// chi.grad = c * 2 chi (v - mu) dV   (system.py:850-853)
// c = N_e / (mean(chi^2) vol) from the reduced sum of chi^2, left on the device (system.py:833-834)
static __global__ void closure_scale_kernel(const acc_t* __restrict__ sumsq, acc_t* __restrict__ cscale, acc_t n_elec,
                                     acc_t vol_over_npts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) cscale[0] = n_elec / (sumsq[0] * vol_over_npts);
}
// the two steps in ONE launch (one block): sum chi^2 from the partials (the order of reduce_partials_kernel), then c
static __global__ __launch_bounds__(kRedThreads) void closure_scale_reduce_kernel(const acc_t* __restrict__ partial, int rows,
                                                                           acc_t* __restrict__ sumsq, acc_t* __restrict__ cscale,
                                                                           acc_t n_elec, acc_t vol_over_npts) {
    acc_t acc = 0.0;
    for (int r = threadIdx.x; r < rows; r += kRedThreads) acc += partial[r];
    __shared__ acc_t red[kRedThreads];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = kRedThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sumsq[0] = red[0];
        cscale[0] = n_elec / (red[0] * vol_over_npts);
    }
}

// Stabilised Wang-Teter style functional T_TF f(X), X = T_NL / T_TF, f = exp (functionals.py:771-782): weights of the two
// potentials from the reduced sums of a first (energy-only) combine pass ...
static __global__ void wts_weights_kernel(const acc_t* __restrict__ sums, acc_t* __restrict__ w) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const acc_t X = sums[4] / sums[2], fx = ::exp(X);
        w[0] = fx * (1.0 - X);      // f - f' X
        w[1] = fx;                  // f' / f'(0)
        w[2] = fx;
    }
}
// ... and the reported sums after the second pass: [2] <- T_TF f(X), [4] <- 0
static __global__ void wts_finalize_kernel(acc_t* __restrict__ sums, const acc_t* __restrict__ w) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sums[2] *= w[2];
        sums[4] = 0.0;
    }
}

static __global__ void chi_grad_kernel(const real* __restrict__ chi, const real* __restrict__ v, real* __restrict__ g,
                                long long npts, real c2dV_host, const acc_t* __restrict__ cscale_dev, real two_dV,
                                real mu_host, const acc_t* __restrict__ vn_dev = nullptr, acc_t dV = 0.0, acc_t n_elec = 1.0,
                                const real* __restrict__ v2 = nullptr, const acc_t* __restrict__ vn2_dev = nullptr) {
    const real c2dV = cscale_dev ? (real)(cscale_dev[0] * two_dV) : c2dV_host;
    // mu = (sum(v n) dV) / N_e: from the host, or formed here from the device-resident sum (no host round trip: the
    // graph-captured evaluation); same operations in the same order as the host form
    // (vn2_dev: the share of sum(v n) of a part of the potential formed on another stream -- added here in the order the host adds it)
    const real mu = vn_dev ? (real)(((vn2_dev ? vn_dev[0] + vn2_dev[0] : vn_dev[0]) * dV) / n_elec) : mu_host;
    // v2: a part of the potential kept in its own array (the WGC99 part formed beside the combine kernel)
    // Round 5: 16-byte accesses, two per array in flight per thread (one 8-byte pair per thread and iteration left the 2 048-block
    // grid at 4.4 TB/s on 4-GB arrays, 0.55 of the peak; unaligned views take pairs as before)
    constexpr int NV = 16 / (int)sizeof(real);
    typedef real vec_t __attribute__((ext_vector_type(NV)));
    const long long stride = (long long)gridDim.x * blockDim.x, gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long done = 0;
    const unsigned long long al = reinterpret_cast<unsigned long long>(chi) | reinterpret_cast<unsigned long long>(v) |
                                  reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(v2);
    if ((al & 15ull) == 0) {
        const vec_t* xv = reinterpret_cast<const vec_t*>(chi);
        const vec_t* wv = reinterpret_cast<const vec_t*>(v);
        const vec_t* w2v = reinterpret_cast<const vec_t*>(v2);
        vec_t* gv = reinterpret_cast<vec_t*>(g);
        const long long nv = npts / NV;
        auto one = [&](const vec_t& x, vec_t w, const vec_t& w2) {
            vec_t r;
#pragma unroll
            for (int e = 0; e < NV; ++e) r[e] = c2dV * x[e] * ((v2 ? w[e] + w2[e] : w[e]) - mu);
            return r;
        };
        long long i = gid;
        for (; i + stride < nv; i += 2 * stride) {
            const vec_t x0 = xv[i], x1 = xv[i + stride], w0 = wv[i], w1 = wv[i + stride];
            const vec_t u0 = v2 ? w2v[i] : w0, u1 = v2 ? w2v[i + stride] : w1;
            gv[i] = one(x0, w0, u0);                 // (plain stores: the optimiser's sweeps read g next)
            gv[i + stride] = one(x1, w1, u1);
        }
        for (; i < nv; i += stride) {
            const vec_t x0 = xv[i], w0 = wv[i];
            const vec_t u0 = v2 ? w2v[i] : w0;
            gv[i] = one(x0, w0, u0);
        }
        done = nv * NV;
    }
    for (long long i = done + gid; i < npts; i += stride) g[i] = c2dV * chi[i] * (v[i] + (v2 ? v2[i] : (real)0.0) - mu);
}

}  // namespace ofdft
