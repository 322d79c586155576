// 3-D real<->half-spectrum FFT passes for gfx950 (fp64).
//
// Internal half-spectrum layout ("block-8"): the kz axis is cut into blocks of 8; the main part
// holds kz < nzm (nzm = largest multiple of 8 <= nzc) as [kz/8][n0][n1][8] so that every (x,y)
// owns a full 128-byte line per block, x- and y-lines of a block are gathered from contiguous
// 128-B..32-KB chunks, and the z-pass writes 1-KB runs; the remaining nzc-nzm planes (exactly the
// Nyquist plane for power-of-two n2) are stored dense as [kz-nzm][n0][n1].  No padding: the
// spectrum has exactly nzc*n0*n1 elements.
//
// Fast path (all three extents powers of two, 8..1024 on axes 0/1, 16..2048 on axis 2):
//   z: real row -> N2/2-point complex FFT in registers + LDS, split post-processing, r2c / c2r
//   y, x: strided complex line FFTs, tile of lines contiguous in memory, in place
// Generic path (any extents): naive O(N^2)-per-line DFT kernels with the same layout/semantics.
#pragma once
#include "fft_radix.h"

namespace ofdft {

struct SpecGeom {
    int n0, n1, n2, nzc, nzm;
    long long nrows;       // n0*n1
    long long main_count;  // nzm*n0*n1
    long long total;       // nzc*n0*n1
};

__host__ __device__ inline long long spec_index(const SpecGeom& g, int x, int y, int kz) {
    const long long row = (long long)x * g.n1 + y;
    if (kz < g.nzm) return (((long long)(kz >> 3)) * g.nrows + row) * 8 + (kz & 7);
    return g.main_count + (long long)(kz - g.nzm) * g.nrows + row;
}

__device__ inline void spec_decode(const SpecGeom& g, long long i, int& x, int& y, int& kz) {
    if (i < g.main_count) {
        const int kin = (int)(i & 7);
        const long long r = i >> 3;
        const long long row = r % g.nrows;
        kz = (int)(r / g.nrows) * 8 + kin;
        x = (int)(row / g.n1);
        y = (int)(row % g.n1);
    } else {
        const long long r = i - g.main_count;
        const long long row = r % g.nrows;
        kz = g.nzm + (int)(r / g.nrows);
        x = (int)(row / g.n1);
        y = (int)(row % g.n1);
    }
}

// ----------------------------------------------------------------------------------------------
// complex line pass (x or y axis), in place.  base(L) = (L / d) * sb + (L % d) * sl; stride se.
struct LineMap {
    long long sb;
    long long se;
    int d;
    int sl;
    int nlines;
    int lf;  // lines that vary fastest over the lanes (memory-contiguous direction)
};

template <int LEN> struct PassCfg {
    static constexpr int P = Plan<LEN>::P;
    static constexpr int E = Plan<LEN>::E;
    static constexpr int TPB = (8 * P > 256) ? 8 * P : 256;
    static constexpr int LPW = TPB / P;
    static constexpr size_t LDS = (Plan<LEN>::NST > 1) ? sizeof(double) * LPW * LineBuf<LEN>::STRIDE : 0;
};

template <int LEN, bool INV>
__global__ __launch_bounds__(PassCfg<LEN>::TPB) void cpass_kernel(cplx* __restrict__ data, LineMap m,
                                                                   const cplx* __restrict__ tw) {
    constexpr int P = PassCfg<LEN>::P, E = PassCfg<LEN>::E, LPW = PassCfg<LEN>::LPW;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int l_lo = tid % m.lf;
    const int j = (tid / m.lf) % P;
    const int l = (tid / (m.lf * P)) * m.lf + l_lo;
    const long long L = (long long)blockIdx.x * LPW + l;
    const bool valid = L < m.nlines;
    cplx* p = data + (valid ? (L / m.d) * m.sb + (L % m.d) * (long long)m.sl : 0);
    cplx v[E];
    if (valid) {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = p[(long long)(j + P * q) * m.se];
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = make_double2(0.0, 0.0);
    }
    line_fft<LEN, INV>(v, j, lds + l * LineBuf<LEN>::STRIDE, tw);
    if (valid) {
#pragma unroll
        for (int q = 0; q < E; ++q) p[(long long)(j + P * q) * m.se] = v[q];
    }
}

// ----------------------------------------------------------------------------------------------
// z pass, forward r2c: real rows [nrows][N2] -> half spectrum.  M = N2/2.
template <int M> struct ZCfg {
    static constexpr int P = Plan<M>::P;
    static constexpr int E = Plan<M>::E;
    static constexpr int TPB = (8 * P > 256) ? 8 * P : 256;
    static constexpr int RPW = TPB / P;                 // rows per workgroup
    static constexpr int RS = LineBuf<M>::STRIDE;       // LDS doubles per row
    static constexpr size_t LDS = sizeof(double) * RPW * RS;
};

struct PreIdentity {
    __device__ __forceinline__ double operator()(double a, long long) const { return a; }
};

template <int M, class Pre>
__global__ __launch_bounds__(ZCfg<M>::TPB) void zfwd_kernel(const double* __restrict__ in, cplx* __restrict__ spec,
                                                            SpecGeom g, const cplx* __restrict__ twM,
                                                            const cplx* __restrict__ twN, Pre pre) {
    constexpr int P = ZCfg<M>::P, E = ZCfg<M>::E, RPW = ZCfg<M>::RPW, RS = ZCfg<M>::RS, TPB = ZCfg<M>::TPB;
    constexpr int N2 = 2 * M;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int j = tid % P, r = tid / P;
    const long long row0 = (long long)blockIdx.x * RPW;
    const long long row = row0 + r;
    const bool valid = row < g.nrows;
    cplx v[E];
    if (valid) {
        const double2* src = reinterpret_cast<const double2*>(in + row * N2);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            double2 t = src[j + P * q];
            const long long e = row * N2 + 2 * (j + P * q);
            v[q] = make_double2(pre(t.x, e), pre(t.y, e + 1));
        }
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = make_double2(0.0, 0.0);
    }
    double* mine = lds + r * RS;
    line_fft<M, false>(v, j, mine, twM);

    // ---- split post-processing through LDS: X[k] = Ev[k] + W_N^k Od[k]
    double cr_k[E], cr_m[E], c0r = 0.0, c0i = 0.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < E; ++q) mine[lpad(j + P * q)] = v[q].x;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        const int k = 8 * b + kin, mk = (M - k) & (M - 1);
        cr_k[it] = lds[rr * RS + lpad(k)];
        cr_m[it] = lds[rr * RS + lpad(mk)];
    }
    if (tid < RPW) c0r = lds[tid * RS];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < E; ++q) mine[lpad(j + P * q)] = v[q].y;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        const int k = 8 * b + kin, mk = (M - k) & (M - 1);
        const double ci_k = lds[rr * RS + lpad(k)];
        const double ci_m = lds[rr * RS + lpad(mk)];
        const cplx ev = make_double2(0.5 * (cr_k[it] + cr_m[it]), 0.5 * (ci_k - ci_m));
        const cplx od = make_double2(0.5 * (ci_k + ci_m), -0.5 * (cr_k[it] - cr_m[it]));
        const cplx X = cadd(ev, cmul(twN[k], od));
        const long long rg = row0 + rr;
        if (rg < g.nrows) spec[((long long)b * g.nrows + rg) * 8 + kin] = X;
    }
    if (tid < RPW) {
        c0i = lds[tid * RS];
        const long long rg = row0 + tid;
        if (rg < g.nrows) spec[g.main_count + rg] = make_double2(c0r - c0i, 0.0);
    }
}

struct PostScale {
    double s;
    __device__ __forceinline__ double operator()(double a, long long) const { return a * s; }
};

// z pass, inverse c2r (imaginary parts of the kz=0 and Nyquist entries are ignored, as irfftn does)
template <int M, class Post>
__global__ __launch_bounds__(ZCfg<M>::TPB) void zinv_kernel(const cplx* __restrict__ spec, double* __restrict__ out,
                                                            SpecGeom g, const cplx* __restrict__ twM,
                                                            const cplx* __restrict__ twN, Post post) {
    constexpr int P = ZCfg<M>::P, E = ZCfg<M>::E, RPW = ZCfg<M>::RPW, RS = ZCfg<M>::RS, TPB = ZCfg<M>::TPB;
    constexpr int N2 = 2 * M;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int j = tid % P, r = tid / P;
    const long long row0 = (long long)blockIdx.x * RPW;
    const long long row = row0 + r;
    const bool valid = row < g.nrows;
    // ---- stage the rows' spectra into LDS (coalesced 128-B lines), real parts then imaginary parts
    cplx xs[E];
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        const long long rg = row0 + rr;
        xs[it] = (rg < g.nrows) ? spec[((long long)b * g.nrows + rg) * 8 + kin] : make_double2(0.0, 0.0);
    }
    const double nyq = (valid && j == 0) ? spec[g.main_count + row].x : 0.0;
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        lds[rr * RS + lpad(8 * b + kin)] = xs[it].x;
    }
    __syncthreads();
    double* mine = lds + r * RS;
    double xr_k[E], xr_m[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int k = j + P * q, mk = (M - k) & (M - 1);
        xr_k[q] = mine[lpad(k)];
        xr_m[q] = mine[lpad(mk)];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        lds[rr * RS + lpad(8 * b + kin)] = xs[it].y;
    }
    __syncthreads();
    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int k = j + P * q, mk = (M - k) & (M - 1);
        const double xi_k = mine[lpad(k)], xi_m = mine[lpad(mk)];
        if (k == 0) {
            v[q] = make_double2(xr_k[q] + nyq, xr_k[q] - nyq);
        } else {
            const cplx ev = make_double2(xr_k[q] + xr_m[q], xi_k - xi_m);
            const cplx d = make_double2(xr_k[q] - xr_m[q], xi_k + xi_m);
            const cplx od = cmul(d, cconj(twN[k]));
            v[q] = make_double2(ev.x - od.y, ev.y + od.x);
        }
    }
    line_fft<M, true>(v, j, mine, twM);
    if (valid) {
        double2* dst = reinterpret_cast<double2*>(out + row * N2);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const long long e = row * N2 + 2 * (j + P * q);
            dst[j + P * q] = make_double2(post(v[q].x, e), post(v[q].y, e + 1));
        }
    }
}

// ----------------------------------------------------------------------------------------------
// generic (any extent) naive DFT kernels -- correctness path for non power-of-two grids
__global__ void gen_r2c_z_kernel(const double* __restrict__ in, cplx* __restrict__ spec, SpecGeom g,
                                 const cplx* __restrict__ tw2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    const double* rowp = in + ((long long)x * g.n1 + y) * g.n2;
    double sr = 0.0, si = 0.0;
    int t = 0;
    for (int z = 0; z < g.n2; ++z) {
        const cplx w = tw2[t];
        sr += rowp[z] * w.x;
        si += rowp[z] * w.y;
        t += kz;
        if (t >= g.n2) t -= g.n2;
    }
    spec[i] = make_double2(sr, si);
}

// out-of-place complex DFT along axis 0 (x) or 1 (y)
__global__ void gen_c2c_kernel(const cplx* __restrict__ in, cplx* __restrict__ out, SpecGeom g, int axis, int inv,
                               const cplx* __restrict__ tw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    const int n = axis == 0 ? g.n0 : g.n1;
    const int k = axis == 0 ? x : y;
    double sr = 0.0, si = 0.0;
    int t = 0;
    for (int e = 0; e < n; ++e) {
        const cplx a = in[axis == 0 ? spec_index(g, e, y, kz) : spec_index(g, x, e, kz)];
        cplx w = tw[t];
        if (inv) w.y = -w.y;
        sr += a.x * w.x - a.y * w.y;
        si += a.x * w.y + a.y * w.x;
        t += k;
        if (t >= n) t -= n;
    }
    out[i] = make_double2(sr, si);
}

template <class Post>
__global__ void gen_c2r_z_kernel(const cplx* __restrict__ spec, double* __restrict__ out, SpecGeom g,
                                 const cplx* __restrict__ tw2, Post post) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npts = g.nrows * g.n2;
    if (i >= npts) return;
    const int z = (int)(i % g.n2);
    const long long row = i / g.n2;
    const int x = (int)(row / g.n1), y = (int)(row % g.n1);
    double acc = spec[spec_index(g, x, y, 0)].x;
    const int kmax = (g.n2 - 1) / 2;   // strictly-interior frequencies
    int t = 0;
    for (int k = 1; k <= kmax; ++k) {
        t += z;
        if (t >= g.n2) t -= g.n2;
        const cplx a = spec[spec_index(g, x, y, k)];
        const cplx w = tw2[t];        // forward table: exp(-i phi); inverse needs exp(+i phi)
        acc += 2.0 * (a.x * w.x + a.y * w.y);
    }
    if ((g.n2 & 1) == 0) acc += spec[spec_index(g, x, y, g.n2 / 2)].x * ((z & 1) ? -1.0 : 1.0);
    out[i] = post(acc, i);
}

// standard [n0][n1][nzc] <-> internal layout
__global__ void spec_to_internal_kernel(const cplx* __restrict__ stdl, cplx* __restrict__ intl, SpecGeom g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    intl[i] = stdl[((long long)x * g.n1 + y) * g.nzc + kz];
}
__global__ void spec_to_standard_kernel(const cplx* __restrict__ intl, cplx* __restrict__ stdl, SpecGeom g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    stdl[((long long)x * g.n1 + y) * g.nzc + kz] = intl[i];
}

}  // namespace ofdft
