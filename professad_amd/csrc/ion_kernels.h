// Ionic (external) potential on the grid from ion positions: exact or particle-mesh-Ewald structure factor times
// the interpolated reciprocal-space pseudopotential (reference: src/professad/ion_utils.py:49-286,
// src/professad/system.py:183-194).  fp64, gfx950.
#pragma once
#include "pointwise_kernels.h"

namespace ofdft {

constexpr int kMaxPmeOrder = 32;

// cardinal B-spline values [M_n(x+i), i = 0..n-1], 0 <= x < 1 (ion_utils.py:140-204: the recursion of its docstring)
__device__ __forceinline__ void bspline_values(double x, int order, double* M) {
    M[0] = x;
    M[1] = 1.0 - x;
    for (int n = 3; n <= order; ++n) {
        M[n - 1] = 0.0;
        for (int i = n - 1; i >= 1; --i) M[i] = ((x + i) * M[i] + (n - x - i) * M[i - 1]) / (n - 1);
        M[0] = x / (n - 1) * M[0];
    }
}

// Q(l0,l1,l2) += M0 M1 M2 with periodic wrap (ion_utils.py:249-273); one workgroup per ion
// slab-decomposed contexts: Q is this rank's x-slab (planes x0 .. x0 + nxl of the n0 global ones); every rank walks all
// ions and keeps the stencil points that fall on its planes -- no halo exchange, no communication
__global__ __launch_bounds__(256) void pme_spread_kernel(const double* __restrict__ frac, int nion, int order,
                                                         double* __restrict__ Q, int n0, int n1, int n2, int x0, int nxl) {
    __shared__ double M[3][kMaxPmeOrder];
    __shared__ int L[3][kMaxPmeOrder];
    const int a = blockIdx.x;
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const int N = d == 0 ? n0 : (d == 1 ? n1 : n2);
        const double u = frac[3 * a + d] * N;
        const long long fl = (long long)floor(u);
        bspline_values(u - (double)fl, order, M[d]);
        for (int i = 0; i < order; ++i) {
            long long l = (i - fl) % N;
            if (l < 0) l += N;
            L[d][i] = (int)l;
        }
    }
    __syncthreads();
    const int tot = order * order * order;
    for (int t = threadIdx.x; t < tot; t += blockDim.x) {
        const int i2 = t % order, i1 = (t / order) % order, i0 = t / (order * order);
        const int xl = L[0][i0] - x0;
        if (xl < 0 || xl >= nxl) continue;
        atomicAdd(&Q[((long long)xl * n1 + L[1][i1]) * n2 + L[2][i2]], M[0][i0] * M[1][i1] * M[2][i2]);
    }
}

struct RecpotTable {
    const double* ks;    // [n] uniform grid from 0
    const double* y;     // [n] table with the Coulomb tail 4 pi z/k^2 added for k > 0 (ion_utils.py:67-68)
    const double* m;     // [n] Hermite slopes (functional_tools.py:309-310)
    int n;
    double z;
    double inv_dk;
};

// interpolate_recpot at |k| (ion_utils.py:74-81; functional_tools.py:311-334)
__device__ __forceinline__ double recpot_value(const RecpotTable& t, double kabs) {
    const double xs = fmin(kabs, t.ks[t.n - 1]);
    int idx = (int)ceil(xs * t.inv_dk) - 1;
    idx = max(0, min(idx, t.n - 2));
    while (idx > 0 && t.ks[idx] >= xs) --idx;            // searchsorted(x[1:], xs): #entries of x[1:] below xs
    while (idx < t.n - 2 && t.ks[idx + 1] < xs) ++idx;
    const double dx = t.ks[idx + 1] - t.ks[idx];
    const double u = (xs - t.ks[idx]) / dx, u2 = u * u, u3 = u2 * u;
    const double v = (1.0 - 3.0 * u2 + 2.0 * u3) * t.y[idx] + (u - 2.0 * u2 + u3) * t.m[idx] * dx +
                     (3.0 * u2 - 2.0 * u3) * t.y[idx + 1] + (-u2 + u3) * t.m[idx + 1] * dx;
    return (kabs != 0.0) ? v - 4.0 * kPi * t.z / (kabs * kabs) : v;
}

// F^ = conj(b0 b1 b2 Q^) v~(|k|) / vol  (PME, ion_utils.py:275-286,118) -- or, with cart != nullptr, the exact
// structure factor sum_i exp(-i k.r_i) (ion_utils.py:121-137)
__global__ void ion_potential_spec_kernel(const cplx* __restrict__ Qk, cplx* __restrict__ out, KGeom kg,
                                          const cplx* __restrict__ b0, const cplx* __restrict__ b1,
                                          const cplx* __restrict__ b2, const double* __restrict__ cart, int nion,
                                          RecpotTable tab, double inv_vol) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        spec_decode(kg.g, i, x, y, z);
        double kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        cplx S;
        if (cart) {
            double sr = 0.0, si = 0.0;
            for (int a = 0; a < nion; ++a) {
                double sn, cs;
                sincos(kx * cart[3 * a] + ky * cart[3 * a + 1] + kz * cart[3 * a + 2], &sn, &cs);
                sr += cs;
                si -= sn;
            }
            S = make_double2(sr, si);
        } else {
            const cplx b = cmul(cmul(b0[x], b1[y]), b2[z]);
            S = cconj(cmul(b, Qk[i]));
        }
        const double f = recpot_value(tab, (k2 != 0.0) ? sqrt(k2) : 0.0) * inv_vol;
        out[i] = make_double2(S.x * f, S.y * f);
    }
}

}  // namespace ofdft

namespace ofdft {

// Exact-structure-factor ion-electron forces: F_a = (dV/vol) sum_k w_k v~(k) k (cos(k.R) Im n^ + sin(k.R) Re n^)
// (= -d/dR_a of dV sum_r n v_ext; reference system.py:913-923 by autograd).  grid = (k chunks, ions).
__global__ __launch_bounds__(kRedThreads) void ion_force_exact_kernel(const cplx* __restrict__ nk, KGeom kg,
                                                                      const double* __restrict__ cart, RecpotTable tab,
                                                                      double* __restrict__ partial) {
    const int a = blockIdx.y;
    const double rx = cart[3 * a], ry = cart[3 * a + 1], rz = cart[3 * a + 2];
    double acc[3] = {0.0, 0.0, 0.0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        spec_decode(kg.g, i, x, y, z);
        double kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        if (k2 == 0.0) continue;
        const double w = (z == 0 || ((kg.g.n2 & 1) == 0 && z == kg.g.n2 / 2)) ? 1.0 : 2.0;
        double sn, cs;
        sincos(kx * rx + ky * ry + kz * rz, &sn, &cs);
        const cplx n = nk[i];
        const double f = w * recpot_value(tab, sqrt(k2)) * (cs * n.y + sn * n.x);
        acc[0] += kx * f;
        acc[1] += ky * f;
        acc[2] += kz * f;
    }
    // partial[(a * gridDim.x + blockIdx.x) * 3 + c]
    __shared__ double red[kRedThreads / 64][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        double v = acc[s];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[s] = v;
    }
    if ((threadIdx.x & 63) == 0)
        for (int s = 0; s < 3; ++s) red[threadIdx.x >> 6][s] = acc[s];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int s = 0; s < 3; ++s) {
            double t = 0.0;
            for (int w = 0; w < kRedThreads / 64; ++w) t += red[w][s];
            partial[((long long)a * gridDim.x + blockIdx.x) * 3 + s] = t;
        }
}

// theta^ = v~(k) conj(b(k) n^(k)) / vol  (the adjoint of the PME potential build applied to the density)
__global__ void pme_theta_spec_kernel(const cplx* __restrict__ nk, cplx* __restrict__ out, KGeom kg,
                                      const cplx* __restrict__ b0, const cplx* __restrict__ b1,
                                      const cplx* __restrict__ b2, RecpotTable tab, double inv_vol) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        spec_decode(kg.g, i, x, y, z);
        double kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        const cplx b = cmul(cmul(b0[x], b1[y]), b2[z]);
        const cplx s = cconj(cmul(b, nk[i]));
        const double f = recpot_value(tab, (k2 != 0.0) ? sqrt(k2) : 0.0) * inv_vol;
        out[i] = make_double2(s.x * f, s.y * f);
    }
}

// G[a][d] = sum over the ion's B-spline stencil of theta(l) * d/du_d (M0 M1 M2); one workgroup per ion
// (slab-decomposed contexts: theta is this rank's x-slab; G then holds the partial sums over its planes)
__global__ __launch_bounds__(256) void pme_gather_kernel(const double* __restrict__ frac, int nion, int order,
                                                         const double* __restrict__ theta, int n0, int n1, int n2,
                                                         double* __restrict__ G, int x0, int nxl) {
    __shared__ double M[3][kMaxPmeOrder], D[3][kMaxPmeOrder];
    __shared__ int L[3][kMaxPmeOrder];
    __shared__ double red[4][3];
    const int a = blockIdx.x;
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const int N = d == 0 ? n0 : (d == 1 ? n1 : n2);
        const double u = frac[3 * a + d] * N;
        const long long fl = (long long)floor(u);
        const double x = u - (double)fl;
        double P[kMaxPmeOrder];
        if (order > 2) {
            bspline_values(x, order - 1, P);
        } else {
            P[0] = 1.0;              // M_1 = indicator of [0,1): M_1(x) = 1, M_1(x+1) = 0
        }
        bspline_values(x, order, M[d]);
        for (int i = 0; i < order; ++i) {
            const double hi = (i < order - 1) ? P[i] : 0.0, lo = (i > 0) ? P[i - 1] : 0.0;
            D[d][i] = hi - lo;       // d/dx M_n(x+i) = M_{n-1}(x+i) - M_{n-1}(x+i-1)
            long long l = (i - fl) % N;
            if (l < 0) l += N;
            L[d][i] = (int)l;
        }
    }
    __syncthreads();
    double acc[3] = {0.0, 0.0, 0.0};
    const int tot = order * order * order;
    for (int t = threadIdx.x; t < tot; t += blockDim.x) {
        const int i2 = t % order, i1 = (t / order) % order, i0 = t / (order * order);
        const int xl = L[0][i0] - x0;
        if (xl < 0 || xl >= nxl) continue;
        const double th = theta[((long long)xl * n1 + L[1][i1]) * n2 + L[2][i2]];
        acc[0] += th * D[0][i0] * M[1][i1] * M[2][i2];
        acc[1] += th * M[0][i0] * D[1][i1] * M[2][i2];
        acc[2] += th * M[0][i0] * M[1][i1] * D[2][i2];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        double v = acc[s];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[s] = v;
    }
    if ((threadIdx.x & 63) == 0)
        for (int s = 0; s < 3; ++s) red[threadIdx.x >> 6][s] = acc[s];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int s = 0; s < 3; ++s) G[3 * a + s] = red[0][s] + red[1][s] + red[2][s] + red[3][s];
}

}  // namespace ofdft

namespace ofdft {

// Ion-ion real-space damped pair sum (ion_utils.py:293-333).  grid = (chunks, ions): block (c, i) scans its share of
// the (j, lattice shift) combinations and accumulates for ion i:
//   [0] sum Z_i Z_j erfc(r/Rd)/r   [1] sum Z_j (neighbour charge)   [2..4] sum f'(r) d / r   [5..10] sum f'(r) d_a d_b / r
// over 0 < r = |R_j + shift - R_i| <= Rc (xx, yy, zz, xy, xz, yz);  f(r) = Z_i Z_j erfc(r/Rd)/r.
constexpr int kIonIonScalars = 11;
struct IonIonGeom {
    double box[9];       // rows = lattice vectors
    int nmax[3];         // shifts -nmax..nmax per axis
    double Rc, Rd;
};
__global__ __launch_bounds__(kRedThreads) void ion_ion_kernel(const double* __restrict__ cart, const double* __restrict__ Z,
                                                              int nion, IonIonGeom g, double* __restrict__ partial) {
    const int i = blockIdx.y;
    const double xi = cart[3 * i], yi = cart[3 * i + 1], zi = cart[3 * i + 2], Zi = Z[i];
    const int w0 = 2 * g.nmax[0] + 1, w1 = 2 * g.nmax[1] + 1, w2 = 2 * g.nmax[2] + 1;
    const long long total = (long long)nion * w0 * w1 * w2;
    const double inv_rd = 1.0 / g.Rd, two_over = 2.0 / (sqrt(kPi) * g.Rd);
    double acc[kIonIonScalars];
#pragma unroll
    for (int s = 0; s < kIonIonScalars; ++s) acc[s] = 0.0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t % nion);
        long long u = t / nion;
        const int s2 = (int)(u % w2) - g.nmax[2];
        u /= w2;
        const int s1 = (int)(u % w1) - g.nmax[1];
        const int s0 = (int)(u / w1) - g.nmax[0];
        const double dx = cart[3 * j] + s0 * g.box[0] + s1 * g.box[3] + s2 * g.box[6] - xi;
        const double dy = cart[3 * j + 1] + s0 * g.box[1] + s1 * g.box[4] + s2 * g.box[7] - yi;
        const double dz = cart[3 * j + 2] + s0 * g.box[2] + s1 * g.box[5] + s2 * g.box[8] - zi;
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 > 1e-24 && r2 <= g.Rc * g.Rc) {
            const double r = sqrt(r2), ir = 1.0 / r, zz = Zi * Z[j];
            const double ec = erfc(r * inv_rd) * ir;
            acc[0] += zz * ec;
            acc[1] += Z[j];
            const double fpr = zz * (-two_over * exp(-r2 * inv_rd * inv_rd) * ir - ec * ir) * ir;     // f'(r) / r
            acc[2] += fpr * dx;
            acc[3] += fpr * dy;
            acc[4] += fpr * dz;
            acc[5] += fpr * dx * dx;
            acc[6] += fpr * dy * dy;
            acc[7] += fpr * dz * dz;
            acc[8] += fpr * dx * dy;
            acc[9] += fpr * dx * dz;
            acc[10] += fpr * dy * dz;
        }
    }
    block_reduce_store<kIonIonScalars>(acc, partial + (long long)i * gridDim.x * kIonIonScalars);
}

}  // namespace ofdft
