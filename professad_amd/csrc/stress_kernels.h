// Per-term stress tensors sigma_ij = (1/Omega) dE/d eps_ij at fixed electron number (density ~ 1/volume): what the
// reference's get_stress (functional_tools.py:73-101) / System.__compute_stress (system.py:925-935) return by autograd.
// Closed forms: Hartree, TF, WT, LDA, PBE as in the reference's analytic test forms (tests/tools_for_tests.py:212-307,
// 367-472); vW, WGC99 and ion-electron in the exact discrete forms derived in oracle/stress.py.  Every kernel is a
// reduction to 7 numbers (xx, yy, zz, xy, xz, yz, + an energy-like scalar) over the half spectrum, or to 19 over the grid.
#pragma once
#include "ion_kernels.h"

namespace ofdft {

constexpr int kStressSpecScalars = 7;
constexpr int kStressRealScalars = 28;

__device__ __forceinline__ double half_weight(const SpecGeom& g, int z) {
    return (z == 0 || ((g.n2 & 1) == 0 && z == g.n2 / 2)) ? 1.0 : 2.0;
}

__device__ __forceinline__ void add_kk(double (&acc)[kStressSpecScalars], double f, double kx, double ky, double kz) {
    acc[0] += f * kx * kx;
    acc[1] += f * ky * ky;
    acc[2] += f * kz * kz;
    acc[3] += f * kx * ky;
    acc[4] += f * kx * kz;
    acc[5] += f * ky * kz;
}

enum { STRESS_HARTREE = 0, STRESS_VW = 1, STRESS_WT = 2, STRESS_HESS = 3 };

// a, b: UNNORMALISED spectra (b only for WT); scale = 1/N^2.
//   HARTREE: acc_ij += w 4 pi |n~|^2 k_i k_j / k^4,              acc[6] += w 4 pi |n~|^2 / k^2        (tools_for_tests.py:212-238)
//   VW:      acc_ij += -w |s~|^2 k_i k_j                                                              (oracle/stress.py)
//   WT:      acc_ij += w Re(a~ b~*) aux3(eta) (k_i k_j / k^2 - delta_ij / 3),  acc[6] += w Re(a~ b~*) shape(eta)   (:262-307)
//   HESS:    acc_ij += w Re(a~ b~*) k_i k_j     = -N^2 mean(b F^-1[-k_i k_j a~])    (Hessian term of a Laplacian-dependent GGA)
template <int OP>
__global__ __launch_bounds__(kRedThreads) void stress_spec_kernel(const cplx* __restrict__ a, const cplx* __restrict__ b,
                                                                  KGeom kg, double scale, double inv2kf,
                                                                  double* __restrict__ partial) {
    double acc[kStressSpecScalars] = {0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        spec_decode(kg.g, i, x, y, z);
        double kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        if (k2 == 0.0) continue;
        const double w = half_weight(kg.g, z) * scale;
        const cplx av = a[i];
        if (OP == STRESS_HARTREE) {
            const double f = w * 4.0 * kPi * (av.x * av.x + av.y * av.y) / (k2 * k2);
            add_kk(acc, f, kx, ky, kz);
            acc[6] += f * k2;
        } else if (OP == STRESS_VW) {
            add_kk(acc, -w * (av.x * av.x + av.y * av.y), kx, ky, kz);
        } else if (OP == STRESS_HESS) {
            const cplx bv = b[i];
            add_kk(acc, w * (av.x * bv.x + av.y * bv.y), kx, ky, kz);
        } else {
            const cplx bv = b[i];
            const double re = w * (av.x * bv.x + av.y * bv.y);
            const double eta = sqrt(k2) * inv2kf;
            const double lg = log(fabs((1.0 + eta) / (1.0 - eta)));
            const double lind = 0.5 + (1.0 - eta * eta) / (4.0 * eta) * lg;
            const double aux3 = eta / (lind * lind) * (0.5 / eta - 0.25 * (1.0 + 1.0 / (eta * eta)) * lg) + 6.0 * eta * eta;
            const double f = re * aux3 / k2;
            add_kk(acc, f, kx, ky, kz);
            const double third = re * aux3 / 3.0;
            acc[0] -= third;
            acc[1] -= third;
            acc[2] -= third;
            acc[6] += re * lindhard_shape(eta);
        }
    }
    block_reduce_store<kStressSpecScalars>(acc, partial);
}

// WGC99 nonlocal part (oracle/stress.py::wgc99_nl): spectra of A, B, C, P, Q, S (unnormalised), scale = 1/N^2
//   acc_ij += G (delta_ij / 3 - k_i k_j / k^2),  G = (w0' X0 + K1' X1 + K2' X2 + K3' X3) eta;   acc[6] += w0 X0 + K1 X1 + K2 X2 + K3 X3
struct WgcSpectra { const cplx* s[6]; };
__global__ __launch_bounds__(kRedThreads) void stress_wgc_kernel(WgcSpectra sp, KGeom kg, WgcSeries s, double scale,
                                                                 double* __restrict__ partial) {
    double acc[kStressSpecScalars] = {0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < kg.g.total; i += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        spec_decode(kg.g, i, x, y, z);
        double kx, ky, kz, k2;
        kvec_xyz(kg, x, y, z, kx, ky, kz, k2);
        if (k2 == 0.0) continue;
        const double w = half_weight(kg.g, z) * scale;
        const double eta = sqrt(k2) * s.inv2kf;
        double w0, w1, w2, w3;
        wgc_series<true>(eta, s, w0, w1, w2, w3);
        w0 *= s.pref;
        w1 *= s.pref;
        w2 *= s.pref;
        w3 *= s.pref;
        const double i6 = 1.0 / (6.0 * s.nref), i36 = 1.0 / (36.0 * s.nref * s.nref);
        const double K1 = -eta * w1 * i6;
        const double K2 = (eta * eta * w2 + (7.0 - s.gamma) * eta * w1) * i36;
        const double K3 = (eta * eta * w2 + (1.0 + s.gamma) * eta * w1) * i36;
        const double dK1 = -(w1 + eta * w2) * i6;
        const double common = 2.0 * eta * w2 + eta * eta * w3;
        const double dK2 = (common + (7.0 - s.gamma) * (w1 + eta * w2)) * i36;
        const double dK3 = (common + (1.0 + s.gamma) * (w1 + eta * w2)) * i36;
        const cplx A = sp.s[0][i], B = sp.s[1][i], C = sp.s[2][i], P = sp.s[3][i], Q = sp.s[4][i], S = sp.s[5][i];
        const double X0 = w * (P.x * A.x + P.y * A.y);
        const double X1 = w * (P.x * B.x + P.y * B.y + Q.x * A.x + Q.y * A.y);
        const double X2 = w * (P.x * C.x + P.y * C.y + S.x * A.x + S.y * A.y);
        const double X3 = w * (Q.x * B.x + Q.y * B.y);
        const double G = (w1 * X0 + dK1 * X1 + dK2 * X2 + dK3 * X3) * eta;
        add_kk(acc, -G / k2, kx, ky, kz);
        acc[0] += G / 3.0;
        acc[1] += G / 3.0;
        acc[2] += G / 3.0;
        acc[6] += w0 * X0 + K1 * X1 + K2 * X2 + K3 * X3;
    }
    block_reduce_store<kStressSpecScalars>(acc, partial);
}

// interpolate_recpot and its derivative with respect to |k| (zero slope beyond the table end: torch.minimum clamp)
__device__ __forceinline__ double recpot_value_d(const RecpotTable& t, double kabs, double& dv) {
    const double xs = fmin(kabs, t.ks[t.n - 1]);
    int idx = (int)ceil(xs * t.inv_dk) - 1;
    idx = max(0, min(idx, t.n - 2));
    while (idx > 0 && t.ks[idx] >= xs) --idx;
    while (idx < t.n - 2 && t.ks[idx + 1] < xs) ++idx;
    const double dx = t.ks[idx + 1] - t.ks[idx];
    const double u = (xs - t.ks[idx]) / dx, u2 = u * u, u3 = u2 * u;
    const double v = (1.0 - 3.0 * u2 + 2.0 * u3) * t.y[idx] + (u - 2.0 * u2 + u3) * t.m[idx] * dx +
                     (3.0 * u2 - 2.0 * u3) * t.y[idx + 1] + (-u2 + u3) * t.m[idx + 1] * dx;
    const double d = ((6.0 * u2 - 6.0 * u) * t.y[idx] + (3.0 * u2 - 4.0 * u + 1.0) * t.m[idx] * dx +
                      (-6.0 * u2 + 6.0 * u) * t.y[idx + 1] + (3.0 * u2 - 2.0 * u) * t.m[idx + 1] * dx) / dx;
    dv = (kabs < t.ks[t.n - 1]) ? d : 0.0;
    if (kabs != 0.0) {
        dv += 8.0 * kPi * t.z / (kabs * kabs * kabs);
        return v - 4.0 * kPi * t.z / (kabs * kabs);
    }
    dv = 0.0;
    return v;
}

// ion-electron: acc_ij += core v~'(k) k_i k_j / |k|,  acc[6] += core v~(k),  core = w Re(S_k conj(n^_k)) (oracle/stress.py)
// S exact (cart != nullptr) or PME (conj(b Q^)); nk unnormalised.
__global__ __launch_bounds__(kRedThreads) void stress_ion_kernel(const cplx* __restrict__ nk, const cplx* __restrict__ Qk,
                                                                 KGeom kg, const cplx* __restrict__ b0,
                                                                 const cplx* __restrict__ b1, const cplx* __restrict__ b2,
                                                                 const double* __restrict__ cart, int nion, RecpotTable tab,
                                                                 double* __restrict__ partial) {
    double acc[kStressSpecScalars] = {0, 0, 0, 0, 0, 0, 0};
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
            S = cconj(cmul(cmul(cmul(b0[x], b1[y]), b2[z]), Qk[i]));
        }
        const cplx n = nk[i];
        const double core = half_weight(kg.g, z) * (S.x * n.x + S.y * n.y);
        const double kabs = (k2 != 0.0) ? sqrt(k2) : 0.0;
        double dv;
        const double v = recpot_value_d(tab, kabs, dv);
        acc[6] += core * v;
        if (k2 != 0.0) add_kk(acc, core * dv / kabs, kx, ky, kz);
    }
    block_reduce_store<kStressSpecScalars>(acc, partial);
}

// real-space sums: [0] n^(5/3); [1] LDA-x (e - v n); [2] LDA-c (e - v n); PBE-x: [3..8] d_i n d_j n df/dg, [9] |grad n|^2 df/dg,
// [10] f - n df/dn; PBE-c: [11..16], [17], [18]; GGA kinetic (Pauli part): [19..24], [25], [26]
// (tools_for_tests.py:241-243, 367-472; the kinetic GGA has the same form, :46-118); [27] G tau_TF of vWGTF
__global__ __launch_bounds__(kRedThreads) void stress_real_kernel(const double* __restrict__ n, const double* __restrict__ gx,
                                                                  const double* __restrict__ gy, const double* __restrict__ gz,
                                                                  long long npts, unsigned mask, GgaSel sel,
                                                                  double gtf_inv_n0, int gtf_kind, double* __restrict__ partial,
                                                                  double* __restrict__ lapn = nullptr) {
    double acc[kStressRealScalars];
#pragma unroll
    for (int i = 0; i < kStressRealScalars; ++i) acc[i] = 0.0;
    const bool do_px = mask & OFDFT_PBE_X, do_pc = mask & OFDFT_PBE_C, do_pk = mask & OFDFT_GGA_K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
        const double d = n[i];
        if (mask & OFDFT_TF) acc[0] += cbrt(d * d) * d;
        if (mask & OFDFT_VWGTF) {
            double e, v;
            vwgtf_point(d, cbrt(d), 0.3 * cbrt(9.0 * kPi * kPi * kPi * kPi), gtf_inv_n0, gtf_kind, e, v);
            acc[27] += e;
        }
        if (mask & (OFDFT_LDA_X | OFDFT_PZ_C | OFDFT_PW_C | OFDFT_CHACHIYO_C)) {
            const XcLocal r = lda_point(d, mask);
            acc[1] += r.ex - r.vx * d;
            acc[2] += r.ec - r.vc * d;
        }
        if (do_px || do_pc || do_pk) {
            const double a = gx[i], b = gy[i], c = gz[i];
            const double g2 = a * a + b * b + c * c;
            for (int which = 0; which < 3; ++which) {
                if (which == 0 ? !do_px : (which == 1 ? !do_pc : !do_pk)) continue;
                const GgaSel one{which == 0, which == 1, which == 2, sel.kkind, sel.kmu, 0.0, 0.0, 0.0};
                PbePoint p = {0.0, 0.0, 0.0, 0.0, 0.0};
                double* o = acc + 3 + 8 * which;
                if (which == 2 && lapn) {
                    // f(n, g, l = lap n):  sigma_ij = delta_ij mean(f - n f_n - 2 g f_g - l f_l) - 2 mean(f_g d_i n d_j n)
                    //                                 - 2 mean(f_l d_i d_j n); the last term is reduced in k-space
                    //                                 (stress_spec_kernel<STRESS_HESS>) from the spectrum of f_l, left in lapn
                    GgaSel full = sel;
                    full.x = full.c = 0;
                    full.k = 1;
                    double dfdl;
                    const double l = lapn[i];
                    pg_laplacian_point(d, g2, l, full, p, dfdl);
                    o[7] -= l * dfdl;
                    lapn[i] = dfdl;
                } else {
                    p = pbe_point(d, g2, one);
                }
                o[0] += a * a * p.dfdg;
                o[1] += b * b * p.dfdg;
                o[2] += c * c * p.dfdg;
                o[3] += a * b * p.dfdg;
                o[4] += a * c * p.dfdg;
                o[5] += b * c * p.dfdg;
                o[6] += g2 * p.dfdg;
                o[7] += (which == 0 ? p.fx : (which == 1 ? p.fc : p.fk)) - d * p.dfdn;
            }
        }
    }
    block_reduce_store<kStressRealScalars>(acc, partial);
}

}  // namespace ofdft
