// Limited-memory BFGS on the device in "vector-free" form: per inner iteration ONE multi-dot sweep over the history
// (all inner products the two-loop recursion needs) and ONE multi-axpy sweep (direction, parameter update, gradient
// copy), instead of the 2m dependent dot / axpy pairs of the textbook recursion.  The recursion itself then runs on
// the (2m+1)-dimensional coefficient vector on the host (professad_amd/optimize.py), which on several GPUs needs
// only one small all-reduce per iteration.  Semantics follow the reference's fixed-step optimiser
// (src/professad/_optimizers/lbfgs/lbfgsnew.py:594-663: y = g - g_prev, s = t d, curvature test, two-loop, H_diag).
#pragma once
#include "pointwise_kernels.h"

namespace ofdft {

// history vectors are read once per sweep and the next sweep comes ~2 GB of traffic later: stream them (nt) so that chi,
// the gradient and the closure's own working set keep the caches
#ifndef OFDFT_LBFGS_NT
#define OFDFT_LBFGS_NT 1
#endif
__device__ __forceinline__ cplx lb_load2(const real* p, long long i) {
#if OFDFT_LBFGS_NT
    const real2_t t = __builtin_nontemporal_load(reinterpret_cast<const real2_t*>(p) + i);
    return mkc(t.x, t.y);
#else
    return reinterpret_cast<const cplx*>(p)[i];
#endif
}
__device__ __forceinline__ void lb_store2(real* p, long long i, cplx v) {
#if OFDFT_LBFGS_NT
    real2_t t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<real2_t*>(p) + i);
#else
    reinterpret_cast<cplx*>(p)[i] = v;
#endif
}

constexpr int kLbfgsMaxHist = 8;

struct LbfgsVecs {
    const real* S[kLbfgsMaxHist];   // stored steps, oldest first
    const real* Y[kLbfgsMaxHist];   // stored gradient differences
};
struct LbfgsCoef {
    double cs[kLbfgsMaxHist], cy[kLbfgsMaxHist], cg;
};

// number of scalars the sweep produces for K stored pairs:
//   for j < K: (s.S_j, y.S_j, g.S_j), then (s.Y_j, y.Y_j, g.Y_j); tail: s.s, s.y, y.y, g.s, g.y, g.g, |g|_1
__host__ __device__ constexpr int lbfgs_nscal(int K) { return 6 * K + 7; }

// Sweep 1: forms the candidate pair y = g - g_prev, s = t d (written to s_new / y_new; zeros when there is no
// previous step) and accumulates every inner product of {s, y, g} with the stored vectors and with each other.
template <int K>
__global__ __launch_bounds__(kRedThreads) void lbfgs_dots_kernel(LbfgsVecs v, const real* __restrict__ g,
                                                                 const real* __restrict__ g_prev,
                                                                 const real* __restrict__ d, real t, int have_prev,
                                                                 real* __restrict__ s_new, real* __restrict__ y_new,
                                                                 long long n, acc_t* __restrict__ partial) {
    constexpr int NS = lbfgs_nscal(K);
    acc_t acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    const long long n2 = n >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2 + (n & 1); i += (long long)gridDim.x * blockDim.x) {
        const bool tail = i == n2;          // odd length: the last element alone
        cplx gi, si = mkc(0.0, 0.0), yi = mkc(0.0, 0.0);
        if (!tail) gi = reinterpret_cast<const cplx*>(g)[i];
        else gi = mkc(g[n - 1], 0.0);
        if (have_prev) {
            cplx gp, di;
            if (!tail) {
                gp = lb_load2(g_prev, i);
                di = lb_load2(d, i);
            } else {
                gp = mkc(g_prev[n - 1], 0.0);
                di = mkc(d[n - 1], 0.0);
            }
            yi = mkc(gi.x - gp.x, gi.y - gp.y);
            si = mkc(t * di.x, t * di.y);
            if (!tail) {
                lb_store2(s_new, i, si);
                lb_store2(y_new, i, yi);
            } else {
                s_new[n - 1] = si.x;
                y_new[n - 1] = yi.x;
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            cplx a, b;
            if (!tail) {
                a = lb_load2(v.S[j], i);
                b = lb_load2(v.Y[j], i);
            } else {
                a = mkc(v.S[j][n - 1], 0.0);
                b = mkc(v.Y[j][n - 1], 0.0);
            }
            acc[3 * j + 0] += si.x * a.x + si.y * a.y;
            acc[3 * j + 1] += yi.x * a.x + yi.y * a.y;
            acc[3 * j + 2] += gi.x * a.x + gi.y * a.y;
            acc[3 * K + 3 * j + 0] += si.x * b.x + si.y * b.y;
            acc[3 * K + 3 * j + 1] += yi.x * b.x + yi.y * b.y;
            acc[3 * K + 3 * j + 2] += gi.x * b.x + gi.y * b.y;
        }
        acc[6 * K + 0] += si.x * si.x + si.y * si.y;
        acc[6 * K + 1] += si.x * yi.x + si.y * yi.y;
        acc[6 * K + 2] += yi.x * yi.x + yi.y * yi.y;
        acc[6 * K + 3] += gi.x * si.x + gi.y * si.y;
        acc[6 * K + 4] += gi.x * yi.x + gi.y * yi.y;
        acc[6 * K + 5] += gi.x * gi.x + gi.y * gi.y;
        acc[6 * K + 6] += fabs(gi.x) + fabs(gi.y);
    }
    block_reduce_store<NS>(acc, partial);
}

// Sweep 2: d = cg g + sum_j cs_j S_j + cy_j Y_j;  x += t d;  g_prev = g;  partial sums of |t d|.
template <int K>
__global__ __launch_bounds__(kRedThreads) void lbfgs_update_kernel(LbfgsVecs v, LbfgsCoef c, const real* __restrict__ g,
                                                                   real t, real* __restrict__ d, real* __restrict__ x,
                                                                   real* __restrict__ g_prev, long long n,
                                                                   acc_t* __restrict__ partial) {
    acc_t acc[1] = {0.0};
    const long long n2 = n >> 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        const cplx gi = reinterpret_cast<const cplx*>(g)[i];
        cplx di = mkc(c.cg * gi.x, c.cg * gi.y);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const cplx a = lb_load2(v.S[j], i), b = lb_load2(v.Y[j], i);
            di.x += c.cs[j] * a.x + c.cy[j] * b.x;
            di.y += c.cs[j] * a.y + c.cy[j] * b.y;
        }
        cplx xi = reinterpret_cast<cplx*>(x)[i];
        xi.x += t * di.x;
        xi.y += t * di.y;
        lb_store2(d, i, di);
        reinterpret_cast<cplx*>(x)[i] = xi;
        lb_store2(g_prev, i, gi);
        acc[0] += fabs(t * di.x) + fabs(t * di.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long i = n - 1;
        real di = c.cg * g[i];
        for (int j = 0; j < K; ++j) di += c.cs[j] * v.S[j][i] + c.cy[j] * v.Y[j][i];
        d[i] = di;
        x[i] += t * di;
        g_prev[i] = g[i];
        acc[0] += fabs(t * di);
    }
    block_reduce_store<1>(acc, partial);
}

}  // namespace ofdft
