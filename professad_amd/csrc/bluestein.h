// Arbitrary-length line transforms by Bluestein's chirp-z algorithm on top of the power-of-two register/LDS line FFT:
//   X_k = w_k * sum_n (x_n w_n) conj(w)_{k-n},  w_n = exp(-i pi n^2 / N)
// i.e. one cyclic convolution of padded length M >= 2N - 1 (two M-point FFTs per line, the filter spectrum precomputed).
// This is the engine's path for grid extents that are not powers of two -- which is what the reference's own
// System.ecut2shape (system.py:75-89, always odd extents) produces: O(N log N) per line instead of the O(N^2) of the
// plain DFT kernels (kept for extents above 512).
#pragma once
#include "fft_kernels.h"

namespace ofdft {

// mode: 0 = complex lines along x (axis 0) or y (axis 1) of the internal half-spectrum layout, in place
//       1 = real rows -> half spectrum along z (r2c);  2 = half spectrum -> real rows along z (c2r, times `scale`)
struct BsArgs {
    int N;          // line length
    int mode, axis, inv;
    long long nlines;
    real scale;
};

// one launch may cover several arrays (blockIdx.y): the unfused pipeline transforms its spectra in groups of three (gradient,
// flux, the WGC99 triples) and small grids are bound by their launch count, not by bytes
constexpr int kBsBatch = 4;
struct BsIo {
    cplx* spec[kBsBatch];
    const real* rin[kBsBatch];
    real* rout[kBsBatch];
};

template <int M>
__global__ __launch_bounds__(PassCfg<M>::TPB) void bluestein_kernel(BsIo io, SpecGeom g, BsArgs b,
                                                                    const cplx* __restrict__ chirp,   // w_n, n < N
                                                                    const cplx* __restrict__ filt,    // FFT_M(b) / M
                                                                    const cplx* __restrict__ twM) {
    constexpr int P = PassCfg<M>::P, E = PassCfg<M>::E, LPW = PassCfg<M>::LPW;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    cplx* __restrict__ spec = io.spec[0];
    const real* __restrict__ rin = io.rin[0];
    real* __restrict__ rout = io.rout[0];
#pragma unroll
    for (int a = 1; a < kBsBatch; ++a)          // (select without dynamic indexing into the kernel arguments)
        if ((int)blockIdx.y == a) {
            spec = io.spec[a];
            rin = io.rin[a];
            rout = io.rout[a];
        }
    // complex lines (mode 0): consecutive lanes take consecutive LINES, enumerated kz-fastest, so that a wave's accesses
    // for one element index fall on the 8 x 16-B runs of the block-8 layout; z rows (modes 1, 2): consecutive lanes take
    // consecutive elements of one contiguous row
    const int j = b.mode == 0 ? tid / LPW : tid % P;
    const int l = b.mode == 0 ? tid % LPW : tid / P;
    const long long L = (long long)blockIdx.x * LPW + l;
    const bool valid = L < b.nlines;
    const int N = b.N;
    // line coordinates
    int x = 0, y = 0, kz = 0;
    if (valid) {
        if (b.mode == 0) {
            kz = (int)(L % g.nzc);
            if (b.axis == 0) y = (int)(L / g.nzc);
            else x = (int)(L / g.nzc);
        } else {
            x = (int)(L / g.n1);
            y = (int)(L % g.n1);
        }
    }
    const bool conj_in = b.inv != 0;      // inverse transform = conjugated chirps and filter
    // Round 3: the kernel sat 60 % of its wave cycles waiting (profiles/r03_sq_chirpz_255.md): of its 30 loads per wave only 8
    // fetch data -- chirp (twice), filter and the stage twiddles were requested one by one where they were used, each a
    // dependent L2 round trip between barrier-separated phases.  Now every table value a lane needs is requested up front
    // with its data (the chirp once: it serves the pre- AND the post-multiplication; M >= 2 N - 1 means only the lower half
    // of the register slots ever holds data) and the M twiddles are staged in LDS behind the line buffers.
    constexpr int EH = (E + 1) / 2;           // slots that can hold an element e = j + P q < N <= (M + 1) / 2
    cplx v[E], wch[EH], fl[E];
    cplx* tw_l = reinterpret_cast<cplx*>(lds + LPW * LineBuf<M>::STRIDE);
    constexpr int TWC = (M + PassCfg<M>::TPB - 1) / PassCfg<M>::TPB;
    cplx twr[TWC];
#pragma unroll
    for (int t = 0; t < TWC; ++t) {
        const int i = tid + t * PassCfg<M>::TPB;
        twr[t] = twM[i < M ? i : 0];
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int e = j + P * q;
        cplx a = mkc(0.0, 0.0);
        if (q < EH && valid && e < N) {
            if (b.mode == 0) {
                a = spec[b.axis == 0 ? spec_index(g, e, y, kz) : spec_index(g, x, e, kz)];
            } else if (b.mode == 1) {
                a = mkc(rin[((long long)x * g.n1 + y) * g.n2 + e], 0.0);
            } else {            // rebuild the Hermitian line; imaginary parts of k = 0 (and Nyquist) are ignored like irfftn
                const int k = (e < g.nzc) ? e : N - e;
                a = spec[spec_index(g, x, y, k)];
                if (e >= g.nzc) a.y = -a.y;
                if (e == 0 || (2 * e == N)) a.y = 0.0;
            }
        }
        v[q] = a;
    }
#pragma unroll
    for (int q = 0; q < EH; ++q) {
        const int e = j + P * q;
        wch[q] = chirp[e < N ? e : 0];
        if (conj_in) wch[q].y = -wch[q].y;
    }
    // (the filter values: up front for the short transforms; for M >= 256 the 2 E more registers cost more occupancy than the
    // one round trip they save -- 255^3 11.4 -> 13.2 ms with them prefetched -- so there they are requested after the first FFT)
    constexpr bool PREF = M <= 128;
    if constexpr (PREF) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            fl[q] = filt[j + P * q];
            if (conj_in) fl[q].y = -fl[q].y;
        }
    }
#pragma unroll
    for (int t = 0; t < TWC; ++t) {
        const int i = tid + t * PassCfg<M>::TPB;
        if (i < M) tw_l[i] = twr[t];
    }
#pragma unroll
    for (int q = 0; q < EH; ++q) v[q] = cmul(v[q], wch[q]);       // (slots without an element hold zero)
    real* mine = lds + l * LineBuf<M>::STRIDE;
    __syncthreads();                                               // the staged twiddles
    line_fft<M, false>(v, j, mine, tw_l);
    if constexpr (!PREF) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            fl[q] = filt[j + P * q];
            if (conj_in) fl[q].y = -fl[q].y;
        }
    }
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = cmul(v[q], fl[q]);
    __syncthreads();
    line_fft<M, true>(v, j, mine, tw_l);
#pragma unroll
    for (int q = 0; q < EH; ++q) {
        const int e = j + P * q;
        if (!valid || e >= N) continue;
        const cplx r = cmul(v[q], wch[q]);
        if (b.mode == 0) {
            spec[b.axis == 0 ? spec_index(g, e, y, kz) : spec_index(g, x, e, kz)] = r;
        } else if (b.mode == 1) {
            if (e < g.nzc) spec[spec_index(g, x, y, e)] = r;
        } else {
            rout[((long long)x * g.n1 + y) * g.n2 + e] = r.x * b.scale;
        }
    }
}

}  // namespace ofdft
