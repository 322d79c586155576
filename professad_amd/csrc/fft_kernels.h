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
    int blk0 = 0;          // first workgroup of a z-kernel launch that covers only a range of rows (x-chunked pipeline)
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
    // optional restriction to a range of an outer index (x-chunked y passes): group G = L / d enumerates (b, xl) with
    // xl < gc; it stands for group (G / gc) * gn + g0 + G % gc of the full array.  gc == 0: no remapping.
    int gc = 0, gn = 0, g0 = 0;
    int blk0 = 0;   // first workgroup (in units of the kernel's lines-per-workgroup) of a launch that covers a line range
    int kz0 = 0;    // fused x pass on a kz chunk of an exchange buffer: kz of line 0 (the lines are numbered inside the chunk)
};

#ifndef OFDFT_CPASS_TPB
#define OFDFT_CPASS_TPB 256      // threads per workgroup of the line passes (measured: 128 and 512 are not faster at 256^3)
#endif
template <int LEN> struct PassCfg {
    static constexpr int P = Plan<LEN>::P;
    static constexpr int E = Plan<LEN>::E;
    static constexpr int TPB = (8 * P > OFDFT_CPASS_TPB) ? 8 * P : OFDFT_CPASS_TPB;
    static constexpr int LPW = TPB / P;
    static constexpr int LSTR = kCXMul * LineBuf<LEN>::STRIDE;      // reals per line buffer (fp32 build: complex elements, StageP CX)
    static constexpr size_t LDS = (Plan<LEN>::NST > 1) ? sizeof(real) * LPW * LSTR : 0;
};

// uniform base of a tile (first line of the workgroup) and the per-lane byte offset of line L, element j
__device__ __forceinline__ long long line_base(const LineMap& m, long long L) {
    long long grp = L / m.d;
    if (m.gc) grp = (grp / m.gc) * m.gn + m.g0 + grp % m.gc;
    return grp * m.sb + (L % m.d) * (long long)m.sl;
}

struct ArrList { cplx* p[16]; };

// Cache policy of the big streaming accesses (aux = 2: nt).  A pass reads every spectrum line once and its successor
// reads it only after >= one whole array has gone through the caches, so keeping the lines only evicts what IS reused
// (kernel tables, twiddles, the half-line partner tiles of the fused x pass).  Measured at 256^3: y pass nt loads +
// stores 3.62 -> 3.32 ms per evaluation.
#ifndef OFDFT_CPASS_LD_AUX
#define OFDFT_CPASS_LD_AUX 2
#endif
#ifndef OFDFT_CPASS_ST_AUX
#define OFDFT_CPASS_ST_AUX 2
#endif
#ifndef OFDFT_XF_LD_AUX
#define OFDFT_XF_LD_AUX 0
#endif
#ifndef OFDFT_XF_BATCH_TABLES
#define OFDFT_XF_BATCH_TABLES 1    // table-driven fused x pass (WGC99): request a half's coefficients in one batch
#endif
#ifndef OFDFT_XF_ST_AUX
#define OFDFT_XF_ST_AUX 2     // fused x pass: nt stores (-8 % on the WGC99 kernel); nt LOADS cost +14 % (partner half-lines in L2)
#endif
#ifndef OFDFT_ZS_LD_AUX
#define OFDFT_ZS_LD_AUX 0
#endif
#ifndef OFDFT_ZS_ST_AUX
#define OFDFT_ZS_ST_AUX 0
#endif
#ifndef OFDFT_ZR_LD_AUX
#define OFDFT_ZR_LD_AUX 0
#endif
#ifndef OFDFT_ZR_ST_AUX
#define OFDFT_ZR_ST_AUX 0
#endif

// OFDFT_CPASS_TILES = 2: a workgroup takes two consecutive tiles and requests both tiles' lines up front, so that the second
// tile's loads are in flight during the first tile's transform and the first tile's stores during the second's (a 256^3 y pass
// is 8 256 waves -- what the chip holds at once -- so with one tile per workgroup every workgroup loads, transforms and stores
// in lock step: the kernel's time is the SUM of the three phases)
#ifndef OFDFT_CPASS_TILES
#define OFDFT_CPASS_TILES 1
#endif
template <int LEN, bool INV>
__global__ __launch_bounds__(PassCfg<LEN>::TPB) void cpass_kernel(ArrList arrs, LineMap m_main, LineMap m_rem,
                                                                   int main_blocks, long long rem_offset,
                                                                   const cplx* __restrict__ tw) {
    cplx* data0 = arrs.p[blockIdx.y];         // one launch may cover several spectra (grid.y)
    constexpr int P = PassCfg<LEN>::P, E = PassCfg<LEN>::E, LPW = PassCfg<LEN>::LPW, T = OFDFT_CPASS_TILES;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    using PL = Plan<LEN>;
    cplx v[T][E];
    cplx* ubs[T];
    unsigned voffs[T];
    bool valids[T];
    long long se_us[T];
    int js[T], ls[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int blk = (int)blockIdx.x * T + t;    // (the launcher rounds the main part up to whole groups of T tiles)
        // one grid covers the block-8 main part and the dense remainder planes (Nyquist plane)
        const bool in_rem = blk >= main_blocks;
        const LineMap m = in_rem ? m_rem : m_main;
        cplx* data = in_rem ? data0 + rem_offset : data0;
        const int bid = in_rem ? blk - main_blocks : blk + m.blk0;
        const int l_lo = tid % m.lf;
        const int j = (tid / m.lf) % P;
        const int l = (tid / (m.lf * P)) * m.lf + l_lo;
        const long long L0 = (long long)bid * LPW;
        const long long L = L0 + l;
        const bool valid = L < m.nlines;
        const long long b0 = uniform64(line_base(m, L0));
        ubs[t] = data + b0;                                     // wave-uniform
        voffs[t] = valid ? (unsigned)((line_base(m, L) - b0 + (long long)j * m.se) * kCB) : 0u;
        se_us[t] = uniform64(m.se);                             // uniform element stride; slot q holds element j + cin(q) / j + cout(q)
        valids[t] = valid;
        js[t] = j;
        ls[t] = l;
#pragma unroll
        for (int q = 0; q < E; ++q)
            v[t][q] = (PL::slot_in(q) && valid && PL::lane_in(j, q)) ? buf_load_c_aux<OFDFT_CPASS_LD_AUX>(ubs[t] + PL::cin(q) * se_us[t], voffs[t])
                                                                       : mkc(0.0, 0.0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t > 0) __syncthreads();                             // the line buffers are reused
        line_fft<LEN, INV, kCX>(v[t], js[t], lds + ls[t] * PassCfg<LEN>::LSTR, tw);
        if (valids[t]) {
#pragma unroll
            for (int q = 0; q < E; ++q)
                if (PL::slot_out(q) && PL::lane_out(js[t], q)) buf_store_c_aux<OFDFT_CPASS_ST_AUX>(ubs[t] + PL::cout(q) * se_us[t], voffs[t], v[t][q]);
        }
    }
}

// Register slots from the exit pattern of a transform (element j + cout(q)) to its entry pattern (element j + cin(q)),
// through the line buffer: needed between a forward and an inverse transform of the same line when the plan is not EXACT
// (extents with factors 3 / 5; for the power-of-two plans the two patterns coincide and this is a no-op).
template <class PL, bool WAVE>
__device__ __forceinline__ void repattern_out_to_in(cplx (&v)[PL::E], int j, real* line) {
    if constexpr (!PL::EXACT) {
        constexpr int E = PL::E;
        real re[E];
        exchange_sync<WAVE>();
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(j, q)) line[lpos<PL>(j + PL::cout(q))] = v[q].x;
        exchange_sync<WAVE>();
#pragma unroll
        for (int q = 0; q < E; ++q) re[q] = (PL::slot_in(q) && PL::lane_in(j, q)) ? line[lpos<PL>(j + PL::cin(q))] : (real)0.0;
        exchange_sync<WAVE>();
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(j, q)) line[lpos<PL>(j + PL::cout(q))] = v[q].y;
        exchange_sync<WAVE>();
#pragma unroll
        for (int q = 0; q < E; ++q)
            v[q] = (PL::slot_in(q) && PL::lane_in(j, q)) ? mkc(re[q], line[lpos<PL>(j + PL::cin(q))]) : mkc(0.0, 0.0);
        exchange_sync<WAVE>();
    }
}

// ----------------------------------------------------------------------------------------------
// index derivative along y in ONE pass: forward FFT of the line, times i f_b (integer frequency, Nyquist positive as in
// functional_tools.py:152-154) times `scale`, inverse FFT.  Out of place or in place (in == out).
// fwd != nullptr (round 4): the forward transform of the line is stored there as well (in place: fwd == in) -- the density
// spectrum needs BOTH its y-forward (for the x pass) and D_b n, which were two passes reading the same array.
template <int LEN>
__global__ __launch_bounds__(PassCfg<LEN>::TPB) void yderiv_kernel(const cplx* in, cplx* out,
                                                                   LineMap m_main, LineMap m_rem, int main_blocks,
                                                                   long long rem_offset, const cplx* __restrict__ tw,
                                                                   real scale, cplx* fwd) {
    constexpr int P = PassCfg<LEN>::P, E = PassCfg<LEN>::E, LPW = PassCfg<LEN>::LPW;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    const bool in_rem = (int)blockIdx.x >= main_blocks;
    const LineMap m = in_rem ? m_rem : m_main;
    const long long roff = in_rem ? rem_offset : 0;
    const int bid = in_rem ? (int)blockIdx.x - main_blocks : (int)blockIdx.x + m.blk0;
    const int l_lo = tid % m.lf;
    const int j = (tid / m.lf) % P;
    const int l = (tid / (m.lf * P)) * m.lf + l_lo;
    const long long L0 = (long long)bid * LPW;
    const long long L = L0 + l;
    const bool valid = L < m.nlines;
    const long long b0 = uniform64(line_base(m, L0));
    const unsigned voff = valid ? (unsigned)((line_base(m, L) - b0 + (long long)j * m.se) * kCB) : 0u;
    const long long se_u = uniform64(m.se);
    using PL = Plan<LEN>;
    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q)
        v[q] = (PL::slot_in(q) && valid && PL::lane_in(j, q)) ? buf_load_c(in + roff + b0 + PL::cin(q) * se_u, voff) : mkc(0.0, 0.0);
    real* mine = lds + l * PassCfg<LEN>::LSTR;
    line_fft<LEN, false, kCX>(v, j, mine, tw);
    if (fwd && valid) {       // (every load of the workgroup's lines precedes the transform's barriers: in place is safe)
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(j, q)) buf_store_c_aux<OFDFT_CPASS_ST_AUX>(fwd + roff + b0 + PL::cout(q) * se_u, voff, v[q]);
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int e = j + PL::cout(q);
        const real f = scale * (real)(e <= LEN / 2 ? e : e - LEN);
        v[q] = mkc(-f * v[q].y, f * v[q].x);
    }
    __syncthreads();
    repattern_out_to_in<PL, false>(v, j, mine);
    line_fft<LEN, true, kCX>(v, j, mine, tw);
    if (valid) {
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(j, q)) buf_store_c_aux<OFDFT_CPASS_ST_AUX>(out + roff + b0 + PL::cout(q) * se_u, voff, v[q]);
    }
}

// ----------------------------------------------------------------------------------------------
// y pass of the slab-decomposed path, writing / reading the all-to-all buffers directly (no pack / un-pack copies).
// Exchange layout (the same for both directions, so the fused x pass can work in place on what it received):
//   buffer = [peer][xl][array][ main: (b, yl, kin) | planes: (plane, yl) ]      xl: x inside a rank's x-slab
// i.e. per (peer, xl) one record of `narr` arrays of arr_sz = nb*nyl*8 + nrem*nyl elements.  On the y-slab rank
// (peer-major == x-major) this is a regular [x][array][...] array whose x lines have stride `rec`; on the x-slab
// rank a y line is piecewise: y = peer*nyl + yl.
struct XchgGeom {
    int nxl, nyl, log_nyl, nb, nrem;
    long long arr_sz;      // elements of one array in one x-plane record
    long long rec;         // narr * arr_sz
    long long chunk;       // nxl * rec: elements per peer
    int kb0 = 0;           // first kz block of the chunk this geometry describes (chunk-major exchange buffers; nb = its block count)
};
template <int LEN, bool INV>
__global__ __launch_bounds__(PassCfg<LEN>::TPB) void ypass_xchg_kernel(ArrList arrs, cplx* __restrict__ buf, XchgGeom xg,
                                                                       LineMap m_main, LineMap m_rem, int main_blocks,
                                                                       long long rem_offset,
                                                                       const cplx* __restrict__ tw) {
    constexpr int P = PassCfg<LEN>::P, E = PassCfg<LEN>::E, LPW = PassCfg<LEN>::LPW;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    const int a = blockIdx.y;
    const bool in_rem = (int)blockIdx.x >= main_blocks;
    const LineMap m = in_rem ? m_rem : m_main;
    cplx* data = arrs.p[a] + (in_rem ? rem_offset : 0);
    const int bid = in_rem ? (int)blockIdx.x - main_blocks : (int)blockIdx.x + m.blk0;
    const int l_lo = tid % m.lf;
    const int j = (tid / m.lf) % P;
    const int l = (tid / (m.lf * P)) * m.lf + l_lo;
    const long long L0 = (long long)bid * LPW;
    const long long L = L0 + l;
    const bool valid = L < m.nlines;
    const long long b0 = uniform64(line_base(m, L0));
    cplx* ub = data + b0;
    const unsigned voff = valid ? (unsigned)((line_base(m, L) - b0 + (long long)j * m.se) * kCB) : 0u;
    // buffer side: line part (per lane) and element part (y = element index -> peer, yl)
    using PL = Plan<LEN>;
    long long lb;
    int es;
    if (!in_rem) {               // L = (b*nxl + xl)*8 + kin
        const long long r = L >> 3;
        const int kin = (int)(L & 7), xl = (int)(r % xg.nxl);
        const long long b = r / xg.nxl - xg.kb0;       // block inside the chunk
        lb = xl * xg.rec + a * xg.arr_sz + b * xg.nyl * 8 + kin;
        es = 8;
    } else {                     // L = plane*nxl + xl
        const int xl = (int)(L % xg.nxl);
        const long long plane = L / xg.nxl;
        lb = xl * xg.rec + a * xg.arr_sz + (long long)xg.nb * xg.nyl * 8 + plane * xg.nyl;
        es = 1;
    }
    // position of line element e in the exchange buffer (power-of-two plans: nyl is a power of two as well)
    auto bpos = [&](int e) -> long long {
        int peer, yl;
        if constexpr (PL::EXACT) {
            peer = e >> xg.log_nyl;
            yl = e & (xg.nyl - 1);
        } else {
            peer = e / xg.nyl;
            yl = e - peer * xg.nyl;
        }
        return lb + (long long)peer * xg.chunk + (long long)yl * es;
    };
    const long long se_u = uniform64(m.se);
    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const bool on = valid && PL::slot_in(q) && PL::lane_in(j, q);
        if (INV) v[q] = on ? nt_load_c(buf + bpos(j + PL::cin(q))) : mkc(0.0, 0.0);
        else v[q] = on ? buf_load_c_aux<OFDFT_CPASS_LD_AUX>(ub + PL::cin(q) * se_u, voff) : mkc(0.0, 0.0);
    }
    line_fft<LEN, INV, kCX>(v, j, lds + l * PassCfg<LEN>::LSTR, tw);
    if (valid) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            if (!(PL::slot_out(q) && PL::lane_out(j, q))) continue;
            if (INV) buf_store_c_aux<OFDFT_CPASS_ST_AUX>(ub + PL::cout(q) * se_u, voff, v[q]);
            else nt_store_c(buf + bpos(j + PL::cout(q)), v[q]);
        }
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
    static constexpr size_t LDS = sizeof(real) * RPW * RS;
};

struct PreIdentity {
    __device__ __forceinline__ real operator()(real a, long long) const { return a; }
};

template <int M, class Pre>
__global__ __launch_bounds__(ZCfg<M>::TPB) void zfwd_kernel(const real* __restrict__ in, cplx* __restrict__ spec,
                                                            SpecGeom g, const cplx* __restrict__ twM,
                                                            const cplx* __restrict__ twN, Pre pre) {
    constexpr int P = ZCfg<M>::P, E = ZCfg<M>::E, RPW = ZCfg<M>::RPW, RS = ZCfg<M>::RS, TPB = ZCfg<M>::TPB;
    constexpr int N2 = 2 * M;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    const int j = tid % P, r = tid / P;
    const long long row0 = (long long)blockIdx.x * RPW;
    const long long row = row0 + r;
    const bool valid = row < g.nrows;
    cplx v[E];
    if (valid) {
        const cplx* ub = reinterpret_cast<const cplx*>(in + row0 * N2);       // wave-uniform
        const unsigned voff = (unsigned)((r * M + j) * kCB);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cplx t = buf_load_c(ub + q * P, voff);
            const long long e = row * N2 + 2 * (j + P * q);
            v[q] = mkc(pre(t.x, e), pre(t.y, e + 1));
        }
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mkc(0.0, 0.0);
    }
    real* mine = lds + r * RS;
    line_fft<M, false>(v, j, mine, twM);

    // ---- split post-processing through LDS: X[k] = Ev[k] + W_N^k Od[k]
    real cr_k[E], cr_m[E], c0r = 0.0, c0i = 0.0;
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
        const real ci_k = lds[rr * RS + lpad(k)];
        const real ci_m = lds[rr * RS + lpad(mk)];
        const cplx ev = mkc(0.5 * (cr_k[it] + cr_m[it]), 0.5 * (ci_k - ci_m));
        const cplx od = mkc(0.5 * (ci_k + ci_m), -0.5 * (cr_k[it] - cr_m[it]));
        const cplx X = cadd(ev, cmul(twN[k], od));
        const long long rg = row0 + rr;
        // block b of rows row0.. is one contiguous 128-B-per-row run: uniform base + (rr*8+kin)*16
        const int bu = (it * TPB) / (8 * RPW);          // compile-time after unrolling: uniform part of b
        if (rg < g.nrows)
            buf_store_c(spec + ((long long)bu * g.nrows + row0) * 8,
                        (unsigned)((((long long)(b - bu) * g.nrows + rr) * 8 + kin) * kCB), X);
    }
    if (tid < RPW) {
        c0i = lds[tid * RS];
        const long long rg = row0 + tid;
        if (rg < g.nrows) spec[g.main_count + rg] = mkc(c0r - c0i, 0.0);
    }
}

struct PostScale {
    real s;
    __device__ __forceinline__ real operator()(real a, long long) const { return a * s; }
};

// z pass, inverse c2r (imaginary parts of the kz=0 and Nyquist entries are ignored, as irfftn does)
template <int M, class Post>
__global__ __launch_bounds__(ZCfg<M>::TPB) void zinv_kernel(const cplx* __restrict__ spec, real* __restrict__ out,
                                                            SpecGeom g, const cplx* __restrict__ twM,
                                                            const cplx* __restrict__ twN, Post post) {
    constexpr int P = ZCfg<M>::P, E = ZCfg<M>::E, RPW = ZCfg<M>::RPW, RS = ZCfg<M>::RS, TPB = ZCfg<M>::TPB;
    constexpr int N2 = 2 * M;
    extern __shared__ __attribute__((aligned(16))) real lds[];
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
        const int bu = (it * TPB) / (8 * RPW);
        xs[it] = (rg < g.nrows) ? buf_load_c(spec + ((long long)bu * g.nrows + row0) * 8,
                                             (unsigned)((((long long)(b - bu) * g.nrows + rr) * 8 + kin) * kCB))
                                : mkc(0.0, 0.0);
    }
    const real nyq = (valid && j == 0) ? spec[g.main_count + row].x : 0.0;
#pragma unroll
    for (int it = 0; it < E; ++it) {
        const int idx = tid + it * TPB;
        const int kin = idx & 7, rr = (idx >> 3) % RPW, b = idx / (8 * RPW);
        lds[rr * RS + lpad(8 * b + kin)] = xs[it].x;
    }
    __syncthreads();
    real* mine = lds + r * RS;
    real xr_k[E], xr_m[E];
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
        const real xi_k = mine[lpad(k)], xi_m = mine[lpad(mk)];
        if (k == 0) {
            v[q] = mkc(xr_k[q] + nyq, xr_k[q] - nyq);
        } else {
            const cplx ev = mkc(xr_k[q] + xr_m[q], xi_k - xi_m);
            const cplx d = mkc(xr_k[q] - xr_m[q], xi_k + xi_m);
            const cplx od = cmul(d, cconj(twN[k]));
            v[q] = mkc(ev.x - od.y, ev.y + od.x);
        }
    }
    line_fft<M, true>(v, j, mine, twM);
    if (valid) {
        cplx* ub = reinterpret_cast<cplx*>(out + row0 * N2);                  // wave-uniform
        const unsigned voff = (unsigned)((r * M + j) * kCB);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const long long e = row * N2 + 2 * (j + P * q);
            buf_store_c(ub + q * P, voff, mkc(post(v[q].x, e), post(v[q].y, e + 1)));
        }
    }
}

// ----------------------------------------------------------------------------------------------
// generic (any extent) naive DFT kernels -- correctness path for non power-of-two grids
static __global__ void gen_r2c_z_kernel(const real* __restrict__ in, cplx* __restrict__ spec, SpecGeom g,
                                 const cplx* __restrict__ tw2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    const real* rowp = in + ((long long)x * g.n1 + y) * g.n2;
    real sr = 0.0, si = 0.0;
    int t = 0;
    for (int z = 0; z < g.n2; ++z) {
        const cplx w = tw2[t];
        sr += rowp[z] * w.x;
        si += rowp[z] * w.y;
        t += kz;
        if (t >= g.n2) t -= g.n2;
    }
    spec[i] = mkc(sr, si);
}

// out-of-place complex DFT along axis 0 (x) or 1 (y)
static __global__ void gen_c2c_kernel(const cplx* __restrict__ in, cplx* __restrict__ out, SpecGeom g, int axis, int inv,
                               const cplx* __restrict__ tw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    const int n = axis == 0 ? g.n0 : g.n1;
    const int k = axis == 0 ? x : y;
    real sr = 0.0, si = 0.0;
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
    out[i] = mkc(sr, si);
}

template <class Post>
__global__ void gen_c2r_z_kernel(const cplx* __restrict__ spec, real* __restrict__ out, SpecGeom g,
                                 const cplx* __restrict__ tw2, Post post) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npts = g.nrows * g.n2;
    if (i >= npts) return;
    const int z = (int)(i % g.n2);
    const long long row = i / g.n2;
    const int x = (int)(row / g.n1), y = (int)(row % g.n1);
    real acc = spec[spec_index(g, x, y, 0)].x;
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
static __global__ void spec_to_internal_kernel(const cplx* __restrict__ stdl, cplx* __restrict__ intl, SpecGeom g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    intl[i] = stdl[((long long)x * g.n1 + y) * g.nzc + kz];
}
static __global__ void spec_to_standard_kernel(const cplx* __restrict__ intl, cplx* __restrict__ stdl, SpecGeom g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.total) return;
    int x, y, kz;
    spec_decode(g, i, x, y, kz);
    stdl[((long long)x * g.n1 + y) * g.nzc + kz] = intl[i];
}

}  // namespace ofdft

// ----------------------------------------------------------------------------------------------
// Fused x pass:  forward x-FFT of NIN spectra  ->  k-space mixing  ->  inverse x-FFT of NOUT spectra.
// One workgroup owns LPW memory-contiguous x-lines; G = max(NIN,NOUT) thread groups work array-parallel:
// group g transforms input g, the groups trade their spectra through LDS (real and imaginary parts in
// turn -- every mixing coefficient is either real or purely imaginary, so the parts never need each
// other), group g then builds output g and transforms it back.  HBM traffic: NIN reads + NOUT writes of a
// spectrum instead of 2*NIN + (mix) + 2*NOUT for separate forward / multiply / inverse passes.
namespace ofdft {

constexpr int kMaxXf = 4;
struct XfIo {
    const cplx* in[kMaxXf];
    cplx* out[kMaxXf];
};

// pick element g (wave-uniform) of a small kernel-argument pointer array without dynamic indexing
template <class T> __device__ __forceinline__ T* xf_pick(T* const (&a)[kMaxXf], int g) {
    T* p = a[0];
    if (g == 1) p = a[1];
    if (g == 2) p = a[2];
    if (g == 3) p = a[3];
    return p;
}

template <int LEN, int G, int NOUT> struct XfCfg {
    static constexpr int P = Plan<LEN>::P;
    static constexpr int E = Plan<LEN>::E;
    // lines per workgroup: >= 8 (128-B runs) and a whole number of waves per group (LPW*P % 64 == 0)
    // multi-group workgroups are kept small (4 lines = 64-B runs) so that several fit a CU and their
    // load / transform / store phases overlap; single-group ones take 8 lines (128-B runs)
    // measured at 256^3: 4-line tiles (more, smaller workgroups; pairs placed on one XCD) win when several
    // spectra are written (1->4: 0.20 -> 0.17 ms, 3->3: 0.38 -> 0.30 ms) and lose for 3->1 (0.20 -> 0.46 ms)
#ifndef OFDFT_XF_WANT4
#define OFDFT_XF_WANT4 1
#endif
    // (line counts for 16-byte elements; the fp32 build doubles them to keep the same 64-B / 128-B runs)
    static constexpr int WANT = ((OFDFT_XF_WANT4 && G > 1 && NOUT > 1) ? 4 : 8) * (16 / (int)sizeof(cplx));
    // ... halved until the workgroup fits 1024 threads (fp32 with the 32-lane plans of the 250..480-point lines: 3 groups x 16
    // lines x 32 lanes would be 1536)
    static constexpr int lpw_fit(int lpw) { return (G * lpw * P > 1024 && lpw > 1 && ((lpw / 2) * P) % 64 == 0) ? lpw_fit(lpw / 2) : lpw; }
    static constexpr int LPW = lpw_fit((P >= 64) ? 4 : ((WANT * P >= 64) ? WANT : 64 / P));
    static constexpr int TPB = G * LPW * P;
    static_assert(TPB <= 1024, "fused x pass: workgroup too large");
    // the mix trades the spectra through the line buffers in two halves of the register slots: real parts of a half at
    // positions j + P qq (qq < EH), imaginary parts RGN further on; 2 RGN = LEN for the power-of-two plans
    static constexpr int EH = (E + 1) / 2;
    static constexpr int RGN = P * EH;
    static constexpr int STRIDE = (2 * RGN + ((2 * RGN) >> 4) + 2 > LineBuf<LEN>::STRIDE) ? 2 * RGN + ((2 * RGN) >> 4) + 2
                                                                                         : LineBuf<LEN>::STRIDE;
    static constexpr size_t LDS = sizeof(real) * G * LPW * STRIDE;
};


// Mix functor contract (all indices compile-time, so every workgroup runs straight-line code):
//   static constexpr bool imag(int o)              coefficient of output o is i*c (else c) ...
//   template<int O,int I> static constexpr bool imag_oi()   ... per contribution (mixes with real AND imaginary terms in one output)
//   template<int O,int I> static constexpr bool present()   whether input I contributes to output O
//   template<int O,int I> real coef(x, y, kz, uoff, loff)  the real number c at that k-point; uoff + loff =
//        element offset of the k-point in a spectrum array (uoff wave-uniform) for buffer-load table lookups
template <int LEN, int LPW_> struct XfMixCtx {
    int j, l, y, kz;
    long long b0, tse;  // uniform offset of the tile and uniform element stride of the k-point tables along the line
    unsigned loff;      // per-lane element offset of the thread's first point (table lookups)
};

// (re, im) += coef<O,I>(k) * input_I, for I = I0..NIN-1 (compile-time recursion; absent terms vanish).  Both parts
// of the inputs are in LDS at once (re at pos, im at pos + RGN), so every coefficient is fetched ONCE.
template <int LEN, int LPW, int STRIDE, int RGN, int NIN, int O, int I, class Mix>
__device__ __forceinline__ void xf_mix_inputs(real& acr, real& aci, const real* lds, const XfMixCtx<LEN, LPW>& c,
                                              const Mix& mix, int q, int x, int pos) {
    if constexpr (I < NIN) {
        if constexpr (Mix::template present<O, I>()) {
            const real cf = mix.template coef<O, I>(x, c.y, c.kz, c.b0 + Plan<LEN>::cout(q) * c.tse, c.loff);
            const real* lb = lds + (I * LPW + c.l) * STRIDE;
            if constexpr (Mix::template imag_oi<O, I>()) {       // (i c)(re + i im) = -c im + i c re
                acr -= cf * lb[lpad(pos + RGN)];
                aci += cf * lb[lpad(pos)];
            } else {
                acr += cf * lb[lpad(pos)];
                aci += cf * lb[lpad(pos + RGN)];
            }
        }
        xf_mix_inputs<LEN, LPW, STRIDE, RGN, NIN, O, I + 1, Mix>(acr, aci, lds, c, mix, q, x, pos);
    }
}

// coefficient functors whose coef() is a table LOAD declare `static constexpr bool kTables = true` (MixWgc)
template <class Mix, class = void> struct mix_has_tables : std::false_type {};
template <class Mix> struct mix_has_tables<Mix, std::void_t<decltype(Mix::kTables)>> : std::bool_constant<Mix::kTables> {};

// table-driven mixing: all coefficients of the half are requested first (one batch of loads in flight), then used
template <int LEN, int LPW, int NIN, int O, int I, class Mix>
__device__ __forceinline__ void xf_fetch_coefs(real (&cf)[NIN], const XfMixCtx<LEN, LPW>& c, const Mix& mix, int q, int x) {
    if constexpr (I < NIN) {
        if constexpr (Mix::template present<O, I>()) cf[I] = mix.template coef<O, I>(x, c.y, c.kz, c.b0 + Plan<LEN>::cout(q) * c.tse, c.loff);
        else cf[I] = 0.0;
        xf_fetch_coefs<LEN, LPW, NIN, O, I + 1, Mix>(cf, c, mix, q, x);
    }
}
template <int LEN, int LPW, int STRIDE, int RGN, int NIN, int O, int I, class Mix>
__device__ __forceinline__ void xf_apply_coefs(real& acr, real& aci, const real (&cf)[NIN], const real* lds,
                                               const XfMixCtx<LEN, LPW>& c, int pos) {
    if constexpr (I < NIN) {
        if constexpr (Mix::template present<O, I>()) {
            const real* lb = lds + (I * LPW + c.l) * STRIDE;
            if constexpr (Mix::template imag_oi<O, I>()) {
                acr -= cf[I] * lb[lpad(pos + RGN)];
                aci += cf[I] * lb[lpad(pos)];
            } else {
                acr += cf[I] * lb[lpad(pos)];
                aci += cf[I] * lb[lpad(pos + RGN)];
            }
        }
        xf_apply_coefs<LEN, LPW, STRIDE, RGN, NIN, O, I + 1, Mix>(acr, aci, cf, lds, c, pos);
    }
}

// one output (compile-time O) from all inputs for the half of the thread's register slots with index HALF
template <int LEN, int LPW, int STRIDE, int NIN, int O, class Mix, int HALF>
__device__ __forceinline__ void xf_mix_part(cplx (&o)[Plan<LEN>::E], const real* lds, const XfMixCtx<LEN, LPW>& c,
                                            const Mix& mix) {
    using PL = Plan<LEN>;
    constexpr int P = PL::P, E = PL::E, EH = (E + 1) / 2, RGN = P * EH;
    if constexpr (mix_has_tables<Mix>::value && OFDFT_XF_BATCH_TABLES) {
        // Left to the compiler, the table loads came one at a time, each waited for before the next was issued: ~24
        // dependent HBM / L2 round trips per thread, i.e. the workgroup's whole lifetime.
        real cf[EH][NIN];
#pragma unroll
        for (int qq = 0; qq < EH; ++qq) {
            const int q = qq + HALF * EH;
            if (q < E && PL::slot_out(q)) xf_fetch_coefs<LEN, LPW, NIN, O, 0, Mix>(cf[qq], c, mix, q, c.j + PL::cout(q));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 0; qq < EH; ++qq) {
            const int q = qq + HALF * EH;
            if (q < E && PL::slot_out(q)) {
                real acr = 0.0, aci = 0.0;
                xf_apply_coefs<LEN, LPW, STRIDE, RGN, NIN, O, 0, Mix>(acr, aci, cf[qq], lds, c, c.j + P * qq);
                o[q] = mkc(acr, aci);
            }
        }
    } else {
#pragma unroll
        for (int qq = 0; qq < EH; ++qq) {
            const int q = qq + HALF * EH;
            if (q < E && PL::slot_out(q)) {
                real acr = 0.0, aci = 0.0;
                xf_mix_inputs<LEN, LPW, STRIDE, RGN, NIN, O, 0, Mix>(acr, aci, lds, c, mix, q, c.j + PL::cout(q), c.j + P * qq);
                o[q] = mkc(acr, aci);
            }
        }
    }
}

template <int LEN, int LPW, int STRIDE, int NIN, int NOUT, class Mix, int IMPART>
__device__ __forceinline__ void xf_mix_dispatch(int grp, cplx (&o)[Plan<LEN>::E], const real* lds,
                                                const XfMixCtx<LEN, LPW>& c, const Mix& mix) {
    // grp is wave-uniform: a scalar branch into straight-line code specialised per output
    if (grp == 0) xf_mix_part<LEN, LPW, STRIDE, NIN, 0, Mix, IMPART>(o, lds, c, mix);
    if constexpr (NOUT > 1) { if (grp == 1) xf_mix_part<LEN, LPW, STRIDE, NIN, 1, Mix, IMPART>(o, lds, c, mix); }
    if constexpr (NOUT > 2) { if (grp == 2) xf_mix_part<LEN, LPW, STRIDE, NIN, 2, Mix, IMPART>(o, lds, c, mix); }
    if constexpr (NOUT > 3) { if (grp == 3) xf_mix_part<LEN, LPW, STRIDE, NIN, 3, Mix, IMPART>(o, lds, c, mix); }
}

#ifndef OFDFT_XF_MINWAVES
#define OFDFT_XF_MINWAVES 1
#endif
// strides of the output arrays and of k-point tables along the line when they differ from the inputs' (0: the same).
// Used by the slab-decomposed path, where inputs and outputs live in exchange buffers with different record sizes.
struct XfStride { long long se_out, tse; };

template <int LEN, int NIN, int NOUT, class Mix>
__global__ __launch_bounds__((XfCfg<LEN, (NIN > NOUT ? NIN : NOUT), NOUT>::TPB), OFDFT_XF_MINWAVES) void xfused_kernel(
    XfIo io, LineMap m_main, LineMap m_rem, int main_blocks, SpecGeom g, const cplx* __restrict__ tw, Mix mix,
    XfStride xs) {
    constexpr int G = NIN > NOUT ? NIN : NOUT;
    using Cfg = XfCfg<LEN, G, NOUT>;
    using PL = Plan<LEN>;
    constexpr int P = Cfg::P, E = Cfg::E, LPW = Cfg::LPW, STRIDE = Cfg::STRIDE, EH = Cfg::EH, RGN = Cfg::RGN;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    static_assert((LPW * P) % 64 == 0, "a thread group must be whole waves");
    const int grp = __builtin_amdgcn_readfirstlane(tid / (LPW * P));   // wave-uniform by construction
    const int t = tid % (LPW * P);
    const int l = t % LPW;          // lines fastest over the lanes: memory-contiguous direction
    const int j = t / LPW;
    const bool is_rem = (int)blockIdx.x >= main_blocks;    // one grid: main part, then the remainder planes
    const LineMap m = is_rem ? m_rem : m_main;
    int bid = is_rem ? (int)blockIdx.x - main_blocks : (int)blockIdx.x;
    if (LPW * sizeof(cplx) < 128 && !is_rem && bid < (main_blocks & ~15)) {
        // two half-line tiles (4 lines of fp64, 8 of fp32 elements) share every 128-B line: put the pair on ONE XCD (blocks are dealt round-robin over
        // the 8 XCDs, so blocks b and b+8 share an L2) -- speed only, never correctness
        bid = (bid & ~15) + ((bid & 7) << 1) + ((bid >> 3) & 1);
    }
    if (!is_rem) bid += m.blk0;          // launch over a range of kz blocks (multiple of 16 workgroups)
    const long long L = (long long)bid * LPW + l;
    const bool valid = L < m.nlines;
    const long long base = valid ? (L / m.d) * m.sb + (L % m.d) * (long long)m.sl : 0;
    // k-point coordinates of this line
    int y, kz;
    if (is_rem) {
        y = (int)(L % g.n1);
        kz = g.nzm + (int)(L / g.n1);
    } else {
        const int c = (int)(L % m.d);
        y = c >> 3;
        kz = m.kz0 + (int)(L / m.d) * 8 + (c & 7);
    }
    const long long region = is_rem ? g.main_count : 0;   // offset of this launch's region in the full array

    const long long L0 = (long long)bid * LPW;
    const long long b0 = uniform64(region + line_base(m, L0));                 // wave-uniform
    const unsigned voff = valid ? (unsigned)((base - line_base(m, L0) + (long long)j * m.se) * kCB) : 0u;
    const long long se_u = uniform64(m.se);
    const long long se_o = uniform64(xs.se_out ? xs.se_out : m.se), se_t = uniform64(xs.tse ? xs.tse : m.se);
    const unsigned voff_o = valid ? (unsigned)((base - line_base(m, L0) + (long long)j * se_o) * kCB) : 0u;
    const unsigned tloff = valid ? (unsigned)(base - line_base(m, L0) + (long long)j * se_t) : 0u;
    cplx v[E];
    if (valid && grp < NIN) {
        // the group index is wave-uniform only when a group is a whole number of waves; select the pointer
        // with scalar-friendly code: each wave belongs to exactly one group when LPW*P % 64 == 0
        const cplx* ub = xf_pick(io.in, grp) + b0;
#pragma unroll
        for (int q = 0; q < E; ++q)
            v[q] = (PL::slot_in(q) && PL::lane_in(j, q)) ? buf_load_c_aux<OFDFT_XF_LD_AUX>(ub + PL::cin(q) * se_u, voff) : mkc(0.0, 0.0);
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mkc(0.0, 0.0);
    }
    real* mine = lds + (grp * LPW + l) * STRIDE;
    line_fft<LEN, false>(v, j, mine, tw);

    cplx o[E];
#pragma unroll
    for (int q = 0; q < E; ++q) o[q] = mkc(0.0, 0.0);
    XfMixCtx<LEN, LPW> mc{j, l, y, kz, b0, se_t, tloff};
    // ---- mix in two halves of the register slots (the k-points x = j + cout(q)): the line buffer holds the real parts of
    // a half at positions j + P qq and the imaginary parts RGN further on
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < EH; ++qq) {
        mine[lpad(j + P * qq)] = v[qq].x;
        mine[lpad(j + P * qq + RGN)] = v[qq].y;
    }
    __syncthreads();
    xf_mix_dispatch<LEN, LPW, STRIDE, NIN, NOUT, Mix, 0>(grp, o, lds, mc, mix);
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < EH; ++qq) {
        if (qq + EH < E) {
            mine[lpad(j + P * qq)] = v[qq + EH].x;
            mine[lpad(j + P * qq + RGN)] = v[qq + EH].y;
        }
    }
    __syncthreads();
    xf_mix_dispatch<LEN, LPW, STRIDE, NIN, NOUT, Mix, 1>(grp, o, lds, mc, mix);
    if constexpr (!PL::EXACT) {      // k-space results sit in the exit pattern of the forward transform
        __syncthreads();
        repattern_out_to_in<PL, false>(o, j, mine);
    }
    line_fft<LEN, true>(o, j, mine, tw);
    if (valid && grp < NOUT) {
        cplx* ub = xf_pick(io.out, grp) + b0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            if (!(PL::slot_out(q) && PL::lane_out(j, q))) continue;
            // nt stores pay in the block-8 layout (one GPU); in the exchange layout (x stride = a whole record) they cost 16 %
            if (xs.se_out) buf_store_c(ub + PL::cout(q) * se_o, voff_o, o[q]);
            else buf_store_c_aux<OFDFT_XF_ST_AUX>(ub + PL::cout(q) * se_o, voff_o, o[q]);
        }
    }
}

}  // namespace ofdft
