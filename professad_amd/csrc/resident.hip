// One persistent kernel for grids that fit on chip (cubic 16^3 / 32^3 / 64^3): the whole closure evaluation
//   chi -> n = c chi^2 -> (E terms, mu, dE/dchi)
// of the local + Hartree + von Weizsaecker + Wang-Teter + GGA (PBE x / c, LKT, mu-only Pauli-Gaussian) term sets in ONE
// launch.  gfx950 only.
//
// On these grids the staged pipeline is latency: 12-25 dependent launches of a few microseconds of work each (the graph
// replay removes the host from between them, not the launch gaps).  Here N workgroups stay resident and walk the
// evaluation in phases separated by grid barriers (agent-scope release / acquire on one counter):
//   A   workgroup = x plane:  f(chi) rows -> z-forward (registers) -> planes in LDS -> y-forward -> T[slot][ky][x][kz]
//       (f = chi^2, |chi|, chi^(2 beta), chi^(2 alpha): the closure scale c = N_e / (sum chi^2 dV) is not known yet, and
//        does not have to be -- the transforms are linear, c^p is applied in phase C)            + partial sum chi^2
//   B   workgroup = ky slab:  per (input, output) pair: x-forward -> multiply by 4 pi / k^2 | -k^2 | Lindhard kernel | i k_j
//       -> x-inverse -> T[output slot]
//   C   workgroup = x plane:  y-inverse (LDS) -> z-inverse (registers) -> R[slot][x][y][z]
//       GGA only:  grad n -> pbe_point -> df/dn, flux_j (R) -> their z- / y-forward -> T        (barrier)
//       B2  ky slab: x-forward of the three flux spectra, sum_j i k_j F_j, x-inverse            (barrier)
//       C2  x plane: y- / z-inverse of the divergence -> R
//       potential + energy integrands (combine_point, the very function of the staged pipeline) -> v(r), partial sums
//   D   workgroup = x plane:  mu from the ordered sum of the partials, chi.grad = 2 c dV chi (v - mu)
// Three barriers (five with a GGA term).  Every array is read and written in contiguous rows of kz (272 / 528 bytes); the
// spectra (<= 2.2 MB each) never leave the L2 / Infinity Cache.  The barrier spins are bounded (~2 s) and report through
// the sums.  Reference path: system.py:830-838 (closure), functionals.py:46,72,223,245,646-651,1597-1635 (terms).
#include "engine_ctx.h"

namespace ofdft {

constexpr int kResSlots = 16;        // doubles per workgroup in the partials: [0..9] combine sums, [10] sum chi^2, [11..13] GGA sums

constexpr int kResT = 15;            // spectrum slots: 0 chi^2 (Hartree in / out), 1 |chi| (vW), 2 chi^(2 beta), 3 chi^(2 alpha),
                                     // 4-6 grad n -> flux -> (4) divergence, 7 v_H when the chi^2 spectrum also feeds the gradient,
                                     // 9-11 WGC99 A, B, C -> u0, u1, u2, 12-14 P, Q, S -> gA, gB, gC (functionals.py:974-981)
constexpr int kResR = 15;            // real-space slots: the same, and 8 = df/dn of the GGA terms

struct ResOp { int in, out, ck; };   // phase B: T[out] = F_x^-1[coef_ck(k) F_x[T[in]]]; ck: 0 4 pi / k^2, 1 -k^2, 2 Lindhard, 3-5 i k_x / i k_y / i k_z

struct ResArgs {
    const real* chi;
    const real* vext;
    real* v;
    real* grad;
    cplx* T;                 // kResT spectra [slot][ky][x][kz]
    real* R;                 // kResR real-space arrays [slot][x][y][z] (unscaled transforms; flux and df/dn as computed)
    acc_t* part;             // [N][kResSlots]: [0..9] combine sums, [10] sum chi^2, [11..13] GGA energy sums
    acc_t* reduced;          // the context's pinned host mirror of the sums, written by the kernel: [0..12] sums, [13] barrier time-out flag
    unsigned* sync;          // [0] barrier counter, [1] count-out word
    unsigned epoch0;         // counter value before this launch
    unsigned* done;          // pinned host word: the last workgroup to finish stores done_target there (the host may spin on it
                             // instead of waiting for the stream: a few microseconds less per evaluation)
    unsigned done_target;    // value of the device-side count-out word sync[1] once every workgroup of this launch has left
    int act[4];              // which of the four scale-free inputs f(chi) the term set needs (their forward spectra land in slots 0..3)
    int kinds[10], in_slots[10], narr;      // phase A work list: input kind (0..3 as above; 4..9 the six WGC99 inputs, which need the
                             // closure scale: n^e theta^m / m!, e = beta | alpha, theta = n - n_ref) and the slot of its spectrum
    int need_c;              // WGC99 present: sum chi^2 is reduced before phase A (one more barrier)
    int nw;                  // WGC99 triples mixed in phase B (0 or 2: slots 9-11 and 12-14, in place)
    MixWgc wtab;             // its kernel tables ((w0, K1) and K2 per k-point, the staged pipeline's spectrum order)
    SpecGeom wg;             // ... and that order
    ResOp bop[8];            // phase B work list
    int nb;
    int outs[14], nout;      // slots transformed back to real space in phase C
    int vh_slot;             // where v_H comes back (0, or 7 with a GGA term)
    int gga;                 // gradient-dependent terms present (two more phases)
    int from_den;            // the input array is the density itself (ofdft_energy_potential): no closure scale, no mu / chi.grad
    int flux_slots[3];       // {4, 5, 6}: the flux components (the divergence comes back in the first of them)
    GgaSel sel;
    KGeom kg;
    CombineArgs ca;
    acc_t nel, vol_over_npts, dV;
    real inv_n, lind_p0, lind_p1;
};

// -DOFDFT_RES_CLOCK=1: workgroup 0 writes a phase clock into the host mirror (ofdft_query 16..27, tools/resident_probe.py);
// off by default -- the stores cross the bus and every barrier waits for them
#ifndef OFDFT_RES_CLOCK
#define OFDFT_RES_CLOCK 0
#endif
#ifndef OFDFT_RES_THREADS
#define OFDFT_RES_THREADS 512
#endif
constexpr int kResThreads = OFDFT_RES_THREADS;

template <int N> struct ResCfg {
    static constexpr int T = kResThreads, WAVES = T / 64;
    static constexpr int M = N / 2, NZH = M + 1, PS = NZH;       // odd row stride of the LDS planes (complex elements)
    static constexpr int AG = N >= 64 ? 2 : 4;                   // spectra whose planes sit in LDS together (a group)
    // z rows: M/4 lanes per row (ZPlan<M, 4>); x / y lines: N/8 lanes per line (ZPlan<N, 8>); the lanes of a row / line
    // sit in one wavefront, so the transforms synchronise at wave level only
    static constexpr int EZ = 4, PZ = M / EZ, RPWV = 64 / PZ;
    static constexpr int EL = 8, PL = N / EL, LPWV = 64 / PL;
    static constexpr int up(int n, int m) { return (n + m - 1) / m * m; }
    static constexpr int lo(int a, int b) { return a < b ? a : b; }
    static constexpr int ROWS = lo(T / PZ, up(AG * N, RPWV));    // row slots of a pass (whole waves)
    static constexpr int LINES = lo(T / PL, up(4 * NZH, LPWV));  // line slots of a pass
    static constexpr int RSZ = (z_cx<EZ>() ? 2 : 1) * line_stride<ZPlan<M, EZ>>();     // reals per z-row buffer (layout of that plan; the z helpers of the fp32 build exchange complex elements)
    static constexpr int RB_Z = ROWS * RSZ, RB_L = LINES * LineBuf<N>::STRIDE;
    static constexpr int RB = RB_Z > RB_L ? RB_Z : RB_L;         // reals of the row / line exchange buffers
    static constexpr size_t LDS_FFT = sizeof(real) * RB + sizeof(cplx) * (M + N) + sizeof(cplx) * AG * N * PS;
    static constexpr size_t LDS_RED = sizeof(double) * kCombineScalars * (T + T / 32);        // the ordered sums of phase C reuse the space
    static constexpr size_t LDS = LDS_FFT > LDS_RED ? LDS_FFT : LDS_RED;
};

// all workgroups of the evaluation have arrived `phase` times since epoch0.  One thread per workgroup talks to the other
// XCDs: its agent-scope release (after the workgroup's stores have reached the L2: workgroup-scope release + barrier)
// writes the L2's dirty lines back once, its acquire invalidates what the CU / the L2 hold from other XCDs once.
__device__ __forceinline__ void res_barrier(unsigned* ctr, unsigned target, int* timed_out) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {
                // shared word: ANY workgroup that gave up makes the whole evaluation fail (workgroup 0 may be the late one
                // and find every counter already past its targets)
                __hip_atomic_fetch_or(ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *timed_out = 1;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// sum over the workgroups, in workgroup order, of slots `slot` .. `slot + cnt - 1` of the partials -> tot[0..cnt-1] (LDS).
// The N * cnt loads (remote L2s / memory: a microsecond each) are issued by as many threads at once; the ordered sums then
// read LDS.
template <int N>
__device__ __forceinline__ void res_totals(const acc_t* part, int slot, int cnt, acc_t* tot, acc_t* stage) {
    __syncthreads();
    for (int i = threadIdx.x; i < N * cnt; i += kResThreads) {
        const int s = i / N, g = i - s * N;
        stage[i] = __builtin_nontemporal_load(part + g * kResSlots + slot + s);
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
        acc_t t = 0.0;
        for (int g = 0; g < N; ++g) t += stage[threadIdx.x * N + g];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// line transforms of the planes held in LDS: lines (array slot s < ns, kz) along y, in place (inverse) or out to T (forward)
template <int N, bool INV>
__device__ __forceinline__ void res_ylines(cplx* plane, int ns, real* rowbuf, const cplx* twN, cplx* Tout, const int* kind_of,
                                           int x) {
    using C = ResCfg<N>;
    constexpr int NZH = C::NZH, PS = C::PS, EL = C::EL, PL = C::PL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ll = lane % C::LPWV, lj = lane / C::LPWV;
    const int nlines = ns * NZH;
    for (int L0 = 0; L0 < nlines; L0 += C::LINES) {
        const int slot = wave * C::LPWV + ll;
        if (wave * C::LPWV >= C::LINES || L0 + wave * C::LPWV >= nlines) continue;      // wave-uniform: no line for this wave
        const int L = L0 + slot;
        const bool valid = L < nlines;
        const int s = valid ? L / NZH : 0, kz = valid ? L - s * NZH : 0;
        real* mine = rowbuf + slot * LineBuf<N>::STRIDE;
        cplx* pl = plane + s * (N * PS);
        cplx u[EL];
#pragma unroll
        for (int q = 0; q < EL; ++q) u[q] = valid ? pl[(lj + PL * q) * PS + kz] : mkc(0.0, 0.0);
        wave_line_fft<N, EL, INV>(u, lj, mine, twN);
        exchange_sync<true>();
        if (valid) {
            if constexpr (INV) {
#pragma unroll
                for (int q = 0; q < EL; ++q) pl[(lj + PL * q) * PS + kz] = u[q];
            } else {
                cplx* o = Tout + ((long long)kind_of[s] * N * N + x) * NZH + kz;
#pragma unroll
                for (int q = 0; q < EL; ++q) o[(long long)(lj + PL * q) * N * NZH] = u[q];
            }
        }
    }
}

// NS sums over the workgroup's threads, through LDS in a fixed order (dependent 64-bit wave shuffles took longer than the
// physics): [s][thread] -> 16 chunks per sum -> out[0..NS-1].  `scratch` = the dynamic LDS (dead buffers), `stage` >= 16 NS.
template <int NS>
__device__ __forceinline__ void res_block_sums(const acc_t (&acc)[NS], real* scratch, acc_t* stage, acc_t* out) {
    constexpr int T = kResThreads, SP = T + T / 32, CH = T / 16;
    const int tid = threadIdx.x;
    acc_t* st2 = reinterpret_cast<acc_t*>(scratch);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) st2[s * SP + tid + tid / 32] = acc[s];
    __syncthreads();
    if (tid < 16 * NS) {
        const int s = tid / 16, ch = tid - s * 16;
        acc_t t = 0.0;
        for (int i = 0; i < CH; ++i) {
            const int k = ch * CH + i;
            t += st2[s * SP + k + k / 32];
        }
        stage[tid] = t;
    }
    __syncthreads();
    if (tid < NS) {
        acc_t t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += stage[tid * 16 + i];
        out[tid] = t;
    }
    __syncthreads();
}

template <int N>
__global__ __launch_bounds__(kResThreads, 1) void resident_closure_kernel(ResArgs A, const cplx* __restrict__ twM_g,
                                                                           const cplx* __restrict__ twN_g) {
    using C = ResCfg<N>;
    constexpr int T = C::T, M = C::M, NZH = C::NZH, PS = C::PS, EZ = C::EZ, PZ = C::PZ, EL = C::EL, PL = C::PL, AG = C::AG;
    const int bid = (int)blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    __shared__ acc_t red[C::WAVES][kCombineScalars];
    __shared__ acc_t tot[16];
    __shared__ acc_t stage[16 * 64];
    real* rowbuf = lds;
    cplx* twM = reinterpret_cast<cplx*>(lds + C::RB);
    cplx* twN = twM + M;                                  // W_N^k, k < N: the r2c post-processing reads k < M, the x / y lines all of it
    cplx* plane = twN + N;                                // AG planes [y][kz]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zj = lane % PZ, zrw = lane / PZ;
    const int zslot = wave * C::RPWV + zrw;
    real* zmine = rowbuf + (zslot < C::ROWS ? zslot : 0) * C::RSZ;
    const int ll = lane % C::LPWV, lj = lane / C::LPWV;
    int timed_out = 0;
    // phase clock of workgroup 0 (100 MHz ticks since kernel entry) -> reduced[16..22]
    const unsigned long long clk0 = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&](int i) {
#if OFDFT_RES_CLOCK
        if (bid == 0 && tid == 0) A.reduced[16 + i] = (acc_t)(__builtin_amdgcn_s_memrealtime() - clk0);
#endif
    };
    (void)clk0;
    auto stage_tables = [&]() {                           // (published by the next __syncthreads)
        for (int i = tid; i < M; i += T) twM[i] = twM_g[i];
        for (int i = tid; i < N; i += T) twN[i] = twN_g[i];
    };
    stage_tables();
    const real be = A.ca.wt_beta, al = A.ca.wt_alpha;
    const int x = bid;                                    // phases A, C, D: this workgroup's x plane
    const cplx* chi_pl = reinterpret_cast<const cplx*>(A.chi + (long long)x * N * N);

    // ---- shared pieces ---------------------------------------------------------------------------------------------
    // rows of this plane: real pairs from `load(s, y, e)` (s = position in the slot list) -> z-forward -> planes in LDS
    auto rows_forward = [&](int ns, auto load) {
        const int nrows = ns * N;
        for (int R0 = 0; R0 < nrows; R0 += C::ROWS) {
            if (wave * C::RPWV >= C::ROWS || R0 + wave * C::RPWV >= nrows) continue;       // wave-uniform
            const int R = R0 + zslot;
            const bool valid = R < nrows;
            const int sI = valid ? R / N : 0, y = valid ? R - sI * N : 0;
            cplx v[EZ];
#pragma unroll
            for (int q = 0; q < EZ; ++q) v[q] = valid ? load(sI, y, zj + PZ * q) : mkc(0.0, 0.0);
            real nyq;
            const ZLane<M, EZ> z(zj, zrw, zmine, valid);
            z_forward_regs<M, EZ>(v, z, twM, twN, nyq);
            if (valid) {
                cplx* pr = plane + sI * (N * PS) + y * PS;
#pragma unroll
                for (int q = 0; q < EZ; ++q) pr[zj + PZ * q] = v[q];
                if (zj == 0) pr[M] = mkc(nyq, 0.0);
            }
        }
    };
    // planes of `ns` spectra (T slots slots[0..ns-1]) of this x plane -> y-inverse -> z-inverse -> R[slot]
    auto planes_inverse = [&](const int* slots, int ns, bool first_group) {
        constexpr int NLD = (AG * N * NZH + T - 1) / T;
        cplx ld[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * T;
            if (i < ns * N * NZH) {
                const int sI = i / (N * NZH), r = i - sI * (N * NZH);
                const int ky = r / NZH, kz = r - ky * NZH;
                ld[k] = A.T[(((long long)slots[sI] * N + ky) * N + x) * NZH + kz];
            }
        }
        if (first_group && tid < N) stage[tid] = __builtin_nontemporal_load(A.part + tid * kResSlots + 10);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + k * T;
            if (i < ns * N * NZH) {
                const int sI = i / (N * NZH), r = i - sI * (N * NZH);
                const int ky = r / NZH, kz = r - ky * NZH;
                plane[sI * (N * PS) + ky * PS + kz] = ld[k];
            }
        }
        __syncthreads();
        res_ylines<N, true>(plane, ns, rowbuf, twN, nullptr, nullptr, x);
        __syncthreads();
        const int nrows = ns * N;
        for (int R0 = 0; R0 < nrows; R0 += C::ROWS) {
            if (wave * C::RPWV >= C::ROWS || R0 + wave * C::RPWV >= nrows) continue;       // wave-uniform
            const int R = R0 + zslot;
            const bool valid = R < nrows;
            const int sI = valid ? R / N : 0, y = valid ? R - sI * N : 0;
            const cplx* pr = plane + sI * (N * PS) + y * PS;
            cplx v[EZ];
#pragma unroll
            for (int q = 0; q < EZ; ++q) v[q] = valid ? pr[zj + PZ * q] : mkc(0.0, 0.0);
            const real nyq = (valid && zj == 0) ? pr[M].x : (real)0.0;
            const ZLane<M, EZ> z(zj, zrw, zmine, valid);
            z_inverse_regs<M, EZ>(v, z, twM, twN, nyq);
            if (valid) {
                cplx* o = reinterpret_cast<cplx*>(A.R) + (((long long)slots[sI] * N + x) * N + y) * M;
#pragma unroll
                for (int q = 0; q < EZ; ++q) o[zj + PZ * q] = v[q];
            }
        }
        __syncthreads();
    };
    unsigned nbar = 0;                                    // barriers passed so far
    auto barrier = [&]() {
        ++nbar;
        res_barrier(A.sync, A.epoch0 + nbar * (unsigned)N, &timed_out);
    };

    acc_t cscale = 0.0;
    bool have_c = false;
    // ------------------------------------------------------------------ phase A
    {
        acc_t s2 = 0.0;
        for (int i = tid; i < N * M; i += T) {           // (density input: the sum is formed and ignored -- the scale is 1)
            const cplx c = chi_pl[i];
            s2 += (acc_t)c.x * c.x + (acc_t)c.y * c.y;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off, 64);
        if (lane == 0) red[wave][0] = s2;
        __syncthreads();                                  // (also publishes the twiddle tables)
        if (tid == 0) {
            acc_t t = 0.0;
            for (int w = 0; w < C::WAVES; ++w) t += red[w][0];
            A.part[bid * kResSlots + 10] = t;
        }
        if (A.need_c) {       // the WGC99 inputs are not homogeneous in the closure scale (theta = n - n_ref): reduce sum chi^2 first
            barrier();
            res_totals<N>(A.part, 10, 1, tot + kCombineScalars, stage);
            cscale = A.from_den ? (acc_t)1.0 : A.nel / (tot[kCombineScalars] * A.vol_over_npts);       // system.py:833-834
            have_c = true;
        }
        const real cs0 = (real)cscale, wal = A.ca.wgc_alpha, wbe = A.ca.wgc_beta, nref = A.ca.nref;
        for (int g0 = 0; g0 < A.narr; g0 += AG) {
            const int ns = (A.narr - g0) < AG ? (A.narr - g0) : AG;
            rows_forward(ns, [&](int sI, int y, int e) {
                const int kind = A.kinds[g0 + sI];
                const cplx c = chi_pl[y * M + e];
                const real x2 = A.from_den ? c.x : c.x * c.x, y2 = A.from_den ? c.y : c.y * c.y;       // n / c
                if (kind == 0) return mkc(x2, y2);
                if (kind == 1)                                                                          // sqrt(n / c)
                    return A.from_den ? mkc(x2 > 0.0 ? sqrt(x2) : (real)0.0, y2 > 0.0 ? sqrt(y2) : (real)0.0) : mkc(fabs(c.x), fabs(c.y));
                if (kind < 4) {
                    const real ex = kind == 2 ? be : al;
                    return mkc(x2 > 0.0 ? fm::pow_pos(x2, ex) : (real)0.0, y2 > 0.0 ? fm::pow_pos(y2, ex) : (real)0.0);
                }
                // WGC99: n^e theta^m / m!  (functionals.py:974-981), on the scaled density
                const real n0 = cs0 * x2, n1 = cs0 * y2, ex = kind < 7 ? wbe : wal;
                const int m = (kind - 4) % 3;
                real a0 = n0 > 0.0 ? fm::pow_pos(n0, ex) : (real)0.0, a1 = n1 > 0.0 ? fm::pow_pos(n1, ex) : (real)0.0;
                if (m >= 1) { a0 *= n0 - nref; a1 *= n1 - nref; }
                if (m == 2) { a0 *= 0.5 * (n0 - nref); a1 *= 0.5 * (n1 - nref); }
                return mkc(a0, a1);
            });
            __syncthreads();
            res_ylines<N, false>(plane, ns, rowbuf, twN, A.T, A.in_slots + g0, x);
            __syncthreads();
        }
    }
    stamp(0);
    barrier();
    stamp(1);

    // ------------------------------------------------------------------ phase B: slab ky = bid, lines along x
    {
        const int ky = bid;
        const int nlines = A.nb * NZH;
        for (int L0 = 0; L0 < nlines; L0 += C::LINES) {
            const int slot = wave * C::LPWV + ll;
            if (wave * C::LPWV >= C::LINES || L0 + wave * C::LPWV >= nlines) continue;          // wave-uniform
            const int L = L0 + slot;
            const bool valid = L < nlines;
            const int oi = valid ? L / NZH : 0, kz = valid ? L - oi * NZH : 0;
            const ResOp op = A.bop[oi];
            real* mine = rowbuf + slot * LineBuf<N>::STRIDE;
            const cplx* src = A.T + ((long long)op.in * N + ky) * N * NZH + kz;
            cplx* dst = A.T + ((long long)op.out * N + ky) * N * NZH + kz;
            cplx u[EL];
#pragma unroll
            for (int q = 0; q < EL; ++q) u[q] = valid ? src[(lj + PL * q) * NZH] : mkc(0.0, 0.0);
            wave_line_fft<N, EL, false>(u, lj, mine, twN);
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EL; ++q) {
                real kx_, ky_, kz_, k2;
                kvec_xyz(A.kg, lj + PL * q, ky, kz, kx_, ky_, kz_, k2);
                if (op.ck >= 3) {                                                              // i k_j: functional_tools.py:166-183
                    const real kj = op.ck == 3 ? kx_ : (op.ck == 4 ? ky_ : kz_);
                    u[q] = mkc(-kj * u[q].y, kj * u[q].x);
                } else {
                    real cf;
                    if (op.ck == 0) cf = (k2 != 0.0) ? 4.0 * kPiR / k2 : 0.0;                  // functionals.py:72
                    else if (op.ck == 1) cf = -k2;                                             // functionals.py:245
                    else cf = A.lind_p0 * (real)lindhard_shape((k2 != 0.0) ? sqrt(k2) * A.lind_p1 : 0.0);     // functionals.py:646-651
                    u[q] = mkc(cf * u[q].x, cf * u[q].y);
                }
            }
            wave_line_fft<N, EL, true>(u, lj, mine, twN);
            exchange_sync<true>();
            if (valid) {
#pragma unroll
                for (int q = 0; q < EL; ++q) dst[(lj + PL * q) * NZH] = u[q];
            }
        }
    }
    // WGC99: (A, B, C) -> (w0 A + K1 B + K2 C, K1 A + K3 B, K2 A) and the same for (P, Q, S), in place (the tables
    // carry every prefactor, as in the staged pipeline's MixWgc)
    if (A.nw) {
        const int ky = bid;
        const int nlines = A.nw * NZH;
        for (int L0 = 0; L0 < nlines; L0 += C::LINES) {
            const int slot = wave * C::LPWV + ll;
            if (wave * C::LPWV >= C::LINES || L0 + wave * C::LPWV >= nlines) continue;          // wave-uniform
            const int L = L0 + slot;
            const bool valid = L < nlines;
            const int tr = valid ? L / NZH : 0, kz = valid ? L - tr * NZH : 0;
            real* mine = rowbuf + slot * LineBuf<N>::STRIDE;
            cplx* base = A.T + ((long long)(9 + 3 * tr) * N + ky) * N * NZH + kz;
            constexpr long long SS = (long long)N * N * NZH;          // one slot
            cplx u[3][EL];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int q = 0; q < EL; ++q) u[a][q] = valid ? base[a * SS + (lj + PL * q) * NZH] : mkc(0.0, 0.0);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                wave_line_fft<N, EL, false>(u[a], lj, mine, twN);
                exchange_sync<true>();
            }
#pragma unroll
            for (int q = 0; q < EL; ++q) {
                const long long ti = spec_index(A.wg, lj + PL * q, ky, kz);
                const cplx t01 = A.wtab.t01[ti];
                const real w0 = t01.x, K1 = t01.y, K2 = A.wtab.t2[ti], K3 = K2 + A.wtab.ck * K1;
                const cplx a0 = u[0][q], a1 = u[1][q], a2 = u[2][q];
                u[0][q] = mkc(w0 * a0.x + K1 * a1.x + K2 * a2.x, w0 * a0.y + K1 * a1.y + K2 * a2.y);
                u[1][q] = mkc(K1 * a0.x + K3 * a1.x, K1 * a0.y + K3 * a1.y);
                u[2][q] = mkc(K2 * a0.x, K2 * a0.y);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                wave_line_fft<N, EL, true>(u[a], lj, mine, twN);
                exchange_sync<true>();
            }
            if (valid) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int q = 0; q < EL; ++q) base[a * SS + (lj + PL * q) * NZH] = u[a][q];
                }
            }
        }
    }
    stamp(2);
    barrier();
    stamp(3);

    // ------------------------------------------------------------------ phase C
    // the loads that cross XCDs -- this plane of the first group's spectra and the partial sums of chi^2 -- go out together
    if (A.nout == 0 && !have_c) {                         // purely local term set: nothing to transform back
        res_totals<N>(A.part, 10, 1, tot + kCombineScalars, stage);
        cscale = A.nel / (tot[kCombineScalars] * A.vol_over_npts);
    }
    for (int g0 = 0; g0 < A.nout; g0 += AG) {
        const int ns = (A.nout - g0) < AG ? (A.nout - g0) : AG;
        planes_inverse(A.outs + g0, ns, g0 == 0);
        if (g0 == 0) {
            if (!have_c) {
                acc_t t = 0.0;
                for (int g = 0; g < N; ++g) t += stage[g];             // same order in every thread and workgroup
                cscale = A.nel / (t * A.vol_over_npts);                // system.py:833-834
            }
            __syncthreads();                                           // (stage is reused below)
        }
    }
    stamp(7);
    if (A.from_den) cscale = 1.0;
    const real cs = (real)cscale;
    const long long po = (long long)x * N * M;
    constexpr long long AS = (long long)N * N * M;            // pairs per real-space array
    cplx* rp = reinterpret_cast<cplx*>(A.R) + po;
    if (A.gga) {
        // ---- GGA mid stage on this plane: grad n -> energy densities, df/dn, flux_j = df/d|grad n|^2 d_j n
        //      (tools_for_tests.py:155-207; the staged pipeline's pbe_point)
        const real fg = A.inv_n * cs;
        acc_t pacc[kPbeScalars] = {0.0, 0.0, 0.0};
        for (int i = tid; i < N * M; i += T) {
            const cplx c = chi_pl[i];
            const cplx gx = rp[4 * AS + i], gy = rp[5 * AS + i], gz = rp[6 * AS + i];
            const real a0 = fg * gx.x, b0 = fg * gy.x, c0 = fg * gz.x, a1 = fg * gx.y, b1 = fg * gy.y, c1 = fg * gz.y;
            const PbePoint p0 = pbe_point(A.from_den ? c.x : cs * c.x * c.x, a0 * a0 + b0 * b0 + c0 * c0, A.sel);
            const PbePoint p1 = pbe_point(A.from_den ? c.y : cs * c.y * c.y, a1 * a1 + b1 * b1 + c1 * c1, A.sel);
            pacc[0] += p0.fx + p1.fx;
            pacc[1] += p0.fc + p1.fc;
            pacc[2] += p0.fk + p1.fk;
            rp[8 * AS + i] = mkc(p0.dfdn, p1.dfdn);
            rp[4 * AS + i] = mkc(p0.dfdg * a0, p1.dfdg * a1);
            rp[5 * AS + i] = mkc(p0.dfdg * b0, p1.dfdg * b1);
            rp[6 * AS + i] = mkc(p0.dfdg * c0, p1.dfdg * c1);
        }
        res_block_sums<kPbeScalars>(pacc, lds, stage, A.part + bid * kResSlots + 11);
        stage_tables();                                   // the ordered sums borrowed the dynamic LDS, twiddle tables included
        __syncthreads();
        // flux components of this plane -> z- and y-forward -> T[4..6]
        for (int g0 = 0; g0 < 3; g0 += AG) {
            const int ns = (3 - g0) < AG ? (3 - g0) : AG;
            rows_forward(ns, [&](int sI, int y, int e) { return rp[(4 + g0 + sI) * AS + y * M + e]; });
            __syncthreads();
            res_ylines<N, false>(plane, ns, rowbuf, twN, A.T, A.flux_slots + g0, x);
            __syncthreads();
        }
        barrier();
        // ---- phase B2: slab ky = bid: divergence spectrum sum_j i k_j F_j  ->  T[4]
        {
            const int ky = bid;
            for (int L0 = 0; L0 < NZH; L0 += C::LINES) {
                const int slot = wave * C::LPWV + ll;
                if (wave * C::LPWV >= C::LINES || L0 + wave * C::LPWV >= NZH) continue;             // wave-uniform
                const int kz = L0 + slot;
                const bool valid = kz < NZH;
                real* mine = rowbuf + slot * LineBuf<N>::STRIDE;
                cplx acc[EL];
#pragma unroll
                for (int q = 0; q < EL; ++q) acc[q] = mkc(0.0, 0.0);
#pragma unroll 1
                for (int jc = 0; jc < 3; ++jc) {
                    const cplx* src = A.T + ((long long)(4 + jc) * N + ky) * N * NZH + (valid ? kz : 0);
                    cplx u[EL];
#pragma unroll
                    for (int q = 0; q < EL; ++q) u[q] = valid ? src[(lj + PL * q) * NZH] : mkc(0.0, 0.0);
                    wave_line_fft<N, EL, false>(u, lj, mine, twN);
                    exchange_sync<true>();
#pragma unroll
                    for (int q = 0; q < EL; ++q) {
                        real kx_, ky_, kz_, k2;
                        kvec_xyz(A.kg, lj + PL * q, ky, valid ? kz : 0, kx_, ky_, kz_, k2);
                        const real kj = jc == 0 ? kx_ : (jc == 1 ? ky_ : kz_);
                        acc[q] = mkc(acc[q].x - kj * u[q].y, acc[q].y + kj * u[q].x);
                    }
                }
                wave_line_fft<N, EL, true>(acc, lj, mine, twN);
                exchange_sync<true>();
                if (valid) {
                    cplx* dst = A.T + ((long long)4 * N + ky) * N * NZH + kz;
#pragma unroll
                    for (int q = 0; q < EL; ++q) dst[(lj + PL * q) * NZH] = acc[q];
                }
            }
        }
        barrier();
        // ---- phase C2: the divergence back on this plane -> R[4]
        planes_inverse(A.flux_slots, 1, false);
    }
    {
        // potential and energy integrands: the staged pipeline's combine_point on n = c chi^2 and the convolutions
        const real f0 = A.inv_n * cs, f1 = A.inv_n * sqrt(cs);
        const real f2 = A.act[2] ? A.inv_n * (real)::pow((double)cs, (double)be) : (real)0.0;
        const real f3 = A.act[3] ? A.inv_n * (real)::pow((double)cs, (double)al) : (real)0.0;
        const bool has_h = A.ca.mask & 2u;
        acc_t acc[kCombineScalars];
#pragma unroll
        for (int s = 0; s < kCombineScalars; ++s) acc[s] = 0.0;
        const cplx* ep = reinterpret_cast<const cplx*>(A.vext ? A.vext : A.chi) + po;
        cplx* vp = reinterpret_cast<cplx*>(A.v) + po;
        const long long hs = (long long)A.vh_slot * AS;
        stamp(10);
        for (int i = tid; i < N * M; i += T) {
            const cplx c = chi_pl[i], ve = ep[i];
            const cplx r0 = has_h ? rp[hs + i] : mkc(0.0, 0.0), r1 = A.act[1] ? rp[AS + i] : mkc(0.0, 0.0);
            const cplx r2 = A.act[2] ? rp[2 * AS + i] : mkc(0.0, 0.0), r3 = A.act[3] ? rp[3 * AS + i] : mkc(0.0, 0.0);
            const cplx dn = A.gga ? rp[8 * AS + i] : mkc(0.0, 0.0), dv = A.gga ? rp[4 * AS + i] : mkc(0.0, 0.0);
            cplx wv[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) wv[a] = A.nw ? rp[(9 + a) * AS + i] : mkc(0.0, 0.0);
            real vv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const real ch = h ? c.y : c.x;
                CombinePoint p{};
                p.n = A.from_den ? ch : cs * ch * ch;
                p.vext = h ? ve.y : ve.x;
                p.vh = f0 * (h ? r0.y : r0.x);
                p.lap = f1 * (h ? r1.y : r1.x);
                p.cb = f2 * (h ? r2.y : r2.x);
                p.cva = f3 * (h ? r3.y : r3.x);
                p.dfdn = h ? dn.y : dn.x;
                p.div = A.inv_n * (h ? dv.y : dv.x);
                p.u0 = A.inv_n * (h ? wv[0].y : wv[0].x);
                p.u1 = A.inv_n * (h ? wv[1].y : wv[1].x);
                p.u2 = A.inv_n * (h ? wv[2].y : wv[2].x);
                p.gA = A.inv_n * (h ? wv[3].y : wv[3].x);
                p.gB = A.inv_n * (h ? wv[4].y : wv[4].x);
                p.gC = A.inv_n * (h ? wv[5].y : wv[5].x);
                vv[h] = combine_point(A.ca, p, kCtf, acc);
            }
            if (A.v) vp[i] = mkc(vv[0], vv[1]);
        }
        stamp(11);
        res_block_sums<kCombineScalars>(acc, lds, stage, A.part + bid * kResSlots);
    }
    stamp(4);
    barrier();
    stamp(5);

    // ------------------------------------------------------------------ phase D: mu and chi.grad
    res_totals<N>(A.part, 0, 14, tot, stage);
    if (bid == 0) {          // (a workgroup that never arrives stalls every barrier, this one's included: one flag suffices)
        if (tid < kCombineScalars) A.reduced[tid] = tot[tid];
        else if (tid < 13) A.reduced[tid] = A.gga ? tot[tid + 1] : 0.0;        // GGA energy sums (partials [11..13])
        if (tid == 0) {       // (sync[2]: set by any workgroup whose barrier ran out of patience; read after the last barrier)
            const unsigned any = __hip_atomic_load(A.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.reduced[13] = (timed_out || any) ? 1.0 : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // system scope: the sums are in host memory before this workgroup counts out
    }
    if (A.grad) {
        const real mu = (real)((tot[8] * A.dV) / A.nel);                                        // system.py:851
        const real c2dV = (real)(cscale * (2.0 * A.dV));                                        // system.py:836-837,853
        const cplx* vp = reinterpret_cast<const cplx*>(A.v) + po;
        cplx* gp = reinterpret_cast<cplx*>(A.grad) + po;
        for (int i = tid; i < N * M; i += T) {       // the thread that wrote v[i] reads it
            const cplx c = chi_pl[i], w = vp[i];
            gp[i] = mkc(c2dV * c.x * (w.x - mu), c2dV * c.y * (w.y - mu));
        }
    }
    stamp(6);
    __syncthreads();
    if (tid == 0) {
        // count out on the device; the last workgroup tells the host (one write over the bus, not N atomics)
        const unsigned old = __hip_atomic_fetch_add(A.sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == A.done_target) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope
            __hip_atomic_store(A.done, A.done_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace ofdft

namespace eng {

// term sets the resident kernel serves: everything local, Hartree, von Weizsaecker, Wang-Teter, the gradient-dependent
// terms without a Laplacian
bool resident_serves(const ofdft_ctx* c) {
    if (!c->resident || c->nranks != 1 || !c->fast) return false;
    if (!(c->n0 == c->n1 && c->n1 == c->n2 && (c->n0 == 16 || c->n0 == 32 || c->n0 == 64))) return false;
    if ((c->mask & OFDFT_WGC99_NL) && c->n0 > 32) return false;      // (six more spectra: beyond 32^3 the graph replay is faster)
    if ((c->mask & kGgaAny) && (gga_needs_laplacian(c) || (c->n0 > 32 && sizeof(real) == 8))) return false;      // (64^3 fp64 with a GGA term: five phases of
                                                                                          //  4096-point planes per workgroup measured slower than the graph replay, 0.160 vs 0.153 ms; fp32: 0.118 vs 0.150)
    if (wts_active(c)) return false;
    return c->mask != 0;
}

// The grid barrier needs all N workgroups resident at once (512 threads and up to ~135 KB of LDS each: one per CU).  The
// kernel is an ordinary launch, so co-residency is checked before the first one: occupancy x CU count >= N, else the
// context stops using the kernel (the callers fall through to the graph replay / the staged pipeline).  What the check
// cannot see (CU masks of the process, other streams holding CUs) is caught by the barrier's time limit, after which
// the callers re-run the evaluation on the staged path and switch the kernel off as well.
template <int N>
static bool res_fits(ofdft_ctx* c) {
    int per_cu = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_closure_kernel<N>, kResThreads, ResCfg<N>::LDS) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return (long long)per_cu * ncu >= N;
}

template <int N>
static int launch_res(ofdft_ctx* c, const ResArgs& a, hipStream_t st) {
    cplx *twM, *twN;
    if (int rc = get_twiddle(c, N / 2, &twM)) return rc;
    if (int rc = get_twiddle(c, N, &twN)) return rc;
    int nwg = N;
    if (c->test_fault == 1) {         // test hook: a launch that cannot pass its grid barriers
        nwg = N - 1;
        c->test_fault = 0;
    }
    OFDFT_LAUNCH(c, st, "resident", (resident_closure_kernel<N>), dim3(nwg), dim3(kResThreads), ResCfg<N>::LDS, a, (const cplx*)twM,
                 (const cplx*)twN);
    return 0;
}

int resident_closure(ofdft_ctx* c, const real* chi, const real* vext, double nel, real* v, real* grad, hipStream_t st, bool from_den) {
    const int N = c->n0;
    const unsigned mask = c->mask;
    if ((mask & OFDFT_ION_ELECTRON) && !vext) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");
    if (c->res_fits < 0) c->res_fits = (N == 16 ? res_fits<16>(c) : N == 32 ? res_fits<32>(c) : res_fits<64>(c)) ? 1 : 0;
    if (!c->res_fits) {           // the N workgroups cannot be co-resident on this device / partition: never launch
        c->resident = 0;
        return kResidentDeclined;
    }
    ResArgs a{};
    a.chi = chi;
    a.vext = vext;
    a.v = v;
    a.grad = grad;
    void* p;
    if (int rc = get_ws(c, "res:T", sizeof(cplx) * kResT * (size_t)N * N * (N / 2 + 1), &p)) return rc;
    a.T = (cplx*)p;
    if (int rc = get_ws(c, "res:R", sizeof(real) * kResR * (size_t)N * N * N, &p)) return rc;
    a.R = (real*)p;
    if (int rc = get_ws(c, "res:part", sizeof(double) * 64 * kResSlots, &p)) return rc;
    a.part = (acc_t*)p;
    if (!c->res_sync) {
        HIP_TRY(c, hipMalloc((void**)&c->res_sync, 64));
        HIP_TRY(c, hipMemset(c->res_sync, 0, 64));
        HIP_TRY(c, hipHostMalloc((void**)&c->res_done, 64));
        *c->res_done = 0;
        c->res_done_target = 0;
        c->res_epoch = 0;
    }
    const bool gga = mask & kGgaAny, har = mask & OFDFT_HARTREE, vw = mask & OFDFT_VW, wt = mask & OFDFT_WT_NL;
    const double al = c->params[OFDFT_P_WT_ALPHA], be = c->params[OFDFT_P_WT_BETA];
    a.done = c->res_done;
    a.reduced = c->h_partial;          // pinned, device-visible: no copy command behind the kernel
    a.sync = c->res_sync;
    a.epoch0 = c->res_epoch;
    c->res_epoch += ((gga ? 5u : 3u) + ((mask & OFDFT_WGC99_NL) ? 1u : 0u)) * (unsigned)N;
    c->res_done_target += (unsigned)N;
    a.done_target = c->res_done_target;
    // inputs f(chi) -> forward spectra in slots 0..3
    a.act[0] = (har || gga) ? 1 : 0;
    a.act[1] = vw ? 1 : 0;
    a.act[2] = wt ? 1 : 0;
    a.act[3] = (wt && al != be) ? 1 : 0;
    const bool wgc = mask & OFDFT_WGC99_NL;
    for (int k = 0; k < 4; ++k)
        if (a.act[k]) {
            a.in_slots[a.narr] = k;
            a.kinds[a.narr++] = k;
        }
    if (wgc) {
        for (int k = 4; k < 10; ++k) {
            a.in_slots[a.narr] = 9 + (k - 4);
            a.kinds[a.narr++] = k;
        }
        a.need_c = 1;
        a.nw = 2;
    }
    // phase B work list and the slots that come back to real space
    a.gga = gga ? 1 : 0;
    a.from_den = from_den ? 1 : 0;
    a.vh_slot = gga ? 7 : 0;          // (with a gradient the chi^2 spectrum has several readers: v_H may not overwrite it)
    a.flux_slots[0] = 4; a.flux_slots[1] = 5; a.flux_slots[2] = 6;
    auto op = [&](int in, int out, int ck) { a.bop[a.nb++] = ResOp{in, out, ck}; a.outs[a.nout++] = out; };
    if (har) op(0, a.vh_slot, 0);
    if (vw) op(1, 1, 1);
    if (wt) op(2, 2, 2);
    if (a.act[3]) op(3, 3, 2);
    if (gga)
        for (int j = 0; j < 3; ++j) op(0, 4 + j, 3 + j);
    if (wgc) {
        for (int k = 0; k < 6; ++k) a.outs[a.nout++] = 9 + k;
        const double wal = c->params[OFDFT_P_WGC_ALPHA], wbe = c->params[OFDFT_P_WGC_BETA];
        double nref;
        if (int rc = ensure_wgc_tables(c, std::llround(nel), st, &nref)) return rc;     // functionals.py:952 (rounded N_e)
        a.wtab = wgc_tab(c);
        a.wg = c->g;
        a.ca.wgc_alpha = wal;
        a.ca.wgc_beta = wbe;
        a.ca.nref = nref;
        a.ca.wgc_sum_53 = (std::fabs(wal + wbe - kFiveThirds) < 4e-16) ? 1 : 0;
    }
    a.sel = gga_sel(c);
    a.kg = c->kg;
    CombineArgs& ca = a.ca;
    ca.mask = mask;
    ca.npts = c->npts;
    ca.gtf_kind = (int)c->params[OFDFT_P_VWGTF_KIND];
    ca.gtf_inv_n0 = (mask & OFDFT_VWGTF) ? c->vol / (double)std::llround(nel) : 0.0;           // functionals.py:268-270
    ca.conv_a = nullptr;
    if (wt) {
        const double nbar = nel / c->vol;                                                      // functionals.py:646-647
        const double kf = std::cbrt(3.0 * kPi * kPi * nbar);
        a.lind_p0 = 5.0 / (9.0 * al * be * std::pow(nbar, al + be - kFiveThirds));
        a.lind_p1 = 1.0 / (2.0 * kf);
        ca.wt_alpha = al;
        ca.wt_beta = be;
        ca.wt_nbar_pa = std::pow(nbar, al);
        ca.wt_is_56 = (al == kFiveSixths && be == kFiveSixths) ? 1 : 0;
        if (al != be) ca.conv_a = chi;      // any non-null pointer: combine_point only asks whether the second convolution exists
    }
    a.nel = nel;
    a.vol_over_npts = c->vol / (double)c->npts;
    a.dV = c->dV;
    a.inv_n = 1.0 / (double)c->npts;
    c->fft_count += a.narr + a.nout + (gga ? 4 : 0);
    switch (N) {
        case 16: return launch_res<16>(c, a, st);
        case 32: return launch_res<32>(c, a, st);
        case 64: return launch_res<64>(c, a, st);
    }
    return fail(c, OFDFT_EINVAL, "resident kernel: unsupported grid");
}

}  // namespace eng
