// Wave-local z passes with fused real-space math (fp64, gfx950).
//
// A real row of N2 = 2M points is transformed by the M/8 lanes that own it (ZPlan<M>, 8 complex = 16 real
// points per lane); the lanes of a row sit in ONE wavefront, so every LDS exchange is ordered by the wave's
// own program order and the kernels contain no s_barrier.  A 256-thread workgroup is four independent waves.
// Because a lane keeps only 16 points, the kernels have registers left to do the pointwise physics while a
// row is on chip:
//   zf_density_kernel   chi|n row -> spectra of n and sqrt(n)                 (1 read, 2 spectrum writes)
//   zf_powers_kernel    chi|n row -> spectra of up to six n^e theta^m/m! terms (WT / WGC99 inputs, ONE pow)
//   zpbe_kernel         (d_x n, d_y n, d_z n)^ rows -> real space -> PBE -> flux rows -> spectra, in place
//   zi_combine_kernel   all convolution spectra of a row -> real space -> potential + energy integrands
// so none of the real-space intermediates (sqrt n, n^beta theta..., grad n, flux, v_H, u_i, g_i, div) ever
// exists in HBM.
#pragma once
#include "pointwise_kernels.h"

namespace ofdft {

#ifndef OFDFT_ZPBE_WAVES
#define OFDFT_ZPBE_WAVES 2
#endif
#ifndef OFDFT_ZIWGC_WAVES
#define OFDFT_ZIWGC_WAVES 2
#endif
#ifndef OFDFT_Z_PREFETCH
#define OFDFT_Z_PREFETCH 1     // depth-one software pipeline over the spectra a fused z kernel consumes (z_issue_row)
#endif
#ifndef OFDFT_Z_PIPE_BIG_F32
#define OFDFT_Z_PIPE_BIG_F32 0  // fp32 build: the depth-one pipeline of zi_combine also for rows of 1024 points (see zi_combine_waves)
#endif
#ifndef OFDFT_ZI_ROOTS_ONCE
#define OFDFT_ZI_ROOTS_ONCE 0   // zi_combine: n^(-1/6) of every point kept in registers for all sections (1) or formed per section (0)
#endif
#ifndef OFDFT_Z_LDS_TWIDDLES
#define OFDFT_Z_LDS_TWIDDLES 1
#endif

// (round 4, measured negative: whole-complex LDS exchange in the fp32 z kernels, as in the y / x passes -- zpbe2 0.154 -> 0.168 ms,
// zi_combine 0.142 -> 0.149, zf_powers 0.099 -> 0.106, only the 8-point zf_density gains, 0.068 -> 0.062: the radix-4 rows' XOR
// layout is tuned to the b32 bank rules.  Knob kept, default off.)
#ifndef OFDFT_Z_CX
#define OFDFT_Z_CX 0
#endif

// OFDFT_Z_CX=2: only the rows with 8 points per lane (radix-8 plans in the padded layout: 31-45 % of their LDS cycles are bank
// conflicts with 4-byte accesses, profiles/r05_sq_counters_1024_f32_cfg2_before.md) exchange whole complex numbers
template <int E> constexpr bool z_cx() { return kCX && (OFDFT_Z_CX == 1 || (OFDFT_Z_CX == 2 && E == 8)); }
template <int M, int E_> struct ZW {
    using PL = ZPlan<M, E_>;
    static constexpr int E = E_;
    static constexpr int P = PL::P;                 // lanes per row
    static constexpr int RPWV = 64 / P;             // rows per wave
    static constexpr int TPB = 256;
    static constexpr int RPB = RPWV * (TPB / 64);   // rows per block
    // LDS reals per row (fp32 build, OFDFT_Z_CX: the row transforms exchange whole complex numbers -- twice the reals per row)
    static constexpr int RS = (z_cx<E_>() ? kCXMul : 1) * line_stride<PL>();
    static constexpr int ROWS = RPB * RS;           // reals of the row buffers; the staged twiddle tables follow them
#if OFDFT_Z_LDS_TWIDDLES
    static constexpr size_t LDS = sizeof(real) * ROWS + sizeof(cplx) * 2 * M;
#else
    static constexpr size_t LDS = sizeof(real) * ROWS;
#endif
    static constexpr int N2 = 2 * M;
};

// waves per SIMD the fused z kernels are compiled for: the power-of-two rows up to 512 keep 4 points per lane (two or three
// waves); rows of 1024 (8 points) and rows with factors 3 / 5 (5..9 points per lane) need a whole SIMD's registers
// Round 3 (tools/shape_probe.py, A/B on one box): plans with 5 and 6 points per lane keep two waves without spilling -- the GGA
// mid stage at 240^3 59 -> 39 ps per point, evaluation 3.23 -> 3.11 ms, 120^3 0.553 -> 0.526 ms; the GGA mid stage alone also
// with 8 / 9 points (270^3: 53 -> 34 ps per point, evaluation 5.28 -> 5.12 ms), where the power and combine kernels spill
// (zf_powers at 320^3 13.7 -> 27.5 ps per point) and keep one wave.
#ifndef OFDFT_Z_EMAX
#define OFDFT_Z_EMAX 6          // most points per lane that still get the kernel's wanted waves per SIMD
#endif
#ifndef OFDFT_Z_EMAX_PBE
#define OFDFT_Z_EMAX_PBE 9      // ... for the GGA mid stage (zpbe2)
#endif
// (fp32 build: half the registers per point -- rows of 1024 keep the wanted waves, OFDFT_Z_F32_BIG_WAVES; config 5's grid)
#ifndef OFDFT_Z_F32_BIG_WAVES
#define OFDFT_Z_F32_BIG_WAVES 1
#endif
template <int M, int E, int EMAX = OFDFT_Z_EMAX> constexpr int z_waves(int want) {
    if (sizeof(real) == 4 && OFDFT_Z_F32_BIG_WAVES && M >= 512 && E <= 8) return want > 2 ? 2 : want;
    return (M >= 512 || E > EMAX) ? (want > 2 ? 2 : 1) : want;
}

// zf_powers: the single-output form holds one row of points and its transform only
template <int M, int E, bool ONE> constexpr int z_powers_waves() {
    if (!ONE) return z_waves<M, E>(3);
    if (M >= 512 || E > OFDFT_Z_EMAX) return (sizeof(real) == 4 && E <= 8) ? 4 : z_waves<M, E>(3);
    return 4;
}

// lane geometry of the z kernels
template <int M, int E> struct ZLane {
    int j;            // lane's position inside its row group
    int rw;           // row inside the wave
    long long row_u;  // first row of this wave (wave-uniform)
    long long row;    // this lane's row
    bool valid;
    real* mine;     // LDS buffer of this row
    __device__ __forceinline__ ZLane(const SpecGeom& g, real* lds) {
        using W = ZW<M, E>;
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        j = lane % W::P;
        rw = lane / W::P;
        row_u = uniform64((long long)(blockIdx.x + g.blk0) * W::RPB + (long long)wave * W::RPWV);
        row = row_u + rw;
        valid = row < g.nrows;
        mine = lds + (wave * W::RPWV + rw) * W::RS;
    }
    // a row group placed by the caller (the resident small-grid kernel): only j, rw, valid and mine are meaningful
    __device__ __forceinline__ ZLane(int j_, int rw_, real* mine_, bool valid_) : j(j_), rw(rw_), row_u(0), row(rw_), valid(valid_), mine(mine_) {}
};

// The row transforms read ~9 twiddle factors per row and array; as global loads each of them was a dependent L2 round
// trip whose s_waitcnt vmcnt(0) also drained the row loads in flight (the vm counter retires in order).  The z kernels
// therefore copy the two tables (W_M^k and W_2M^k, k < M) into LDS once per workgroup and read them with ds_read.
// Two phases, so that the copy costs no latency of its own: the table elements are REQUESTED first thing in the kernel,
// the kernel then issues its first row loads, and only then are the elements written to LDS (z_tw_commit, one barrier).
template <int M, int E> struct ZTwReq {
    static constexpr int CNT = (M + ZW<M, E>::TPB - 1) / ZW<M, E>::TPB;
    cplx m[CNT], n[CNT];
};
template <int M, int E>
__device__ __forceinline__ ZTwReq<M, E> z_tw_request(const cplx* __restrict__ twM_g, const cplx* __restrict__ twN_g) {
    ZTwReq<M, E> r;
#if OFDFT_Z_LDS_TWIDDLES
#pragma unroll
    for (int c = 0; c < ZTwReq<M, E>::CNT; ++c) {
        const int i = threadIdx.x + c * ZW<M, E>::TPB;
        r.m[c] = twM_g[i < M ? i : 0];
        r.n[c] = twN_g[i < M ? i : 0];
    }
#endif
    return r;
}
template <int M, int E>
__device__ __forceinline__ void z_tw_commit(real* lds, const ZTwReq<M, E>& r, const cplx* __restrict__ twM_g,
                                            const cplx* __restrict__ twN_g, const cplx*& twM, const cplx*& twN) {
#if OFDFT_Z_LDS_TWIDDLES
    cplx* t = reinterpret_cast<cplx*>(lds + ZW<M, E>::ROWS);
#pragma unroll
    for (int c = 0; c < ZTwReq<M, E>::CNT; ++c) {
        const int i = threadIdx.x + c * ZW<M, E>::TPB;
        if (i < M) {
            t[i] = r.m[c];
            t[M + i] = r.n[c];
        }
    }
    __syncthreads();
    twM = t;
    twN = t + M;
#else
    twM = twM_g;
    twN = twN_g;
#endif
}

// ---- real rows: a register slot holds the pair (a[2 e], a[2 e + 1]) of element e = j + cin(q) -- rows that go INTO a
// forward transform -- or e = j + cout(q) -- rows combined with what comes OUT of an inverse transform (the two patterns
// coincide, e = j + P q, for the power-of-two plans)
template <int M, int E, bool OUT = false>
__device__ __forceinline__ void z_load_real(cplx (&v)[E], const ZLane<M, E>& z, const real* __restrict__ a) {
    using W = ZW<M, E>;
    using PL = typename W::PL;
    const cplx* ub = reinterpret_cast<const cplx*>(a + z.row_u * W::N2);
    const unsigned voff = (unsigned)((z.rw * M + z.j) * kCB);
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const bool on = OUT ? (PL::slot_out(q) && PL::lane_out(z.j, q)) : (PL::slot_in(q) && PL::lane_in(z.j, q));
        v[q] = (z.valid && on) ? buf_load_c_aux<OFDFT_ZR_LD_AUX>(ub + (OUT ? PL::cout(q) : PL::cin(q)), voff) : mkc(0.0, 0.0);
    }
}
template <int M, int E>
__device__ __forceinline__ void z_store_real(const cplx (&v)[E], const ZLane<M, E>& z, real* __restrict__ a) {
    using W = ZW<M, E>;
    using PL = typename W::PL;
    cplx* ub = reinterpret_cast<cplx*>(a + z.row_u * W::N2);
    const unsigned voff = (unsigned)((z.rw * M + z.j) * kCB);
    if (z.valid) {
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(z.j, q)) buf_store_c_aux<OFDFT_ZR_ST_AUX>(ub + PL::cout(q), voff, v[q]);
    }
}

// element offset of coefficient kz of row `row` in the block-8 layout (general form; the power-of-two rows use the split
// uniform / per-lane form below)
__device__ __forceinline__ long long z_spec_off(const SpecGeom& g, long long row, int kz) {
    return kz < g.nzm ? (((long long)(kz >> 3)) * g.nrows + row) * 8 + (kz & 7) : g.main_count + (long long)(kz - g.nzm) * g.nrows + row;
}
// partner index of the split r2c / c2r post-processing
template <int M> __device__ __forceinline__ int z_mirror(int k) {
    if constexpr ((M & (M - 1)) == 0) return (M - k) & (M - 1);
    else return k == 0 ? 0 : M - k;
}

// spectrum element k = j + P q of row `row` lives at ((k>>3)*nrows + row)*8 + (k&7); the (P q)>>3 part of
// the block index is wave-uniform (folded into the base), the rest is the per-lane offset
template <int M, int E> __device__ __forceinline__ unsigned z_spec_voff(const ZLane<M, E>& z, const SpecGeom& g) {
    return (unsigned)((((long long)(z.j >> 3) * g.nrows + z.rw) * 8 + (z.j & 7)) * kCB);
}
template <int M, int E, int Q> __device__ __forceinline__ long long z_spec_ubase(const ZLane<M, E>& z, const SpecGeom& g) {
    constexpr int bu = (ZW<M, E>::P * Q) >> 3;
    constexpr int kin_u = (ZW<M, E>::P * Q) & 7;     // non-zero only when P < 8
    return ((long long)bu * g.nrows + z.row_u) * 8 + kin_u;
}

// forward: real pairs in v -> half-spectrum row written to `spec` (block-8 layout + Nyquist plane)
template <int M, int E, bool TT = (OFDFT_Z_TWTAB != 0)>
__device__ __forceinline__ void z_forward_store(cplx (&v)[E], const ZLane<M, E>& z, cplx* __restrict__ spec,
                                                const SpecGeom& g, const cplx* __restrict__ twM,
                                                const cplx* __restrict__ twN) {
    using W = ZW<M, E>;
    using PL = typename W::PL;
    constexpr int P = W::P;
    wave_line_fft<M, E, false, TT, z_cx<E>()>(v, z.j, z.mine, twM);
    real cr_m[E], c0r;
    exchange_sync<true>();
    if constexpr (PL::EXACT) {
#pragma unroll
        for (int q = 0; q < E; ++q) z.mine[lpos<PL>(z.j + P * q)] = v[q].x;
        exchange_sync<true>();
#pragma unroll
        for (int q = 0; q < E; ++q) cr_m[q] = z.mine[lpos<PL>((M - (z.j + P * q)) & (M - 1))];
        c0r = z.mine[0];
        exchange_sync<true>();
#pragma unroll
        for (int q = 0; q < E; ++q) z.mine[lpos<PL>(z.j + P * q)] = v[q].y;
        exchange_sync<true>();
        const unsigned voff = z_spec_voff<M, E>(z, g);
        static_for<E>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int k = z.j + P * q;
            const real ci_m = z.mine[lpos<PL>((M - k) & (M - 1))];
            const cplx ev = mkc(0.5 * (v[q].x + cr_m[q]), 0.5 * (v[q].y - ci_m));
            const cplx od = mkc(0.5 * (v[q].y + ci_m), -0.5 * (v[q].x - cr_m[q]));
            const cplx X = cadd(ev, cmul(twN[k], od));
            if (z.valid) buf_store_c_aux<OFDFT_ZS_ST_AUX>(spec + z_spec_ubase<M, E, q>(z, g), voff, X);
        });
    } else {          // rows with factors 3 / 5: slot q holds coefficient k = j + cout(q); general addressing
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(z.j, q)) z.mine[lpos<PL>(z.j + PL::cout(q))] = v[q].x;
        exchange_sync<true>();
#pragma unroll
        for (int q = 0; q < E; ++q) cr_m[q] = (PL::slot_out(q) && PL::lane_out(z.j, q)) ? z.mine[lpos<PL>(z_mirror<M>(z.j + PL::cout(q)))] : (real)0.0;
        c0r = z.mine[0];
        exchange_sync<true>();
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (PL::slot_out(q) && PL::lane_out(z.j, q)) z.mine[lpos<PL>(z.j + PL::cout(q))] = v[q].y;
        exchange_sync<true>();
#pragma unroll
        for (int q = 0; q < E; ++q) {
            if (!(PL::slot_out(q) && PL::lane_out(z.j, q))) continue;
            const int k = z.j + PL::cout(q);
            const real ci_m = z.mine[lpos<PL>(z_mirror<M>(k))];
            const cplx ev = mkc(0.5 * (v[q].x + cr_m[q]), 0.5 * (v[q].y - ci_m));
            const cplx od = mkc(0.5 * (v[q].y + ci_m), -0.5 * (v[q].x - cr_m[q]));
            const cplx X = cadd(ev, cmul(twN[k], od));
            if (z.valid) spec[z_spec_off(g, z.row, k)] = X;
        }
    }
    if (z.j == 0 && z.valid) {
        const real c0i = z.mine[0];
        spec[z_spec_off(g, z.row, M)] = mkc(c0r - c0i, 0.0);
    }
    exchange_sync<true>();
}

// the loads of an inverse row transform: coefficient k = j + cin(q) of row `row` into slot q, the Nyquist coefficient (lane
// j == 0) into nyq.  Issued one array AHEAD of the transform that consumes them by the fused kernels (software pipeline of
// depth one: E more complex registers buy a second row of HBM requests in flight per wave)
template <int M, int E>
__device__ __forceinline__ void z_issue_row(cplx (&v)[E], real& nyq, const ZLane<M, E>& z, const cplx* __restrict__ spec,
                                            const SpecGeom& g) {
    using PL = typename ZW<M, E>::PL;
    if constexpr (PL::EXACT) {
        const unsigned voff = z_spec_voff<M, E>(z, g);
        static_for<E>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            v[q] = z.valid ? buf_load_c_aux<OFDFT_ZS_LD_AUX>(spec + z_spec_ubase<M, E, q>(z, g), voff) : mkc(0.0, 0.0);
        });
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q)
            v[q] = (z.valid && PL::slot_in(q) && PL::lane_in(z.j, q)) ? spec[z_spec_off(g, z.row, z.j + PL::cin(q))] : mkc(0.0, 0.0);
    }
    nyq = (z.valid && z.j == 0) ? spec[z_spec_off(g, z.row, M)].x : 0.0;
}

// ---- row transforms that stay on chip (the spectrum lives in registers) -----------------------------------------
// forward: real pairs in v (entry pattern) -> v[q] = coefficient k = j + cout(q) (k < M) of the row's half spectrum; nyq
// (lane j == 0) = coefficient M (real)
template <int M, int E, bool TT = (OFDFT_Z_TWTAB != 0)>
__device__ __forceinline__ void z_forward_regs(cplx (&v)[E], const ZLane<M, E>& z, const cplx* __restrict__ twM,
                                               const cplx* __restrict__ twN, real& nyq) {
    using W = ZW<M, E>;
    using PL = typename W::PL;
    wave_line_fft<M, E, false, TT, z_cx<E>()>(v, z.j, z.mine, twM);
    real cr_m[E], c0r;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q)
        if (PL::slot_out(q) && PL::lane_out(z.j, q)) z.mine[lpos<PL>(z.j + PL::cout(q))] = v[q].x;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q) cr_m[q] = (PL::slot_out(q) && PL::lane_out(z.j, q)) ? z.mine[lpos<PL>(z_mirror<M>(z.j + PL::cout(q)))] : (real)0.0;
    c0r = z.mine[0];
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q)
        if (PL::slot_out(q) && PL::lane_out(z.j, q)) z.mine[lpos<PL>(z.j + PL::cout(q))] = v[q].y;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q) {
        if (!(PL::slot_out(q) && PL::lane_out(z.j, q))) continue;
        const int k = z.j + PL::cout(q);
        const real ci_m = z.mine[lpos<PL>(z_mirror<M>(k))];
        const cplx ev = mkc(0.5 * (v[q].x + cr_m[q]), 0.5 * (v[q].y - ci_m));
        const cplx od = mkc(0.5 * (v[q].y + ci_m), -0.5 * (v[q].x - cr_m[q]));
        v[q] = cadd(ev, cmul(twN[k], od));
    }
    nyq = c0r - z.mine[0];
    exchange_sync<true>();
}

// inverse of the above: v[q] = coefficient k = j + cin(q), nyq = coefficient M -> unscaled real pairs (exit pattern);
// imaginary parts of the kz = 0 / Nyquist coefficients are ignored, as irfftn does
template <int M, int E, bool TT = (OFDFT_Z_TWTAB != 0)>
__device__ __forceinline__ void z_inverse_regs(cplx (&v)[E], const ZLane<M, E>& z, const cplx* __restrict__ twM,
                                               const cplx* __restrict__ twN, real nyq) {
    using W = ZW<M, E>;
    using PL = typename W::PL;
    real xr_m[E];
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q)
        if (PL::slot_in(q) && PL::lane_in(z.j, q)) z.mine[lpos<PL>(z.j + PL::cin(q))] = v[q].x;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q) xr_m[q] = (PL::slot_in(q) && PL::lane_in(z.j, q)) ? z.mine[lpos<PL>(z_mirror<M>(z.j + PL::cin(q)))] : (real)0.0;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q)
        if (PL::slot_in(q) && PL::lane_in(z.j, q)) z.mine[lpos<PL>(z.j + PL::cin(q))] = v[q].y;
    exchange_sync<true>();
#pragma unroll
    for (int q = 0; q < E; ++q) {
        if (!(PL::slot_in(q) && PL::lane_in(z.j, q))) continue;
        const int k = z.j + PL::cin(q);
        const real xi_m = z.mine[lpos<PL>(z_mirror<M>(k))];
        const cplx x = v[q];
        if (k == 0) {
            v[q] = mkc(x.x + nyq, x.x - nyq);
        } else {
            const cplx ev = mkc(x.x + xr_m[q], x.y - xi_m);
            const cplx d = mkc(x.x - xr_m[q], x.y + xi_m);
            const cplx od = cmul(d, cconj(twN[k]));
            v[q] = mkc(ev.x - od.y, ev.y + od.x);
        }
    }
    exchange_sync<true>();
    wave_line_fft<M, E, true, TT, z_cx<E>()>(v, z.j, z.mine, twM);
    exchange_sync<true>();
    if constexpr (!PL::EXACT) {      // slots no butterfly of the last stage wrote: keep them out of every sum downstream
#pragma unroll
        for (int q = 0; q < E; ++q)
            if (!(PL::slot_out(q) && PL::lane_out(z.j, q))) v[q] = mkc(0.0, 0.0);
    }
}

// inverse: half-spectrum row of `spec` -> unscaled real pairs in v (imaginary parts of kz = 0 / Nyquist ignored)
template <int M, int E, bool TT = (OFDFT_Z_TWTAB != 0)>
__device__ __forceinline__ void z_load_inverse(cplx (&v)[E], const ZLane<M, E>& z, const cplx* __restrict__ spec,
                                               const SpecGeom& g, const cplx* __restrict__ twM,
                                               const cplx* __restrict__ twN) {
    // keep this row's loads below the previous array's work: hoisting the loads of ALL arrays to the kernel top
    // (the compiler's default) costs 16 VGPRs per array and collapses the occupancy that hides their latency
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    real nyq;
    z_issue_row<M, E>(v, nyq, z, spec, g);
    z_inverse_regs<M, E, TT>(v, z, twM, twN, nyq);
}

// index derivative along the row, D_c f = F^-1[i f_c F[f]] with the integer frequency f_c = k (rfftfreq,
// functional_tools.py:155): real pairs in (entry pattern) -> N2 x (derivative) out (exit pattern).  The k = 0 and Nyquist
// terms are purely imaginary after the multiplication and drop out of the real inverse, as they do in the reference's irfftn.
template <int M, int E, bool TT = (OFDFT_Z_TWTAB != 0)>
__device__ __forceinline__ void z_deriv_row(cplx (&v)[E], const ZLane<M, E>& z, const cplx* __restrict__ twM,
                                            const cplx* __restrict__ twN) {
    using PL = typename ZW<M, E>::PL;
    real nyq;
    z_forward_regs<M, E, TT>(v, z, twM, twN, nyq);
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const real k = (real)(z.j + PL::cout(q));
        v[q] = mkc(-k * v[q].y, k * v[q].x);
    }
    repattern_out_to_in<PL, true>(v, z.j, z.mine);
    z_inverse_regs<M, E, TT>(v, z, twM, twN, 0.0);
}

// x^y for x > 0 as exp(y log x) with the lean log / exp of fastmath.h: a few ulp for the |y log x| = O(1..10) met here,
// ~52 instead of ~250 instructions of the fully-general pow() (which the unfused pipeline keeps using as an
// independent check).
__device__ __forceinline__ real pow_pos(real x, real y) { return fm::pow_pos(x, y); }

// density of a point from the kernel's source array: n = cscale * x^2 (source = chi) or n = x (source = den)
struct DenSrc {
    const real* src;
    real cscale;
    int from_chi;
    const acc_t* cscale_dev;    // when set, the scale is read from device memory (no host round trip after sum chi^2)
    __device__ __forceinline__ real operator()(real x) const {
        const real c = cscale_dev ? (real)*cscale_dev : cscale;
        return from_chi ? c * x * x : x;
    }
    // the same source with the device-resident scale read ONCE (a kernel calls this first: left in memory, the scale
    // was re-read -- a dependent L2 round trip -- at every point, because the kernel's stores might alias it)
    __device__ __forceinline__ DenSrc resolved() const {
        DenSrc r = *this;
        if (cscale_dev) r.cscale = (real)*cscale_dev;
        r.cscale_dev = nullptr;
        return r;
    }
};

// ------------------------------------------------------------------------------------------------
// plain z passes in the wave-local form (rows whose half length has factors 3 / 5; the power-of-two rows use
// zfwd_kernel / zinv_kernel of fft_kernels.h): real rows -> half spectrum, half spectrum -> real rows times `scale`
template <int M, int E>
__global__ __launch_bounds__(256) void zfwd_w_kernel(const real* __restrict__ in, cplx* __restrict__ spec, SpecGeom g,
                                                     const cplx* __restrict__ twM_g, const cplx* __restrict__ twN_g) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    const ZLane<M, E> z(g, lds);
    cplx v[E];
    z_load_real<M, E>(v, z, in);
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    z_forward_store<M, E>(v, z, spec, g, twM, twN);
}
template <int M, int E>
__global__ __launch_bounds__(256) void zinv_w_kernel(const cplx* __restrict__ spec, real* __restrict__ out, SpecGeom g,
                                                     const cplx* __restrict__ twM_g, const cplx* __restrict__ twN_g, real scale) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    const ZLane<M, E> z(g, lds);
    cplx v[E];
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    z_load_inverse<M, E>(v, z, spec, g, twM, twN);
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = mkc(v[q].x * scale, v[q].y * scale);
    z_store_real<M, E>(v, z, out);
}

// ------------------------------------------------------------------------------------------------
// chi|n -> n^ and (sqrt n)^      (functionals.py:65 rfftn(den); :245 laplacian(k2, sqrt_den))
template <int M, int E>
__global__ __launch_bounds__(256, (ZW<M, E>::PL::EXACT ? 3 : 2)) void zf_density_kernel(DenSrc ds, cplx* __restrict__ out_n, cplx* __restrict__ out_s,
                                                         SpecGeom g, const cplx* __restrict__ twM_g,
                                                         const cplx* __restrict__ twN_g, real* __restrict__ dzn = nullptr) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    ds = ds.resolved();
    const ZLane<M, E> z(g, lds);
    cplx x[E], v[E];
    z_load_real<M, E>(x, z, ds.src);
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    if (out_n) {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mkc(ds(x[q].x), ds(x[q].y));
        z_forward_store<M, E>(v, z, out_n, g, twM, twN);
    }
    if (out_s) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const real a = ds(x[q].x), b = ds(x[q].y);
            v[q] = mkc(a != 0.0 ? sqrt(a) : 0.0, b != 0.0 ? sqrt(b) : 0.0);     // functionals.py:242-243
        }
        z_forward_store<M, E>(v, z, out_s, g, twM, twN);
    }
    if (dzn) {       // D_c n, the index derivative along z, formed on chip (split-derivative form of the GGA chain)
        const real sc = 1.0 / (real)(2 * M);
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mkc(ds(x[q].x), ds(x[q].y));
        z_deriv_row<M, E>(v, z, twM, twN);
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mkc(v[q].x * sc, v[q].y * sc);
        z_store_real<M, E>(v, z, dzn);
    }
}

// chi|n -> spectra of  n^e0 theta^m / m!  (m = 0,1,2; out[0..2])  and  n^e1 theta^m / m!  (out[3..5]).
// WGC99: e0 = beta, e1 = alpha (functionals.py:976-981 and the closed form SURVEY §8a-8); WT: only out[0]
// (and out[3] when alpha != beta), theta unused.  One pow per point when e0 + e1 = 5/3.
struct PowersArgs {
    cplx* out[6];
    real e0, e1, nref;
    int sum53;       // e0 + e1 == 5/3: n^e1 = n^(5/3) / n^e0
};
// ONE: only out[0] is wanted (Wang-Teter with alpha = beta, BASELINE configs 2 and 5): n, its logarithm and the second half are
// dead after the power is formed -- the general instantiation keeps them (176 VGPRs at 8 points per lane in the fp32 build, two
// waves per SIMD; this one fits four)
template <int M, int E, bool ONE = false>
__global__ __launch_bounds__(256, (z_powers_waves<M, E, ONE>())) void zf_powers_kernel(DenSrc ds, PowersArgs pa, SpecGeom g,
                                                        const cplx* __restrict__ twM_g, const cplx* __restrict__ twN_g) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    ds = ds.resolved();
    const ZLane<M, E> z(g, lds);
    cplx n[E], a[E], v[E];
    z_load_real<M, E>(n, z, ds.src);
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    // ONE logarithm per point serves both powers: n^e0 = exp(e0 L) now, n^e1 = exp(e1 L) for the second half (L is kept
    // in `l`; with e0 + e1 = 5/3 the former cbrt + quotient form cost more than the second exp)
    cplx l[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        n[q] = mkc(ds(n[q].x), ds(n[q].y));
        l[q] = mkc(fm::log(n[q].x), fm::log(n[q].y));
        a[q] = mkc(fm::exp(pa.e0 * l[q].x), fm::exp(pa.e0 * l[q].y));
    }
    if constexpr (ONE) {
        z_forward_store<M, E>(a, z, pa.out[0], g, twM, twN);
        return;
    }
    for (int half = 0; half < 2; ++half) {
        if (half == 1) {
            if (!pa.out[3] && !pa.out[4] && !pa.out[5]) break;
#pragma unroll
            for (int q = 0; q < E; ++q) a[q] = mkc(fm::exp(pa.e1 * l[q].x), fm::exp(pa.e1 * l[q].y));
        }
        cplx* const* o = pa.out + 3 * half;
        if (o[0]) {
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = a[q];
            z_forward_store<M, E>(v, z, o[0], g, twM, twN);
        }
        if (o[1]) {
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = mkc(a[q].x * (n[q].x - pa.nref), a[q].y * (n[q].y - pa.nref));
            z_forward_store<M, E>(v, z, o[1], g, twM, twN);
        }
        if (o[2]) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const real tx = n[q].x - pa.nref, ty = n[q].y - pa.nref;
                v[q] = mkc(0.5 * a[q].x * tx * tx, 0.5 * a[q].y * ty * ty);
            }
            z_forward_store<M, E>(v, z, o[2], g, twM, twN);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PBE mid stage on chip: spectra of grad n (x already in real space, y done) -> real space -> f, df/dn,
// flux_j = df/d|grad n|^2 * d_j n -> spectra again, in place.  (functionals.py:1597-1618;
// tests/tools_for_tests.py:155-207)
template <int M, int E>
__global__ __launch_bounds__(256, (z_waves<M, E>(OFDFT_ZPBE_WAVES))) void zpbe_kernel(DenSrc ds, cplx* __restrict__ gx, cplx* __restrict__ gy,
                                                   cplx* __restrict__ gz, real* __restrict__ dfdn, real inv_n,
                                                   GgaSel sel, SpecGeom g, const cplx* __restrict__ twM_g,
                                                   const cplx* __restrict__ twN_g, acc_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    ds = ds.resolved();
    const ZLane<M, E> z(g, lds);
    cplx a[E], b[E], c[E], n[E];
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    z_load_inverse<M, E>(a, z, gx, g, twM, twN);
    z_load_inverse<M, E>(b, z, gy, g, twM, twN);
    z_load_inverse<M, E>(c, z, gz, g, twM, twN);
    z_load_real<M, E, true>(n, z, ds.src);
    acc_t acc[kPbeScalars] = {0.0, 0.0, 0.0};
    cplx d[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        const real ax = a[q].x * inv_n, bx = b[q].x * inv_n, cx = c[q].x * inv_n;
        const real ay = a[q].y * inv_n, by = b[q].y * inv_n, cy = c[q].y * inv_n;
        PbePoint p0 = {0, 0, 0, 0, 0}, p1 = {0, 0, 0, 0, 0};
        if (z.valid && ZW<M, E>::PL::slot_out(q) && ZW<M, E>::PL::lane_out(z.j, q)) {
            p0 = pbe_point(ds(n[q].x), ax * ax + bx * bx + cx * cx, sel);
            p1 = pbe_point(ds(n[q].y), ay * ay + by * by + cy * cy, sel);
        }
        acc[0] += p0.fx + p1.fx;
        acc[1] += p0.fc + p1.fc;
        acc[2] += p0.fk + p1.fk;
        d[q] = mkc(p0.dfdn, p1.dfdn);
        a[q] = mkc(p0.dfdg * ax, p1.dfdg * ay);
        b[q] = mkc(p0.dfdg * bx, p1.dfdg * by);
        c[q] = mkc(p0.dfdg * cx, p1.dfdg * cy);
    }
    z_store_real<M, E>(d, z, dfdn);
    using PL = typename ZW<M, E>::PL;      // (rows with factors 3 / 5: results of an inverse sit in the exit pattern)
    repattern_out_to_in<PL, true>(a, z.j, z.mine);
    z_forward_store<M, E>(a, z, gx, g, twM, twN);
    repattern_out_to_in<PL, true>(b, z.j, z.mine);
    z_forward_store<M, E>(b, z, gy, g, twM, twN);
    repattern_out_to_in<PL, true>(c, z.j, z.mine);
    z_forward_store<M, E>(c, z, gz, g, twM, twN);
    block_reduce_store<kPbeScalars>(acc, partial + (long long)g.blk0 * kPbeScalars);
}

// Split-derivative form of the GGA mid stage.  With the index derivatives D_a, D_b, D_c (Cartesian d_j = sum_axis
// b[axis][j] D_axis) only D_a needs the x transform: D_b comes from a y pass with the i f_b multiply (yderiv_kernel),
// D_c is formed on chip.  In: A = (D_a n)^ rows, B = (D_b n)^ rows, dzn = D_c n (real);  out: the contravariant flux
// components G_a, G_b as spectra (in place of A, B) and df/dn - 2 D_c G_c (real) -- the whole z part of the divergence.
struct Bmat { real b[9]; };
// LAPL: the Laplacian-dependent Pauli-Gaussian members (functionals.py:336-403).  Lsp holds, on entry, the rows of
// (lap n)^ = -k^2 n^ (x and y already back in real space); on exit the rows of the spectrum of df/d(lap n), whose
// Laplacian joins the divergence in the next x pass (MixDerivAL).
// The GGA mid stage is bound by fp64 instruction issue (35 % VALU-busy, 3 700 vector instructions per wave of which ~1 400 are
// its six row transforms): its transforms read every twiddle power from the LDS table (StageP TWTAB) instead of forming
// W^2k, W^3k by complex products -- 8 fp64 instructions less per radix-4 butterfly for two more ds_read_b128
// (256^3, A/B on one box: 0.332 / 0.342 -> 0.323 / 0.319 ms).  OFDFT_ZPBE_TWTAB=0: the product tree as in the other z kernels.
#ifndef OFDFT_ZPBE_TWTAB
#define OFDFT_ZPBE_TWTAB 1
#endif
constexpr bool kZpbeTT = OFDFT_ZPBE_TWTAB != 0;
template <int M, int E, bool LAPL>
__global__ __launch_bounds__(256, (z_waves<M, E, OFDFT_Z_EMAX_PBE>(2))) void zpbe2_kernel(DenSrc ds, cplx* __restrict__ A, cplx* __restrict__ B,
                                                                      const real* __restrict__ dzn,
                                                                      real* __restrict__ dfdn, real inv_n, real inv_nz,
                                                                      GgaSel sel, Bmat bm, SpecGeom g,
                                                                      const cplx* __restrict__ twM_g,
                                                                      const cplx* __restrict__ twN_g,
                                                                      acc_t* __restrict__ partial, cplx* __restrict__ Lsp) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    ds = ds.resolved();
    const ZLane<M, E> z(g, lds);
    cplx a[E], b[E], c[E], n[E];
    cplx lp[LAPL ? E : 1];
#if OFDFT_Z_PREFETCH
    {       // B's row is requested before A's is transformed (depth-one software pipeline, see z_issue_row)
        real nyq_a, nyq_b;
        z_issue_row<M, E>(a, nyq_a, z, A, g);
        z_issue_row<M, E>(b, nyq_b, z, B, g);
        z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
        z_inverse_regs<M, E, kZpbeTT>(a, z, twM, twN, nyq_a);
        if constexpr (LAPL) {
            real nyq_l;
            z_issue_row<M, E>(lp, nyq_l, z, Lsp, g);
            z_inverse_regs<M, E, kZpbeTT>(b, z, twM, twN, nyq_b);
            z_inverse_regs<M, E, kZpbeTT>(lp, z, twM, twN, nyq_l);
        } else {
            z_inverse_regs<M, E, kZpbeTT>(b, z, twM, twN, nyq_b);
        }
    }
#else
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    z_load_inverse<M, E, kZpbeTT>(a, z, A, g, twM, twN);
    z_load_inverse<M, E, kZpbeTT>(b, z, B, g, twM, twN);
    if constexpr (LAPL) z_load_inverse<M, E, kZpbeTT>(lp, z, Lsp, g, twM, twN);
#endif
    z_load_real<M, E, true>(c, z, dzn);
    z_load_real<M, E, true>(n, z, ds.src);
    acc_t acc[kPbeScalars] = {0.0, 0.0, 0.0};
    cplx d[E];
    const bool diag = bm.b[1] == 0.0 && bm.b[2] == 0.0 && bm.b[3] == 0.0 && bm.b[5] == 0.0 && bm.b[6] == 0.0 && bm.b[7] == 0.0;   // (uniform)
#pragma unroll
    for (int q = 0; q < E; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        const real da0 = a[q].x * inv_n, db0 = b[q].x * inv_n, dc0 = c[q].x;
        const real da1 = a[q].y * inv_n, db1 = b[q].y * inv_n, dc1 = c[q].y;
        // Cartesian gradient g_j = b[0][j] D_a n + b[1][j] D_b n + b[2][j] D_c n (orthorhombic cells: the diagonal only -- the
        // same numbers, 24 fp64 instructions less per point pair in a kernel bound by their issue)
        real gx0, gx1, gy0, gy1, gz0, gz1;
        if (diag) {
            gx0 = bm.b[0] * da0; gx1 = bm.b[0] * da1;
            gy0 = bm.b[4] * db0; gy1 = bm.b[4] * db1;
            gz0 = bm.b[8] * dc0; gz1 = bm.b[8] * dc1;
        } else {
            gx0 = bm.b[0] * da0 + bm.b[3] * db0 + bm.b[6] * dc0; gx1 = bm.b[0] * da1 + bm.b[3] * db1 + bm.b[6] * dc1;
            gy0 = bm.b[1] * da0 + bm.b[4] * db0 + bm.b[7] * dc0; gy1 = bm.b[1] * da1 + bm.b[4] * db1 + bm.b[7] * dc1;
            gz0 = bm.b[2] * da0 + bm.b[5] * db0 + bm.b[8] * dc0; gz1 = bm.b[2] * da1 + bm.b[5] * db1 + bm.b[8] * dc1;
        }
        PbePoint p0 = {0, 0, 0, 0, 0}, p1 = {0, 0, 0, 0, 0};
        if constexpr (LAPL) {
            real dl0 = 0.0, dl1 = 0.0;
            if (z.valid && ZW<M, E>::PL::slot_out(q) && ZW<M, E>::PL::lane_out(z.j, q)) {
                GgaSel nk = sel;
                nk.k = 0;          // PBE parts (if any) by pbe_point, the kinetic part with its q dependence below
                const real g20 = gx0 * gx0 + gy0 * gy0 + gz0 * gz0, g21 = gx1 * gx1 + gy1 * gy1 + gz1 * gz1;
                p0 = pbe_point(ds(n[q].x), g20, nk);
                p1 = pbe_point(ds(n[q].y), g21, nk);
                pg_laplacian_point(ds(n[q].x), g20, lp[q].x * inv_n, sel, p0, dl0);
                pg_laplacian_point(ds(n[q].y), g21, lp[q].y * inv_n, sel, p1, dl1);
            }
            lp[q] = mkc(dl0, dl1);
        } else if (z.valid && ZW<M, E>::PL::slot_out(q) && ZW<M, E>::PL::lane_out(z.j, q)) {
            p0 = pbe_point(ds(n[q].x), gx0 * gx0 + gy0 * gy0 + gz0 * gz0, sel);
            p1 = pbe_point(ds(n[q].y), gx1 * gx1 + gy1 * gy1 + gz1 * gz1, sel);
        }
        acc[0] += p0.fx + p1.fx;
        acc[1] += p0.fc + p1.fc;
        acc[2] += p0.fk + p1.fk;
        d[q] = mkc(p0.dfdn, p1.dfdn);
        // contravariant components of the flux F = df/dg grad n:  G_axis = sum_j b[axis][j] F_j
        if (diag) {
            a[q] = mkc(p0.dfdg * (bm.b[0] * gx0), p1.dfdg * (bm.b[0] * gx1));
            b[q] = mkc(p0.dfdg * (bm.b[4] * gy0), p1.dfdg * (bm.b[4] * gy1));
            c[q] = mkc(p0.dfdg * (bm.b[8] * gz0), p1.dfdg * (bm.b[8] * gz1));
        } else {
            a[q] = mkc(p0.dfdg * (bm.b[0] * gx0 + bm.b[1] * gy0 + bm.b[2] * gz0),
                       p1.dfdg * (bm.b[0] * gx1 + bm.b[1] * gy1 + bm.b[2] * gz1));
            b[q] = mkc(p0.dfdg * (bm.b[3] * gx0 + bm.b[4] * gy0 + bm.b[5] * gz0),
                       p1.dfdg * (bm.b[3] * gx1 + bm.b[4] * gy1 + bm.b[5] * gz1));
            c[q] = mkc(p0.dfdg * (bm.b[6] * gx0 + bm.b[7] * gy0 + bm.b[8] * gz0),
                       p1.dfdg * (bm.b[6] * gx1 + bm.b[7] * gy1 + bm.b[8] * gz1));
        }
    }
    using PL = typename ZW<M, E>::PL;      // (rows with factors 3 / 5: pointwise results sit in the exit pattern of the inverses)
    repattern_out_to_in<PL, true>(a, z.j, z.mine);
    z_forward_store<M, E, kZpbeTT>(a, z, A, g, twM, twN);
    repattern_out_to_in<PL, true>(b, z.j, z.mine);
    z_forward_store<M, E, kZpbeTT>(b, z, B, g, twM, twN);
    if constexpr (LAPL) {
        repattern_out_to_in<PL, true>(lp, z.j, z.mine);
        z_forward_store<M, E, kZpbeTT>(lp, z, Lsp, g, twM, twN);
    }
    repattern_out_to_in<PL, true>(c, z.j, z.mine);
    z_deriv_row<M, E, kZpbeTT>(c, z, twM, twN);               // N2 x D_c G_c
#pragma unroll
    for (int q = 0; q < E; ++q) d[q] = mkc(d[q].x - 2.0 * inv_nz * c[q].x, d[q].y - 2.0 * inv_nz * c[q].y);
    z_store_real<M, E>(d, z, dfdn);
    block_reduce_store<kPbeScalars>(acc, partial + (long long)g.blk0 * kPbeScalars);
}

// ------------------------------------------------------------------------------------------------
// (8-point lanes at M = 512 -- rows of 1024 -- need more than 256 registers in the fused kernels: those instantiations
// take a whole SIMD's register file, one wave per SIMD, instead of spilling ~1 KB per thread)
// final stage: every convolution spectrum of a row -> real space -> potential and energy integrands.
struct ZCombineArgs {
    DenSrc ds;
    const real* vext;
    const real* dfdn;       // real array from zpbe
    const cplx* vh;
    const cplx* lap;
    const cplx* conv_b;
    const cplx* conv_a;
    const cplx* u[3];
    const cplx* gw[3];
    const cplx* div;
    const cplx* div2;         // split-derivative form: the D_b part of the divergence (added to div)
    real* v_out;
    const real* v_part;     // split form: the WGC99 potential computed by zi_wgc_kernel (then u / gw are not read here)
    int v_part_deferred;    // closure form: v_part is NOT added here (zi_wgc_kernel runs beside this kernel; chi_grad adds it, and
                            // its share of sum(v n) arrives through zi_wgc's second sum)
    unsigned mask;
    real inv_n;
    real wt_alpha, wt_beta, wt_nbar_pa;
    real wgc_alpha, wgc_beta, nref;
    real gtf_inv_n0;
    int gtf_kind;
    int wt_is_56, wgc_sum_53;
    const acc_t* wts_w;     // device weights (w_tf, w_nl) of the stabilised WT-style functional, or null
};

// WGC99 nonlocal part of one row (SURVEY §8a-8 closed form): adds the potential to vacc, returns the thread's energy sum
template <int M, int E>
__device__ __forceinline__ real wgc_row_section(const cplx (&n)[E], cplx (&vacc)[E], cplx (&w)[E], const ZLane<M, E>& z,
                                                  const ZCombineArgs& a, const SpecGeom& g, const cplx* __restrict__ twM,
                                                  const cplx* __restrict__ twN, real sc, real ctf) {
    cplx t1[E], t2[E];
    real e = 0.0;
#if OFDFT_Z_PREFETCH
    // depth-one software pipeline over the six spectra: the next array's row is requested before this one is transformed
    cplx nx[E];
    real nyq_nx, nyq_w;
    z_issue_row<M, E>(nx, nyq_nx, z, a.u[0], g);
#define OFDFT_ZROW(cur, nxt)                                                          \
    {                                                                                 \
        _Pragma("unroll") for (int q = 0; q < E; ++q) w[q] = nx[q];                   \
        nyq_w = nyq_nx;                                                               \
        if ((nxt) != nullptr) z_issue_row<M, E>(nx, nyq_nx, z, (nxt), g);             \
        z_inverse_regs<M, E>(w, z, twM, twN, nyq_w);                                  \
    }
#else
#define OFDFT_ZROW(cur, nxt) z_load_inverse<M, E>(w, z, (cur), g, twM, twN);
#endif
    OFDFT_ZROW(a.u[0], a.u[1])
#pragma unroll
    for (int q = 0; q < E; ++q) t1[q] = mkc(w[q].x * sc, w[q].y * sc);                 // S_e = u0 + ...
    OFDFT_ZROW(a.u[1], a.u[2])
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const real x0 = w[q].x * sc, x1 = w[q].y * sc;
        t1[q].x += (n[q].x - a.nref) * x0;
        t1[q].y += (n[q].y - a.nref) * x1;
        t2[q] = mkc(x0, x1);                                                            // S_1 = u1 + ...
    }
    OFDFT_ZROW(a.u[2], a.gw[0])
#pragma unroll
    for (int q = 0; q < E; ++q) {
        __builtin_amdgcn_sched_barrier(0);      // one point pair at a time: pow_pos() is register-hungry
        const real x0 = w[q].x * sc, x1 = w[q].y * sc;
        const real h0 = n[q].x - a.nref, h1 = n[q].y - a.nref;
        t1[q].x += 0.5 * h0 * h0 * x0;
        t1[q].y += 0.5 * h1 * h1 * x1;
        t2[q].x += h0 * x0;
        t2[q].y += h1 * x1;
        // fold: e_NL = ctf n^alpha S_e ; v += ctf n^(alpha-1) (alpha S_e + n S_1); keep n^(beta-1) in t2
        // n^(beta-1) and n^(alpha-1) from ONE logarithm per point
        const real l0 = fm::log(n[q].x), l1 = fm::log(n[q].y);
        const real pb0 = fm::exp((a.wgc_beta - 1.0) * l0), pb1 = fm::exp((a.wgc_beta - 1.0) * l1);
        const real pa0 = fm::exp((a.wgc_alpha - 1.0) * l0), pa1 = fm::exp((a.wgc_alpha - 1.0) * l1);
        e += ctf * (pa0 * n[q].x * t1[q].x + pa1 * n[q].y * t1[q].y);
        vacc[q].x += ctf * pa0 * (a.wgc_alpha * t1[q].x + n[q].x * t2[q].x);
        vacc[q].y += ctf * pa1 * (a.wgc_alpha * t1[q].y + n[q].y * t2[q].y);
        t2[q] = mkc(pb0, pb1);
    }
    OFDFT_ZROW(a.gw[0], a.gw[1])
#pragma unroll
    for (int q = 0; q < E; ++q) t1[q] = mkc(a.wgc_beta * w[q].x * sc, a.wgc_beta * w[q].y * sc);
    OFDFT_ZROW(a.gw[1], a.gw[2])
#pragma unroll
    for (int q = 0; q < E; ++q) {
        t1[q].x += (a.wgc_beta * (n[q].x - a.nref) + n[q].x) * w[q].x * sc;
        t1[q].y += (a.wgc_beta * (n[q].y - a.nref) + n[q].y) * w[q].y * sc;
    }
    OFDFT_ZROW(a.gw[2], nullptr)
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const real h0 = n[q].x - a.nref, h1 = n[q].y - a.nref;
        t1[q].x += (0.5 * a.wgc_beta * h0 * h0 + n[q].x * h0) * w[q].x * sc;
        t1[q].y += (0.5 * a.wgc_beta * h1 * h1 + n[q].y * h1) * w[q].y * sc;
        vacc[q].x += ctf * t2[q].x * t1[q].x;
        vacc[q].y += ctf * t2[q].y * t1[q].y;
    }
#undef OFDFT_ZROW
    return e;
}

// parked energy sums of zi_combine (slot numbers): ion-electron, Hartree, vW, Wang-Teter, WGC99
constexpr int kParkIe = 0, kParkH = 1, kParkVw = 2, kParkWt = 3, kParkWgc = 4, kParkSlots = 5;
// waves per SIMD zi_combine is compiled for.  fp32 build, rows of 1024 points, lean instantiation (OFDFT_ZI_WAVES_BIG_F32): with the
// park area at 5 KB the LDS no longer caps the kernel at three workgroups per CU, and FOUR waves per SIMD without the depth-one
// pipeline (125 VGPRs) beat three with it (141): 7.1-7.3 -> 6.4-6.5 ps per point at 1024-point rows; the pipeline squeezed into 128
// registers spills (10.2), five waves spill more (16.2) -- profiles/r05_ab_zi_waves.jsonl
#ifndef OFDFT_ZI_WAVES_BIG_F32
#define OFDFT_ZI_WAVES_BIG_F32 4
#endif
template <int M, int E, bool WGC_INLINE> constexpr int zi_combine_waves() {
    if (sizeof(real) == 4 && M >= 512 && E <= 8 && !WGC_INLINE) return OFDFT_ZI_WAVES_BIG_F32;
    return z_waves<M, E>(2);
}
template <int M, int E, bool WGC_INLINE>
__global__ __launch_bounds__(256, (zi_combine_waves<M, E, WGC_INLINE>())) void zi_combine_kernel(ZCombineArgs a, SpecGeom g, const cplx* __restrict__ twM_g,
                                                         const cplx* __restrict__ twN_g, acc_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    a.ds = a.ds.resolved();
    const ZLane<M, E> z(g, lds);
    const real ctf = kCtf;
    // Each energy sum is touched by ONE section only: it is accumulated in a local and parked in the thread's own LDS
    // slots when the section ends (9 live doubles less through the register-hungry WGC99 section; no barrier needed,
    // a thread only reads what it wrote).  The final reduction order is unchanged.
    // (five slots of the grid precision -- the parked values ARE of that precision: 5 KB per workgroup in the fp32 build, where the ten
    // fp64 slots of rounds 2-4, 20 KB, held the kernel to three workgroups per CU at 1024-point rows)
    real* park = lds + ZW<M, E>::LDS / sizeof(real) + threadIdx.x;
#pragma unroll
    for (int s = 0; s < kParkSlots; ++s) park[s * 256] = 0.0;
    cplx n[E], vacc[E], w[E];
    z_load_real<M, E, true>(n, z, a.ds.src);
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    using PL = typename ZW<M, E>::PL;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        n[q] = (z.valid && PL::slot_out(q) && PL::lane_out(z.j, q)) ? mkc(a.ds(n[q].x), a.ds(n[q].y)) : mkc(1.0, 1.0);
        vacc[q] = mkc(0.0, 0.0);
    }
    const real sc = a.inv_n;
    const real w_tf = a.wts_w ? (real)a.wts_w[0] : (real)1.0, w_nl = a.wts_w ? (real)a.wts_w[1] : (real)1.0;
    // ONE n^(-1/6) per point serves every section below (round 5): the vW, Wang-Teter (alpha = 5/6), Thomas-Fermi and local-XC
    // sections each formed it again -- four times per point in BASELINE configs 2 / 5, a fifth of the kernel's vector instructions
    // at 1024-point rows, where it is bound by their issue (3 250 per wave: 5.4 of its 8.0 ms at 1024^3).  n == 0 -> 0: the
    // guard of functionals.py:242-243 (sqrt n and 1 / sqrt n both read as 0 there)
    constexpr unsigned kRootTerms = 4u | 8u | 16u | (0xFu << 6) | (1u << 13);
    constexpr bool ROOTS_ONCE = OFDFT_ZI_ROOTS_ONCE != 0;
    cplx yv[ROOTS_ONCE ? E : 1];
    if constexpr (ROOTS_ONCE) {
        if (a.mask & kRootTerms) {
#pragma unroll
            for (int q = 0; q < E; ++q)
                yv[q] = mkc(n[q].x != 0.0 ? fm::rsixth(n[q].x) : (real)0.0, n[q].y != 0.0 ? fm::rsixth(n[q].y) : (real)0.0);
        }
    }
    auto root_y = [&](int q, int c) -> real {          // n^(-1/6) of point (q, c); 0 for n == 0
        if constexpr (ROOTS_ONCE) return c ? yv[q].y : yv[q].x;
        else {
            const real nn = c ? n[q].y : n[q].x;
            return nn != 0.0 ? fm::rsixth(nn) : (real)0.0;
        }
    };
    // Depth-one software pipeline over the spectra this kernel consumes (lean instantiation): chain[i] = the i-th
    // spectrum or null; a section takes its row from the registers filled one section earlier and requests the next
    // present one before transforming its own (z_issue_row).
    // (not at M = 512 -- rows of 1024 -- where E = 8: the extra row would push the kernel past 256 registers, i.e. from
    // two waves per SIMD to one)
    // (round 5, fp32 build at M = 512: the pipeline costs 16 VGPRs there; with three workgroups per CU -- the old 20-KB park area's
    // cap -- it lifted the kernel from 0.33 to 0.39 of the peak, but a fourth wave per SIMD without it does better still, 0.43 -> 0.47:
    // OFDFT_Z_PIPE_BIG_F32 / OFDFT_ZI_WAVES_BIG_F32)
    constexpr bool PIPE = OFDFT_Z_PREFETCH && !WGC_INLINE && (M < 512 || (sizeof(real) == 4 && OFDFT_Z_PIPE_BIG_F32));
    const bool gga = (a.mask & (7u << 10)) != 0;
    const cplx* chain[6] = {(a.mask & 2u) ? a.vh : nullptr,  (a.mask & 8u) ? a.lap : nullptr, (a.mask & 16u) ? a.conv_b : nullptr,
                            (a.mask & 16u) ? a.conv_a : nullptr, gga ? a.div : nullptr, gga ? a.div2 : nullptr};
    cplx nx[PIPE ? E : 1];
    real nyq_nx = 0.0;
    auto request_after = [&](auto ic) {       // rows of the first present spectrum after section I -> nx
        constexpr int I = decltype(ic)::value;
        const cplx* p = nullptr;
#pragma unroll
        for (int k = 5; k > I; --k)
            if (chain[k]) p = chain[k];
        if constexpr (PIPE)
            if (p) z_issue_row<M, E>(nx, nyq_nx, z, p, g);
    };
    auto take_row = [&](auto ic, const cplx* spec) {      // section I's row -> w (real space, unscaled)
        if constexpr (PIPE) {
#pragma unroll
            for (int q = 0; q < E; ++q) w[q] = nx[q];
            const real nyq_w = nyq_nx;
            request_after(ic);
            z_inverse_regs<M, E>(w, z, twM, twN, nyq_w);
        } else {
            z_load_inverse<M, E>(w, z, spec, g, twM, twN);
        }
    };
    request_after(std::integral_constant<int, -1>{});
    if (a.mask & 2u) {                                   // Hartree  functionals.py:72
        take_row(std::integral_constant<int, 0>{}, a.vh);
        real e = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const real x0 = w[q].x * sc, x1 = w[q].y * sc;
            e += 0.5 * (n[q].x * x0 + n[q].y * x1);
            vacc[q].x += x0;
            vacc[q].y += x1;
        }
        park[kParkH * 256] = e;
    }
    if (a.mask & 8u) {                                   // vW  functionals.py:245; tools_for_tests.py:23-26
        take_row(std::integral_constant<int, 1>{}, a.lap);
        real e = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            const real x0 = w[q].x * sc, x1 = w[q].y * sc;
            // sqrt(n) = n y^3 and 1 / sqrt(n) = y^3 with y = n^(-1/6) (one root, no quotient); n == 0 -> 0 (functionals.py:242-243)
            const real y0 = root_y(q, 0), y1 = root_y(q, 1);
            const real r0 = y0 * y0 * y0, r1 = y1 * y1 * y1;
            e += -0.5 * (n[q].x * r0 * x0 + n[q].y * r1 * x1);
            vacc[q].x += -0.5 * x0 * r0;
            vacc[q].y += -0.5 * x1 * r1;
        }
        park[kParkVw * 256] = e;
    }
    if (a.mask & 16u) {                                  // WT family  functionals.py:650-651; tools_for_tests.py:29-39
        take_row(std::integral_constant<int, 2>{}, a.conv_b);
        cplx pa1[E];
        real e = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            __builtin_amdgcn_sched_barrier(0);
            const real x0 = w[q].x * sc, x1 = w[q].y * sc;
            pa1[q] = a.wt_is_56 ? mkc(root_y(q, 0), root_y(q, 1)) : mkc(pow_pos(n[q].x, a.wt_alpha - 1.0), pow_pos(n[q].y, a.wt_alpha - 1.0));
            e += ctf * ((pa1[q].x * n[q].x - a.wt_nbar_pa) * x0 + (pa1[q].y * n[q].y - a.wt_nbar_pa) * x1);
            const real f = w_nl * (a.conv_a ? a.wt_alpha : 2.0 * a.wt_alpha);
            vacc[q].x += ctf * f * pa1[q].x * x0;
            vacc[q].y += ctf * f * pa1[q].y * x1;
        }
        park[kParkWt * 256] = e;
        if (a.conv_a) {
            take_row(std::integral_constant<int, 3>{}, a.conv_a);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                vacc[q].x += w_nl * ctf * a.wt_beta * pow_pos(n[q].x, a.wt_beta - 1.0) * w[q].x * sc;
                vacc[q].y += w_nl * ctf * a.wt_beta * pow_pos(n[q].y, a.wt_beta - 1.0) * w[q].y * sc;
            }
        }
    }
    if (a.mask & 32u) {                                  // WGC99
        if (WGC_INLINE) {
            park[kParkWgc * 256] = wgc_row_section<M, E>(n, vacc, w, z, a, g, twM, twN, sc, ctf);
        } else if (!a.v_part_deferred) {                 // computed by zi_wgc_kernel on the nonlocal chain's stream
            z_load_real<M, E, true>(w, z, a.v_part);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                vacc[q].x += w[q].x;
                vacc[q].y += w[q].y;
            }
        }
    }
    if (a.mask & (7u << 10)) {                           // PBE / GGA kinetic: v += df/dn - 2 div  (tools_for_tests.py:168-170)
        take_row(std::integral_constant<int, 4>{}, a.div);
        cplx d[E];
        z_load_real<M, E, true>(d, z, a.dfdn);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            vacc[q].x += d[q].x - 2.0 * w[q].x * sc;
            vacc[q].y += d[q].y - 2.0 * w[q].y * sc;
        }
        if (a.div2) {
            take_row(std::integral_constant<int, 5>{}, a.div2);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                vacc[q].x -= 2.0 * w[q].x * sc;
                vacc[q].y -= 2.0 * w[q].y * sc;
            }
        }
    }
    // ---- local terms and the sum of v n
    if (a.mask & 1u) {
        cplx ve[E];
        z_load_real<M, E, true>(ve, z, a.vext);
        real e = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            e += n[q].x * ve[q].x + n[q].y * ve[q].y;
            vacc[q].x += ve[q].x;
            vacc[q].y += ve[q].y;
        }
        park[kParkIe * 256] = e;
    }
    acc_t acc[kCombineScalars];
#pragma unroll
    for (int s = 0; s < kCombineScalars; ++s) acc[s] = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(PL::slot_out(q) && PL::lane_out(z.j, q))) continue;      // (rows with factors 3 / 5: slots without a grid point)
        // (the local sections of a point share one root in either form)
        const fm::Roots<real> rt0 = fm::roots_from_y(n[q].x, root_y(q, 0)), rt1 = fm::roots_from_y(n[q].y, root_y(q, 1));
        if (a.mask & 4u) {                               // TF  functionals.py:223
            const real c0 = rt0.n13, c1 = rt1.n13;
            acc[2] += ctf * (c0 * c0 * n[q].x + c1 * c1 * n[q].y);
            vacc[q].x += w_tf * (5.0 / 3.0) * ctf * c0 * c0;
            vacc[q].y += w_tf * (5.0 / 3.0) * ctf * c1 * c1;
        }
        if (a.mask & (1u << 13)) {                       // vWGTF1 / 2  functionals.py:251-306
            real e0, v0, e1, v1;
            vwgtf_point(n[q].x, rt0.n13, ctf, a.gtf_inv_n0, a.gtf_kind, e0, v0);
            vwgtf_point(n[q].y, rt1.n13, ctf, a.gtf_inv_n0, a.gtf_kind, e1, v1);
            acc[9] += e0 + e1;
            vacc[q].x += v0;
            vacc[q].y += v1;
        }
        if (a.mask & (0xFu << 6)) {                      // local XC
            const XcLocal x0 = lda_point(n[q].x, a.mask, rt0), x1 = lda_point(n[q].y, a.mask, rt1);
            acc[6] += x0.ex + x1.ex;
            acc[7] += x0.ec + x1.ec;
            vacc[q].x += x0.vx + x0.vc;
            vacc[q].y += x1.vx + x1.vc;
        }
        acc[8] += vacc[q].x * n[q].x + vacc[q].y * n[q].y;
    }
    acc[0] = park[kParkIe * 256];
    acc[1] = park[kParkH * 256];
    acc[3] = park[kParkVw * 256];
    acc[4] = park[kParkWt * 256];
    acc[5] = park[kParkWgc * 256];
    if (!z.valid) {
#pragma unroll
        for (int s = 0; s < kCombineScalars; ++s) acc[s] = 0.0;
    }
    if (a.v_out) z_store_real<M, E>(vacc, z, a.v_out);
    block_reduce_store<kCombineScalars>(acc, partial + (long long)g.blk0 * kCombineScalars);
}

// The WGC99 part of the combine on its own (split form): chi|n row + the six result spectra -> v_part rows and the
// energy partial sums (one per workgroup).  Runs on the nonlocal chain's stream while the other chain still works.
template <int M, int E>
__global__ __launch_bounds__(256, (z_waves<M, E>(OFDFT_ZIWGC_WAVES))) void zi_wgc_kernel(ZCombineArgs a, real* __restrict__ v_part, SpecGeom g,
                                                       const cplx* __restrict__ twM_g, const cplx* __restrict__ twN_g,
                                                       acc_t* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const cplx *twM, *twN;
    const ZTwReq<M, E> twq = z_tw_request<M, E>(twM_g, twN_g);
    a.ds = a.ds.resolved();
    const ZLane<M, E> z(g, lds);
    const real ctf = kCtf;
    cplx n[E], vacc[E], w[E];
    z_load_real<M, E, true>(n, z, a.ds.src);
    z_tw_commit<M, E>(lds, twq, twM_g, twN_g, twM, twN);
    using PL = typename ZW<M, E>::PL;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        n[q] = (z.valid && PL::slot_out(q) && PL::lane_out(z.j, q)) ? mkc(a.ds(n[q].x), a.ds(n[q].y)) : mkc(1.0, 1.0);
        vacc[q] = mkc(0.0, 0.0);
    }
    acc_t acc[2];         // the WGC99 energy integrand, and this part's share of sum(v n) (for mu when the parts are merged later)
    acc[0] = wgc_row_section<M, E>(n, vacc, w, z, a, g, twM, twN, a.inv_n, ctf);
    acc[1] = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q)
        if (PL::slot_out(q) && PL::lane_out(z.j, q)) acc[1] += vacc[q].x * n[q].x + vacc[q].y * n[q].y;
    if (!z.valid) acc[0] = acc[1] = 0.0;
    z_store_real<M, E>(vacc, z, v_part);
    block_reduce_store<2>(acc, partial + (long long)g.blk0 * 2);
}

}  // namespace ofdft
