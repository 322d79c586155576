// Wave-local fused x pass (gfx950):  forward x-FFT of NIN spectra -> k-space mixing -> inverse x-FFT of NOUT spectra.
//
// Second design of the fused x pass.  The first one (xfused_kernel, fft_kernels.h) gives every spectrum its own group
// of waves and trades the transformed lines through LDS for the mix: 12 workgroup barriers per tile, ~5 table loads
// per k-point spread over the groups, one tile of loads in flight per wave; rocprofv3 showed it parked or
// issue-stalled 86 % of its wave cycles at 3.9-4.1 TB/s of traffic (profiles/r02_start_sq_counters.md).
// Here a line of LEN points is owned by P = LEN/8 lanes of ONE wavefront (8 complex points per lane, the z kernels'
// ZPlan), and the SAME lanes hold that line of every input spectrum:
//   * all NIN x 8 loads of a lane (and its k-point table entries) are issued back to back -> 3x the bytes in flight;
//   * every LDS exchange of the line transforms is wave-local (program order, no s_barrier);
//   * the mix is register arithmetic: out_O[k] = sum_I coef_OI(k) in_I[k] for the lane's own 8 k-points -- no LDS
//     traffic, no barrier, each table entry loaded once;
// The only workgroup barrier is the one that publishes the twiddle table staged in LDS.
// A 256-thread workgroup = 4 waves covers 4 x (64/P) memory-adjacent lines (LEN = 256: 8 lines = whole 128-B lines).
#pragma once
#include "fft_kernels.h"

namespace ofdft {

// LDS layout of this kernel's line buffers.  Its lanes interleave the wave's 64 / P lines (lane = j * LPWV + l), so the 32
// lanes of a read group span several line buffers; with the padded layout of the other kernels every exchange access then
// costs 2 LDS cycles (SQ_LDS_BANK_CONFLICT 48.8 % of the LDS cycles, profiles/r02_final_sq_counters.md: the kernel is bound
// by its LDS phases).  Per length and precision: XOR swizzle parameters (fft_radix.h: lpos) and the line-buffer stride that
// make every read group and every write group of every stage hit distinct banks (enumerated against the bank rules of
// MI355X_MICROARCH.md, tools/lds_conflicts.py).  OFDFT_XW_SWIZZLE=0 keeps the padded layout (A/B).
#ifndef OFDFT_XW_SWIZZLE
#define OFDFT_XW_SWIZZLE 1
#endif
template <int LEN> struct XwSwz { static constexpr int XS = 0, XM = 0, XMUL = 0, LMUL = 0, RS = LineBuf<LEN>::STRIDE; };
#if OFDFT_XW_SWIZZLE
#if defined(OFDFT_REAL_F32) && !OFDFT_F32_CX
// (b32 accesses: reads and writes in two 32-lane groups over 32 banks)
template <> struct XwSwz<32> { static constexpr int XS = 3, XM = 1, XMUL = 1, LMUL = 0, RS = 34; };
template <> struct XwSwz<64> { static constexpr int XS = 3, XM = 3, XMUL = 1, LMUL = 0, RS = 68; };
template <> struct XwSwz<128> { static constexpr int XS = 3, XM = 7, XMUL = 1, LMUL = 0, RS = 136; };
template <> struct XwSwz<256> { static constexpr int XS = 3, XM = 15, XMUL = 1, LMUL = 0, RS = 272; };
template <> struct XwSwz<512> { static constexpr int XS = 3, XM = 31, XMUL = 1, LMUL = 0, RS = 512; };
#else
// (b64 accesses -- fp64, and the fp32 build's complex-element exchange, OFDFT_F32_CX: reads in two 32-lane groups over 64 banks,
// writes in four 16-lane groups over 32 banks -- the two rules ask
// for different strides between the interleaved lines, which the line-dependent XOR (LMUL * line) & 31 reconciles)
template <> struct XwSwz<32> { static constexpr int XS = 1, XM = 1, XMUL = 1, LMUL = 1, RS = 48; };
template <> struct XwSwz<64> { static constexpr int XS = 1, XM = 1, XMUL = 1, LMUL = 1, RS = 72; };
template <> struct XwSwz<128> { static constexpr int XS = 1, XM = 15, XMUL = 1, LMUL = 5, RS = 144; };
template <> struct XwSwz<256> { static constexpr int XS = 3, XM = 7, XMUL = 1, LMUL = 8, RS = 272; };
template <> struct XwSwz<512> { static constexpr int XS = 3, XM = 15, XMUL = 1, LMUL = 0, RS = 512; };
#endif
#endif
template <int LEN> struct XwPlan : ZPlan<LEN, 8> {};
template <int LEN> struct LdsLayout<XwPlan<LEN>> : LdsLayoutDefault {
    static constexpr int XS = XwSwz<LEN>::XS, XM = XwSwz<LEN>::XM, XMUL = XwSwz<LEN>::XMUL;
};
template <int LEN, bool INV>
__device__ __forceinline__ void xw_line_fft(cplx (&v)[8], int j, real* line, const cplx* __restrict__ tw, int lx) {
    StageP<XwPlan<LEN>, 0, 1, INV, true, false, kCX>::run(v, j, line, tw, lx);
}

template <int LEN> struct XwCfg {
    static constexpr int E = 8;
    using PL = ZPlan<LEN, E>;
    static constexpr int P = LEN / E;                 // lanes per line (1..64)
    static constexpr int LPWV = 64 / P;               // lines per wave
    // (OFDFT_XW_TPB512=1: 8 waves per workgroup where a line takes a whole wave, LEN = 512, so that a workgroup covers whole
    // 128-byte lines like the 256-point kernel does -- measured neutral at 512^3, 6.10 vs 6.20 ms for the WGC99 pair: the
    // 512-point kernel's 1.5 x cost per point is the one-line-per-wave access shape itself, 16 useful bytes per cache line and
    // load instruction; xpass_impl.h therefore hands 512-point lines to the group-parallel kernel)
#ifndef OFDFT_XW_TPB512
#define OFDFT_XW_TPB512 0
#endif
#ifdef OFDFT_XW_TPB
    static constexpr int TPB = OFDFT_XW_TPB;          // (A/B: threads per workgroup for every length)
#else
    static constexpr int TPB = (P == 64 && OFDFT_XW_TPB512) ? 512 : 256;
#endif
    static constexpr int LPB = LPWV * (TPB / 64);     // lines per workgroup
    static constexpr int RS = kCXMul * XwSwz<LEN>::RS;         // LDS reals per line buffer (XwSwz counts elements: fp32 CX = complex)
    static constexpr int ROWS = LPB * RS;
    static constexpr size_t LDS = sizeof(real) * ROWS + sizeof(cplx) * LEN;     // line buffers + the staged twiddle table
};

// all coefficients of one k-point for a table-driven mix are fetched by ONE call (so that they can be requested early);
// computed mixes (functions of the integer indices) are evaluated where they are used
template <class Mix, class = void> struct mix_has_fold : std::false_type {};
template <class Mix> struct mix_has_fold<Mix, std::void_t<decltype(std::declval<const Mix&>().fold_n0)>> : std::true_type {};
template <class Mix, class = void> struct mix_coef_count { static constexpr int N = 0; };
template <class Mix> struct mix_coef_count<Mix, std::void_t<decltype(Mix::kTableReals)>> { static constexpr int N = Mix::kTableReals; };

template <int NIN, int NOUT, int O, int I, class Mix>
__device__ __forceinline__ void xw_accumulate(real& acr, real& aci, const cplx (&in)[NIN], const Mix& mix, int x, int y, int kz,
                                              long long uoff, unsigned loff) {
    if constexpr (I < NIN) {
        if constexpr (Mix::template present<O, I>()) {
            const real cf = mix.template coef<O, I>(x, y, kz, uoff, loff);
            if constexpr (Mix::template imag_oi<O, I>()) {       // (i c)(re + i im) = -c im + i c re
                acr -= cf * in[I].y;
                aci += cf * in[I].x;
            } else {
                acr += cf * in[I].x;
                aci += cf * in[I].y;
            }
        }
        xw_accumulate<NIN, NOUT, O, I + 1, Mix>(acr, aci, in, mix, x, y, kz, uoff, loff);
    }
}
template <int NIN, int NOUT, int O, class Mix>
__device__ __forceinline__ void xw_outputs(cplx (&out)[NOUT], const cplx (&in)[NIN], const Mix& mix, int x, int y, int kz,
                                           long long uoff, unsigned loff) {
    if constexpr (O < NOUT) {
        real acr = 0.0, aci = 0.0;
        xw_accumulate<NIN, NOUT, O, 0, Mix>(acr, aci, in, mix, x, y, kz, uoff, loff);
        out[O] = mkc(acr, aci);
        xw_outputs<NIN, NOUT, O + 1, Mix>(out, in, mix, x, y, kz, uoff, loff);
    }
}

#ifndef OFDFT_XW_WAVES
#define OFDFT_XW_WAVES 2
#endif
#ifndef OFDFT_XW_LD_AUX
#define OFDFT_XW_LD_AUX 0       // the four waves of a workgroup read adjacent 32-B pieces of the same 128-B lines: keep them cached
#endif
#ifndef OFDFT_XW_ST_AUX
#define OFDFT_XW_ST_AUX 0
#endif

template <int LEN, int NIN, int NOUT, class Mix>
__global__ __launch_bounds__(XwCfg<LEN>::TPB, (NIN + NOUT > 3 ? OFDFT_XW_WAVES : 3)) void xw_kernel(XfIo io, LineMap m_main, LineMap m_rem,
                                                                                     int main_blocks, SpecGeom g,
                                                                                     const cplx* __restrict__ tw_g, Mix mix,
                                                                                     XfStride xs) {
    using Cfg = XwCfg<LEN>;
    constexpr int E = Cfg::E, P = Cfg::P, LPWV = Cfg::LPWV, LPB = Cfg::LPB, G = NIN > NOUT ? NIN : NOUT;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l = lane % LPWV;            // lines fastest over the lanes: the memory-contiguous direction
    const int j = lane / LPWV;
    // ---- twiddle table W_LEN^m: requested now, written to LDS after the data loads have been issued
    constexpr int TWC = (LEN + Cfg::TPB - 1) / Cfg::TPB;
    cplx twr[TWC];
#pragma unroll
    for (int c = 0; c < TWC; ++c) {
        const int i = threadIdx.x + c * Cfg::TPB;
        twr[c] = tw_g[i < LEN ? i : 0];
    }
    const bool is_rem = (int)blockIdx.x >= main_blocks;    // one grid: the block-8 main part, then the remainder planes
    const LineMap m = is_rem ? m_rem : m_main;
    int bid = is_rem ? (int)blockIdx.x - main_blocks : (int)blockIdx.x;
    if (LPB * sizeof(cplx) < 128 && !is_rem && bid < (main_blocks & ~15)) {
        // tiles narrower than a 128-B line: the two workgroups that share every line go to ONE XCD (blocks are dealt
        // round-robin over the 8 XCDs) -- speed only, never correctness
        bid = (bid & ~15) + ((bid & 7) << 1) + ((bid >> 3) & 1);
    }
    if (!is_rem) bid += m.blk0;           // launch over a range of kz blocks
    const long long L0 = (long long)bid * LPB;
    const long long L = L0 + wave * LPWV + l;
    const bool valid = L < m.nlines;
    const long long base = valid ? (L / m.d) * m.sb + (L % m.d) * (long long)m.sl : 0;
    int y, kz;                            // k-point coordinates of this line
    if (is_rem) {
        y = (int)(L % g.n1);
        kz = g.nzm + (int)(L / g.n1);
    } else {
        const int c = (int)(L % m.d);
        y = c >> 3;
        kz = m.kz0 + (int)(L / m.d) * 8 + (c & 7);
    }
    const long long region = is_rem ? g.main_count : 0;
    const long long lb0 = line_base(m, L0);
    const long long b0 = uniform64(region + lb0);                               // workgroup-uniform
    const long long se_o = xs.se_out ? xs.se_out : m.se, se_t = xs.tse ? xs.tse : m.se;
    const unsigned voff = valid ? (unsigned)((base - lb0 + (long long)j * m.se) * kCB) : 0u;
    const unsigned voff_o = valid ? (unsigned)((base - lb0 + (long long)j * se_o) * kCB) : 0u;
    const unsigned tloff = valid ? (unsigned)(base - lb0 + (long long)j * se_t) : 0u;
    const long long qstep = uniform64((long long)P * m.se), qstep_o = uniform64((long long)P * se_o),
                    tqstep = uniform64((long long)P * se_t);

    cplx a[G][E];
    static_for<NIN>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        const cplx* ub = io.in[I] + b0;
#pragma unroll
        for (int q = 0; q < E; ++q) a[I][q] = valid ? buf_load_c_aux<OFDFT_XW_LD_AUX>(ub + q * qstep, voff) : mkc(0.0, 0.0);
    });
    // table-driven mixes: the k-point entries are requested with the data (one batch in flight), not one by one later
#ifndef OFDFT_XW_PREFETCH
#define OFDFT_XW_PREFETCH 1
#endif
    constexpr int NC = OFDFT_XW_PREFETCH ? mix_coef_count<Mix>::N : 0;
    real cfs[E][NC > 0 ? NC : 1];
    if constexpr (NC > 0) {
#pragma unroll
        for (int q = 0; q < E; ++q) mix.fetch(cfs[q], b0 + q * tqstep, tloff, valid);
    }
    // ---- publish the twiddles (the only workgroup barrier of the kernel)
    cplx* tw = reinterpret_cast<cplx*>(lds + Cfg::ROWS);
#pragma unroll
    for (int c = 0; c < TWC; ++c) {
        const int i = threadIdx.x + c * Cfg::TPB;
        if (i < LEN) tw[i] = twr[c];
    }
    __syncthreads();
    real* mine = lds + (wave * LPWV + l) * Cfg::RS;
    const int lx = (l * XwSwz<LEN>::LMUL) & 31;       // line-dependent part of the LDS swizzle
    static_for<NIN>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        xw_line_fft<LEN, false>(a[I], j, mine, tw, lx);
        exchange_sync<true>();
    });
    // ---- mix, in registers: the lane owns k-points x = j + P q of its line in every spectrum
#pragma unroll
    for (int q = 0; q < E; ++q) {
        cplx in[NIN], out[NOUT];
#pragma unroll
        for (int I = 0; I < NIN; ++I) in[I] = a[I][q];
        if constexpr (NC > 0) mix.apply(out, in, cfs[q]);
        else xw_outputs<NIN, NOUT, 0, Mix>(out, in, mix, j + P * q, y, kz, b0 + q * tqstep, tloff);
#pragma unroll
        for (int O = 0; O < NOUT; ++O) a[O][q] = out[O];
    }
    static_for<NOUT>([&](auto oc) {
        constexpr int O = decltype(oc)::value;
        xw_line_fft<LEN, true>(a[O], j, mine, tw, lx);
        exchange_sync<true>();
        if (valid) {
            cplx* ub = io.out[O] + b0;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                if (xs.se_out) buf_store_c(ub + q * qstep_o, voff_o, a[O][q]);
                else buf_store_c_aux<OFDFT_XW_ST_AUX>(ub + q * qstep_o, voff_o, a[O][q]);
            }
        }
    });
}

}  // namespace ofdft
