// Cross-wave fused x pass (gfx950):  forward x-FFT of NIN spectra -> k-space mixing -> inverse x-FFT of NOUT spectra,
// with every global access a whole run of memory-adjacent lines.
//
// Third design of the fused x pass.  The wave-local kernel (xwave.h) gives a line to LEN/8 lanes of one wavefront; its
// lanes therefore touch LEN/8 different x planes per load instruction -- 32 cache lines with 32 useful bytes each at 256
// points per line, 64 cache lines with 16 useful bytes at 512 (one line per wave): the texture path looks up one cache
// line per cycle, so the pass is bound by cache-line look-ups, not by bytes (4.0 TB/s at 256^3, 2.7 TB/s at 512^3 where
// the y pass streams 5.4-5.9; profiles/r03_shape_probe_per_kernel.jsonl).
// Here the LEN-point transform is split as LEN = A x S (four-step form, A = waves per workgroup):
//     x = A b + a,   k = b' + S a'
//     X[b' + S a'] = sum_a W_A^(a a') [ W_LEN^(a b') sum_b f[A b + a] W_S^(b b') ]
//   * wave a of the workgroup loads the residue class x = a (mod A) of ALL the tile's lines: its lanes are
//     (line l fastest, b) -- a load instruction covers S/8 x planes x (512/S) memory-adjacent lines, i.e. whole 128-byte
//     (S = 64) or 64-byte (S = 128) runs: 8 (16) cache-line look-ups per instruction instead of 32 (64);
//   * the S-point sub-transforms over b are wave-local (the wave-local kernel's own plan, layout and swizzle: XwPlan<S>);
//   * the radix-A step over a runs ACROSS the waves through LDS -- the exchange a Stockham stage needs anyway, so the
//     LDS traffic equals the wave-local kernel's (two exchanges per line transform); afterwards thread t owns the
//     k-points k = J + (LEN/8) m, m < 8, of one line of EVERY spectrum (J = t / lines-per-tile): the mix is register
//     arithmetic with one table entry per k-point, exactly as in the wave-local kernel;
//   * the inverse mirrors it (inverse radix-A step, conj twiddles, cross exchange, wave-local inverse sub-transforms) and
//     wave a stores its residue class in the same whole runs.
// One workgroup barrier per spectrum and direction: two cross buffers alternate (A x 512 complex each), and the wave-local
// line buffers of a step live inside the wave's own region of that step's buffer, which nobody else touches before the
// step's barrier -- 64 KB (fp64) + tables per workgroup, two workgroups per CU.
#pragma once
#include "xwave.h"

namespace ofdft {

#ifndef OFDFT_XC_A512
#define OFDFT_XC_A512 8       // (512^3, WGC99 pair: A = 4, 64-byte runs: 4.95 ms; A = 8, whole lines, one 512-thread workgroup per CU: 3.8 ms)
#endif
template <int LEN> struct XcWaves { static constexpr int A = 4; };
#ifndef OFDFT_XC_A256
#ifdef OFDFT_REAL_F32
#define OFDFT_XC_A256 4
#else
#define OFDFT_XC_A256 4
#endif
#endif
template <> struct XcWaves<256> { static constexpr int A = OFDFT_XC_A256; };
template <> struct XcWaves<512> { static constexpr int A = OFDFT_XC_A512; };
template <> struct XcWaves<1024> { static constexpr int A = 8; };

// lines per lane (round 4, fp32 build): a tile of the fp32 kernel carried half the bytes of the fp64 tile behind the same seven
// barriers and the same per-tile address / twiddle work, and streamed 3.6 TB/s where the fp64 kernel reaches 5.2 (SQ counters:
// 62 % of its wave cycles waiting).  With NL = 2 a lane owns the same points of TWO memory-adjacent lines -- one 16-byte access
// per point pair, 128-byte runs, the fp64 tile's bytes per barrier and per byte in flight.
#ifndef OFDFT_XC_NL_F32
#define OFDFT_XC_NL_F32 2
#endif
template <int LEN> struct XcCfg {
    static constexpr int A = XcWaves<LEN>::A;         // waves per workgroup = radix of the cross-wave step
    static constexpr int S = LEN / A;                 // length of the wave-local sub-transforms
    static constexpr int E = 8;
    static constexpr int NL = (sizeof(real) == 4 && LEN <= 1024) ? OFDFT_XC_NL_F32 : 1;     // memory-adjacent lines per lane
    static constexpr int P = S / E;                   // lanes per line in a sub-transform
    static constexpr int LPWV = 64 / P;               // line groups per tile (every wave works on all of them)
    static constexpr int LPB = NL * LPWV;             // lines per tile
    static constexpr int TPB = 64 * A;
    static constexpr int PP = TPB / LPWV;             // threads per line in the mix phase (= LEN / 8)
    static constexpr int RR = E / A;                  // radix-A butterflies per thread and line in the cross step
    static constexpr int RS = kCXMul * XwSwz<S>::RS;  // reals per wave-local line buffer
    static constexpr int WPT = 64 * E;                // points of one line group in a wave's region
    // complex elements of one wave's region of a cross buffer: its NL x 64 E points -- or its line buffers, which live there too
    // (fp32 build with complex-element exchange: 8 x 72 x 8 B = 4.5 KB of line buffers per 4 KB of points)
    static constexpr int WLB = (int)((NL * LPWV * RS * sizeof(real) + sizeof(cplx) - 1) / sizeof(cplx));
    static constexpr int WREG = ((NL * WPT > WLB ? NL * WPT : WLB) + 15) / 16 * 16;
    static constexpr int XB = A * WREG;               // complex elements of a cross buffer (>= lines per tile x LEN)
    static_assert(S >= 16 && P <= 64 && E % A == 0 && (NL == 1 || NL == 2), "split");
    // entries of W_LEN^m the cross step reads: m = a b' <= (A - 1)(S - 1)
    static constexpr int TWN = ((A - 1) * (S - 1) + 1 + 15) / 16 * 16;
    static_assert(TWN <= LEN, "table");
    // cross buffers (two alternate, or ONE with a second barrier per step: xc_one_buffer) + W_LEN + W_S
    static constexpr size_t lds_bytes(bool one) { return sizeof(cplx) * ((one ? 1 : 2) * XB + TWN + S); }
    static constexpr size_t LDS = lds_bytes(false);
};
// One cross buffer instead of two (round 5).  Two buffers let step n + 1 write while stragglers still read step n's -- one
// barrier per step -- but they set the workgroup's LDS footprint, and with it how many workgroups a CU holds: at 1024-point
// lines in the fp32 build (config 5) 157 KB, ONE 8-wave workgroup per CU, whose waves load, transform and store in lock step,
// so nothing is in flight while it computes: the 1 -> 1 passes ran at 0.31 of the peak (13.6 us per 128-KB tile, where its loads,
// its 1.7 us of arithmetic and its stores in sequence take about that long).  With one buffer (a second barrier per step) two
// workgroups fit (80 KB each at 1024 points, exactly) and one's memory phases overlap the other's arithmetic.
// Measured (tools/ab_onebuf.sh, one box, ps per grid point, two alternations; profiles/r05_onebuf_ab.jsonl): 1 -> 1 passes at 1024-point
// lines fp32 3.1 -> 2.0-2.2 (the Lindhard mix 3.9 -> 2.6), at 512 points fp64 4.1-4.4 -> 3.5, at 256 points 3.7 -> 3.6; 1 -> 2 at
// 512 points 6.1 -> 5.7, neutral at 256.  Compiled for four waves per SIMD the WGC99 pass (3 -> 3) spills: 24 -> 55 ps -- it keeps
// two buffers unless OFDFT_XC_ONEBUF=2, which leaves its register budget alone.
// OFDFT_XC_ONEBUF: 0 = never, 1 = passes over at most three spectra (1 -> 1, 1 -> 2, 2 -> 1), 2 = every pass
#ifndef OFDFT_XC_ONEBUF
#define OFDFT_XC_ONEBUF 1
#endif
template <int LEN, int NIN, int NOUT> constexpr bool xc_one_buffer() {
    return OFDFT_XC_ONEBUF == 2 || (OFDFT_XC_ONEBUF == 1 && NIN + NOUT <= 3);
}

#ifndef OFDFT_XC_WAVES
#define OFDFT_XC_WAVES 2
#endif
// waves per SIMD the kernel is compiled for: with one cross buffer twice the workgroups fit the LDS
template <int LEN, int NIN, int NOUT> constexpr int xc_waves() {
    // (the 1 -> 2 pass needs ~144 VGPRs in fp64: compiled for twice the waves it spills 68 bytes per thread.  With A = 4 waves per
    // workgroup three workgroups fit a CU at its natural register count and the spill-free kernel is 7 % faster (256-point lines:
    // 5.75 -> 5.35 ps per point); with A = 8 only the capped kernel gets a second workgroup onto the CU and wins by 5 % (512-point
    // lines: 5.65 against 5.95) -- profiles/r05_ab_spill.jsonl)
    return (xc_one_buffer<LEN, NIN, NOUT>() && (NIN + NOUT <= 2 || XcWaves<LEN>::A >= 8) && OFDFT_XC_WAVES < 4) ? 2 * OFDFT_XC_WAVES
                                                                                                                  : OFDFT_XC_WAVES;
}
// cache policy of the data loads: a tile of whole 128-byte runs is read once by ONE wave -> nt (256^3: the WGC99 pair 452 -> 405 us);
// narrower tiles (64-byte runs: the partner workgroup reads the other half of every cache line) keep the lines cached
// (512^3 with A = 4: nt 4.95 -> 5.26 ms)
#ifndef OFDFT_XC_LD_AUX
#define OFDFT_XC_LD_AUX 2
#endif
#ifndef OFDFT_XC_ST_AUX
#define OFDFT_XC_ST_AUX 2
#endif
#ifndef OFDFT_XC_ST_AUX_HALF
#define OFDFT_XC_ST_AUX_HALF 0     // stores of tiles narrower than a 128-byte line
#endif

// NL complex elements that are adjacent in memory by one access (NL = 2: 16 bytes in the fp32 build)
template <int NL, int AUX> __device__ __forceinline__ void buf_load_cn(cplx (&o)[NL], const cplx* ubase, unsigned voff_bytes) {
    if constexpr (NL == 1) {
        o[0] = buf_load_c_aux<AUX>(ubase, voff_bytes);
    } else if constexpr (sizeof(cplx) == 8) {
        u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ubase), (int)voff_bytes, 0, AUX);
        o[0] = *reinterpret_cast<cplx*>(&t);
        o[1] = *(reinterpret_cast<cplx*>(&t) + 1);
    } else {
        o[0] = buf_load_c_aux<AUX>(ubase, voff_bytes);
        o[1] = buf_load_c_aux<AUX>(ubase + 1, voff_bytes);
    }
}
template <int NL, int AUX> __device__ __forceinline__ void buf_store_cn(cplx* ubase, unsigned voff_bytes, const cplx (&v)[NL]) {
    if constexpr (NL == 1) {
        buf_store_c_aux<AUX>(ubase, voff_bytes, v[0]);
    } else if constexpr (sizeof(cplx) == 8) {
        u32x4 t;
        *reinterpret_cast<cplx*>(&t) = v[0];
        *(reinterpret_cast<cplx*>(&t) + 1) = v[1];
        __builtin_amdgcn_raw_buffer_store_b128(t, make_rsrc(ubase), (int)voff_bytes, 0, AUX);
    } else {
        buf_store_c_aux<AUX>(ubase, voff_bytes, v[0]);
        buf_store_c_aux<AUX>(ubase + 1, voff_bytes, v[1]);
    }
}

template <int LEN, int NIN, int NOUT, class Mix>
__global__ __launch_bounds__(XcCfg<LEN>::TPB, (xc_waves<LEN, NIN, NOUT>())) void xc_kernel(XfIo io, LineMap m_main, LineMap m_rem, int main_blocks,
                                                                              SpecGeom g, const cplx* __restrict__ tw_g, Mix mix,
                                                                              XfStride xs) {
    using Cfg = XcCfg<LEN>;
    constexpr int A = Cfg::A, S = Cfg::S, E = Cfg::E, LPWV = Cfg::LPWV, LPB = Cfg::LPB, PP = Cfg::PP, RR = Cfg::RR,
                  TPB = Cfg::TPB, WREG = Cfg::WREG, WPT = Cfg::WPT, NL = Cfg::NL, G = NIN > NOUT ? NIN : NOUT;
    constexpr int LDA = (LPB * sizeof(cplx) >= 128) ? OFDFT_XC_LD_AUX : 0;
    // ... and so are the stores (round 5): a tile of half lines stored nt sent its 64-byte halves to memory on their own -- the
    // 1024-point fp32 passes wrote 5.1-5.75 GB per launch where 4.3 GB leave the kernel (rocprofv3 WRITE_SIZE,
    // profiles/r05_1024_f32_cfg2_rocprof_serialised.md); cached, the two halves meet in the L2 of the XCD both workgroups run on
    constexpr int STA = (LPB * sizeof(cplx) >= 128) ? OFDFT_XC_ST_AUX : OFDFT_XC_ST_AUX_HALF;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    constexpr bool ONEBUF = xc_one_buffer<LEN, NIN, NOUT>();
    cplx* xb = reinterpret_cast<cplx*>(lds);
    cplx* twN = xb + (ONEBUF ? 1 : 2) * Cfg::XB;         // W_LEN^m, m < TWN
    cplx* twS = twN + Cfg::TWN;           // W_S^m = W_LEN^(A m)
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int a = __builtin_amdgcn_readfirstlane(t >> 6);       // residue class of this wave
    const int l = lane % LPWV;            // line groups fastest over the lanes: the memory-contiguous direction
    const int j = lane / LPWV;            // b = j + P q in the load / store phases
    const int J = t / LPWV;               // k = J + PP m in the mix phase (J = a P + j)
    // ---- twiddle tables: requested now, written to LDS after the data loads have been issued
    constexpr int TWC = (Cfg::TWN + TPB - 1) / TPB;
    cplx twr[TWC];
#pragma unroll
    for (int c = 0; c < TWC; ++c) {
        const int i = t + c * TPB;
        twr[c] = tw_g[i < Cfg::TWN ? i : 0];
    }
    const cplx tws = tw_g[(t < S ? t : 0) * A];
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
    const long long L = L0 + NL * l;      // first line of this lane's group (the launcher guarantees whole groups: nlines % NL == 0)
    const bool valid = L < m.nlines;
    const long long base = valid ? (L / m.d) * m.sb + (L % m.d) * (long long)m.sl : 0;
    int y[NL], kz[NL];                    // k-point coordinates of the group's lines
#pragma unroll
    for (int nl = 0; nl < NL; ++nl) {
        const long long Ln = L + nl;
        if (is_rem) {
            y[nl] = (int)(Ln % g.n1);
            kz[nl] = g.nzm + (int)(Ln / g.n1);
        } else {
            const int c = (int)(Ln % m.d);
            y[nl] = c >> 3;
            kz[nl] = m.kz0 + (int)(Ln / m.d) * 8 + (c & 7);
        }
    }
    const long long region = is_rem ? g.main_count : 0;
    const long long lb0 = line_base(m, L0);
    const long long b0 = uniform64(region + lb0);                               // workgroup-uniform
    const long long se_o = xs.se_out ? xs.se_out : m.se, se_t = xs.tse ? xs.tse : m.se;
    const int xa = A * j + a;             // x of register slot 0 in the load / store phases (slot q: x = xa + PP q)
    const unsigned voff = valid ? (unsigned)((base - lb0 + (long long)xa * m.se) * kCB) : 0u;
    const unsigned voff_o = valid ? (unsigned)((base - lb0 + (long long)xa * se_o) * kCB) : 0u;
    const unsigned tloff = valid ? (unsigned)(base - lb0 + (long long)J * se_t) : 0u;
    const unsigned tl0 = valid ? (unsigned)(base - lb0) : 0u;      // ... without the x term (folded table reads: MixWgc::fold_n0)
    const long long qstep = uniform64((long long)PP * m.se), qstep_o = uniform64((long long)PP * se_o),
                    tqstep = uniform64((long long)PP * se_t);

    cplx v[G][NL][E];
    static_for<NIN>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        const cplx* ub = io.in[I] + b0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            cplx pr[NL];
            if (valid) {
                buf_load_cn<NL, LDA>(pr, ub + q * qstep, voff);
            } else {
#pragma unroll
                for (int nl = 0; nl < NL; ++nl) pr[nl] = mkc(0.0, 0.0);
            }
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) v[I][nl][q] = pr[nl];
        }
    });
    // table-driven mixes: the k-point entries of the MIX phase's slots are requested with the data
    constexpr int NC = OFDFT_XW_PREFETCH ? mix_coef_count<Mix>::N : 0;
    real cfs[NL][E][NC > 0 ? NC : 1];
    if constexpr (NC > 0) {
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int q = 0; q < E; ++q) {
                if constexpr (mix_has_fold<Mix>::value) {       // the entry of x > n0 / 2 lives at n0 - x too: read it there (MixWgcFold)
                    const int xq = J + PP * q, xf = 2 * xq > mix.fold_n0 ? mix.fold_n0 - xq : xq;
                    mix.fetch(cfs[nl][q], b0, tl0 + nl + (unsigned)xf * (unsigned)se_t, valid);
                } else {
                    mix.fetch(cfs[nl][q], b0 + q * tqstep, tloff + nl, valid);
                }
            }
    }
    // ---- publish the twiddles
#pragma unroll
    for (int c = 0; c < TWC; ++c) {
        const int i = t + c * TPB;
        if (i < Cfg::TWN) twN[i] = twr[c];
    }
    if (t < S) twS[t] = tws;
    __syncthreads();
    const int lx = (l * XwSwz<S>::LMUL) & 31;       // line-dependent part of the wave-local LDS swizzle
    static_for<NIN>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        cplx* buf = ONEBUF ? xb : xb + (I & 1) * Cfg::XB;
        if constexpr (ONEBUF && I > 0) __syncthreads();      // the previous step's cross reads are done: the buffer is free again
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            real* mine = reinterpret_cast<real*>(buf + a * WREG) + (nl * LPWV + l) * Cfg::RS;
            xw_line_fft<S, false>(v[I][nl], j, mine, twS, lx);          // G_a[b' = j + P q] of line (l, nl)
        }
        exchange_sync<true>();
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int q = 0; q < E; ++q) buf[a * WREG + nl * WPT + lane + 64 * q] = v[I][nl][q];      // index (b' LPWV + l) of region a
        __syncthreads();
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const int bp = J + PP * r;
                cplx c[A];
#pragma unroll
                for (int aa = 0; aa < A; ++aa) c[aa] = buf[aa * WREG + nl * WPT + t + TPB * r];
#pragma unroll
                for (int aa = 1; aa < A; ++aa) c[aa] = cmul(c[aa], twN[aa * bp]);
                Dft<A, false>::run(c);
#pragma unroll
                for (int ap = 0; ap < A; ++ap) v[I][nl][r + RR * ap] = c[ap];           // X[b' + S a'] = X[J + PP (r + RR a')]
            }
    });
    // ---- mix, in registers: the thread owns k-points x = J + PP m of its lines in every spectrum
#pragma unroll
    for (int nl = 0; nl < NL; ++nl)
#pragma unroll
        for (int q = 0; q < E; ++q) {
            cplx in[NIN], out[NOUT];
#pragma unroll
            for (int I = 0; I < NIN; ++I) in[I] = v[I][nl][q];
            if constexpr (NC > 0) mix.apply(out, in, cfs[nl][q]);
            else xw_outputs<NIN, NOUT, 0, Mix>(out, in, mix, J + PP * q, y[nl], kz[nl], b0 + q * tqstep, tloff + nl);
#pragma unroll
            for (int O = 0; O < NOUT; ++O) v[O][nl][q] = out[O];
        }
    static_for<NOUT>([&](auto oc) {
        constexpr int O = decltype(oc)::value;
        cplx* buf = ONEBUF ? xb : xb + ((NIN + O) & 1) * Cfg::XB;
        if constexpr (ONEBUF) __syncthreads();      // everybody is through with the buffer's previous contents (cross reads / line buffers)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const int bp = J + PP * r;
                cplx c[A];
#pragma unroll
                for (int ap = 0; ap < A; ++ap) c[ap] = v[O][nl][r + RR * ap];
                Dft<A, true>::run(c);
#pragma unroll
                for (int aa = 1; aa < A; ++aa) c[aa] = cmul(c[aa], cconj(twN[aa * bp]));
#pragma unroll
                for (int aa = 0; aa < A; ++aa) buf[aa * WREG + nl * WPT + t + TPB * r] = c[aa];
            }
        __syncthreads();
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int q = 0; q < E; ++q) v[O][nl][q] = buf[a * WREG + nl * WPT + lane + 64 * q];
        exchange_sync<true>();
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) {
            real* mine = reinterpret_cast<real*>(buf + a * WREG) + (nl * LPWV + l) * Cfg::RS;
            xw_line_fft<S, true>(v[O][nl], j, mine, twS, lx);
        }
        exchange_sync<true>();
        if (valid) {
            cplx* ub = io.out[O] + b0;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                cplx pr[NL];
#pragma unroll
                for (int nl = 0; nl < NL; ++nl) pr[nl] = v[O][nl][q];
                if (xs.se_out) buf_store_cn<NL, 0>(ub + q * qstep_o, voff_o, pr);
                else buf_store_cn<NL, STA>(ub + q * qstep_o, voff_o, pr);
            }
        }
    });
}

}  // namespace ofdft
