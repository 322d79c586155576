// Arbitrary-length line transforms by Bluestein's chirp-z algorithm on top of the power-of-two register/LDS line FFT:
//   X_k = w_k * sum_n (x_n w_n) conj(w)_{k-n},  w_n = exp(-i pi n^2 / N)
// i.e. one cyclic convolution of padded length M >= 2N - 1 (two M-point FFTs per line, the filter spectrum precomputed).
// This is the engine's path for grid extents that are not powers of two -- which is what the reference's own
// System.ecut2shape (system.py:75-89, always odd extents) produces: O(N log N) per line instead of the O(N^2) of the
// plain DFT kernels (kept for extents above 512).
#pragma once
#include "fastmath.h"
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
// Pointwise pre-operation of an r2c pass (round 5): the real rows of array a are f_a(rin[a]) -- the forward transforms of
// sqrt n (vW), n^e (Wang-Teter) and n^e theta^m / m! (WGC99, theta = n - n_ref) read the density itself, where a separate
// kernel wrote each of these arrays first (map / wgc_prep: a launch, a read and a write each; on the reference's own 53^3 grid
// 9 % of the evaluation, which is bound by its launches there).  Same expressions as map_kernel / wgc_prep_kernel.
enum { BS_PREP_NONE = 0, BS_PREP_SQRT = 1, BS_PREP_POW0 = 2, BS_PREP_POW1 = 3, BS_PREP_POW2 = 4 };
struct BsIo {
    cplx* spec[kBsBatch];
    const real* rin[kBsBatch];
    real* rout[kBsBatch];
    int prep[kBsBatch];      // BS_PREP_* of array a (r2c passes)
    real pe[kBsBatch];       // exponent e of BS_PREP_POW*
    real pnref;              // n_ref of theta
};
__device__ __forceinline__ real bs_prep(real x, int kind, real e, real nref) {
    if (kind == BS_PREP_NONE) return x;
    if (kind == BS_PREP_SQRT) return (x != 0.0) ? sqrt(x) : (real)0.0;           // functionals.py:242-243
    // (the lean exp(e log x) of fastmath.h: with the library's pow -- ~250 fp64 instructions, three arrays each forming it -- the r2c
    // pass lost more than the separate kernels had cost: 255^3 6.35 -> 6.49 ms)
    const real a = x > 0.0 ? fm::pow_pos(x, e) : (real)0.0, th = x - nref;       // functionals.py:974-981
    return kind == BS_PREP_POW0 ? a : (kind == BS_PREP_POW1 ? a * th : (real)0.5 * a * th * th);
}

// Design of the kernels (round 3, second half; the steps and their A/B measurements: DESIGN.md section 13a):
//  * a line is owned by the lanes of ONE wavefront (ZPlan<M, 8>: 8 points per lane, 64 lanes for M = 512), so the two M-point
//    transforms of a line exchange through LDS without an s_barrier;
//  * persistent workgroups: the twiddle table and the filter spectrum are staged in LDS once per workgroup, a lane's chirp values
//    sit in registers for all of its lines (slot q of lane j holds element j + P q of EVERY line);
//  * every global access is a buffer access with ONE 32-bit byte offset per lane and slot; elements past the end of the line (the
//    zero padding of the convolution) and lines past the end of the launch get an offset beyond the descriptor's range -- such
//    loads return zero and such stores are dropped by the hardware, so the loops carry no branches and no 64-bit address
//    arithmetic; direction and kind of line are template parameters;
//  * z rows go through in PAIRS (z = a + i b: one complex convolution for two real rows, the spectra separate with one more
//    wave-local exchange); complex lines are loaded and stored by consecutive lanes for consecutive LINES (whole 128-B runs of
//    the block-8 layout) and handed to the owner wave through its line buffer;
//  * the x transforms and the spectral multiply of a convolution are one kernel (bluestein_xmix_kernel), and on grids of up to
//    64 points per axis so are the z and y passes of an x plane (bluestein_zy_kernel).
// What bounds them: the fp64 arithmetic of 2 M >= 4 N points per N-point line (~660 VALU instructions per line at M = 512) and the
// LDS store rate (a ds_write_b64 costs ~6 cycles per wave); hence the conflict-free line layouts below and the twiddle powers
// formed by a product tree rather than read from the LDS table (OFDFT_BS_TWTAB).
#ifndef OFDFT_BS_TPB
#define OFDFT_BS_TPB 512
#endif
#ifndef OFDFT_BS_WGS
#define OFDFT_BS_WGS 4            // persistent workgroups per CU and array
#endif
#ifndef OFDFT_BS_XTPB
#define OFDFT_BS_XTPB 512
#endif
#ifndef OFDFT_BS_XWAVES
#define OFDFT_BS_XWAVES 2
#endif
#ifndef OFDFT_BS_ZTPB
#define OFDFT_BS_ZTPB 256
#endif
#ifndef OFDFT_BS_ZWAVES
#define OFDFT_BS_ZWAVES 3
#endif
#ifndef OFDFT_BS_WAVES
#define OFDFT_BS_WAVES 4          // waves per SIMD the register allocation aims at (M = 1024 keeps 16 points per lane: 2)
#endif
// LDS line layouts.  With the padded layout of the other kernels 52 % of the LDS-array cycles of these kernels were bank conflicts
// (SQ counters at 255^3, profiles/r03_sq_chirpz_255_before.md).  XOR layouts per length and precision, position
// i ^ ((XMUL ((i >> XS) & XM)) & 31) ^ (LMUL line & 31), found by enumerating the exchanges of these plans against the bank rules of
// MI355X_MICROARCH.md with a line's lanes CONSECUTIVE in the wave (tools/bs_lds_search.py): every read and write group
// conflict-free (fp64 M = 128: writes 1.125 cycles per group); on hardware 5-7 % conflicts.
template <int M, bool F32 = (sizeof(real) == 4)> struct BsLds { static constexpr int XS = 0, XM = 0, XMUL = 0, LMUL = 0, RS = (LineBuf<M>::STRIDE + 1) & ~1; };
template <> struct BsLds<64, false> { static constexpr int XS = 2, XM = 15, XMUL = 1, LMUL = 1, RS = 72; };
template <> struct BsLds<128, false> { static constexpr int XS = 4, XM = 7, XMUL = 3, LMUL = 0, RS = 144; };
template <> struct BsLds<256, false> { static constexpr int XS = 3, XM = 15, XMUL = 1, LMUL = 0, RS = 256; };
template <> struct BsLds<512, false> { static constexpr int XS = 3, XM = 15, XMUL = 1, LMUL = 0, RS = 512; };
template <> struct BsLds<64, true> { static constexpr int XS = 1, XM = 31, XMUL = 1, LMUL = 1, RS = 72; };
template <> struct BsLds<128, true> { static constexpr int XS = 2, XM = 31, XMUL = 1, LMUL = 1, RS = 144; };
template <> struct BsLds<256, true> { static constexpr int XS = 3, XM = 31, XMUL = 1, LMUL = 0, RS = 256; };
template <> struct BsLds<512, true> { static constexpr int XS = 3, XM = 31, XMUL = 1, LMUL = 0, RS = 512; };
#ifndef OFDFT_BS_TWTAB
#define OFDFT_BS_TWTAB 0          // 1: stage twiddles W^(t k) read from the LDS table (fewer fp64 instructions, more LDS time)
#endif
#ifndef OFDFT_BS_TRANSPOSE
#define OFDFT_BS_TRANSPOSE 1
#endif
#ifndef OFDFT_BS_TRANSPOSE_MINP
#define OFDFT_BS_TRANSPOSE_MINP 64          // lines of one wave (M >= 512); shorter lines share waves and gain nothing (129 x 135 x 127)
#endif
#ifndef OFDFT_BS_XOR
#define OFDFT_BS_XOR 1
#endif
template <int M> struct BsPlan : ZPlan<M, 8> {};
template <int M> struct LdsLayout<BsPlan<M>> : LdsLayoutDefault {
    static constexpr int XS = OFDFT_BS_XOR ? BsLds<M>::XS : 0, XM = OFDFT_BS_XOR ? BsLds<M>::XM : 0, XMUL = OFDFT_BS_XOR ? BsLds<M>::XMUL : 0;
};
template <int M> struct BsPlanPick { using type = BsPlan<M>; };
template <> struct BsPlanPick<1024> { using type = Plan<1024>; };
// CLS 0: complex lines, 1: the z passes (real rows in pairs), 2: the fused forward-x / mix / inverse-x kernel (up to three
// spectra of a line in registers)
template <int M, int CLS> struct BsCfg {
    using PL = typename BsPlanPick<M>::type;
    static constexpr int TPB = CLS == 1 ? OFDFT_BS_ZTPB : (CLS == 2 ? OFDFT_BS_XTPB : OFDFT_BS_TPB);
    static constexpr int STRIDE = (OFDFT_BS_XOR && M <= 512) ? BsLds<M>::RS : ((LineBuf<M>::STRIDE + 1) & ~1);      // (even: the buffers also hold complex entries)
    static constexpr int LMUL = (OFDFT_BS_XOR && M <= 512) ? BsLds<M>::LMUL : 0;
    static constexpr int P = PL::P, E = PL::E, LPW = TPB / P;
    static constexpr int WAVES = M >= 1024 ? 2 : (CLS == 1 ? OFDFT_BS_ZWAVES : (CLS == 2 ? OFDFT_BS_XWAVES : OFDFT_BS_WAVES));
    static_assert(PL::EXACT, "element j + P q in slot q");
    static constexpr size_t LDS = sizeof(real) * LPW * STRIDE + 2 * sizeof(cplx) * M;   // line buffers + the staged twiddle and filter tables
};
constexpr unsigned kBsOob = 0xffffff00u;      // a byte offset no buffer descriptor of this engine covers (make_rsrc: 2 GiB)

// kinds of line (template parameter KIND)
constexpr int BS_CPLX = 0;      // complex lines along x (axis 0) or y (axis 1) of the internal half-spectrum layout, in place
constexpr int BS_R2C = 1;       // real rows -> half spectrum along z
constexpr int BS_C2R = 2;       // half spectrum -> real rows along z (times `scale`)

__device__ __forceinline__ cplx cmul_cj(cplx a, cplx b) {      // a * conj(b)
#if defined(OFDFT_REAL_F32) && OFDFT_F32_PK
    const v2f_t t = (v2f_t){a.x, a.x} * (v2f_t){b.x, -b.y};      // a.x (b.x, -b.y) + a.y (b.y, b.x)
    return v2c(__builtin_elementwise_fma((v2f_t){a.y, a.y}, (v2f_t){b.y, b.x}, t));
#else
    return mkc(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
// the convolution itself: v (element j + P q in slot q, zero beyond the line) -> its N-point DFT, same slots.  The chirp (wch)
// and the filter spectrum (fl_l, in LDS) are in their FORWARD form; the inverse transform conjugates them where they are used
// (PRE / POST = false skip the chirp multiplication before / after the convolution: in the fused x pass the forward transform's
// post-multiplication w_k and the inverse transform's pre-multiplication conj(w_k) cancel, |w_k| = 1, around a mix that is
// diagonal in k)
template <class PL, int M, bool INV, bool PRE = true, bool POST = true>
__device__ __forceinline__ void chirpz_line(cplx (&v)[PL::E], const cplx (&wch)[(PL::E + 1) / 2], int j, real* mine,
                                            const cplx* tw_l, const cplx* fl_l, int lx) {
    constexpr int P = PL::P, E = PL::E, EH = (E + 1) / 2;
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < EH; ++q) v[q] = INV ? cmul_cj(v[q], wch[q]) : cmul(v[q], wch[q]);       // (slots without an element hold zero)
    }
    StageP<PL, 0, 1, false, true, OFDFT_BS_TWTAB != 0>::run(v, j, mine, tw_l, lx);
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = INV ? cmul_cj(v[q], fl_l[j + P * q]) : cmul(v[q], fl_l[j + P * q]);
    exchange_sync<true>();
    StageP<PL, 0, 1, true, true, OFDFT_BS_TWTAB != 0>::run(v, j, mine, tw_l, lx);
    if constexpr (POST) {
#pragma unroll
        for (int q = 0; q < EH; ++q) v[q] = INV ? cmul_cj(v[q], wch[q]) : cmul(v[q], wch[q]);
    }
}

template <int M, int KIND, bool INV>
__global__ __launch_bounds__((BsCfg<M, (KIND != BS_CPLX ? 1 : 0)>::TPB))
__attribute__((amdgpu_waves_per_eu(BsCfg<M, (KIND != BS_CPLX ? 1 : 0)>::WAVES, BsCfg<M, (KIND != BS_CPLX ? 1 : 0)>::WAVES))) void bluestein_kernel(BsIo io, SpecGeom g, BsArgs b,
                                                                  const cplx* __restrict__ chirp,   // w_n, n < N
                                                                  const cplx* __restrict__ filt,    // FFT_M(b) / M
                                                                  const cplx* __restrict__ twM) {
    using Cfg = BsCfg<M, (KIND != BS_CPLX ? 1 : 0)>;
    using PL = typename Cfg::PL;
    constexpr int P = Cfg::P, E = Cfg::E, LPW = Cfg::LPW, TPB = Cfg::TPB;
    constexpr int EH = (E + 1) / 2;           // slots that can hold an element e = j + P q < N <= (M + 1) / 2
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    cplx* __restrict__ spec = io.spec[0];
    const real* __restrict__ rin = io.rin[0];
    real* __restrict__ rout = io.rout[0];
    int prep = io.prep[0];
    real pe = io.pe[0];
#pragma unroll
    for (int a = 1; a < kBsBatch; ++a)          // (select without dynamic indexing into the kernel arguments)
        if ((int)blockIdx.y == a) {
            spec = io.spec[a];
            rin = io.rin[a];
            rout = io.rout[a];
            prep = io.prep[a];
            pe = io.pe[a];
        }
    const int j = tid % P, l = tid / P;
    const int N = b.N;
    // ---- once per workgroup: twiddles and the filter spectrum into LDS, this lane's chirp values
    // into registers (slot q of lane j holds element j + P q of EVERY line).  The filter used to be requested from global
    // memory between the two transforms -- an L2 round trip in the middle of each line's dependent chain.
    cplx* tw_l = reinterpret_cast<cplx*>(lds + LPW * Cfg::STRIDE);
    cplx* fl_l = tw_l + M;
    for (int i = tid; i < M; i += TPB) {
        tw_l[i] = buf_load_c(twM, (unsigned)i * kCB);
        fl_l[i] = buf_load_c(filt, (unsigned)i * kCB);
    }
    cplx wch[EH];
#pragma unroll
    for (int q = 0; q < EH; ++q) {
        const int e = j + P * q;
        wch[q] = buf_load_c(chirp, e < N ? (unsigned)e * kCB : kBsOob);
    }
    real* mine = lds + l * Cfg::STRIDE;
    __syncthreads();
    // ---- persistent workgroups: tiles of LPW lines, no workgroup barrier inside the loop
    const long long ntiles = (b.nlines + LPW - 1) / LPW;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // complex lines (OFDFT_BS_TRANSPOSE): a thread LOADS and STORES for line `lsel`, element es + P q -- consecutive lanes take
        // consecutive lines (kz-fastest), so one instruction of a wave covers whole 128-B runs of the block-8 layout -- and hands the
        // elements to the line's owner wave through that wave's line buffer (N <= M / 2 complex entries, slot e ^ line so that the
        // 8 or 16 lanes of a write group hit distinct banks).  With the owner lanes loading their own line every cache line was
        // requested by 8 waves, 16 bytes each (x pass 2.2 TB/s, y pass 3.1 TB/s at 255^3).
        constexpr bool TR = KIND == BS_CPLX && OFDFT_BS_TRANSPOSE != 0 && P >= OFDFT_BS_TRANSPOSE_MINP;
        constexpr int XMASK = (M / 2 >= 16) ? ((sizeof(real) == 4 && LPW >= 16) ? 15 : 7) : 0;
        const int lg = TR ? tid % LPW : l, jg = TR ? tid / LPW : j;      // line and lane slot of the global accesses
        const long long L = tile * LPW + lg;
        const bool valid = L < b.nlines;
        // element offsets (bytes, 32 bit: the engine takes this path up to 512 points per axis, 1.1 GB per spectrum); complex
        // lines are enumerated kz-fastest: the waves of a workgroup take the kz neighbours of one 128-B run
        unsigned base = 0, stride = 0;              // BS_CPLX: first element and element stride of this lane's line
        // rows: TWO real rows per complex transform, z = a + i b (rows 2 L and 2 L + 1: the convolution costs the same for a
        // complex line, so pairing halves the z passes; the two spectra are separated with one more wave-local exchange)
        const unsigned row = (unsigned)(2 * L);     // x n1 + y of the first row
        const bool valid1 = 2 * L + 1 < g.nrows;
        if (KIND == BS_CPLX) {
            const int kz = (int)(L % g.nzc), r = (int)(L / g.nzc);       // r = y (x lines) or x (y lines)
            const unsigned rowbase = b.axis == 0 ? (unsigned)r : (unsigned)r * (unsigned)g.n1;
            const unsigned rowstep = b.axis == 0 ? (unsigned)g.n1 : 1u;
            if (kz < g.nzm) {
                base = (((unsigned)(kz >> 3) * (unsigned)g.nrows + rowbase) * 8u + (unsigned)(kz & 7)) * kCB;
                stride = rowstep * 8u * kCB;
            } else {
                base = ((unsigned)g.main_count + (unsigned)(kz - g.nzm) * (unsigned)g.nrows + rowbase) * kCB;
                stride = rowstep * kCB;
            }
        }
        cplx v[E];
        unsigned off[EH];
#pragma unroll
        for (int q = 0; q < EH; ++q) {
            const int e = jg + P * q;
            const bool in = valid && e < N;
            if (KIND == BS_CPLX) {
                off[q] = in ? base + (unsigned)e * stride : kBsOob;
                v[q] = buf_load_c(spec, off[q]);
            } else if (KIND == BS_R2C) {
                off[q] = in ? (row * (unsigned)g.n2 + (unsigned)e) * (unsigned)sizeof(real) : kBsOob;
                v[q] = mkc(buf_load_d(rin, off[q]), buf_load_d(rin, in && valid1 ? off[q] + (unsigned)g.n2 * (unsigned)sizeof(real) : kBsOob));
                if (prep != BS_PREP_NONE)       // (uniform; elements of the zero padding and of a missing second row stay zero)
                    v[q] = mkc(in ? bs_prep(v[q].x, prep, pe, io.pnref) : (real)0.0, (in && valid1) ? bs_prep(v[q].y, prep, pe, io.pnref) : (real)0.0);
            } else {            // rebuild the Hermitian lines; imaginary parts of k = 0 (and Nyquist) are ignored like irfftn
                const int k = (e < g.nzc) ? e : N - e;
                const unsigned idx = k < g.nzm ? ((unsigned)(k >> 3) * (unsigned)g.nrows + row) * 8u + (unsigned)(k & 7)
                                               : (unsigned)g.main_count + (unsigned)(k - g.nzm) * (unsigned)g.nrows + row;
                const unsigned step = k < g.nzm ? 8u : 1u;           // the next row's element
                cplx a = buf_load_c(spec, in ? idx * kCB : kBsOob);
                cplx bb = buf_load_c(spec, in && valid1 ? (idx + step) * kCB : kBsOob);
                if (e >= g.nzc) { a.y = -a.y; bb.y = -bb.y; }
                if (e == 0 || 2 * e == N) { a.y = 0.0; bb.y = 0.0; }
                v[q] = mkc(a.x - bb.y, a.y + bb.x);                  // A + i B
                off[q] = in ? (row * (unsigned)g.n2 + (unsigned)e) * (unsigned)sizeof(real) : kBsOob;
            }
        }
        if (TR) {
            cplx* theirs = reinterpret_cast<cplx*>(lds + lg * Cfg::STRIDE);
#pragma unroll
            for (int q = 0; q < EH; ++q) theirs[(jg + P * q) ^ (lg & XMASK)] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EH; ++q) v[q] = reinterpret_cast<cplx*>(mine)[(j + P * q) ^ (l & XMASK)];
            exchange_sync<true>();
        }
#pragma unroll
        for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
        chirpz_line<PL, M, INV>(v, wch, j, mine, tw_l, fl_l, (l * Cfg::LMUL) & 31);
        if (TR) {
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) reinterpret_cast<cplx*>(mine)[(j + P * q) ^ (l & XMASK)] = v[q];
            __syncthreads();
            const cplx* theirs = reinterpret_cast<const cplx*>(lds + lg * Cfg::STRIDE);
#pragma unroll
            for (int q = 0; q < EH; ++q) v[q] = theirs[(jg + P * q) ^ (lg & XMASK)];
        }
        if (KIND == BS_R2C) {
            // Z = A + i B with A, B Hermitian: A_k = (Z_k + conj Z_{N-k}) / 2, B_k = (Z_k - conj Z_{N-k}) / 2i -- the partner
            // element comes through the line buffer (real parts, then imaginary parts; N <= M / 2 entries)
            cplx p[EH];
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q)
                if (j + P * q < N) mine[j + P * q] = v[q].x;
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                p[q].x = mine[e == 0 || e >= N ? 0 : N - e];
            }
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q)
                if (j + P * q < N) mine[j + P * q] = v[q].y;
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                p[q].y = mine[e == 0 || e >= N ? 0 : N - e];
            }
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                const bool in = valid && e < g.nzc;
                const unsigned idx = e < g.nzm ? ((unsigned)(e >> 3) * (unsigned)g.nrows + row) * 8u + (unsigned)(e & 7)
                                               : (unsigned)g.main_count + (unsigned)(e - g.nzm) * (unsigned)g.nrows + row;
                const unsigned step = e < g.nzm ? 8u : 1u;
                const cplx za = mkc((real)0.5 * (v[q].x + p[q].x), (real)0.5 * (v[q].y - p[q].y));
                const cplx zb = mkc((real)0.5 * (v[q].y + p[q].y), (real)-0.5 * (v[q].x - p[q].x));
                buf_store_c(spec, in ? idx * kCB : kBsOob, za);
                buf_store_c(spec, in && valid1 ? (idx + step) * kCB : kBsOob, zb);
            }
        } else {
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                if (KIND == BS_CPLX) {
                    buf_store_c(spec, off[q], v[q]);
                } else {
                    const real r0 = v[q].x * b.scale, r1 = v[q].y * b.scale;
                    const unsigned o1 = (off[q] != kBsOob && valid1) ? off[q] + (unsigned)g.n2 * (unsigned)sizeof(real) : kBsOob;
                    if constexpr (sizeof(real) == 8) {
                        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&r0), make_rsrc(rout), (int)off[q], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&r1), make_rsrc(rout), (int)o1, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned*>(&r0), make_rsrc(rout), (int)off[q], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned*>(&r1), make_rsrc(rout), (int)o1, 0, 0);
                    }
                }
            }
        }
        if (TR) __syncthreads();        // the line buffers are rewritten by other waves in the next tile
        else exchange_sync<true>();     // the line buffer is reused by the next tile
    }
}

// ---- forward-x, k-space mix, inverse-x of the unfused pipeline in ONE kernel (the chirp-z counterpart of xfused_kernel):
// NIN spectra that have been through their z and y passes -> NOUT spectra ready for the inverse y and z passes, out_o =
// sum_i c_oi(k) in_i with the coefficient functors of the fused x pass (pointwise_kernels.h: MixDensity, MixScale, MixDiv,
// MixWgc).  Per convolution this saves one x pass' load + store and the separate spectral multiply kernel (a read and a
// write of every spectrum involved); the four M-point transforms per line and spectrum pair remain.
struct BsMixIo { const cplx* in[4]; cplx* out[4]; };
template <int M, int NIN, int NOUT, class Mix>
__global__ __launch_bounds__((BsCfg<M, 2>::TPB))
__attribute__((amdgpu_waves_per_eu(BsCfg<M, 2>::WAVES, BsCfg<M, 2>::WAVES))) void bluestein_xmix_kernel(
    BsMixIo io, SpecGeom g, int N, long long nlines, const cplx* __restrict__ chirp, const cplx* __restrict__ filt,
    const cplx* __restrict__ twM, Mix mix) {
    using Cfg = BsCfg<M, 2>;
    using PL = typename Cfg::PL;
    constexpr int P = Cfg::P, E = Cfg::E, LPW = Cfg::LPW, TPB = Cfg::TPB;
    constexpr int EH = (E + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    const int j = tid % P, l = tid / P;
    cplx* tw_l = reinterpret_cast<cplx*>(lds + LPW * Cfg::STRIDE);
    cplx* fl_l = tw_l + M;
    for (int i = tid; i < M; i += TPB) {
        tw_l[i] = buf_load_c(twM, (unsigned)i * kCB);
        fl_l[i] = buf_load_c(filt, (unsigned)i * kCB);
    }
    cplx wch[EH];
#pragma unroll
    for (int q = 0; q < EH; ++q) {
        const int e = j + P * q;
        wch[q] = buf_load_c(chirp, e < N ? (unsigned)e * kCB : kBsOob);
    }
    real* mine = lds + l * Cfg::STRIDE;
    cplx* minec = reinterpret_cast<cplx*>(mine);
    constexpr int XMASK = (M / 2 >= 16) ? ((sizeof(real) == 4 && LPW >= 16) ? 15 : 7) : 0;
    const int lg = tid % LPW, jg = tid / LPW;       // line and lane slot of the global accesses (see bluestein_kernel)
    cplx* theirs = reinterpret_cast<cplx*>(lds + lg * Cfg::STRIDE);
    const int lx = (l * Cfg::LMUL) & 31;
    const long long ntiles = (nlines + LPW - 1) / LPW;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // x lines, enumerated kz-fastest: line L = (y, kz)
        unsigned off[EH];
        {
            const long long L = tile * LPW + lg;
            const int kz = (int)(L % g.nzc), y = (int)(L / g.nzc);
            unsigned base, stride;
            if (kz < g.nzm) {
                base = (((unsigned)(kz >> 3) * (unsigned)g.nrows + (unsigned)y) * 8u + (unsigned)(kz & 7)) * kCB;
                stride = (unsigned)g.n1 * 8u * kCB;
            } else {
                base = ((unsigned)g.main_count + (unsigned)(kz - g.nzm) * (unsigned)g.nrows + (unsigned)y) * kCB;
                stride = (unsigned)g.n1 * kCB;
            }
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = jg + P * q;
                off[q] = (L < nlines && e < N) ? base + (unsigned)e * stride : kBsOob;
            }
        }
        // the line this wave transforms (its k-space coordinates and table offsets for the mix)
        const long long Lo = tile * LPW + l;
        const bool valid_o = Lo < nlines;
        const int kzo = (int)(Lo % g.nzc), yo = (int)(Lo / g.nzc);
        cplx X[NIN][EH];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) v[q] = buf_load_c(io.in[i], off[q]);
            if (i) __syncthreads();             // the other waves are done with their line buffers
#pragma unroll
            for (int q = 0; q < EH; ++q) theirs[(jg + P * q) ^ (lg & XMASK)] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EH; ++q) v[q] = minec[(j + P * q) ^ (l & XMASK)];
            exchange_sync<true>();
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, false, true, false>(v, wch, j, mine, tw_l, fl_l, lx);
#pragma unroll
            for (int q = 0; q < EH; ++q) X[i][q] = v[q];
        }
        cplx Y[NOUT][EH];
#pragma unroll
        for (int q = 0; q < EH; ++q) {
            const int e = j + P * q;
            const bool in = valid_o && e < N;
            const unsigned idx = kzo < g.nzm ? ((unsigned)(kzo >> 3) * (unsigned)g.nrows + (unsigned)e * (unsigned)g.n1 + (unsigned)yo) * 8u + (unsigned)(kzo & 7)
                                             : (unsigned)g.main_count + (unsigned)(kzo - g.nzm) * (unsigned)g.nrows + (unsigned)e * (unsigned)g.n1 + (unsigned)yo;
            static_for<NOUT>([&](auto oc) {
                constexpr int O = decltype(oc)::value;
                cplx acc = mkc(0.0, 0.0);
                static_for<NIN>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (Mix::template present<O, I>()) {
                        const real cf = in ? mix.template coef<O, I>(e, yo, kzo, 0LL, idx) : (real)0.0;
                        if constexpr (Mix::template imag_oi<O, I>()) {
                            acc.x -= cf * X[I][q].y;
                            acc.y += cf * X[I][q].x;
                        } else {
                            acc.x += cf * X[I][q].x;
                            acc.y += cf * X[I][q].y;
                        }
                    }
                });
                Y[O][q] = acc;
            });
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) v[q] = Y[o][q];
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, true, false, true>(v, wch, j, mine, tw_l, fl_l, lx);
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) minec[(j + P * q) ^ (l & XMASK)] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EH; ++q) buf_store_c(io.out[o], off[q], theirs[(jg + P * q) ^ (lg & XMASK)]);
            __syncthreads();                    // before the line buffers are used again
        }
    }
}

// ---- small grids: the z rows and the y lines of one x plane in ONE kernel.  Grids of 30 ... 64 points per axis -- what the
// reference's ecut2shape gives its own test systems -- are bound by their chain of dependent launches (~30 per evaluation at 53^3,
// 8-12 us each), not by work; a (y, kz) plane of such a grid fits LDS (64 x 33 complex = 34 KB), so the forward half of a 3-D
// transform is: row pairs -> chirp-z along z -> plane in LDS -> chirp-z along y -> the spectrum (x still in real space), and the
// inverse half the reverse.  Both padded lengths must be the same M <= 128 (one plan, one twiddle table).
template <int M> struct BsZyCfg {
    using PL = typename BsPlanPick<M>::type;
    static constexpr int TPB = 512, P = PL::P, E = PL::E, G = TPB / P;      // G lines / row pairs in flight
    static constexpr int STRIDE = BsCfg<M, 0>::STRIDE, LMUL = BsCfg<M, 0>::LMUL;
    static constexpr size_t lds_bytes(int n1, int nzc) {
        return sizeof(real) * G * STRIDE + 3 * sizeof(cplx) * M + sizeof(cplx) * (size_t)n1 * (size_t)(nzc | 1);
    }
};
template <int M, bool INV>
__global__ __launch_bounds__(512) void bluestein_zy_kernel(BsIo io, SpecGeom g, real scale, const cplx* __restrict__ chirp_z,
                                                           const cplx* __restrict__ filt_z, const cplx* __restrict__ chirp_y,
                                                           const cplx* __restrict__ filt_y, const cplx* __restrict__ twM) {
    using Cfg = BsZyCfg<M>;
    using PL = typename Cfg::PL;
    constexpr int P = Cfg::P, E = Cfg::E, G = Cfg::G, TPB = Cfg::TPB, EH = (E + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) real lds[];
    const int tid = threadIdx.x;
    cplx* __restrict__ spec = io.spec[0];
    const real* __restrict__ rin = io.rin[0];
    real* __restrict__ rout = io.rout[0];
    int prep = io.prep[0];
    real pe = io.pe[0];
#pragma unroll
    for (int a = 1; a < kBsBatch; ++a)
        if ((int)blockIdx.y == a) {
            spec = io.spec[a];
            rin = io.rin[a];
            rout = io.rout[a];
            prep = io.prep[a];
            pe = io.pe[a];
        }
    const int j = tid % P, grp = tid / P;
    const int x = blockIdx.x, N1 = g.n1, N2 = g.n2, nzc = g.nzc, PS = nzc | 1;
    cplx* tw_l = reinterpret_cast<cplx*>(lds + G * Cfg::STRIDE);
    cplx* flz_l = tw_l + M;
    cplx* fly_l = flz_l + M;
    cplx* plane = fly_l + M;                       // [y][PS] complex
    for (int i = tid; i < M; i += TPB) {
        tw_l[i] = buf_load_c(twM, (unsigned)i * kCB);
        flz_l[i] = buf_load_c(filt_z, (unsigned)i * kCB);
        fly_l[i] = buf_load_c(filt_y, (unsigned)i * kCB);
    }
    cplx wz[EH], wy[EH];
#pragma unroll
    for (int q = 0; q < EH; ++q) {
        const int e = j + P * q;
        wz[q] = buf_load_c(chirp_z, e < N2 ? (unsigned)e * kCB : kBsOob);
        wy[q] = buf_load_c(chirp_y, e < N1 ? (unsigned)e * kCB : kBsOob);
    }
    real* mine = lds + grp * Cfg::STRIDE;
    const int lx = (grp * Cfg::LMUL) & 31;
    const unsigned rowx = (unsigned)x * (unsigned)N1;
    const int npair = (N1 + 1) / 2;
    __syncthreads();
    if (!INV) {
        // ---- z: row pairs (y0, y0 + 1) of this plane -> their half spectra, into the LDS plane
        for (int p = grp; p < npair; p += G) {
            const int y0 = 2 * p;
            const bool valid1 = y0 + 1 < N1;
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                const unsigned o = e < N2 ? ((rowx + (unsigned)y0) * (unsigned)N2 + (unsigned)e) * (unsigned)sizeof(real) : kBsOob;
                v[q] = mkc(buf_load_d(rin, o), buf_load_d(rin, (e < N2 && valid1) ? o + (unsigned)N2 * (unsigned)sizeof(real) : kBsOob));
                if (prep != BS_PREP_NONE)
                    v[q] = mkc(e < N2 ? bs_prep(v[q].x, prep, pe, io.pnref) : (real)0.0, (e < N2 && valid1) ? bs_prep(v[q].y, prep, pe, io.pnref) : (real)0.0);
            }
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, false>(v, wz, j, mine, tw_l, flz_l, lx);
            cplx pr[EH];
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q)
                if (j + P * q < N2) mine[j + P * q] = v[q].x;
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                pr[q].x = mine[e == 0 || e >= N2 ? 0 : N2 - e];
            }
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q)
                if (j + P * q < N2) mine[j + P * q] = v[q].y;
            exchange_sync<true>();
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                pr[q].y = mine[e == 0 || e >= N2 ? 0 : N2 - e];
            }
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                if (e < nzc) {
                    plane[y0 * PS + e] = mkc((real)0.5 * (v[q].x + pr[q].x), (real)0.5 * (v[q].y - pr[q].y));
                    if (valid1) plane[(y0 + 1) * PS + e] = mkc((real)0.5 * (v[q].y + pr[q].y), (real)-0.5 * (v[q].x - pr[q].x));
                }
            }
            exchange_sync<true>();
        }
        __syncthreads();
        // ---- y: lines (x, kz) of the plane -> the spectrum
        for (int kz = grp; kz < nzc; kz += G) {
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                v[q] = e < N1 ? plane[e * PS + kz] : mkc(0.0, 0.0);
            }
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, false>(v, wy, j, mine, tw_l, fly_l, lx);
            const unsigned kb = kz < g.nzm ? (unsigned)(kz >> 3) * (unsigned)g.nrows * 8u + (unsigned)(kz & 7)
                                           : (unsigned)g.main_count + (unsigned)(kz - g.nzm) * (unsigned)g.nrows;
            const unsigned ks = kz < g.nzm ? 8u : 1u;
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                buf_store_c(spec, e < N1 ? (kb + (rowx + (unsigned)e) * ks) * kCB : kBsOob, v[q]);
            }
            exchange_sync<true>();
        }
    } else {
        // ---- y inverse: lines (x, kz) of the spectrum -> the LDS plane
        for (int kz = grp; kz < nzc; kz += G) {
            const unsigned kb = kz < g.nzm ? (unsigned)(kz >> 3) * (unsigned)g.nrows * 8u + (unsigned)(kz & 7)
                                           : (unsigned)g.main_count + (unsigned)(kz - g.nzm) * (unsigned)g.nrows;
            const unsigned ks = kz < g.nzm ? 8u : 1u;
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                v[q] = buf_load_c(spec, e < N1 ? (kb + (rowx + (unsigned)e) * ks) * kCB : kBsOob);
            }
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, true>(v, wy, j, mine, tw_l, fly_l, lx);
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                if (e < N1) plane[e * PS + kz] = v[q];
            }
            exchange_sync<true>();
        }
        __syncthreads();
        // ---- z inverse: row pairs rebuilt as A + i B from the plane -> two real rows
        for (int p = grp; p < npair; p += G) {
            const int y0 = 2 * p;
            const bool valid1 = y0 + 1 < N1;
            cplx v[E];
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                cplx a = mkc(0.0, 0.0), bb = mkc(0.0, 0.0);
                if (e < N2) {
                    const int k = (e < nzc) ? e : N2 - e;
                    a = plane[y0 * PS + k];
                    if (valid1) bb = plane[(y0 + 1) * PS + k];
                    if (e >= nzc) { a.y = -a.y; bb.y = -bb.y; }
                    if (e == 0 || 2 * e == N2) { a.y = 0.0; bb.y = 0.0; }
                }
                v[q] = mkc(a.x - bb.y, a.y + bb.x);
            }
#pragma unroll
            for (int q = EH; q < E; ++q) v[q] = mkc(0.0, 0.0);
            chirpz_line<PL, M, true>(v, wz, j, mine, tw_l, flz_l, lx);
#pragma unroll
            for (int q = 0; q < EH; ++q) {
                const int e = j + P * q;
                const unsigned o = e < N2 ? ((rowx + (unsigned)y0) * (unsigned)N2 + (unsigned)e) * (unsigned)sizeof(real) : kBsOob;
                const unsigned o1 = (e < N2 && valid1) ? o + (unsigned)N2 * (unsigned)sizeof(real) : kBsOob;
                const real r0 = v[q].x * scale, r1 = v[q].y * scale;
                if constexpr (sizeof(real) == 8) {
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&r0), make_rsrc(rout), (int)o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&r1), make_rsrc(rout), (int)o1, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned*>(&r0), make_rsrc(rout), (int)o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned*>(&r1), make_rsrc(rout), (int)o1, 0, 0);
                }
            }
            exchange_sync<true>();
        }
    }
}

}  // namespace ofdft
