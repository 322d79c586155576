// Launchers of the fused x passes (xfused_kernel in fft_kernels.h, xw_kernel in xwave.h): included by xpass_a.hip and
// xpass_b.hip, which instantiate them for the mix functors of the pipelines.
#pragma once
#include "engine_ctx.h"
#include "xcross.h"

namespace eng {

template <int LEN, int NIN, int NOUT, class Mix>
int launch_xfused_t(ofdft_ctx* c, const XfIo& io, const Mix& mix, const XfLayout& lay, hipStream_t st, const char* nm) {
    constexpr int G = NIN > NOUT ? NIN : NOUT;
    using Cfg = XfCfg<LEN, G, NOUT>;
    cplx* tw;
    if (int rc = get_twiddle(c, LEN, &tw)) return rc;
    LineMap main, rem;
    SpecGeom gk = c->gx;
    if (lay.se_in) {
        const int nyl = c->xg.nyl;
        const int xnb = lay.xnb >= 0 ? lay.xnb : c->xg.nb, xnrem = lay.xnb >= 0 ? lay.xnrem : c->xg.nrem;   // this chunk's share
        main.d = nyl * 8; main.sb = nyl * 8; main.sl = 1; main.se = lay.se_in; main.nlines = xnb * nyl * 8;
        main.kz0 = lay.kz0;
        rem.d = nyl; rem.sb = nyl; rem.sl = 1; rem.se = lay.se_in; rem.nlines = xnrem * nyl;
        if (main.nlines == 0) main.d = 1;
        if (rem.nlines == 0) rem.d = 1;
        gk.main_count = (long long)xnb * nyl * 8;          // offset of the plane part inside a record
    } else {
        pass_maps(c, 0, main, rem);
    }
    main.lf = rem.lf = Cfg::LPW;
    int line0 = 0;
    if (lay.kb1 > lay.kb0 && !lay.se_in) {              // a range of kz blocks; the remainder planes ride with the last one
        const int per_block = c->gx.n1 * 8;
        line0 = lay.kb0 * per_block;
        main.nlines = lay.kb1 * per_block;
        if (lay.kb1 != c->gx.nzm / 8) rem.nlines = 0;
    }
    main.blk0 = line0 / Cfg::LPW;
    const int mb = (main.nlines - line0 + Cfg::LPW - 1) / Cfg::LPW, rb = (rem.nlines + Cfg::LPW - 1) / Cfg::LPW;
    OFDFT_LAUNCH(c, st, nm, (xfused_kernel<LEN, NIN, NOUT, Mix>), dim3(mb + rb), dim3(Cfg::TPB), Cfg::LDS, io, main, rem, mb,
                 gk, tw, mix, XfStride{lay.se_out, lay.tse});
    return 0;
}

// the wave-local form of the same pass (xwave.h): a line of every spectrum in the lanes of one wave, mix in registers
template <int LEN, int NIN, int NOUT, class Mix>
int launch_xw_t(ofdft_ctx* c, const XfIo& io, const Mix& mix, const XfLayout& lay, hipStream_t st, const char* nm) {
    using Cfg = XwCfg<LEN>;
    cplx* tw;
    if (int rc = get_twiddle(c, LEN, &tw)) return rc;
    LineMap main, rem;
    SpecGeom gk = c->gx;
    if (lay.se_in) {
        const int nyl = c->xg.nyl;
        const int xnb = lay.xnb >= 0 ? lay.xnb : c->xg.nb, xnrem = lay.xnb >= 0 ? lay.xnrem : c->xg.nrem;   // this chunk's share
        main.d = nyl * 8; main.sb = nyl * 8; main.sl = 1; main.se = lay.se_in; main.nlines = xnb * nyl * 8;
        main.kz0 = lay.kz0;
        rem.d = nyl; rem.sb = nyl; rem.sl = 1; rem.se = lay.se_in; rem.nlines = xnrem * nyl;
        if (main.nlines == 0) main.d = 1;
        if (rem.nlines == 0) rem.d = 1;
        gk.main_count = (long long)xnb * nyl * 8;          // offset of the plane part inside a record
    } else {
        pass_maps(c, 0, main, rem);
    }
    main.lf = rem.lf = Cfg::LPB;
    int line0 = 0;
    if (lay.kb1 > lay.kb0 && !lay.se_in) {              // a range of kz blocks; the remainder planes ride with the last one
        const int per_block = c->gx.n1 * 8;
        if (per_block % Cfg::LPB) return fail(c, OFDFT_EINVAL, "kz-range x pass needs whole workgroups per kz block");
        line0 = lay.kb0 * per_block;
        main.nlines = lay.kb1 * per_block;
        if (lay.kb1 != c->gx.nzm / 8) rem.nlines = 0;
    }
    main.blk0 = line0 / Cfg::LPB;
    const int mb = (main.nlines - line0 + Cfg::LPB - 1) / Cfg::LPB, rb = (rem.nlines + Cfg::LPB - 1) / Cfg::LPB;
    OFDFT_LAUNCH(c, st, nm, (xw_kernel<LEN, NIN, NOUT, Mix>), dim3(mb + rb), dim3(Cfg::TPB), Cfg::LDS, io, main, rem, mb, gk,
                 (const cplx*)tw, mix, XfStride{lay.se_out, lay.tse});
    return 0;
}

// the cross-wave form (xcross.h): whole runs of memory-adjacent lines per access, radix-A step across the waves of a workgroup
template <int LEN, int NIN, int NOUT, class Mix>
int launch_xc_t(ofdft_ctx* c, const XfIo& io, const Mix& mix, const XfLayout& lay, hipStream_t st, const char* nm) {
    using Cfg = XcCfg<LEN>;
    cplx* tw;
    if (int rc = get_twiddle(c, LEN, &tw)) return rc;
    LineMap main, rem;
    SpecGeom gk = c->gx;
    if (lay.se_in) {
        const int nyl = c->xg.nyl;
        const int xnb = lay.xnb >= 0 ? lay.xnb : c->xg.nb, xnrem = lay.xnb >= 0 ? lay.xnrem : c->xg.nrem;   // this chunk's share
        main.d = nyl * 8; main.sb = nyl * 8; main.sl = 1; main.se = lay.se_in; main.nlines = xnb * nyl * 8;
        main.kz0 = lay.kz0;
        rem.d = nyl; rem.sb = nyl; rem.sl = 1; rem.se = lay.se_in; rem.nlines = xnrem * nyl;
        if (main.nlines == 0) main.d = 1;
        if (rem.nlines == 0) rem.d = 1;
        gk.main_count = (long long)xnb * nyl * 8;          // offset of the plane part inside a record
    } else {
        pass_maps(c, 0, main, rem);
    }
    main.lf = rem.lf = Cfg::LPB;
    if constexpr (Cfg::NL > 1) {          // lanes own groups of memory-adjacent lines: whole groups only, else the wave-local kernel
        if (main.sl != 1 || rem.sl != 1 || main.d % Cfg::NL || rem.d % Cfg::NL || main.nlines % Cfg::NL || rem.nlines % Cfg::NL)
        {
            if constexpr (LEN <= 512) return launch_xw_t<LEN, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            else return launch_xfused_t<LEN, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        }
    }
    int line0 = 0;
    if (lay.kb1 > lay.kb0 && !lay.se_in) {              // a range of kz blocks; the remainder planes ride with the last one
        const int per_block = c->gx.n1 * 8;
        if (per_block % Cfg::LPB) return fail(c, OFDFT_EINVAL, "kz-range x pass needs whole workgroups per kz block");
        line0 = lay.kb0 * per_block;
        main.nlines = lay.kb1 * per_block;
        if (lay.kb1 != c->gx.nzm / 8) rem.nlines = 0;
    }
    main.blk0 = line0 / Cfg::LPB;
    const int mb = (main.nlines - line0 + Cfg::LPB - 1) / Cfg::LPB, rb = (rem.nlines + Cfg::LPB - 1) / Cfg::LPB;
    constexpr size_t lds = Cfg::lds_bytes(xc_one_buffer<LEN, NIN, NOUT>());
    if constexpr (lds > 64 * 1024) {          // more dynamic LDS than the default limit: declared once per kernel and device
        static bool declared[64] = {};
        const int dv = c->device & 63;
        if (!declared[dv]) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&xc_kernel<LEN, NIN, NOUT, Mix>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            declared[dv] = true;
        }
    }
    OFDFT_LAUNCH(c, st, nm, (xc_kernel<LEN, NIN, NOUT, Mix>), dim3(mb + rb), dim3(Cfg::TPB), lds, io, main, rem, mb, gk,
                 (const cplx*)tw, mix, XfStride{lay.se_out, lay.tse});
    return 0;
}

// does the cross-wave kernel take this pass? (OFDFT_OPT_XWAVE, measured choices: see xfused below)
template <int NIN, int NOUT> bool xc_serves(const ofdft_ctx* c) {
    constexpr bool xc_1024_f32 = sizeof(real) == 4;
    const bool xc_len = (c->n0g == 256 && !(sizeof(real) == 4 && NIN + NOUT == 2)) || c->n0g == 512;
    const bool len_ok = c->n0g == 128 || c->n0g == 256 || c->n0g == 512 || c->n0g == 1024;
    return len_ok && (c->use_xwave == 5 || (c->use_xwave == 6 && NIN + NOUT >= 3) ||
                      (c->use_xwave == 1 && (xc_len || (c->n0g == 1024 && (NIN + NOUT >= 3 || xc_1024_f32)))));
}

// forward-x, k-space mix, inverse-x in one pass over NIN input / NOUT output spectra
template <int NIN, int NOUT, class Mix>
int xfused(ofdft_ctx* c, const XfIo& io, const Mix& mix, hipStream_t st, const char* nm, const XfLayout& lay) {
    // measured at 256^3 (A/B on one box): 3 -> 3 (WGC99) 0.55 -> 0.53 ms, 1 -> 2 0.112 -> 0.097 ms, but 1 -> 1 0.062 -> 0.079 ms
    // (at 512 points per line the wave-local kernel -- one line per wave: 16 useful bytes per cache line and load -- costs 1.5 x its
    // 256-point self per grid point, tools/shape_probe.py, and the group-parallel kernel is faster ALONE, 5.4 vs 6.3 ms for the WGC99
    // pair at 512^3; with the two chains overlapped the evaluation is nevertheless 3-4 % faster with the wave-local one, 24.6-25.0 vs
    // 25.8-25.9 ms in two alternations on one box: option value 4 selects the group-parallel kernel at 512 for A/B)
    // cross-wave kernel (xcross.h; round 4).  Measured on one box, WGC99 pair / 1 -> 2 / 1 -> 1 passes, us per evaluation:
    //   256^3: 537 -> 405 / 103 -> 92 / 61-69 -> 58-62;  512^3: 6200 -> 3800 / 1180 -> 920 / 570-600 -> 535-560;
    //   1024 x 128 x 128: 920 -> 670 / 197 -> 140 but 90-94 -> 101-106 for the 1 -> 1 passes;  128^3: neutral to 3 % slower.
    // option 1 (default): 256- and 512-point lines always, 1024-point lines for passes over >= 3 spectra;
    // 5 = wherever it exists (128..1024), 6 = passes over >= 3 spectra only, 7 = the round-3 choice (no cross-wave kernel)
    // (fp32 build, 256-point lines: the 1 -> 1 passes measured 36-37 us in the group-parallel kernel against 39-40 here)
    // (fp32 build, 1024-point lines -- config 5 -- with two lines per lane: every pass, 1 -> 1 included: 205-220 us against 264-272 in
    // the group-parallel kernel at 1024 x 256 x 256, the Lindhard mix 5.8 against 5.9 ms at 1024^3; profiles/r04_x_stride_probe.jsonl)
    if (xc_serves<NIN, NOUT>(c)) {
        switch (c->n0g) {
            case 128: return launch_xc_t<128, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 256: return launch_xc_t<256, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 512: return launch_xc_t<512, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 1024: return launch_xc_t<1024, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        }
    }
    if (c->use_xwave == 2 || ((c->use_xwave == 1 || c->use_xwave >= 5 || (c->use_xwave == 4 && c->n0g < 512)) && NIN + NOUT >= 3)) {
        switch (c->n0g) {
            case 8: return launch_xw_t<8, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 16: return launch_xw_t<16, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 32: return launch_xw_t<32, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 64: return launch_xw_t<64, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 128: return launch_xw_t<128, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 256: return launch_xw_t<256, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
            case 512: return launch_xw_t<512, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        }
    }
    switch (c->n0g) {
        case 8: return launch_xfused_t<8, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 16: return launch_xfused_t<16, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 32: return launch_xfused_t<32, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 64: return launch_xfused_t<64, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 128: return launch_xfused_t<128, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 256: return launch_xfused_t<256, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 512: return launch_xfused_t<512, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        case 1024: return launch_xfused_t<1024, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
#define X(L) case L: return launch_xfused_t<L, NIN, NOUT, Mix>(c, io, mix, lay, st, nm);
        OFDFT_MIXED_LINES(X)
#undef X
    }
    return fail(c, OFDFT_EINVAL, "unsupported fast FFT length %d", c->n0g);
}

}  // namespace eng
