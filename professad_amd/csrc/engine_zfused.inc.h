// The z-fused pipeline of the engine (five stages, two chains; single- and multi-GPU) and the launchers of the z
// kernels.  Included once by engine.hip inside its anonymous namespace (one translation unit).
// ---------------------------------------------------------------------------------- z-fused pipeline (zpass.h)
void energies_from_sums(const ofdft_ctx* c, const double* sums, const double* pbe_sums, double* E_terms, double* vn_int) {
    const unsigned mask = c->mask;
    const double dV = c->dV;
    if (mask & OFDFT_ION_ELECTRON) E_terms[0] = sums[0] * dV;
    if (mask & OFDFT_HARTREE) E_terms[1] = sums[1] * dV;
    if (mask & OFDFT_TF) E_terms[2] = sums[2] * dV;
    if (mask & OFDFT_VW) E_terms[3] = sums[3] * dV;
    if (mask & OFDFT_WT_NL) E_terms[4] = sums[4] * dV;
    if (mask & OFDFT_WGC99_NL) E_terms[5] = sums[5] * dV;
    if (mask & OFDFT_LDA_X) E_terms[6] = sums[6] * dV;
    int nc = 0;
    for (int b = 7; b <= 9; ++b) nc += (mask >> b) & 1;
    for (int b = 7; b <= 9; ++b)
        if ((mask >> b) & 1) E_terms[b] = sums[7] * dV / nc;
    if (mask & OFDFT_PBE_X) E_terms[10] = pbe_sums[0] * dV;
    if (mask & OFDFT_PBE_C) E_terms[11] = pbe_sums[1] * dV;
    if (mask & OFDFT_GGA_K) E_terms[12] = pbe_sums[2] * dV;
    if (mask & OFDFT_VWGTF) E_terms[13] = sums[9] * dV;
    *vn_int = sums[8] * dV;
}

// Pipeline with every real-space intermediate kept on chip: z kernels compute their inputs from chi|n on the
// fly and consume the convolution results straight out of the inverse transform.  It is written as five
// stages separated by the four points where the spectra change between the x-slab geometry (z, y passes) and
// the x-pass geometry: on one GPU the two coincide and the stages simply run back to back; on several GPUs
// each boundary is one all-to-all over the listed arrays (the host does the collective, see ofdft_dist_*).
//   stage 1  z-forward (+pointwise pre-ops) and y-forward of every input spectrum          -> exchange
//   stage 2  fused x passes (Hartree, gradient, Laplacian, Lindhard / WGC99 mixing)         -> exchange
//   stage 3  y-inverse of the results; PBE mid stage on chip; y-forward of the flux          -> exchange
//   stage 4  fused x pass of the divergence                                                  -> exchange
//   stage 5  y-inverse of the divergence; combine kernel (potential + energy integrands)
struct ZRun {
    DenSrc ds{};
    double nel = 0.0;
    const real* vext = nullptr;
    real* v_out = nullptr;
    ZCombineArgs za{};
    double pbe_sums[kPbeScalars] = {0.0, 0.0, 0.0};
    bool has_h = false, has_g = false, has_vw = false, has_wt = false, has_wgc = false;
    cplx *s_n = nullptr, *s_s = nullptr, *s_vh = nullptr, *s_g[3] = {nullptr, nullptr, nullptr};
    cplx *s_b = nullptr, *s_a = nullptr, *sw[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    real* dfdn = nullptr;
    double wt_pref = 0.0, wt_kf = 1.0;
    // The evaluation is two independent chains that meet only in the combine kernel:
    //   chain 0: density spectrum -> Hartree, grad n -> PBE -> divergence;  sqrt(n) -> Laplacian (vW)
    //   chain 1: the nonlocal KEDF (Wang-Teter powers or the six WGC99 spectra)
    // xlist[k] = spectra of chain k that cross the next geometry boundary (= one all-to-all on several GPUs)
    std::vector<cplx*> xlist[2];
    bool gsplit = false;           // split-derivative form of the GGA chain (only D_a visits the x pass)
    bool lapl = false;             // Laplacian-dependent Pauli-Gaussian member: lap n in, lap(df/dL) out ride the same chain
    cplx* s_l = nullptr;
    real* dzn = nullptr;
    bool wgc_yinv_done = false;    // kz-chunked form: the y-inverse of the WGC99 results already ran next to the x pass
    bool wgc_split = false;        // the WGC99 potential was formed by zi_wgc_kernel (za.v_part)
    bool closure = false;          // single-GPU closure evaluation: only chi.grad leaves the call, so v may stay in two parts
    bool setup_done = false;       // zsetup ran for the evaluation being enqueued (zstage1 of chain 0 consumes it)
    bool late_join = false;        // host-bound sums of a closure evaluation: the share of sum(v n) of the deferred part is added by chi_grad / the host
    bool vpart_deferred = false;   // ... and does: zi_combine does not wait for zi_wgc, chi_grad adds v_part (zstage5)
    std::vector<cplx*> deferred;   // x-chunked pipeline: spectra whose y-inverse runs inside the combine loop
    int stage[2] = {0, 0};
    // kz-chunked exchange (ofdft_dist_step): last (step, chunk) of each chain; what stage 1 sent (read again by every chunk of stage 2)
    int step[2] = {0, 0}, step_chunk[2] = {0, 0};
    std::vector<cplx*> in_list[2];
    int combine_blocks = 0, pbe_blocks = 0;
    hipStream_t sb = nullptr;      // stream of the nonlocal-KEDF chain (== the main stream unless forked)
    hipStream_t sc = nullptr;      // second side stream: vW chain and the second half of the WGC99 chain
    bool forked = false;
};

}  // namespace
struct ofdft_zrun_holder { ZRun r; };
namespace {

ZRun& zrun(ofdft_ctx* c);

// ---- all-to-all buffers of the slab-decomposed path, one pair per chain (both directions reuse the pair):
// chain 0 carries at most 5 spectra (Hartree, grad n, vW leaving stage 2), chain 1 at most 8 (2 Wang-Teter + 6 WGC99)
// x chunks for a loop whose working set is `narr` spectra: the option value is the count for six spectra; more
// spectra -> proportionally more chunks, so that a chunk's working set stays the same share of the Infinity Cache.
// Every chunk must be whole workgroups of every z kernel (at most 256 rows each) -> powers of two that divide n0.
int chunks_for(const ofdft_ctx* c, int narr, int which = 15) {
    if (c->nranks > 1 || c->xchunks == 1 || !(c->xchunk_mask & which) || !all_pow2(c)) return 1;
    if (which == 8 && wts_active(c)) return 1;        // the two-pass combine of the stabilised WT-style functional is not chunked
    // automatic: about 100 MB of spectra per chunk (measured best at 256^3: 8 chunks for the six WGC99 spectra)
    int want = c->xchunks > 1 ? (c->xchunks * narr + 5) / 6
                              : (int)std::min<double>(64.0, (double)narr * sizeof(cplx) * (double)c->g.total / 100e6);
    int n = 1;
    while (n * 2 <= want && c->n0 % (n * 2) == 0 && ((long long)(c->n0 / (n * 2)) * c->n1) % 256 == 0) n *= 2;
    return n;
}

// Stage 1: z-forward (with the pointwise pre-ops) and y-forward of the chain's input spectra.
// (xk: chunk of the kz-chunked exchange -- the z kernels run with chunk 0, every call y-transforms its chunk into the send
// buffer; -1 = all chunks)
// host-side state of an evaluation that both chains read (term flags, the combine kernel's arguments): first thing of stage 1
// of chain 0 -- or, when the nonlocal chain's first kernels are enqueued first (zfused_enqueue), before either
#ifndef OFDFT_YDERIV_FWD
#define OFDFT_YDERIV_FWD 1
#endif
int zsetup(ofdft_ctx* c) {
    ZRun& r = zrun(c);
    const unsigned mask = c->mask;
    r.has_h = mask & OFDFT_HARTREE;
    r.has_g = mask & kGgaAny;
    r.has_vw = mask & OFDFT_VW;
    r.has_wt = mask & OFDFT_WT_NL;
    r.has_wgc = mask & OFDFT_WGC99_NL;
    r.za = ZCombineArgs{};
    r.za.ds = r.ds;
    r.za.vext = r.vext;
    r.za.v_out = r.v_out;
    r.za.mask = mask;
    r.za.inv_n = 1.0 / (double)c->npts_g;
    r.za.gtf_kind = (int)c->params[OFDFT_P_VWGTF_KIND];
    r.za.gtf_inv_n0 = (mask & OFDFT_VWGTF) ? c->vol / (double)std::llround(r.nel) : 0.0;   // functionals.py:268-270
    r.pbe_sums[0] = r.pbe_sums[1] = r.pbe_sums[2] = 0.0;
    r.wgc_yinv_done = false;
    r.s_g[0] = r.s_g[1] = r.s_g[2] = nullptr;
    r.s_n = r.s_s = r.s_vh = r.s_b = r.s_a = nullptr;
    if ((mask & OFDFT_ION_ELECTRON) && !r.vext) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");
    r.setup_done = true;
    return 0;
}

int zstage1(ofdft_ctx* c, hipStream_t st, int chain, int xk = -1) {
    ZRun& r = zrun(c);
    int rc;
    hipStream_t sb = r.forked ? r.sb : st, sc = r.forked ? r.sc : st;
    const bool dx = c->nranks > 1;
    std::vector<cplx*>& xl = r.xlist[chain];
    if (xk <= 0) {
    xl.clear();
    if (chain == 0) {
        if (!r.setup_done && (rc = zsetup(c))) return rc;
        r.setup_done = false;         // (consumed: the next evaluation sets up again)
        if (r.has_h || r.has_g)
            if ((rc = spec_ws(c, "zn", &r.s_n))) return rc;
        if (r.has_vw)
            if ((rc = spec_ws(c, "zs", &r.s_s))) return rc;
        if (r.s_n || r.s_s) {
            cplx* both[2];
            int nb = 0;
            if (r.s_n) both[nb++] = r.s_n;
            if (r.s_s) both[nb++] = r.s_s;
            r.gsplit = r.has_g && c->gga_split;
            r.lapl = r.gsplit && gga_needs_laplacian(c);
            r.s_l = nullptr;
            if (r.gsplit) {
                if ((rc = real_ws(c, "dzn", &r.dzn))) return rc;
                if ((rc = spec_ws(c, "zgx", &r.s_g[0]))) return rc;
                if ((rc = spec_ws(c, "zgy", &r.s_g[1]))) return rc;
                if (r.lapl && (rc = spec_ws(c, "zgl", &r.s_l))) return rc;
            }
            const int nch = r.gsplit ? 1 : chunks_for(c, nb, 1);
            if (nch > 1) {        // x-chunked: a chunk's spectra are y-transformed while still in the Infinity Cache
                for (int ch = 0; ch < nch; ++ch) {
                    if ((rc = launch_zf_density(c, r.ds, r.s_n, r.s_s, st, ch, nch))) return rc;
                    if ((rc = fast_axis_pass_multi<false>(c, 1, both, nb, st, ch * (c->n0 / nch), c->n0 / nch))) return rc;
                }
            } else if ((rc = launch_zf_density(c, r.ds, r.s_n, r.s_s, st, 0, 1, r.gsplit ? r.dzn : nullptr))) {
                return rc;
            }
            // D_b n from the (kz; y, x) spectrum before its in-place y-forward; scaled so that the consumer's 1/N fits
            // (one GPU: the y-forward of n^ rides in the same pass -- both read the same array; OFDFT_YDERIV_FWD=0: two passes)
            const bool yfwd = OFDFT_YDERIV_FWD && r.gsplit && !dx && nch == 1;
            if (r.gsplit && (rc = yderiv(c, r.s_n, r.s_g[1], (double)c->n0g, st, yfwd ? r.s_n : nullptr))) return rc;
            if (r.forked && r.s_s) {          // the vW chain continues on the second side stream
                HIP_TRY(c, hipEventRecord(c->ev_a, st));
                HIP_TRY(c, hipStreamWaitEvent(sc, c->ev_a, 0));
            }
            if (!dx && nch == 1 && r.s_n && !yfwd && (rc = fast_axis_pass<false>(c, 1, r.s_n, st))) return rc;
            if (!dx && nch == 1 && r.s_s && (rc = fast_axis_pass<false>(c, 1, r.s_s, sc))) return rc;
            if (r.s_n) xl.push_back(r.s_n);
            if (r.s_s) xl.push_back(r.s_s);
        }
    } else {
        if (r.has_wt) {
            const double al = c->params[OFDFT_P_WT_ALPHA], be = c->params[OFDFT_P_WT_BETA];
            const double nbar = r.nel / c->vol;                                  // functionals.py:646-647
            r.wt_kf = std::cbrt(3.0 * kPi * kPi * nbar);
            r.wt_pref = 5.0 / (9.0 * al * be * std::pow(nbar, al + be - kFiveThirds));
            if ((rc = spec_ws(c, "zwb", &r.s_b))) return rc;
            if (al != be && (rc = spec_ws(c, "zwa", &r.s_a))) return rc;
            PowersArgs pa{};
            pa.out[0] = r.s_b;
            pa.out[3] = r.s_a;
            pa.e0 = be;
            pa.e1 = al;
            if ((rc = launch_zf_powers(c, r.ds, pa, sb))) return rc;
            for (cplx* sp : {r.s_b, r.s_a}) {
                if (!sp) continue;
                if (!dx && (rc = fast_axis_pass<false>(c, 1, sp, sb))) return rc;
                xl.push_back(sp);
            }
            r.za.wt_alpha = al;
            r.za.wt_beta = be;
            r.za.wt_nbar_pa = std::pow(nbar, al);
            r.za.wt_is_56 = (al == kFiveSixths && be == kFiveSixths) ? 1 : 0;
        }
        if (r.has_wgc) {
            const double al = c->params[OFDFT_P_WGC_ALPHA], be = c->params[OFDFT_P_WGC_BETA];
            const long long nel_r = std::llround(r.nel);                         // functionals.py:952
            double nref;
            if ((rc = ensure_wgc_tables(c, nel_r, sb, &nref))) return rc;
            const char* wn[6] = {"zw0", "zw1", "zw2", "zw3", "zw4", "zw5"};
            PowersArgs pa{};
            for (int i = 0; i < 6; ++i) {
                if ((rc = spec_ws(c, wn[i], &r.sw[i]))) return rc;
                pa.out[i] = r.sw[i];
            }
            pa.e0 = be;
            pa.e1 = al;
            pa.nref = nref;
            pa.sum53 = (std::fabs(al + be - kFiveThirds) < 4e-16) ? 1 : 0;
            // x-chunked form: each chunk's six spectra (6 x C / nchunks) are y-transformed while still in the Infinity Cache
            const int nch = chunks_for(c, 6, 2);
            if (nch > 1) {
                for (int ch = 0; ch < nch; ++ch) {
                    if ((rc = launch_zf_powers(c, r.ds, pa, sb, ch, nch))) return rc;
                    if ((rc = fast_axis_pass_multi<false>(c, 1, r.sw, 6, sb, ch * (c->n0 / nch), c->n0 / nch))) return rc;
                }
            } else if ((rc = launch_zf_powers(c, r.ds, pa, sb))) {
                return rc;
            }
            if (r.forked) {                    // second half (P, Q, S) continues on the second side stream
                HIP_TRY(c, hipEventRecord(c->ev_b, sb));
                HIP_TRY(c, hipStreamWaitEvent(sc, c->ev_b, 0));
            }
            if (!dx && nch == 1 && c->ybatch) {       // a half's three y passes as ONE launch (grid.y = 3): more workgroups than the
                                                      // chip holds at once, so loads, transforms and stores of different tiles overlap
                if ((rc = fast_axis_pass_multi<false>(c, 1, r.sw, 3, sb))) return rc;
                if ((rc = fast_axis_pass_multi<false>(c, 1, r.sw + 3, 3, sc))) return rc;
            }
            for (int i = 0; i < 6; ++i) {
                if (!dx && nch == 1 && !c->ybatch && (rc = fast_axis_pass<false>(c, 1, r.sw[i], i < 3 ? sb : sc))) return rc;
                xl.push_back(r.sw[i]);
            }
            r.za.wgc_alpha = al;
            r.za.wgc_beta = be;
            r.za.nref = nref;
            r.za.wgc_sum_53 = pa.sum53;
        }
    }
    }   // xk <= 0
    if (dx && !xl.empty()) {        // the chain's y-forwards in one launch (per chunk), written in the exchange layout
        cplx *send, *recv;
        if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
        if ((rc = ypass_xchg<false>(c, xl, send, st, xk))) return rc;
    }
    r.stage[chain] = 1;
    return 0;
}

// Stage 2: the fused x passes (forward x, k-space mixing, inverse x).
int zstage2(ofdft_ctx* c, hipStream_t st, int chain, int xk = -1) {
    ZRun& r = zrun(c);
    int rc;
    hipStream_t sb = r.forked ? r.sb : st, sc = r.forked ? r.sc : st;
    // several ranks: the inputs sit in the chain's receive buffer (slot = position in stage 1's list) and the
    // outputs are written to its send buffer in the order they are listed here -- per kz chunk of the exchange layout
    const bool dx = c->nranks > 1;
    if (dx && xk == -1 && c->xc.n > 1) {
        for (int k = 0; k < c->xc.n; ++k)
            if ((rc = zstage2(c, st, chain, k))) return rc;
        return 0;
    }
    std::vector<cplx*>& xl = r.xlist[chain];
    if (xk <= 0) r.in_list[chain] = xl;           // what stage 1 sent: every chunk's pass reads the same slots
    const std::vector<cplx*>& in_list = r.in_list[chain];
    xl.clear();
    cplx *send = nullptr, *recv = nullptr;
    XfLayout lay{};
    const XcView xv = xc_view(c, xk < 0 ? 0 : xk);
    int nout = 0;
    if (dx) {
        if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
        nout = chain == 0 ? (r.has_h ? 1 : 0) + (r.has_g ? (r.gsplit ? (r.lapl ? 2 : 1) : 3) : 0) + (r.s_s ? 1 : 0)
                          : (r.s_b ? 1 : 0) + (r.s_a ? 1 : 0) + (r.has_wgc ? 6 : 0);
        lay = XfLayout{(long long)in_list.size() * xv.arr_sz, nout * xv.arr_sz, xv.arr_sz, 0, 0, xv.nb, xv.nrem, xv.kb0 * 8};
        recv += (long long)in_list.size() * xv.base1;         // this chunk's region of either buffer
        send += (long long)nout * xv.base1;
    }
    auto in_of = [&](cplx* arr) -> cplx* {
        if (!dx) return arr;
        for (size_t i = 0; i < in_list.size(); ++i)
            if (in_list[i] == arr) return recv + (long long)i * xv.arr_sz;
        return nullptr;
    };
    auto out_of = [&](cplx* arr) -> cplx* {       // also records the array as crossing the next boundary
        xl.push_back(arr);
        return dx ? send + (long long)(xl.size() - 1) * xv.arr_sz : arr;
    };
    if (chain == 0) {
        if (r.s_n) {
            XfIo io{};
            io.in[0] = in_of(r.s_n);
            int no = 0;
            if (r.has_h) {
                if ((rc = spec_ws(c, "zvh", &r.s_vh))) return rc;
                io.out[no++] = out_of(r.s_vh);
            }
            if (r.has_g && r.gsplit) {
                io.out[no++] = out_of(r.s_g[0]);          // (D_a n)^ only
                if (r.lapl) io.out[no++] = out_of(r.s_l); // -k^2 n^
            } else if (r.has_g) {
                const char* gn[3] = {"zgx", "zgy", "zgz"};
                for (int k = 0; k < 3; ++k) {
                    if ((rc = spec_ws(c, gn[k], &r.s_g[k]))) return rc;
                    io.out[no++] = out_of(r.s_g[k]);
                }
            }
            if (r.lapl && r.has_h) rc = xfused<1, 3>(c, io, MixDensityA<true, true>{c->kg}, st, "xfused_n", lay);
            else if (r.lapl) rc = xfused<1, 2>(c, io, MixDensityA<false, true>{c->kg}, st, "xfused_n", lay);
            else if (r.gsplit && r.has_h) rc = xfused<1, 2>(c, io, MixDensityA<true>{c->kg}, st, "xfused_n", lay);
            else if (r.gsplit) rc = xfused<1, 1>(c, io, MixDensityA<false>{c->kg}, st, "xfused_n", lay);
            else if (r.has_h && r.has_g) rc = xfused<1, 4>(c, io, MixDensity<true, true>{c->kg}, st, "xfused_n", lay);
            else if (r.has_h) rc = xfused<1, 1>(c, io, MixDensity<true, false>{c->kg}, st, "xfused_n", lay);
            else rc = xfused<1, 3>(c, io, MixDensity<false, true>{c->kg}, st, "xfused_n", lay);
            if (rc) return rc;
        }
        if (r.s_s) {
            XfIo io{};
            io.in[0] = in_of(r.s_s);
            io.out[0] = out_of(r.s_s);
            if ((rc = xfused<1, 1>(c, io, MixScale<SPEC_LAPLACE>{c->kg, 0.0, 0.0}, sc, "xfused_lap", lay))) return rc;
        }
    } else {
        if (r.has_wt) {
            const MixScale<SPEC_LINDHARD> lind{c->kg, (real)r.wt_pref, (real)(1.0 / (2.0 * r.wt_kf))};
            for (cplx* sp : {r.s_b, r.s_a}) {
                if (!sp) continue;
                XfIo io{};
                io.in[0] = in_of(sp);
                io.out[0] = out_of(sp);
                if ((rc = xfused<1, 1>(c, io, lind, sb, "xfused_lind", lay))) return rc;
            }
        }
        if (r.has_wgc) {
            const MixWgc mix = wgc_tab(c, dx ? xv.base1 : 0);    // (the table is chunk-major like the buffers)
            // kz-chunked form (one GPU): the fused x pass of a range of kz blocks is followed at once by the y-inverse
            // of the same range, which then reads the x pass' output from the Infinity Cache
            const int nb = c->g.nzm / 8;
            int nkz = (!dx && (c->xchunk_mask & 16) && c->xchunks != 1) ? chunks_for(c, 6, 16) : 1;
            while (nkz > 1 && nb % nkz) nkz >>= 1;
            r.wgc_yinv_done = nkz > 1;
            for (int half = 0; half < 2; ++half) {
                XfIo io{};
                for (int i = 0; i < 3; ++i) {
                    io.in[i] = in_of(r.sw[3 * half + i]);
                    io.out[i] = out_of(r.sw[3 * half + i]);
                }
                hipStream_t hs = half == 0 ? sb : sc;
                if (nkz == 1) {
                    if ((rc = xfused_wgc(c, io, mix, hs, "xfused_wgc", lay))) return rc;
                    continue;
                }
                for (int ch = 0; ch < nkz; ++ch) {
                    XfLayout lk = lay;
                    lk.kb0 = ch * (nb / nkz);
                    lk.kb1 = (ch + 1) * (nb / nkz);
                    if ((rc = xfused_wgc(c, io, mix, hs, "xfused_wgc", lk))) return rc;
                    if ((rc = fast_axis_pass_multi<true>(c, 1, r.sw + 3 * half, 3, hs, 0, 0, lk.kb0, lk.kb1))) return rc;
                }
            }
        }
    }
    r.stage[chain] = 2;
    return 0;
}

// Stage 3: y-inverse of what came back from the x passes (each completes one c2r except grad n); chain 0 then runs
// the PBE mid stage on chip and starts the three r2c of the flux.
// part 0: the whole stage; kz-chunked exchange: part 1 = the y-inverse of chunk xk only, part 2 = everything after the
// y-inverses (the row kernels with chunk 0; every call y-transforms its chunk of the flux into the send buffer)
int zstage3(ofdft_ctx* c, hipStream_t st, int chain, int part = 0, int xk = -1) {
    ZRun& r = zrun(c);
    int rc;
    hipStream_t sb = r.forked ? r.sb : st, sc = r.forked ? r.sc : st;
    const bool dx = c->nranks > 1;
    std::vector<cplx*>& xl = r.xlist[chain];
    cplx *send = nullptr, *recv = nullptr;
    if (dx && !xl.empty() && part != 2) {        // one launch (per chunk): the chain's y-inverses, read from the exchange layout
        if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
        if ((rc = ypass_xchg<true>(c, xl, recv, st, xk))) return rc;
    }
    if (part == 1) return 0;
    if (part == 2 && xk > 0) {                   // a later chunk of the flux: the y-forward into the send buffer only
        if (dx && chain == 0 && !xl.empty()) {
            if ((rc = dist_buffers(c, 0, &send, &recv))) return rc;
            if ((rc = ypass_xchg<false>(c, xl, send, st, xk))) return rc;
        }
        return 0;
    }
    const int yk = part == 2 ? 0 : -1;           // chunks the y-forwards below cover
    // x-chunked pipeline: the y-inverse of every spectrum the combine kernel consumes moves into the combine loop
    // (stage 5) and that of grad n into the PBE loop below, so the consumer reads the lines from the Infinity Cache
    const bool chunked = chunks_for(c, 6, 8) > 1, pbe_chunked = !r.gsplit && chunks_for(c, 6, 4) > 1;
    const bool wbatch = !dx && c->ybatch && chain == 1 && r.has_wgc && !r.wgc_yinv_done && !chunked && xl.size() >= 6;
    if (wbatch) {           // the six WGC99 results: one batched y-inverse per half (see stage 1)
        if ((rc = fast_axis_pass_multi<true>(c, 1, r.sw, 3, sb))) return rc;
        if ((rc = fast_axis_pass_multi<true>(c, 1, r.sw + 3, 3, sc))) return rc;
    }
    for (cplx* sp : xl) {
        const bool on_b = sp == r.s_b || sp == r.s_a || sp == r.sw[0] || sp == r.sw[1] || sp == r.sw[2];
        const bool on_c = sp == r.s_s || sp == r.sw[3] || sp == r.sw[4] || sp == r.sw[5];
        const bool is_g = sp == r.s_g[0] || sp == r.s_g[1] || sp == r.s_g[2] || (r.s_l && sp == r.s_l);
        const bool is_w = sp == r.sw[0] || sp == r.sw[1] || sp == r.sw[2] || sp == r.sw[3] || sp == r.sw[4] || sp == r.sw[5];
        if (is_w && (r.wgc_yinv_done || wbatch)) {
            // already y-inverted next to the x pass / by the batched launches above
        } else if (is_g ? pbe_chunked : chunked) {
            if (!is_g) r.deferred.push_back(sp);
        } else if (!dx && (rc = fast_axis_pass<true>(c, 1, sp, on_b ? sb : (on_c ? sc : st)))) {
            return rc;
        }
        if (!is_g) c->fft_count++;
    }
    xl.clear();
    if (chain == 1) {
        if (r.has_wt) {
            r.za.conv_b = r.s_b;
            r.za.conv_a = r.s_a;
        }
        if (r.has_wgc)
            for (int i = 0; i < 3; ++i) {
                r.za.u[i] = r.sw[i];
                r.za.gw[i] = r.sw[3 + i];
            }
        r.wgc_split = false;
        if (r.has_wgc && !chunked && c->split_combine && !graph_eligible(c)) {
            // both halves of the chain are done -> its part of the combine runs here, beside the other chain's PBE tail
            real* vp;
            acc_t* part2;
            int blocks = 0;
            if ((rc = real_ws(c, "vpart", &vp))) return rc;
            if ((rc = get_ws(c, "zwgc:part", sizeof(double) * 2 * (size_t)(c->partial_rows + kRedMidRows), (void**)&part2))) return rc;
            if (r.forked) {
                // (its own event: one event recorded on two different streams inside a stream capture crashes the runtime)
                HIP_TRY(c, hipEventRecord(c->ev_c, sc));
                HIP_TRY(c, hipStreamWaitEvent(sb, c->ev_c, 0));
            }
            if ((rc = launch_zi_wgc(c, r.za, vp, part2, &blocks, sb))) return rc;
            OFDFT_REDUCE(c, sb, (const acc_t*)part2, blocks, 2, c->d_scal + 2, c->h_partial + kNSums);        // [2] energy sum, [3] this part's sum(v n); host mirror
            r.za.v_part = vp;
            r.wgc_split = true;
            // closure evaluations leave v in two parts: the combine kernel then has nothing to wait for on this stream
            r.vpart_deferred = r.closure && r.forked && c->defer_vpart;
            r.za.v_part_deferred = r.vpart_deferred ? 1 : 0;
        }
        r.stage[1] = 3;
        return 0;
    }
    if (r.has_h) r.za.vh = r.s_vh;
    if (r.s_s) r.za.lap = r.s_s;
    if (r.has_g && r.gsplit) {
        // split-derivative form: A = (D_a n) came back from the x pass and was y-inverted above, B = (D_b n) is local
        if ((rc = real_ws(c, "dfdn", &r.dfdn))) return rc;
        if ((rc = launch_zpbe2(c, r.ds, r.s_g[0], r.s_g[1], r.dzn, r.dfdn, r.za.inv_n, &r.pbe_blocks, st, r.lapl ? r.s_l : nullptr)))
            return rc;
        OFDFT_REDUCE(c, st, c->d_partial, r.pbe_blocks, kPbeScalars, c->d_reduced + kCombineScalars, c->h_partial + kCombineScalars);
        // D_b G_b in one y pass, in place (scaled like B); only G_a goes on to the x pass
        if ((rc = yderiv(c, r.s_g[1], r.s_g[1], (double)c->n0g, st))) return rc;
        if (!dx && (rc = fast_axis_pass<false>(c, 1, r.s_g[0], st))) return rc;
        xl.push_back(r.s_g[0]);
        if (r.lapl) {             // (df/dL)^ goes on to the x pass beside G_a
            if (!dx && (rc = fast_axis_pass<false>(c, 1, r.s_l, st))) return rc;
            xl.push_back(r.s_l);
        }
        if (dx) {
            if ((rc = dist_buffers(c, 0, &send, &recv))) return rc;
            if ((rc = ypass_xchg<false>(c, xl, send, st, yk))) return rc;
        }
    } else if (r.has_g) {
        if ((rc = real_ws(c, "dfdn", &r.dfdn))) return rc;
        const int nch = chunks_for(c, 6, 4);       // 3 spectra in, 3 out, the density and df/dn rows
        for (int ch = 0; ch < nch; ++ch) {
            const int x0 = ch * (c->n0 / nch), cx = c->n0 / nch;
            if (pbe_chunked && (rc = fast_axis_pass_multi<true>(c, 1, r.s_g, 3, st, x0, cx))) return rc;
            if ((rc = launch_zpbe(c, r.ds, r.s_g[0], r.s_g[1], r.s_g[2], r.dfdn, r.za.inv_n, &r.pbe_blocks, st, ch, nch)))
                return rc;
            if (pbe_chunked && (rc = fast_axis_pass_multi<false>(c, 1, r.s_g, 3, st, x0, cx))) return rc;
        }
        // no host round trip in the middle of the evaluation: reduce on the device, read with the final sums
        OFDFT_REDUCE(c, st, c->d_partial, r.pbe_blocks, kPbeScalars, c->d_reduced + kCombineScalars, c->h_partial + kCombineScalars);
        for (int k = 0; k < 3; ++k) {
            if (!dx && !pbe_chunked && (rc = fast_axis_pass<false>(c, 1, r.s_g[k], st))) return rc;
            xl.push_back(r.s_g[k]);
        }
        if (dx) {
            if ((rc = dist_buffers(c, 0, &send, &recv))) return rc;
            if ((rc = ypass_xchg<false>(c, xl, send, st, yk))) return rc;
        }
    }
    r.stage[0] = 3;
    return 0;
}

// Stage 4: fused x pass of the divergence (chain 0 only).
int zstage4(ofdft_ctx* c, hipStream_t st, int chain, int xk = -1) {
    ZRun& r = zrun(c);
    if (c->nranks > 1 && xk == -1 && c->xc.n > 1) {
        for (int k = 0; k < c->xc.n; ++k)
            if (int rc = zstage4(c, st, chain, k)) return rc;
        return 0;
    }
    r.xlist[chain].clear();
    const XcView xv = xc_view(c, xk < 0 ? 0 : xk);
    if (chain == 0 && r.has_g && r.gsplit) {
        XfIo dio{};
        XfLayout lay{};
        dio.in[0] = dio.out[0] = r.s_g[0];
        const int nin = r.lapl ? 2 : 1;
        if (c->nranks > 1) {     // receive buffer slot 0 -> send buffer slot 0 (of this kz chunk)
            cplx *send, *recv;
            if (int rc = dist_buffers(c, 0, &send, &recv)) return rc;
            dio.in[0] = recv + (long long)nin * xv.base1;
            dio.out[0] = send + xv.base1;
            lay = XfLayout{xv.arr_sz, xv.arr_sz, xv.arr_sz, 0, 0, xv.nb, xv.nrem, xv.kb0 * 8};
        }
        if (r.lapl) {            // i f_a G_a^ + (k^2 / 2) (df/dL)^ -> the spectrum the combine subtracts twice
            dio.in[1] = r.s_l;
            if (c->nranks > 1) {
                dio.in[1] = dio.in[0] + xv.arr_sz;
                lay.se_in = 2 * xv.arr_sz;
            }
            if (int rc = xfused<2, 1>(c, dio, MixDerivAL{c->kg}, st, "xfused_div", lay)) return rc;
        } else if (int rc = xfused<1, 1>(c, dio, MixDerivA{c->kg}, st, "xfused_div", lay)) {
            return rc;
        }
        r.xlist[0].push_back(r.s_g[0]);
    } else if (chain == 0 && r.has_g) {
        XfIo dio{};
        XfLayout lay{};
        for (int k = 0; k < 3; ++k) dio.in[k] = r.s_g[k];
        dio.out[0] = r.s_n;      // n^ is no longer needed
        if (c->nranks > 1) {     // receive buffer slots 0..2 -> send buffer slot 0 (of this kz chunk)
            cplx *send, *recv;
            if (int rc = dist_buffers(c, 0, &send, &recv)) return rc;
            for (int k = 0; k < 3; ++k) dio.in[k] = recv + 3 * xv.base1 + k * xv.arr_sz;
            dio.out[0] = send + xv.base1;
            lay = XfLayout{3 * xv.arr_sz, xv.arr_sz, xv.arr_sz, 0, 0, xv.nb, xv.nrem, xv.kb0 * 8};
        }
        if (int rc = xfused<3, 1>(c, dio, MixDiv{c->kg}, st, "xfused_div", lay)) return rc;
        r.xlist[0].push_back(r.s_n);
    }
    r.stage[chain] = 4;
    return 0;
}

// the local sums of an evaluation from the pinned host mirror (after the stream that copied them has been drained)
// what the host has to fold when it reads the pinned mirror of an evaluation's sums (the reducing kernels write it directly)
constexpr int kCollectWgcSplit = 1;     // the split WGC99 kernel's energy sum sits in slot kNSums
constexpr int kCollectVnShare = 2;      // ... and its share of sum(v n) in slot kNSums + 1 (closure form: chi_grad added it on the device)
constexpr int kCollectNoGga = 4;        // no GGA term: the three GGA slots are not written
int collect_flags(const ZRun& r, bool late_join) {
    return (r.wgc_split ? kCollectWgcSplit : 0) | (late_join ? kCollectVnShare : 0) | (r.has_g ? 0 : kCollectNoGga);
}
void zfused_collect(const ofdft_ctx* c, int flags, double* sums) {
    for (int i = 0; i < kNSums; ++i) sums[i] = c->h_partial[i];
    if (flags & kCollectNoGga)
        for (int i = kCombineScalars; i < kNSums; ++i) sums[i] = 0.0;
    if (flags & kCollectWgcSplit) sums[5] += c->h_partial[kNSums];
    if (flags & kCollectVnShare) sums[8] += c->h_partial[kNSums + 1];
}

// local sums: sums[0..8] combine scalars, sums[9..10] PBE x / c
// `sums` == nullptr: the caller reduces the device-resident sums itself (slab-decomposed path).  `defer`: the sums are
// copied to the pinned host mirror but nothing waits here (zfused_collect reads them after the caller's stream sync --
// the graph-capturable form)
// (part 1: only the y-inverse of kz chunk xk of the divergence out of the exchange buffer; part 2: everything else; 0: both)
int zstage5(ofdft_ctx* c, double* sums, hipStream_t st, bool defer = false, int part = 0, int xk = -1) {
    ZRun& r = zrun(c);
    int rc;
    const bool chunked = chunks_for(c, 6, 8) > 1;
    if (part == 1) {
        if (r.has_g && c->nranks > 1) {
            cplx *send, *recv;
            if ((rc = dist_buffers(c, 0, &send, &recv))) return rc;
            if ((rc = ypass_xchg<true>(c, {r.gsplit ? r.s_g[0] : r.s_n}, recv, st, xk))) return rc;
        }
        return 0;
    }
    if (r.has_g) {
        cplx* dsp = r.gsplit ? r.s_g[0] : r.s_n;       // the spectrum that carries the (x part of the) divergence
        if (c->nranks > 1) {
            cplx *send, *recv;
            if ((rc = dist_buffers(c, 0, &send, &recv))) return rc;
            if (part != 2 && (rc = ypass_xchg<true>(c, {dsp}, recv, st))) return rc;
        } else if (chunked) {
            r.deferred.push_back(dsp);
        } else if ((rc = fast_axis_pass<true>(c, 1, dsp, st))) {
            return rc;
        }
        c->fft_count++;
        r.za.div = dsp;
        r.za.div2 = r.gsplit ? r.s_g[1] : nullptr;
        r.za.dfdn = r.dfdn;
    }
    r.xlist[0].clear();
    r.xlist[1].clear();
    const bool late_join = r.forked && r.vpart_deferred && r.wgc_split;
    if (r.forked) {       // the combine needs both chains -- unless the nonlocal chain's only product is the deferred v_part
        if (!late_join) {
            HIP_TRY(c, hipEventRecord(c->ev_join, r.sb));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
        }
        HIP_TRY(c, hipEventRecord(c->ev_join2, r.sc));
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join2, 0));
    }
    const bool wts = wts_active(c);
    r.za.wts_w = nullptr;
    if (wts) {            // energies-only pass -> device-resident weights f - f' X, f' (no host round trip: graph-capturable)
        ZCombineArgs z1 = r.za;
        z1.v_out = nullptr;
        if ((rc = launch_zi_combine(c, z1, &r.combine_blocks, st))) return rc;
        OFDFT_REDUCE(c, st, c->d_partial, r.combine_blocks, kCombineScalars, c->d_reduced);
        OFDFT_LAUNCH(c, st, "reduce", wts_weights_kernel, dim3(1), dim3(64), 0, (const acc_t*)c->d_reduced, c->d_scal + 4);
        r.za.wts_w = c->d_scal + 4;
    }
    if (chunked) {        // y-inverse of a chunk of every result spectrum, then the combine kernel on the same x planes
        const int narr = (int)r.deferred.size();
        const int nch = chunks_for(c, narr + 2, 8);       // + the real rows (chi, v_ext, df/dn, v) the kernel touches
        for (int ch = 0; ch < nch; ++ch) {
            if (narr && (rc = fast_axis_pass_multi<true>(c, 1, r.deferred.data(), narr, st, ch * (c->n0 / nch), c->n0 / nch)))
                return rc;
            if ((rc = launch_zi_combine(c, r.za, &r.combine_blocks, st, ch, nch))) return rc;
        }
        r.deferred.clear();
    } else if ((rc = launch_zi_combine(c, r.za, &r.combine_blocks, st))) {
        return rc;
    }
    // (host-bound sums: the reduce kernel writes the pinned mirror itself -- no copy command behind it; the stabilised
    // WT-style functional rewrites two of the sums afterwards and keeps the copy)
    OFDFT_REDUCE(c, st, c->d_partial, r.combine_blocks, kCombineScalars, c->d_reduced, (sums && !wts) ? c->h_partial : (acc_t*)nullptr);
    if (wts) OFDFT_LAUNCH(c, st, "reduce", wts_finalize_kernel, dim3(1), dim3(64), 0, c->d_reduced, (const acc_t*)(c->d_scal + 4));
    if (late_join) {      // now the nonlocal chain: its share of sum(v n) joins the combine's (mu is formed from the total)
        HIP_TRY(c, hipEventRecord(c->ev_join, r.sb));
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
        // host-bound sums: chi_grad and the host each add the share (slot 3 of the device scalars / its pinned mirror) themselves
        if (!sums)
            OFDFT_LAUNCH(c, st, "reduce", (axpy_kernel<acc_t>), dim3(1), dim3(64), 0, (const acc_t*)(c->d_scal + 3), c->d_reduced + 8,
                         (long long)1, 1);
    }
    r.late_join = late_join && sums != nullptr;
    if (!r.has_g) HIP_TRY(c, hipMemsetAsync(c->d_reduced + kCombineScalars, 0, kPbeScalars * sizeof(double), st));
    r.stage[0] = r.stage[1] = 5;
    if (!sums) {                  // the caller reduces the device-resident sums (c->d_reduced) itself
        if (r.wgc_split)          // fold in the energy sum of the split WGC99 kernel
            OFDFT_LAUNCH(c, st, "reduce", (axpy_kernel<acc_t>), dim3(1), dim3(64), 0, (const acc_t*)(c->d_scal + 2), c->d_reduced + 5,
                         (long long)1, 1);
        return 0;
    }
    if (wts) HIP_TRY(c, hipMemcpyAsync(c->h_partial, c->d_reduced, sizeof(double) * kCombineScalars, hipMemcpyDeviceToHost, st));
    if (defer) return 0;
    HIP_TRY(c, hipStreamSynchronize(st));
    zfused_collect(c, collect_flags(r, r.late_join), sums);
    return 0;
}

// stages 1-5 of one evaluation enqueued on `st` (and the side streams forked from / joined to it); with `defer` nothing
// touches the host: the sequence can be captured into a hipGraph
int zfused_enqueue(ofdft_ctx* c, const DenSrc& ds, double nel, const real* vext, real* v_out, double* sums, hipStream_t st,
                   bool defer) {
    if ((c->mask & OFDFT_ION_ELECTRON) && !vext) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");
    if (!ds.src) return fail(c, OFDFT_EINVAL, "null density pointer");
    ZRun& r = zrun(c);
    r.ds = ds;
    r.nel = nel;
    r.vext = vext;
    r.v_out = v_out;
    r.deferred.clear();
    r.closure = defer;            // (only the closure enqueues in deferred form; its consumer is chi_grad)
    r.vpart_deferred = false;
    r.za.v_part_deferred = 0;
    int rc;
    // Forking the nonlocal-KEDF chain (and the vW / second WGC99 half) onto their own streams lets their
    // latency-bound fused kernels overlap the other chain's bandwidth-bound passes.
    r.forked = c->use_side_stream && c->side_stream && c->side_stream2 && (c->mask & (OFDFT_WT_NL | OFDFT_WGC99_NL)) &&
               (c->mask & (OFDFT_HARTREE | OFDFT_VW | kGgaAny));
    if (r.forked) {
        r.sb = c->side_stream;
        r.sc = c->side_stream2;
        HIP_TRY(c, hipEventRecord(c->ev_fork, st));
        HIP_TRY(c, hipStreamWaitEvent(r.sb, c->ev_fork, 0));
        HIP_TRY(c, hipStreamWaitEvent(r.sc, c->ev_fork, 0));
    }
    // one GPU, forked: the nonlocal chain's first kernels go to the device BEFORE the other chain's four launches (its stream
    // sat idle for ~30 us of host enqueue time at the head of every evaluation; profiles/r04_timeline_*.md)
    r.setup_done = false;
    rc = zsetup(c);
    const bool nl_first = r.forked && c->nranks == 1;
    for (int i = 0; i < 2 && !rc; ++i) rc = zstage1(c, st, nl_first ? 1 - i : i);
    for (int chain = 0; chain < 2 && !rc; ++chain) rc = zstage2(c, st, chain);
    for (int chain = 0; chain < 2 && !rc; ++chain) rc = zstage3(c, st, chain);
    if (!rc) rc = zstage4(c, st, 0);
    if (!rc) rc = zstage5(c, sums, st, defer);
    // (a call that failed between zsetup and chain 0's stage 1 -- the nonlocal chain goes first here -- must not leave the
    // flag set: a later staged evaluation entering through zstage1(chain 0) would skip its own set-up)
    r.setup_done = false;
    return rc;
}

int run_terms_zfused(ofdft_ctx* c, const DenSrc& ds, double nel, const real* vext, double* E_terms, real* v_out,
                     double* vn_int, hipStream_t st) {
    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
    double sums[kNSums];
    if (int rc = zfused_enqueue(c, ds, nel, vext, v_out, sums, st, false)) return rc;
    energies_from_sums(c, sums, sums + kCombineScalars, E_terms, vn_int);
    return 0;
}

ZRun& zrun(ofdft_ctx* c) {
    if (!c->zr) c->zr = new ofdft_zrun_holder();
    return c->zr->r;
}

// timed = false: no event pair around the call (OFDFT_Q_KERNEL_MS then reads 0): two stream markers are a visible share of
// a 30-microsecond evaluation
int begin_call(ofdft_ctx* c, hipStream_t st, bool timed = true) {
    if (!c) return OFDFT_EINVAL;      // (the ABI function that called this holds the DeviceScope)
    if (!c->cell_set) return fail(c, OFDFT_ESTATE, "ofdft_set_cell has not been called");
    c->fft_count = 0;
    c->launch_count = 0;
    c->ypass_count = 0.0;
    c->yfwd_fused = 0;
    if (timed) HIP_TRY(c, hipEventRecord(c->ev0, st));
    return 0;
}
int end_call(ofdft_ctx* c, hipStream_t st, bool timed = true) {
    if (timed) HIP_TRY(c, hipEventRecord(c->ev1, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (timed) HIP_TRY(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    else c->last_ms = 0.f;
    c->ms_pending = false;
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return 0;
}

}  // namespace

