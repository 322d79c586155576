// C ABI of the routines around the hot path: ionic potential, ion-electron forces, per-term stress, ion-electron stress and
// the ion-ion sum.  Included once by engine.hip inside its extern "C" block (one
// translation unit: the kernels in the headers are not inline); uses the engine's context, workspaces and FFT drivers.
// ------------------------------------------------------------------------------ ionic potential
struct IonPrep {
    std::vector<double> frac, cart, slopes;
    std::vector<cplx> hb;
    double *d_frac = nullptr, *d_cart = nullptr;
    cplx *d_b0 = nullptr, *d_b1 = nullptr, *d_b2 = nullptr;
    RecpotTable tab{};
};

// shared host-side preparation of the ionic-potential entry points: wrapped fractional coordinates, Cartesian
// coordinates, Hermite slopes, PME b factors; uploads everything on `st` (caller syncs before `p` dies)
static int ion_prepare(ofdft_ctx* c, IonPrep& p, const double* frac_host, int nions, const double* tab_k,
                       const double* tab_v, int ntab, double z_ion, int pme_order, hipStream_t st) {
    if (!c->cell_set) return fail(c, OFDFT_ESTATE, "ofdft_set_cell has not been called");
    if (c->nranks > 1 && !(c->a2a && c->allreduce))
        return fail(c, OFDFT_ESTATE, "slab-decomposed context: the ionic-potential entry points need ofdft_set_collectives");
    if (nions < 1 || ntab < 2) return fail(c, OFDFT_EINVAL, "need at least one ion and two table points");
    if (pme_order != 0 && (pme_order < 2 || pme_order > kMaxPmeOrder || (pme_order & 1)))
        return fail(c, OFDFT_EINVAL, "Requires even order n >= 2 (<= %d)", kMaxPmeOrder);       // ion_utils.py:116
    OFDFT_ON_DEVICE(c, c->device);
    p.frac.resize(3 * (size_t)nions);
    p.cart.resize(3 * (size_t)nions);
    for (int a = 0; a < nions; ++a) {
        for (int d = 0; d < 3; ++d) {
            double f = frac_host[3 * a + d];
            f -= std::floor(f);
            f -= std::floor(f);                                                      // ion_utils.py:241-242
            p.frac[3 * a + d] = f;
        }
        for (int d = 0; d < 3; ++d)     // cart = frac @ box (un-wrapped, as the reference's exact sum uses)
            p.cart[3 * a + d] = frac_host[3 * a] * c->box[d] + frac_host[3 * a + 1] * c->box[3 + d] +
                                frac_host[3 * a + 2] * c->box[6 + d];
    }
    p.slopes.resize(ntab);
    {
        std::vector<double> m(ntab - 1);
        for (int i = 0; i + 1 < ntab; ++i) m[i] = (tab_v[i + 1] - tab_v[i]) / (tab_k[i + 1] - tab_k[i]);
        p.slopes[0] = m[0];
        for (int i = 1; i + 1 < ntab; ++i) p.slopes[i] = (m[i] + m[i - 1]) / 2;
        p.slopes[ntab - 1] = m[ntab - 2];
    }
    double *d_k, *d_y, *d_m;
    if (int rc = get_ws(c, "i:frac", sizeof(double) * p.frac.size(), (void**)&p.d_frac)) return rc;
    if (int rc = get_ws(c, "i:cart", sizeof(double) * p.cart.size(), (void**)&p.d_cart)) return rc;
    if (int rc = get_ws(c, "i:k", sizeof(double) * ntab, (void**)&d_k)) return rc;
    if (int rc = get_ws(c, "i:y", sizeof(double) * ntab, (void**)&d_y)) return rc;
    if (int rc = get_ws(c, "i:m", sizeof(double) * ntab, (void**)&d_m)) return rc;
    HIP_TRY(c, hipMemcpyAsync(p.d_frac, p.frac.data(), sizeof(double) * p.frac.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(p.d_cart, p.cart.data(), sizeof(double) * p.cart.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_k, tab_k, sizeof(double) * ntab, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_y, tab_v, sizeof(double) * ntab, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_m, p.slopes.data(), sizeof(double) * ntab, hipMemcpyHostToDevice, st));
    p.tab = RecpotTable{d_k, d_y, d_m, ntab, z_ion, 1.0 / (tab_k[1] - tab_k[0])};
    if (pme_order != 0) {
        // b(m) = exp(2 pi i m (n-1)/N) / sum_i M_n(i) exp(2 pi i m (i-1)/N)        ion_utils.py:207-215
        std::vector<double> M(pme_order, 0.0);
        M[1] = 1.0;
        for (int n = 3; n <= pme_order; ++n) {
            for (int i = n - 1; i >= 1; --i) M[i] = (i * M[i] + (double)(n - i) * M[i - 1]) / (n - 1);
            M[0] = 0.0;
        }
        const int cnt[3] = {c->n0g, c->n1g, c->g.nzc}, Ns[3] = {c->n0g, c->n1g, c->n2};      // global extents
        p.hb.resize((size_t)cnt[0] + cnt[1] + cnt[2]);
        size_t off = 0;
        for (int d = 0; d < 3; ++d) {
            for (int m = 0; m < cnt[d]; ++m) {
                double br = 0.0, bi = 0.0;
                for (int i = 0; i < pme_order; ++i) {
                    const double ph = 2.0 * kPi * m * (i - 1.0) / Ns[d];
                    br += M[i] * std::cos(ph);
                    bi += M[i] * std::sin(ph);
                }
                const double ph = 2.0 * kPi * m * (pme_order - 1.0) / Ns[d];
                const double nr = std::cos(ph), ni = std::sin(ph), den = br * br + bi * bi;
                p.hb[off + m] = make_double2((nr * br + ni * bi) / den, (ni * br - nr * bi) / den);
            }
            off += cnt[d];
        }
        cplx* d_b;
        if (int rc = get_ws(c, "i:b", sizeof(cplx) * p.hb.size(), (void**)&d_b)) return rc;
        HIP_TRY(c, hipMemcpyAsync(d_b, p.hb.data(), sizeof(cplx) * p.hb.size(), hipMemcpyHostToDevice, st));
        p.d_b0 = d_b;
        p.d_b1 = d_b + cnt[0] + c->kg.y0;      // the k-space kernels index b1 by the LOCAL y of the y-slab
        p.d_b2 = d_b + cnt[0] + cnt[1];
    }
    return 0;
}

int ofdft_ionic_potential(ofdft_ctx* c, const double* frac_host, int nions, const double* tab_k, const double* tab_v,
                          int ntab, double z_ion, int pme_order, void* vext_dev, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !frac_host || !tab_k || !tab_v || !vext_dev) return OFDFT_EINVAL;
    IonPrep p;
    if (int rc = ion_prepare(c, p, frac_host, nions, tab_k, tab_v, ntab, z_ion, pme_order, st)) return rc;
    cplx *sQ, *sF;
    double* tmp;
    if (int rc = spec_ws(c, "i:F", &sF)) return rc;
    if (int rc = real_ws(c, "i:tmp", &tmp)) return rc;
    const int sp_grid = grid_for(c->g.total);
    if (pme_order == 0) {
        OFDFT_LAUNCH(c, st, "ion_spec", ion_potential_spec_kernel, dim3(sp_grid), dim3(256), 0, (const cplx*)nullptr, sF, c->kg,
                     (const cplx*)nullptr, (const cplx*)nullptr, (const cplx*)nullptr, (const double*)p.d_cart, nions, p.tab,
                     1.0 / c->vol);
    } else {
        if (int rc = spec_ws(c, "i:Q", &sQ)) return rc;
        HIP_TRY(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)c->npts, st));
        OFDFT_LAUNCH(c, st, "pme_spread", pme_spread_kernel, dim3(nions), dim3(256), 0, (const double*)p.d_frac, nions,
                     pme_order, tmp, c->n0g, c->n1g, c->n2, c->rank * c->n0, c->n0);
        if (int rc = rfftn_internal(c, tmp, sQ, st)) return rc;
        OFDFT_LAUNCH(c, st, "ion_spec", ion_potential_spec_kernel, dim3(sp_grid), dim3(256), 0, (const cplx*)sQ, sF, c->kg,
                     (const cplx*)p.d_b0, (const cplx*)p.d_b1, (const cplx*)p.d_b2, (const double*)nullptr, nions, p.tab,
                     1.0 / c->vol);
    }
    if (int rc = irfftn_internal(c, sF, tmp, 1.0, st)) return rc;              // norm='forward': no 1/N  (ion_utils.py:118)
    OFDFT_LAUNCH(c, st, "axpy", (axpy_kernel<double>), dim3(grid_for(c->npts)), dim3(256), 0, (const double*)tmp, (double*)vext_dev,
                 c->npts, accumulate);
    HIP_TRY(c, hipStreamSynchronize(st));      // `p` (host staging) must outlive the async copies
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

// F_a = -dU/dR_a, U = int n v_ext for one species (the ion-electron part of System.forces, system.py:913-923)
int ofdft_ion_electron_forces(ofdft_ctx* c, const void* den_dev, const double* frac_host, int nions, const double* tab_k,
                              const double* tab_v, int ntab, double z_ion, int pme_order, double* forces_host,
                              void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !den_dev || !frac_host || !tab_k || !tab_v || !forces_host) return OFDFT_EINVAL;
    IonPrep p;
    if (int rc = ion_prepare(c, p, frac_host, nions, tab_k, tab_v, ntab, z_ion, pme_order, st)) return rc;
    cplx *sN, *sT;
    if (int rc = spec_ws(c, "i:Q", &sN)) return rc;
    if (int rc = rfftn_internal(c, (const double*)den_dev, sN, st)) return rc;
    const double pref = c->dV / c->vol;
    if (pme_order == 0) {
        const int kb = grid_for(c->g.total, kRedThreads, 64);
        double* d_part;
        std::vector<double> h((size_t)nions * kb * 3);
        if (int rc = get_ws(c, "i:fpart", sizeof(double) * h.size(), (void**)&d_part)) return rc;
        OFDFT_LAUNCH(c, st, "ion_force", ion_force_exact_kernel, dim3(kb, nions), dim3(kRedThreads), 0, (const cplx*)sN, c->kg,
                     (const double*)p.d_cart, p.tab, d_part);
        HIP_TRY(c, hipMemcpyAsync(h.data(), d_part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        for (int a = 0; a < nions; ++a)
            for (int s3 = 0; s3 < 3; ++s3) {
                long double t = 0.0L;
                for (int b = 0; b < kb; ++b) t += h[((size_t)a * kb + b) * 3 + s3];
                forces_host[3 * a + s3] = pref * (double)t;
            }
        if (int rc = global_sums(c, forces_host, 3 * nions)) return rc;      // each rank summed its own k-points
    } else {
        double *theta, *d_G;
        std::vector<double> G(3 * (size_t)nions);
        if (int rc = spec_ws(c, "i:F", &sT)) return rc;
        if (int rc = real_ws(c, "i:tmp", &theta)) return rc;
        if (int rc = get_ws(c, "i:G", sizeof(double) * G.size(), (void**)&d_G)) return rc;
        OFDFT_LAUNCH(c, st, "pme_theta", pme_theta_spec_kernel, dim3(grid_for(c->g.total)), dim3(256), 0, (const cplx*)sN, sT,
                     c->kg, (const cplx*)p.d_b0, (const cplx*)p.d_b1, (const cplx*)p.d_b2, p.tab, 1.0 / c->vol);
        if (int rc = irfftn_internal(c, sT, theta, 1.0, st)) return rc;
        OFDFT_LAUNCH(c, st, "pme_gather", pme_gather_kernel, dim3(nions), dim3(256), 0, (const double*)p.d_frac, nions,
                     pme_order, (const double*)theta, c->n0g, c->n1g, c->n2, d_G, c->rank * c->n0, c->n0);
        HIP_TRY(c, hipMemcpyAsync(G.data(), d_G, sizeof(double) * G.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (int rc = global_sums(c, G.data(), (int)G.size())) return rc;     // each rank gathered over its own planes
        // dU/dR_j = dV sum_d G_d N_d d(frac_d)/d(cart_j),  frac = cart @ inv(box)  ->  d frac_d / d cart_j = inv(box)[j][d]
        const double* a9 = c->box;
        const double det = a9[0] * (a9[4] * a9[8] - a9[5] * a9[7]) - a9[1] * (a9[3] * a9[8] - a9[5] * a9[6]) +
                           a9[2] * (a9[3] * a9[7] - a9[4] * a9[6]);
        double inv[9];
        inv[0] = (a9[4] * a9[8] - a9[5] * a9[7]) / det;
        inv[1] = (a9[2] * a9[7] - a9[1] * a9[8]) / det;
        inv[2] = (a9[1] * a9[5] - a9[2] * a9[4]) / det;
        inv[3] = (a9[5] * a9[6] - a9[3] * a9[8]) / det;
        inv[4] = (a9[0] * a9[8] - a9[2] * a9[6]) / det;
        inv[5] = (a9[2] * a9[3] - a9[0] * a9[5]) / det;
        inv[6] = (a9[3] * a9[7] - a9[4] * a9[6]) / det;
        inv[7] = (a9[1] * a9[6] - a9[0] * a9[7]) / det;
        inv[8] = (a9[0] * a9[4] - a9[1] * a9[3]) / det;
        const int Ns[3] = {c->n0g, c->n1g, c->n2};
        for (int a = 0; a < nions; ++a)
            for (int j = 0; j < 3; ++j) {
                double t = 0.0;
                for (int d = 0; d < 3; ++d) t += inv[3 * j + d] * Ns[d] * G[3 * a + d];
                forces_host[3 * a + j] = -c->dV * t;
            }
    }
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

// ------------------------------------------------------------------------------ stress
namespace {

void sym_store(double* out9, const double* c6, double diag) {
    out9[0] = c6[0] + diag; out9[4] = c6[1] + diag; out9[8] = c6[2] + diag;
    out9[1] = out9[3] = c6[3];
    out9[2] = out9[6] = c6[4];
    out9[5] = out9[7] = c6[5];
}

}  // namespace

// Per-term stress tensors for the active terms, sigma_terms_host[OFDFT_NTERMS][9] (row-major 3x3, Ha/bohr^3); the
// ion-electron entry stays zero (its potential depends on the ions: ofdft_ion_electron_stress).
int ofdft_stress(ofdft_ctx* c, const void* den_dev, double* sig, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    if (int rc = begin_call(c, st)) return rc;
    if (!den_dev || !sig) return fail(c, OFDFT_EINVAL, "null argument");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    if (c->nranks > 1 && !(c->a2a && c->allreduce))
        return fail(c, OFDFT_ESTATE, "slab-decomposed context: ofdft_stress needs ofdft_set_collectives");
    if (wts_active(c) && c->nranks > 1)
        return fail(c, OFDFT_EINVAL, "the stabilised Wang-Teter style functional (OFDFT_P_WTS_KIND) is served by single-GPU contexts");
    const double* den = (const double*)den_dev;
    const unsigned mask = c->mask;
    const long long npts = c->npts;
    // slab-decomposed contexts: npts = this rank's points (kernel extents); every normalisation uses the GLOBAL grid, every
    // reduced number is summed over the ranks (global_sums: the host's all-reduce) before it is used
    const double invN = 1.0 / (double)c->npts_g, invN2 = invN * invN;
    for (int i = 0; i < OFDFT_NTERMS * 9; ++i) sig[i] = 0.0;
    const int sp_blocks = grid_for(c->g.total, kRedThreads, kRedBlocks), pw_grid = grid_for(npts / 2 + 1);
    double s7[kStressSpecScalars];
    double nsum;
    if (int rc = device_sum(c, den, false, &nsum, st)) return rc;
    if (int rc = global_sums(c, &nsum, 1)) return rc;
    const double nbar = nsum * invN;                    // N_e / vol, un-rounded (functionals.py:634)
    cplx *s0 = nullptr, *s1 = nullptr, *s2 = nullptr, *s3 = nullptr;
    double *gx = nullptr, *gy = nullptr, *gz = nullptr, *lapn = nullptr;
    if (int rc = spec_ws(c, "s0", &s0)) return rc;
    if (int rc = spec_ws(c, "s1", &s1)) return rc;
    if (mask & (OFDFT_HARTREE | kGgaAny)) {
        if (int rc = rfftn_internal(c, den, s0, st)) return rc;
        if (mask & OFDFT_HARTREE) {
            OFDFT_LAUNCH(c, st, "stress_spec", (stress_spec_kernel<STRESS_HARTREE>), dim3(sp_blocks), dim3(kRedThreads), 0,
                         (const cplx*)s0, (const cplx*)nullptr, c->kg, invN2, 0.0, c->d_partial);
            if (int rc = fetch_partials(c, sp_blocks, kStressSpecScalars, s7, st)) return rc;
            if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
            sym_store(sig + 9 * 1, s7, -0.5 * s7[6]);                 // -E_H / vol on the diagonal
        }
        if (mask & kGgaAny) {
            if (int rc = spec_ws(c, "s2", &s2)) return rc;
            if (int rc = spec_ws(c, "s3", &s3)) return rc;
            if (int rc = real_ws(c, "gx", &gx)) return rc;
            if (int rc = real_ws(c, "gy", &gy)) return rc;
            if (int rc = real_ws(c, "gz", &gz)) return rc;
            if (gga_needs_laplacian(c)) {        // lap n = F^-1[-k^2 n^] (functional_tools.py:209-227)
                if (int rc = real_ws(c, "lapn", &lapn)) return rc;
                OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_LAPLACE>), dim3(grid_for(c->g.total)), dim3(256), 0, s0, s1,
                             c->kg, 0.0, 0.0);
                if (int rc = irfftn_internal(c, s1, lapn, invN, st)) return rc;
            }
            OFDFT_LAUNCH(c, st, "spec_grad", spec_grad_kernel, dim3(grid_for(c->g.total)), dim3(256), 0, s0, s1, s2, s3, c->kg);
            if (int rc = irfftn_internal(c, s1, gx, invN, st)) return rc;
            if (int rc = irfftn_internal(c, s2, gy, invN, st)) return rc;
            if (int rc = irfftn_internal(c, s3, gz, invN, st)) return rc;
        }
    }
    if (mask & (OFDFT_TF | OFDFT_VWGTF | OFDFT_LDA_X | OFDFT_PZ_C | OFDFT_PW_C | OFDFT_CHACHIYO_C | kGgaAny)) {
        const int blocks = grid_for(npts, kRedThreads, kRedBlocks);
        double r[kStressRealScalars];
        OFDFT_LAUNCH(c, st, "stress_real", stress_real_kernel, dim3(blocks), dim3(kRedThreads), 0, den, (const double*)gx,
                     (const double*)gy, (const double*)gz, npts, mask, gga_sel(c),
                     (mask & OFDFT_VWGTF) ? c->vol / (double)std::llround(nsum * c->dV) : 0.0, (int)c->params[OFDFT_P_VWGTF_KIND],
                     c->d_partial, lapn);
        if (int rc = fetch_partials(c, blocks, kStressRealScalars, r, st)) return rc;
        if (int rc = global_sums(c, r, kStressRealScalars)) return rc;
        const double zero6[6] = {0, 0, 0, 0, 0, 0};
        const double ctf = 0.3 * std::pow(3.0 * kPi * kPi, 2.0 / 3.0);
        if (mask & OFDFT_TF) sym_store(sig + 9 * 2, zero6, -2.0 / 3.0 * ctf * r[0] * invN);      // tools_for_tests.py:241-243
        if (mask & OFDFT_VWGTF) sym_store(sig + 9 * 13, zero6, -2.0 / 3.0 * r[27] * invN);      // E ~ J^(-2/3) at fixed n / n0
        if (mask & OFDFT_LDA_X) sym_store(sig + 9 * 6, zero6, r[1] * invN);                    // :367-370
        int nc = 0;
        for (int b = 7; b <= 9; ++b) nc += (mask >> b) & 1;
        for (int b = 7; b <= 9; ++b)
            if ((mask >> b) & 1) sym_store(sig + 9 * b, zero6, r[2] * invN / nc);
        for (int which = 0; which < 3; ++which) {                                              // :393-472 (same form for the kinetic GGA)
            if (!(mask & (which == 0 ? OFDFT_PBE_X : (which == 1 ? OFDFT_PBE_C : OFDFT_GGA_K)))) continue;
            const double* o = r + 3 + 8 * which;
            double c6[6];
            for (int k = 0; k < 6; ++k) c6[k] = -2.0 * o[k] * invN;
            for (int k = 0; k < 3; ++k) c6[k] += -2.0 * o[6] * invN;
            sym_store(sig + 9 * (10 + which), c6, o[7] * invN);
        }
        if (lapn) {       // Hessian term of the q-dependent Pauli-Gaussian members: -2 mean(f_l d_i d_j n) = (2 / N^2) sum_k w k_i k_j Re(n^ conj(f_l^))
            if (int rc = rfftn_internal(c, lapn, s1, st)) return rc;          // lapn now holds f_l = df/d(lap n)
            OFDFT_LAUNCH(c, st, "stress_spec", (stress_spec_kernel<STRESS_HESS>), dim3(sp_blocks), dim3(kRedThreads), 0, (const cplx*)s0,
                         (const cplx*)s1, c->kg, invN2, 0.0, c->d_partial);
            if (int rc = fetch_partials(c, sp_blocks, kStressSpecScalars, s7, st)) return rc;
            if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
            double* o = sig + 9 * 12;
            o[0] += 2.0 * s7[0]; o[4] += 2.0 * s7[1]; o[8] += 2.0 * s7[2];
            o[1] += 2.0 * s7[3]; o[3] += 2.0 * s7[3];
            o[2] += 2.0 * s7[4]; o[6] += 2.0 * s7[4];
            o[5] += 2.0 * s7[5]; o[7] += 2.0 * s7[5];
        }
    }
    if (mask & OFDFT_VW) {
        double* tmp;
        if (int rc = real_ws(c, "t0", &tmp)) return rc;
        OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_SQRT>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, 0.0);
        if (int rc = rfftn_internal(c, tmp, s0, st)) return rc;
        OFDFT_LAUNCH(c, st, "stress_spec", (stress_spec_kernel<STRESS_VW>), dim3(sp_blocks), dim3(kRedThreads), 0, (const cplx*)s0,
                     (const cplx*)nullptr, c->kg, invN2, 0.0, c->d_partial);
        if (int rc = fetch_partials(c, sp_blocks, kStressSpecScalars, s7, st)) return rc;
        if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
        sym_store(sig + 9 * 3, s7, 0.0);
    }
    if (mask & OFDFT_WT_NL) {
        const double al = c->params[OFDFT_P_WT_ALPHA], be = c->params[OFDFT_P_WT_BETA];
        const double kf = std::cbrt(3.0 * kPi * kPi * nbar);
        const double ctf = 0.3 * std::pow(3.0 * kPi * kPi, 2.0 / 3.0);
        const double pref = ctf * 5.0 / (9.0 * al * be * std::pow(nbar, al + be - 5.0 / 3.0));
        double* tmp;
        if (int rc = real_ws(c, "t0", &tmp)) return rc;
        OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, be);
        if (int rc = rfftn_internal(c, tmp, s0, st)) return rc;
        cplx* sa = s0;
        if (al != be) {
            OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, al);
            if (int rc = rfftn_internal(c, tmp, s1, st)) return rc;
            sa = s1;
        }
        OFDFT_LAUNCH(c, st, "stress_spec", (stress_spec_kernel<STRESS_WT>), dim3(sp_blocks), dim3(kRedThreads), 0, (const cplx*)sa,
                     (const cplx*)s0, c->kg, invN2, 1.0 / (2.0 * kf), c->d_partial);
        if (int rc = fetch_partials(c, sp_blocks, kStressSpecScalars, s7, st)) return rc;
        if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
        double c6[6];
        for (int k = 0; k < 6; ++k) c6[k] = pref * s7[k];
        sym_store(sig + 9 * 4, c6, -2.0 / 3.0 * pref * s7[6]);                                  // -2/3 T_NL / vol
    }
    if (wts_active(c)) {      // T = T_TF f(X): sigma = sigma_TF (f - f' X) + sigma_NL f'(X) / f'(0)   (tools_for_tests.py:310-364), f = exp
        // both stresses carry -2/3 E / vol on their diagonals: T_TF / vol = -3/2 sig_TF[0] exactly; T_NL / vol from the trace
        // of the Wang-Teter tensor (its k-dependent part is traceless)
        const double tf_over_vol = -1.5 * sig[9 * 2], nl_over_vol = -0.5 * (sig[9 * 4] + sig[9 * 4 + 4] + sig[9 * 4 + 8]);
        const double X = nl_over_vol / tf_over_vol, fx = std::exp(X);
        for (int k = 0; k < 9; ++k) {
            sig[9 * 2 + k] *= fx * (1.0 - X);
            sig[9 * 4 + k] *= fx;
        }
    }
    if (mask & OFDFT_WGC99_NL) {
        const double al = c->params[OFDFT_P_WGC_ALPHA], be = c->params[OFDFT_P_WGC_BETA];
        const long long nel_r = std::llround(nsum * c->dV);                                      // functionals.py:952
        WgcSeries ser{};
        if (int rc = wgc_series_setup(c, nel_r, st, &ser)) return rc;
        if (ser.v == 0.0) return fail(c, OFDFT_EINVAL, "WGC99 stress: degenerate kernel parameters (v = 0) not supported");
        const char* wn[6] = {"zw0", "zw1", "zw2", "zw3", "zw4", "zw5"};
        WgcSpectra sp{};
        double* t[3];
        if (int rc = real_ws(c, "t0", &t[0])) return rc;
        if (int rc = real_ws(c, "t1", &t[1])) return rc;
        if (int rc = real_ws(c, "t2", &t[2])) return rc;
        for (int pass = 0; pass < 2; ++pass) {
            OFDFT_LAUNCH(c, st, "wgc_prep", wgc_prep_kernel, dim3(pw_grid), dim3(256), 0, den, t[0], t[1], t[2], npts,
                         pass == 0 ? be : al, ser.nref);
            for (int k = 0; k < 3; ++k) {
                cplx* w;
                if (int rc = spec_ws(c, wn[3 * pass + k], &w)) return rc;
                if (int rc = rfftn_internal(c, t[k], w, st)) return rc;
                sp.s[3 * pass + k] = w;
            }
        }
        OFDFT_LAUNCH(c, st, "stress_wgc", stress_wgc_kernel, dim3(sp_blocks), dim3(kRedThreads), 0, sp, c->kg, ser, invN2,
                     c->d_partial);
        if (int rc = fetch_partials(c, sp_blocks, kStressSpecScalars, s7, st)) return rc;
        if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
        const double ctf = 0.3 * std::pow(3.0 * kPi * kPi, 2.0 / 3.0);
        double c6[6];
        for (int k = 0; k < 6; ++k) c6[k] = ctf * s7[k];
        sym_store(sig + 9 * 5, c6, -2.0 / 3.0 * ctf * s7[6]);
    }
    return end_call(c, st);
}

// Ion-electron stress of one species for a given density, the potential being rebuilt from the ions at fixed fractional
// coordinates (what System.__compute_stress differentiates, system.py:925-935).  sigma_host[9], row-major.
int ofdft_ion_electron_stress(ofdft_ctx* c, const void* den_dev, const double* frac_host, int nions, const double* tab_k,
                              const double* tab_v, int ntab, double z_ion, int pme_order, double* sigma_host, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !den_dev || !frac_host || !tab_k || !tab_v || !sigma_host) return OFDFT_EINVAL;
    IonPrep p;
    if (int rc = ion_prepare(c, p, frac_host, nions, tab_k, tab_v, ntab, z_ion, pme_order, st)) return rc;
    cplx *sN, *sQ = nullptr;
    if (int rc = spec_ws(c, "i:F", &sN)) return rc;
    if (int rc = rfftn_internal(c, (const double*)den_dev, sN, st)) return rc;
    if (pme_order != 0) {
        double* tmp;
        if (int rc = spec_ws(c, "i:Q", &sQ)) return rc;
        if (int rc = real_ws(c, "i:tmp", &tmp)) return rc;
        HIP_TRY(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)c->npts, st));
        OFDFT_LAUNCH(c, st, "pme_spread", pme_spread_kernel, dim3(nions), dim3(256), 0, (const double*)p.d_frac, nions,
                     pme_order, tmp, c->n0g, c->n1g, c->n2, c->rank * c->n0, c->n0);
        if (int rc = rfftn_internal(c, tmp, sQ, st)) return rc;
    }
    const int blocks = grid_for(c->g.total, kRedThreads, kRedBlocks);
    OFDFT_LAUNCH(c, st, "stress_ion", stress_ion_kernel, dim3(blocks), dim3(kRedThreads), 0, (const cplx*)sN, (const cplx*)sQ,
                 c->kg, (const cplx*)p.d_b0, (const cplx*)p.d_b1, (const cplx*)p.d_b2,
                 pme_order == 0 ? (const double*)p.d_cart : (const double*)nullptr, nions, p.tab, c->d_partial);
    double s7[kStressSpecScalars];
    if (int rc = fetch_partials(c, blocks, kStressSpecScalars, s7, st)) return rc;
    if (int rc = global_sums(c, s7, kStressSpecScalars)) return rc;
    const double invN = 1.0 / (double)c->npts_g;
    double c6[6];
    for (int k = 0; k < 6; ++k) c6[k] = -s7[k] * invN / c->vol;
    sym_store(sigma_host, c6, -s7[6] * invN / c->vol);
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

// Ion-ion interaction energy, forces and stress (ion_utils.py:293-333 with the parameter heuristics of
// System.__ion_ion_interaction, system.py:733-754; forces / stress = what autograd yields, system.py:913-935).
// Rc <= 0 selects the reference's default (Rd = 2 h_max, Rc = 3 Rd^2 / h_max); forces_host [nions][3] and
// stress_host [9] may be NULL.
int ofdft_ion_ion(ofdft_ctx* c, const double* frac_host, const double* charges_host, int nions, double Rc, double* E_host,
                  double* forces_host, double* stress_host, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !frac_host || !charges_host || !E_host || nions < 1) return OFDFT_EINVAL;
    if (!c->cell_set) return fail(c, OFDFT_ESTATE, "ofdft_set_cell has not been called");
    OFDFT_ON_DEVICE(c, c->device);
    const double* B = c->box;
    // interplanar spacings h_d = 1 / |row d of inv(B^T)| = vol / |cross of the other two lattice vectors|
    double h[3];
    for (int d = 0; d < 3; ++d) {
        const double* u = B + 3 * ((d + 1) % 3);
        const double* v = B + 3 * ((d + 2) % 3);
        const double cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
        h[d] = c->vol / std::sqrt(cx * cx + cy * cy + cz * cz);
    }
    const double h_max = std::max(h[0], std::max(h[1], h[2]));
    double Rd;
    if (Rc <= 0.0) {
        Rd = 2.0 * h_max;
        Rc = 3.0 * Rd * Rd / h_max;
    } else {
        Rd = std::sqrt(h_max * Rc / 3.0);
    }
    IonIonGeom g{};
    std::memcpy(g.box, B, sizeof(g.box));
    g.Rc = Rc;
    g.Rd = Rd;
    std::vector<double> cart(3 * (size_t)nions);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int a = 0; a < nions; ++a)
        for (int d = 0; d < 3; ++d) {
            cart[3 * a + d] = frac_host[3 * a] * B[d] + frac_host[3 * a + 1] * B[3 + d] + frac_host[3 * a + 2] * B[6 + d];
            lo[d] = std::min(lo[d], frac_host[3 * a + d]);
            hi[d] = std::max(hi[d], frac_host[3 * a + d]);
        }
    long long nshift = 1;
    for (int d = 0; d < 3; ++d) {
        g.nmax[d] = (int)std::ceil(Rc / h[d] + (hi[d] - lo[d]));
        nshift *= 2 * g.nmax[d] + 1;
    }
    const long long total = nshift * nions;
    const int chunks = (int)std::min<long long>(256, (total + kRedThreads * 8 - 1) / (kRedThreads * 8));
    double *d_cart, *d_z, *d_part;
    const size_t np = (size_t)nions * chunks * kIonIonScalars;
    if (int rc = get_ws(c, "ii:cart", sizeof(double) * cart.size(), (void**)&d_cart)) return rc;
    if (int rc = get_ws(c, "ii:z", sizeof(double) * nions, (void**)&d_z)) return rc;
    if (int rc = get_ws(c, "ii:part", sizeof(double) * np, (void**)&d_part)) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_cart, cart.data(), sizeof(double) * cart.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_z, charges_host, sizeof(double) * nions, hipMemcpyHostToDevice, st));
    OFDFT_LAUNCH(c, st, "ion_ion", ion_ion_kernel, dim3(chunks, nions), dim3(kRedThreads), 0, (const double*)d_cart,
                 (const double*)d_z, nions, g, d_part);
    std::vector<double> hp(np);
    HIP_TRY(c, hipMemcpyAsync(hp.data(), d_part, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    double ztot = 0.0;
    for (int a = 0; a < nions; ++a) ztot += charges_host[a];
    const double rho = ztot / c->vol, spi = std::sqrt(kPi);
    long double E = 0.0L, sig6[6] = {0, 0, 0, 0, 0, 0}, corr = 0.0L;
    for (int a = 0; a < nions; ++a) {
        long double s[kIonIonScalars];
        for (int k = 0; k < kIonIonScalars; ++k) {
            s[k] = 0.0L;
            for (int b = 0; b < chunks; ++b) s[k] += hp[((size_t)a * chunks + b) * kIonIonScalars + k];
        }
        const double Z = charges_host[a];
        const double Q = Z + (double)s[1];
        const double aux = 0.75 / kPi * Q / rho;
        const double Ra = std::cbrt(aux);
        const double ex = std::exp(-Ra * Ra / (Rd * Rd)), er = std::erf(Ra / Rd);
        E += 0.5L * s[0] - kPi * Z * rho * Ra * Ra + kPi * Z * rho * (Ra * Ra - 0.5 * Rd * Rd) * er + spi * Z * rho * Ra * Rd * ex -
             Z * Z / spi / Rd;
        if (forces_host)
            for (int d = 0; d < 3; ++d) forces_host[3 * a + d] = (double)s[2 + d];
        for (int k = 0; k < 6; ++k) sig6[k] += 0.5L * s[5 + k];
        const double e_rho = -kPi * Z * Ra * Ra + kPi * Z * (Ra * Ra - 0.5 * Rd * Rd) * er + spi * Z * Ra * Rd * ex;
        const double dE_dRa = -2.0 * kPi * Z * rho * Ra + 2.0 * kPi * Z * rho * Ra * er +
                              kPi * Z * rho * (Ra * Ra - 0.5 * Rd * Rd) * 2.0 / (spi * Rd) * ex +
                              spi * Z * rho * Rd * ex * (1.0 - 2.0 * Ra * Ra / (Rd * Rd));
        corr += -rho * e_rho + dE_dRa * Ra / 3.0;
    }
    *E_host = (double)E;
    if (stress_host) {
        double c6[6];
        for (int k = 0; k < 6; ++k) c6[k] = (double)sig6[k] / c->vol;
        sym_store(stress_host, c6, (double)corr / c->vol);
    }
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

