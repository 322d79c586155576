// C-ABI implementation of the OFDFT energy/gradient engine (include/ofdft_hip.h).  gfx950 only.
// Core translation unit: context, workspaces, pipelines, entry points (engine_ctx.h lists the other sources).
#include "engine_ctx.h"
#include <atomic>
#include <chrono>
#ifndef OFDFT_REAL_F32
#include "ion_kernels.h"
#include "stress_kernels.h"
#endif

namespace {
thread_local char g_create_error[512] = "";
}  // namespace

namespace eng {

int fail(ofdft_ctx* c, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_error, 512, fmt, ap);
    va_end(ap);
    return code;
}

int get_twiddle(ofdft_ctx* c, int n, cplx** out) {
    auto it = c->tw.find(n);
    if (it != c->tw.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<cplx> h(n);
    for (int m = 0; m < n; ++m) {
        const long double ph = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
        h[m] = mkc((double)cosl(ph), (double)sinl(ph));
    }
    // exact values on the axes
    h[0] = mkc(1.0, 0.0);
    if (n % 2 == 0) h[n / 2] = mkc(-1.0, 0.0);
    if (n % 4 == 0) {
        h[n / 4] = mkc(0.0, -1.0);
        h[3 * n / 4] = mkc(0.0, 1.0);
    }
    cplx* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, sizeof(cplx) * n));
    HIP_TRY(c, hipMemcpy(d, h.data(), sizeof(cplx) * n, hipMemcpyHostToDevice));
    c->tw[n] = d;
    c->ws_bytes += sizeof(cplx) * n;
    *out = d;
    return 0;
}

int get_ws(ofdft_ctx* c, const std::string& name, size_t bytes, void** out) {
    DevBuf& b = c->ws[name];
    if (b.bytes < bytes) {
        if (b.p) {
            if (!b.borrowed) HIP_TRY(c, hipFree(b.p));
            c->ws_bytes -= b.bytes;
        }
        b.p = nullptr;
        b.bytes = 0;
        b.borrowed = false;
        HIP_TRY(c, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
        c->ws_bytes += bytes;
        c->version++;        // captured graphs hold workspace addresses
    }
    *out = b.p;
    return 0;
}
int real_ws(ofdft_ctx* c, const char* name, real** out) {
    return get_ws(c, std::string("r:") + name, sizeof(real) * (size_t)c->npts, (void**)out);
}
int spec_ws(ofdft_ctx* c, const char* name, cplx** out) {
    return get_ws(c, std::string("s:") + name, sizeof(cplx) * (size_t)c->g.total, (void**)out);
}

// ---------------------------------------------------------------------------------- profiling
hipEvent_t prof_event(ofdft_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}
void prof_begin(ofdft_ctx* c, hipStream_t st, const char* name) {
    if (!c->profiling) return;
    hipEvent_t e = prof_event(c);
    if (!e) return;
    (void)hipEventRecord(e, st);
    c->pending.push_back({name, c->ev_used - 1, 0});
}
void prof_end(ofdft_ctx* c, hipStream_t st) {
    if (!c->profiling || c->pending.empty()) return;
    hipEvent_t e = prof_event(c);
    if (!e) return;
    (void)hipEventRecord(e, st);
    c->pending.back().b = c->ev_used - 1;
}
// after a stream sync: fold the pending event pairs into the per-name accumulators
void prof_collect(ofdft_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (p.b && hipEventElapsedTime(&ms, c->ev_pool[p.a], c->ev_pool[p.b]) == hipSuccess) {
            auto& a = c->prof[p.name];
            a.ms += ms;
            a.launches += 1;
        }
    }
    c->pending.clear();
    c->ev_used = 0;
}


// sized by what the active terms send at most across one geometry boundary (the buffers only ever grow): chain 0 carries
// {n^, (sqrt n)^} -> {vH, D_a n | grad n (3), lap} -> flux (1 | 3) -> divergence (1); chain 1 the Wang-Teter powers (1-2) and / or
// the six WGC99 spectra
size_t dist_buffer_bytes(ofdft_ctx* c, int chain) {
    const unsigned m = c->mask;
    const bool g = m & kGgaAny, h = m & OFDFT_HARTREE, vw = m & OFDFT_VW;
    const int ng = g ? (c->gga_split ? 1 : 3) : 0;
    const int nl = (gga_needs_laplacian(c) && c->gga_split) ? 1 : 0;      // lap n back, df/dL forth
    int narr;
    if (chain == 0) {
        narr = std::max(((h || g) ? 1 : 0) + (vw ? 1 : 0), (h ? 1 : 0) + ng + nl + (vw ? 1 : 0));
        narr = std::max(narr, ng + nl);
    } else {
        narr = ((m & OFDFT_WT_NL) ? (c->params[OFDFT_P_WT_ALPHA] != c->params[OFDFT_P_WT_BETA] ? 2 : 1) : 0) +
               ((m & OFDFT_WGC99_NL) ? 6 : 0);
    }
    if (narr < 1) narr = 1;
    return sizeof(cplx) * (size_t)c->g.total * narr;
}
// Two send and two receive buffers per chain, alternating from one exchange to the next (c->recv_parity[chain] = p, flipped by
// whoever issues the last chunk of an exchange, reset when an evaluation begins): a stage READS the receive buffer R[p] of the
// exchange before it and WRITES the send buffer S[p ^ 1], whose exchange lands in the peers' R[p ^ 1].  With the kz-chunked
// exchange a stage's first chunks are on their way back while its last chunks are still being read (and, on the send side,
// still being sent) -- in ONE buffer chunk k's result region would overlap chunk k + 1's input region whenever the two
// exchanges carry different numbers of spectra.  A buffer is reused two exchanges later, by which time every peer has consumed it
// (a peer sends exchange e + 1 only after it has read all of exchange e; transitively all of e - 1 has been delivered and read).
int dist_buffers(ofdft_ctx* c, int chain, cplx** send, cplx** recv) {
    const size_t bytes = dist_buffer_bytes(c, chain);
    const int p = c->recv_parity[chain];
    const char* sn[2][2] = {{"x:send0", "x:send0b"}, {"x:send1", "x:send1b"}};
    const char* rn[2][2] = {{"x:recv0", "x:recv0b"}, {"x:recv1", "x:recv1b"}};
    if (int rc = get_ws(c, sn[chain][p ^ 1], bytes, (void**)send)) return rc;
    return get_ws(c, rn[chain][p], bytes, (void**)recv);
}
// the receive buffer the exchange of what is being written now (S[p ^ 1]) lands in
int dist_recv_next(ofdft_ctx* c, int chain, cplx** recv) {
    const char* rn[2][2] = {{"x:recv0", "x:recv0b"}, {"x:recv1", "x:recv1b"}};
    return get_ws(c, rn[chain][c->recv_parity[chain] ^ 1], dist_buffer_bytes(c, chain), (void**)recv);
}


int dist_exchange(ofdft_ctx* c, cplx* send, cplx* recv, hipStream_t st) {
    if (!c->a2a) return fail(c, OFDFT_ESTATE, "slab-decomposed context without collectives: call ofdft_set_collectives first");
    const unsigned long long bytes = (unsigned long long)(sizeof(cplx) * (size_t)c->xg.nxl * c->xg.arr_sz);
    if (int rc = c->a2a(c->coll_user, send, recv, bytes, (void*)st)) return fail(c, OFDFT_EHIP, "all-to-all callback failed (%d)", rc);
    return 0;
}
// in-place sum over the ranks of host numbers (no-op on one rank)
int global_sums(ofdft_ctx* c, double* v, int n) {
    if (c->nranks == 1) return 0;
    if (!c->allreduce) return fail(c, OFDFT_ESTATE, "slab-decomposed context without collectives: call ofdft_set_collectives first");
    if (int rc = c->allreduce(c->coll_user, v, n)) return fail(c, OFDFT_EHIP, "all-reduce callback failed (%d)", rc);
    return 0;
}

}  // namespace eng

using namespace eng;

namespace {

struct ProfRec { const char* name; hipEvent_t a, b; };
void graph_drop(ofdft_ctx* c);
void ipc_release(ofdft_ctx* c);
void resident_give_up(ofdft_ctx* c);

// ---------------------------------------------------------------------------------- reductions
// copy `rows` x `ns` partials to the host and sum them in a fixed order
int fetch_partials(ofdft_ctx* c, int rows, int ns, double* sums, hipStream_t st) {
    if (rows > kRedBlocks) {     // many rows: reduce on the device first (fixed order), copy ns numbers
        OFDFT_REDUCE(c, st, c->d_partial, rows, ns, c->d_reduced);
        HIP_TRY(c, hipMemcpyAsync(c->h_partial, c->d_reduced, sizeof(double) * ns, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        for (int s = 0; s < ns; ++s) sums[s] = c->h_partial[s];
        return 0;
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_partial, c->d_partial, sizeof(double) * rows * ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    for (int s = 0; s < ns; ++s) {
        long double t = 0.0L;
        for (int r = 0; r < rows; ++r) t += (long double)c->h_partial[(size_t)r * ns + s];
        sums[s] = (double)t;
    }
    return 0;
}

int device_sum(ofdft_ctx* c, const real* a, bool square, double* out, hipStream_t st) {
    const int blocks = grid_for(c->npts / 2 + 1, kRedThreads, kRedBlocks);
    if (square)
        OFDFT_LAUNCH(c, st, "sum", (sum_kernel<true>), dim3(blocks), dim3(kRedThreads), 0, a, c->npts, c->d_partial);
    else
        OFDFT_LAUNCH(c, st, "sum", (sum_kernel<false>), dim3(blocks), dim3(kRedThreads), 0, a, c->npts, c->d_partial);
    return fetch_partials(c, blocks, 1, out, st);
}

// ---------------------------------------------------------------------------------- WGC tables
void wgc_series_coeffs(int nt, std::vector<double>& A, std::vector<double>& B) {
    // functionals.py:817-843
    std::vector<double> a(nt + 1, 0.0), b(nt, 0.0);
    a[0] = 3.0;
    for (int idx = 1; idx <= nt; ++idx) {
        const int i = idx - 1;
        double s = 0.0;
        for (int j = -1; j < i; ++j) s += -3.0 * a[j + 1] / (4.0 * (i - j + 1) * (i - j + 1) - 1.0);
        a[idx] = s;
    }
    A.assign(nt, 0.0);
    A[0] = a[1] - 1.0;
    for (int i = 1; i < nt; ++i) A[i] = a[i + 1];
    b[0] = 1.0;
    for (int i = 1; i < nt; ++i) {
        double s = 0.0;
        for (int j = 0; j < i; ++j) s += b[j] / (4.0 * (i - j) * (i - j) - 1.0);
        b[i] = s;
    }
    B.assign(nt, 0.0);
    B[0] = 0.0;
    if (nt > 1) B[1] = b[1] - 3.0;
    for (int i = 2; i < nt; ++i) B[i] = b[i];
}

// series constants of the WGC99 kernel for n_ref = kappa round(N_e) / vol (functionals.py:845-939); uploads the
// coefficient arrays
int wgc_series_setup(ofdft_ctx* c, long long nel_rounded, hipStream_t st, WgcSeries* out) {
    const double al = c->params[OFDFT_P_WGC_ALPHA], be = c->params[OFDFT_P_WGC_BETA];
    const double ga = c->params[OFDFT_P_WGC_GAMMA], ka = c->params[OFDFT_P_WGC_KAPPA];
    const double nref = ka * ((double)nel_rounded / c->vol);
    const int nt = 100;
    const double u = 3.0 * (al + be) - ga / 2.0, v = u * u - 36.0 * al * be;
    std::vector<double> A, B, coef(2 * nt);
    wgc_series_coeffs(nt, A, B);
    double Sd = 0.0, Ss = 0.0;
    for (int i = 0; i < nt; ++i) {
        const double da = (u + 2.0 * i) * (u + 2.0 * i) - v, db = (u - 2.0 * i) * (u - 2.0 * i) - v;
        coef[i] = A[i] / da;
        coef[nt + i] = B[i] / db;
        Sd += coef[i] - coef[nt + i];
        Ss += i * (coef[i] + coef[nt + i]);
    }
    Ss *= -2.0;
    const double sgn = (u > 0) - (u < 0);
    WgcSeries s{};
    s.u = u;
    s.v = v;
    if (v > 0) {
        const double rv = std::sqrt(v);
        s.c1 = sgn * ((rv - u) * Sd + Ss);
        s.c2 = sgn * ((rv + u) * Sd - Ss) / (2.0 * rv);
    } else if (v == 0) {
        s.c1 = sgn * Sd;
        s.c2 = sgn * (Ss - u * Sd);
    } else {
        s.c1 = sgn * Sd;
        s.c2 = sgn * (Ss - u * Sd) / std::sqrt(-v);
    }
    s.gamma = ga;
    s.nref = nref;
    s.pref = 20.0 * std::pow(nref, kFiveThirds - al - be);
    s.inv2kf = 1.0 / (2.0 * std::cbrt(3.0 * kPi * kPi * nref));
    s.nt = nt;
    if (!c->d_wgc_coef) HIP_TRY(c, hipMalloc((void**)&c->d_wgc_coef, sizeof(double) * 2 * nt));
    HIP_TRY(c, hipMemcpyAsync(c->d_wgc_coef, coef.data(), sizeof(double) * 2 * nt, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipStreamSynchronize(st));   // coef is a stack-backed vector
    s.ca = c->d_wgc_coef;
    s.cb = c->d_wgc_coef + nt;
    *out = s;
    return 0;
}

}  // namespace
namespace eng {
// (also called by the persistent small-grid kernel's launcher, resident.hip)
int ensure_wgc_tables(ofdft_ctx* c, long long nel_rounded, hipStream_t st, double* nref_out) {
    const double nref = c->params[OFDFT_P_WGC_KAPPA] * ((double)nel_rounded / c->vol);
    *nref_out = nref;
    if (c->wgc_valid && c->wgc_key_nel == nel_rounded) return 0;
    WgcSeries s{};
    if (int rc = wgc_series_setup(c, nel_rounded, st, &s)) return rc;
    cplx* t01;
    const size_t tb = sizeof(real) * (size_t)c->g.total;
    if (int rc = get_ws(c, "t:wgc", 3 * tb, (void**)&t01)) return rc;      // (w0, K1) per k-point, then K2 per k-point (wgc_tab)
    real* t2 = reinterpret_cast<real*>(t01 + c->g.total);
    TabMap tm{c->nranks > 1 ? 1 : 0, c->xg.nyl, c->g.nzm, c->xg.arr_sz};
    tm.nch = c->xc.n;
    tm.n0g = c->n0g;
    tm.nrem = c->xg.nrem;
    for (int k = 0; k <= c->xc.n; ++k) tm.kb[k] = c->xc.kb[k];
    if (c->xc.n <= 1) tm.kb[1] = c->xg.nb;
    OFDFT_LAUNCH(c, st, "wgc_table", wgc_table_kernel, dim3(grid_for(c->g.total, 256, 4096)), dim3(256), 0, t01, t2, c->kg, s, tm);
    c->wgc_ck = (3.0 - c->params[OFDFT_P_WGC_GAMMA]) / (3.0 * nref);
    c->wgc_key_nel = nel_rounded;
    c->wgc_valid = true;
    return 0;
}
}  // namespace eng
namespace {

// ---------------------------------------------------------------------------------- combine / energies
// launch the combine kernel and turn its partial sums into per-term energies
// defer (the closure of the unfused / chirp-z pipelines, round 4): nothing waits here -- the sums are reduced on the device in a fixed
// order (c->d_reduced, mirrored into the pinned host block by the reduce kernel), chi_grad forms mu from them on the device and the
// caller turns the mirror into energies after ITS one synchronisation (an evaluation used to wait four times: sum chi^2, the GGA
// sums, the combine sums, the end).  Not for the two-pass stabilised WT-style functional, whose weights pass through the host.
int finish_terms(ofdft_ctx* c, const CombineArgs& ca, const double* pbe_sums, double* E_terms, double* vn_int,
                 hipStream_t st, bool defer = false) {
    const unsigned mask = c->mask;
    const long long npts = c->npts;
    const int blocks = grid_for(npts / 2 + 1, kRedThreads, kRedBlocks);
    CombineArgs cb = ca;     // every pointer valid: unused inputs alias the density (their terms are masked off)
    const real* d = cb.n;
    if (!cb.vext) cb.vext = d;
    if (!cb.vh) cb.vh = d;
    if (!cb.lap_s) cb.lap_s = d;
    if (!cb.conv_b) cb.conv_b = d;
    if (!cb.u0) cb.u0 = cb.u1 = cb.u2 = cb.gA = cb.gB = cb.gC = d;
    if (!cb.dfdn) cb.dfdn = cb.div = d;
    double sums[kCombineScalars];
    double wts_f = 1.0;
    if (wts_active(c)) {      // first pass: energies only -> X = T_NL / T_TF -> weights of the two potentials
        CombineArgs c1 = cb;
        c1.v_out = nullptr;
        OFDFT_LAUNCH(c, st, "combine", combine_kernel, dim3(blocks), dim3(kRedThreads), 0, c1, c->d_partial);
        if (int rc = fetch_partials(c, blocks, kCombineScalars, sums, st)) return rc;
        const double X = sums[4] / sums[2];
        wts_f = std::exp(X);
        cb.w_tf = (real)(wts_f * (1.0 - X));
        cb.w_nl = (real)wts_f;
    }
    OFDFT_LAUNCH(c, st, "combine", combine_kernel, dim3(blocks), dim3(kRedThreads), 0, cb, c->d_partial);
    if (defer && !wts_active(c)) {
        OFDFT_REDUCE(c, st, c->d_partial, blocks, kCombineScalars, c->d_reduced, c->h_partial);
        return 0;
    }
    if (int rc = fetch_partials(c, blocks, kCombineScalars, sums, st)) return rc;
    if (wts_active(c)) {
        sums[2] *= wts_f;
        sums[4] = 0.0;
    }
    const double dV = c->dV;
    if (mask & OFDFT_ION_ELECTRON) E_terms[0] = sums[0] * dV;
    if (mask & OFDFT_HARTREE) E_terms[1] = sums[1] * dV;
    if (mask & OFDFT_TF) E_terms[2] = sums[2] * dV;
    if (mask & OFDFT_VW) E_terms[3] = sums[3] * dV;
    if (mask & OFDFT_WT_NL) E_terms[4] = sums[4] * dV;
    if (mask & OFDFT_WGC99_NL) E_terms[5] = sums[5] * dV;
    if (mask & OFDFT_LDA_X) E_terms[6] = sums[6] * dV;
    // one local correlation flavour is expected; if several are set their sum is split evenly
    {
        int nc = 0;
        for (int b = 7; b <= 9; ++b) nc += (mask >> b) & 1;
        for (int b = 7; b <= 9; ++b)
            if ((mask >> b) & 1) E_terms[b] = sums[7] * dV / nc;
    }
    if (mask & OFDFT_PBE_X) E_terms[10] = pbe_sums[0] * dV;
    if (mask & OFDFT_PBE_C) E_terms[11] = pbe_sums[1] * dV;
    if (mask & OFDFT_GGA_K) E_terms[12] = pbe_sums[2] * dV;
    if (mask & OFDFT_VWGTF) E_terms[13] = sums[9] * dV;
    *vn_int = sums[8] * dV;
    return 0;
}


// ---------------------------------------------------------------------------------- the energy pipeline
// den: density on device.  Fills E_terms (host), writes v_out (device, may be NULL), returns sum(v n) dV.
// nel_known > 0: the caller knows N_e = int n (the closure: n = N_e chi^2 / int chi^2) -- no device sum of the density;
// defer: see finish_terms (the GGA sums are reduced on the device as well)
int run_terms_unfused(ofdft_ctx* c, const real* den, const real* vext, double* E_terms, real* v_out,
                      double* vn_int, hipStream_t st, double nel_known = 0.0, bool defer = false) {
    const unsigned mask = c->mask;
    const long long npts = c->npts;
    const double inv_n = 1.0 / (double)npts;
    const int pw_grid = grid_for(npts / 2 + 1);
    const int sp_grid = grid_for(c->g.total);
    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
    if ((mask & OFDFT_ION_ELECTRON) && !vext) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");

    defer = defer && !wts_active(c);
    double nsum = 0.0;
    if (nel_known <= 0.0 && (mask & (OFDFT_WT_NL | OFDFT_WGC99_NL | OFDFT_VWGTF)))
        if (int rc = device_sum(c, den, false, &nsum, st)) return rc;
    const double nel = nel_known > 0.0 ? nel_known : nsum * inv_n * c->vol;       // mean(den) * vol   functionals.py:634,646,952

    CombineArgs ca{};
    ca.n = den;
    ca.vext = vext;
    ca.v_out = v_out;
    ca.npts = npts;
    ca.mask = mask;
    ca.gtf_kind = (int)c->params[OFDFT_P_VWGTF_KIND];
    ca.gtf_inv_n0 = (mask & OFDFT_VWGTF) ? c->vol / (double)std::llround(nel) : 0.0;      // n0 = round(N_e) / vol (functionals.py:268-270)
    double pbe_sums[kPbeScalars] = {0.0, 0.0, 0.0};

    cplx *s0 = nullptr, *s1 = nullptr, *s2 = nullptr, *s3 = nullptr;
    if (int rc = spec_ws(c, "s0", &s0)) return rc;

    // The transforms of the density-derived inputs are independent of one another, and so are the inverse transforms of the
    // convolution results: they run as BATCHES (rfftn_internal_multi / irfftn_internal_multi: on the chirp-z path one launch per
    // pass for up to kBsBatch arrays -- small odd grids are bound by their launch count).  Phase A: pointwise inputs + forward
    // batch; B: spectral multiplies; C: inverse batches; D: the GGA mid stage and its flux / divergence transforms.
    const bool has_n = mask & (OFDFT_HARTREE | kGgaAny), has_h = mask & OFDFT_HARTREE, has_g = mask & kGgaAny, has_vw = mask & OFDFT_VW,
               has_wt = mask & OFDFT_WT_NL;
    const double wal = c->params[OFDFT_P_WT_ALPHA], wbe = c->params[OFDFT_P_WT_BETA];
    const bool wt2 = has_wt && wal != wbe;
    real *t_sqrt = nullptr, *t_pb = nullptr, *t_pa = nullptr;
    cplx *s_vw = nullptr, *s_wb = nullptr, *s_wa = nullptr, *s4 = nullptr;
    real *vh = nullptr, *gx = nullptr, *gy = nullptr, *gz = nullptr, *dfdn = nullptr, *dv = nullptr, *lapn = nullptr, *lap = nullptr,
         *cb = nullptr, *cva = nullptr;
    const bool lapl = has_g && gga_needs_laplacian(c);
    // chirp-z path: the x transforms and the spectral multiplies of every convolution below run as ONE kernel per convolution
    // (bluestein_xmix); the Laplacian-dependent GGA members keep the three-pass form
    const bool xm = bluestein_xmix_ok(c) && !lapl;
    // ---- A: forward batch
    {
        const real* fin[4];
        cplx* fout[4];
        int nf = 0;
        // chirp-z path with the fused x pass (xm): sqrt n and the powers are formed by the r2c pass as it loads the density
        // (BsPrep) -- no map launches, no intermediate arrays
        BsPrep prep;
        if (has_n) {
            fin[nf] = den;
            fout[nf++] = s0;
        }
        if (has_vw) {
            if (int rc = spec_ws(c, "svw", &s_vw)) return rc;
            if (xm) {
                prep.kind[nf] = BS_PREP_SQRT;
                fin[nf] = den;
            } else {
                if (int rc = real_ws(c, "t0", &t_sqrt)) return rc;
                OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_SQRT>), dim3(pw_grid), dim3(256), 0, den, t_sqrt, npts, 0.0);
                fin[nf] = t_sqrt;
            }
            fout[nf++] = s_vw;
        }
        if (has_wt) {
            if (int rc = spec_ws(c, "swb", &s_wb)) return rc;
            if (xm) {
                prep.kind[nf] = BS_PREP_POW0;
                prep.e[nf] = wbe;
                fin[nf] = den;
            } else {
                if (int rc = real_ws(c, "t1", &t_pb)) return rc;
                OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, t_pb, npts, wbe);
                fin[nf] = t_pb;
            }
            fout[nf++] = s_wb;
            if (wt2) {
                if (int rc = spec_ws(c, "swa", &s_wa)) return rc;
                if (xm) {
                    prep.kind[nf] = BS_PREP_POW0;
                    prep.e[nf] = wal;
                    fin[nf] = den;
                } else {
                    if (int rc = real_ws(c, "t2", &t_pa)) return rc;
                    OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, t_pa, npts, wal);
                    fin[nf] = t_pa;
                }
                fout[nf++] = s_wa;
            }
        }
        if (nf)
            if (int rc = (xm ? bluestein_fwd_zy_multi(c, fin, fout, nf, st, &prep) : rfftn_internal_multi(c, fin, fout, nf, st))) return rc;
    }
    // ---- B: spectral multiplies; C: inverse batches (at most kBsBatch arrays each)
    {
        cplx* iin[8];
        real* iout[8];
        int ni = 0;
        if (has_n && (has_h || has_g))
            if (int rc = spec_ws(c, "s1", &s1)) return rc;
        if (has_g) {
            if (int rc = spec_ws(c, "s2", &s2)) return rc;
            if (int rc = spec_ws(c, "s3", &s3)) return rc;
            if (int rc = real_ws(c, "gx", &gx)) return rc;
            if (int rc = real_ws(c, "gy", &gy)) return rc;
            if (int rc = real_ws(c, "gz", &gz)) return rc;
            if (int rc = real_ws(c, "dfdn", &dfdn)) return rc;
            if (int rc = real_ws(c, "div", &dv)) return rc;
        }
        cplx* svh = nullptr;            // spectrum of the Hartree potential (its own array when s1 carries a gradient component)
        if (has_h) {
            svh = s1;
            if (has_g)
                if (int rc = spec_ws(c, "svh", &svh)) return rc;       // (s1 carries a gradient component then)
            if (int rc = real_ws(c, "vh", &vh)) return rc;
            if (!xm)
                OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_HARTREE>), dim3(sp_grid), dim3(256), 0, s0, svh, c->kg, 0.0, 0.0);
            else if (!has_g) {
                const cplx* xi[1] = {s0};
                cplx* xo[1] = {svh};
                if (int rc = bluestein_xmix<1, 1>(c, xi, xo, MixScale<SPEC_HARTREE>{c->kg, (real)0.0, (real)0.0}, st)) return rc;
            }
            iin[ni] = svh;
            iout[ni++] = vh;
            ca.vh = vh;
        }
        if (has_g) {
            if (!xm)
                OFDFT_LAUNCH(c, st, "spec_grad", spec_grad_kernel, dim3(sp_grid), dim3(256), 0, s0, s1, s2, s3, c->kg);
            else if (has_h) {          // n^ -> v_H^ and the three gradient components: one forward-x, four inverse-x
                const cplx* xi[1] = {s0};
                cplx* xo[4] = {svh, s1, s2, s3};
                if (int rc = bluestein_xmix<1, 4>(c, xi, xo, MixDensity<true, true>{c->kg}, st)) return rc;
            } else {
                const cplx* xi[1] = {s0};
                cplx* xo[3] = {s1, s2, s3};
                if (int rc = bluestein_xmix<1, 3>(c, xi, xo, MixDensity<false, true>{c->kg}, st)) return rc;
            }
            iin[ni] = s1; iout[ni++] = gx;
            iin[ni] = s2; iout[ni++] = gy;
            iin[ni] = s3; iout[ni++] = gz;
            if (lapl) {              // lap n = F^-1[-k^2 n^]  (reduced Laplacian q, functional_tools.py:271-287)
                if (int rc = real_ws(c, "lapn", &lapn)) return rc;
                if (int rc = spec_ws(c, "s4", &s4)) return rc;
                OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_LAPLACE>), dim3(sp_grid), dim3(256), 0, s0, s4, c->kg, 0.0, 0.0);
                iin[ni] = s4; iout[ni++] = lapn;
            }
        }
        if (has_vw) {
            if (int rc = real_ws(c, "lap", &lap)) return rc;
            if (!xm)
                OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_LAPLACE>), dim3(sp_grid), dim3(256), 0, s_vw, s_vw, c->kg, 0.0, 0.0);
            else {
                const cplx* xi[1] = {s_vw};
                cplx* xo[1] = {s_vw};
                if (int rc = bluestein_xmix<1, 1>(c, xi, xo, MixScale<SPEC_LAPLACE>{c->kg, (real)0.0, (real)0.0}, st)) return rc;
            }
            iin[ni] = s_vw; iout[ni++] = lap;
            ca.lap_s = lap;
        }
        if (has_wt) {
            const double nbar = nel / c->vol;                                    // functionals.py:646-647
            const double kf = std::cbrt(3.0 * kPi * kPi * nbar);
            const double pref = 5.0 / (9.0 * wal * wbe * std::pow(nbar, wal + wbe - kFiveThirds));
            if (int rc = real_ws(c, "conv_b", &cb)) return rc;
            if (!xm)
                OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_LINDHARD>), dim3(sp_grid), dim3(256), 0, s_wb, s_wb, c->kg, pref,
                                   1.0 / (2.0 * kf));
            else {
                const cplx* xi[1] = {s_wb};
                cplx* xo[1] = {s_wb};
                if (int rc = bluestein_xmix<1, 1>(c, xi, xo, MixScale<SPEC_LINDHARD>{c->kg, (real)pref, (real)(1.0 / (2.0 * kf))}, st)) return rc;
            }
            iin[ni] = s_wb; iout[ni++] = cb;
            ca.conv_b = cb;
            ca.conv_a = nullptr;
            if (wt2) {
                if (int rc = real_ws(c, "conv_a", &cva)) return rc;
                if (!xm)
                    OFDFT_LAUNCH(c, st, "spec_scale", (spec_scale_kernel<SPEC_LINDHARD>), dim3(sp_grid), dim3(256), 0, s_wa, s_wa, c->kg, pref,
                                       1.0 / (2.0 * kf));
                else {
                    const cplx* xi[1] = {s_wa};
                    cplx* xo[1] = {s_wa};
                    if (int rc = bluestein_xmix<1, 1>(c, xi, xo, MixScale<SPEC_LINDHARD>{c->kg, (real)pref, (real)(1.0 / (2.0 * kf))}, st)) return rc;
                }
                iin[ni] = s_wa; iout[ni++] = cva;
                ca.conv_a = cva;
            }
            ca.wt_alpha = wal;
            ca.wt_beta = wbe;
            ca.wt_nbar_pa = std::pow(nbar, wal);
            ca.wt_is_56 = (wal == kFiveSixths && wbe == kFiveSixths) ? 1 : 0;
        }
        for (int b0 = 0; b0 < ni; b0 += 4)
            if (int rc = (xm ? bluestein_inv_yz_multi(c, iin + b0, iout + b0, std::min(4, ni - b0), inv_n, st)
                             : irfftn_internal_multi(c, iin + b0, iout + b0, std::min(4, ni - b0), inv_n, st)))
                return rc;
    }
    // ---- D: GGA mid stage (needs grad n [, lap n] in real space), flux forward batch, divergence inverse
    if (has_g) {
        const int blocks = grid_for(npts / 2 + 1, kRedThreads, kRedBlocks);
        OFDFT_LAUNCH(c, st, "pbe", pbe_kernel, dim3(blocks), dim3(kRedThreads), 0, den, gx, gy, gz, dfdn, npts,
                           gga_sel(c), c->d_partial, lapn);
        if (defer) {
            OFDFT_REDUCE(c, st, c->d_partial, blocks, kPbeScalars, c->d_reduced + kCombineScalars, c->h_partial + kCombineScalars);
        } else if (int rc = fetch_partials(c, blocks, kPbeScalars, pbe_sums, st)) {
            return rc;
        }
        {
            cplx* sg[4] = {s1, s2, s3, s4};
            const real* rg[4] = {gx, gy, gz, lapn};
            if (int rc = (xm ? bluestein_fwd_zy_multi(c, rg, sg, 3, st) : rfftn_internal_multi(c, rg, sg, lapl ? 4 : 3, st))) return rc;
        }
        if (xm) {
            const cplx* xi[3] = {s1, s2, s3};
            cplx* xo[1] = {s0};
            real* dvo[1] = {dv};
            if (int rc = bluestein_xmix<3, 1>(c, xi, xo, MixDiv{c->kg}, st)) return rc;
            if (int rc = bluestein_inv_yz_multi(c, xo, dvo, 1, inv_n, st)) return rc;
        } else {
        OFDFT_LAUNCH(c, st, "spec_div", spec_div_kernel, dim3(sp_grid), dim3(256), 0, s1, s2, s3, s0, c->kg);
        if (lapl)        // v += lap(df/dL): the combine forms v += df/dn - 2 div, so div -= lap(df/dL) / 2, i.e. s0 += k^2 (df/dL)^ / 2
            OFDFT_LAUNCH(c, st, "spec_scale", spec_add_lap_kernel, dim3(sp_grid), dim3(256), 0, (const cplx*)s4, s0, c->kg, 0.5);
        if (int rc = irfftn_internal(c, s0, dv, inv_n, st)) return rc;
        }
        ca.dfdn = dfdn;
        ca.div = dv;
    }
    if (mask & OFDFT_WGC99_NL) {
        const double al = c->params[OFDFT_P_WGC_ALPHA], be = c->params[OFDFT_P_WGC_BETA];
        const long long nel_r = std::llround(nel);                           // functionals.py:952
        double nref;
        if (int rc = ensure_wgc_tables(c, nel_r, st, &nref)) return rc;
        real *t0, *t1, *t2, *o[6];
        if (int rc = real_ws(c, "t0", &t0)) return rc;
        if (int rc = real_ws(c, "t1", &t1)) return rc;
        if (int rc = real_ws(c, "t2", &t2)) return rc;
        const char* names[6] = {"u0", "u1", "u2", "gA", "gB", "gC"};
        for (int i = 0; i < 6; ++i)
            if (int rc = real_ws(c, names[i], &o[i])) return rc;
        if (int rc = spec_ws(c, "s1", &s1)) return rc;
        if (int rc = spec_ws(c, "s2", &s2)) return rc;
        const MixWgc wmix = wgc_tab(c);
        for (int pass = 0; pass < 2; ++pass) {
            cplx* sw[3] = {s0, s1, s2};
            const real* tw3[3] = {t0, t1, t2};
            if (xm) {          // A = n^e, B = A theta, C = A theta^2 / 2 formed by the r2c pass from the density (BsPrep)
                const real* d3[3] = {den, den, den};
                BsPrep prep;
                for (int k = 0; k < 3; ++k) {
                    prep.kind[k] = BS_PREP_POW0 + k;
                    prep.e[k] = pass == 0 ? be : al;
                }
                prep.nref = nref;
                if (int rc = bluestein_fwd_zy_multi(c, d3, sw, 3, st, &prep)) return rc;
                if (int rc = bluestein_xmix<3, 3>(c, sw, sw, wmix, st)) return rc;
                if (int rc = bluestein_inv_yz_multi(c, sw, o + 3 * pass, 3, inv_n, st)) return rc;
                continue;
            }
            OFDFT_LAUNCH(c, st, "wgc_prep", wgc_prep_kernel, dim3(pw_grid), dim3(256), 0, den, t0, t1, t2, npts, pass == 0 ? be : al,
                               nref);
            if (int rc = rfftn_internal_multi(c, tw3, sw, 3, st)) return rc;
            OFDFT_LAUNCH(c, st, "spec_wgc_mix", spec_wgc_mix_kernel, dim3(sp_grid), dim3(256), 0, s0, s1, s2, wmix.t01, wmix.t2, wmix.ck, c->g.total);
            if (int rc = irfftn_internal_multi(c, sw, o + 3 * pass, 3, inv_n, st)) return rc;
        }
        ca.u0 = o[0]; ca.u1 = o[1]; ca.u2 = o[2];
        ca.gA = o[3]; ca.gB = o[4]; ca.gC = o[5];
        ca.wgc_alpha = al;
        ca.wgc_beta = be;
        ca.nref = nref;
        ca.wgc_sum_53 = (std::fabs(al + be - kFiveThirds) < 4e-16) ? 1 : 0;
    }
    return finish_terms(c, ca, pbe_sums, E_terms, vn_int, st, defer);
}


// Fused pipeline (power-of-two grids): every spectral multiply rides inside the x pass (xfused_kernel), so a
// forward/inverse FFT pair costs  z + y + (x fused) + y + z  = 5 passes instead of 6 + a multiply pass.
int run_terms_fast(ofdft_ctx* c, const real* den, const real* vext, double* E_terms, real* v_out,
                   double* vn_int, hipStream_t st) {
    const unsigned mask = c->mask;
    const long long npts = c->npts;
    const double inv_n = 1.0 / (double)npts;
    const int pw_grid = grid_for(npts / 2 + 1);
    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
    if ((mask & OFDFT_ION_ELECTRON) && !vext) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");

    double nsum = 0.0;
    if (mask & (OFDFT_WT_NL | OFDFT_WGC99_NL | OFDFT_VWGTF))
        if (int rc = device_sum(c, den, false, &nsum, st)) return rc;
    const double nel = nsum * inv_n * c->vol;

    CombineArgs ca{};
    ca.n = den;
    ca.vext = vext;
    ca.v_out = v_out;
    ca.npts = npts;
    ca.mask = mask;
    ca.gtf_kind = (int)c->params[OFDFT_P_VWGTF_KIND];
    ca.gtf_inv_n0 = (mask & OFDFT_VWGTF) ? c->vol / (double)std::llround(nel) : 0.0;      // n0 = round(N_e) / vol (functionals.py:268-270)
    double pbe_sums[kPbeScalars] = {0.0, 0.0, 0.0};
    cplx* s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const char* sn[5] = {"s0", "s1", "s2", "s3", "s4"};
    if (int rc = spec_ws(c, sn[0], &s[0])) return rc;

    const bool has_h = mask & OFDFT_HARTREE, has_g = mask & kGgaAny;
    if (has_h || has_g) {
        if (int rc = fwd_zy(c, den, s[0], st)) return rc;
        XfIo io{};
        io.in[0] = s[0];
        int no = 0;
        real *vh = nullptr, *gr[3] = {nullptr, nullptr, nullptr}, *dfdn = nullptr, *dv = nullptr;
        if (has_h) {
            if (int rc = spec_ws(c, sn[1], &s[1])) return rc;
            if (int rc = real_ws(c, "vh", &vh)) return rc;
            io.out[no++] = s[1];
        }
        if (has_g) {
            const char* gn[3] = {"gx", "gy", "gz"};
            for (int k = 0; k < 3; ++k) {
                if (int rc = spec_ws(c, sn[2 + k], &s[2 + k])) return rc;
                if (int rc = real_ws(c, gn[k], &gr[k])) return rc;
                io.out[no++] = s[2 + k];
            }
            if (int rc = real_ws(c, "dfdn", &dfdn)) return rc;
            if (int rc = real_ws(c, "div", &dv)) return rc;
        }
        int rc;
        if (has_h && has_g) rc = xfused<1, 4>(c, io, MixDensity<true, true>{c->kg}, st, "xfused_n");
        else if (has_h) rc = xfused<1, 1>(c, io, MixDensity<true, false>{c->kg}, st, "xfused_n");
        else rc = xfused<1, 3>(c, io, MixDensity<false, true>{c->kg}, st, "xfused_n");
        if (rc) return rc;
        if (has_h) {
            if ((rc = inv_yz(c, s[1], vh, inv_n, st))) return rc;
            ca.vh = vh;
        }
        if (has_g) {
            for (int k = 0; k < 3; ++k)
                if ((rc = inv_yz(c, s[2 + k], gr[k], inv_n, st))) return rc;
            const int blocks = grid_for(npts / 2 + 1, kRedThreads, kRedBlocks);
            OFDFT_LAUNCH(c, st, "pbe", pbe_kernel, dim3(blocks), dim3(kRedThreads), 0, den, gr[0], gr[1], gr[2], dfdn, npts,
                         gga_sel(c), c->d_partial);
            if ((rc = fetch_partials(c, blocks, kPbeScalars, pbe_sums, st))) return rc;
            for (int k = 0; k < 3; ++k)
                if ((rc = fwd_zy(c, gr[k], s[2 + k], st))) return rc;
            XfIo dio{};
            for (int k = 0; k < 3; ++k) dio.in[k] = s[2 + k];
            dio.out[0] = s[0];
            if ((rc = xfused<3, 1>(c, dio, MixDiv{c->kg}, st, "xfused_div"))) return rc;
            if ((rc = inv_yz(c, s[0], dv, inv_n, st))) return rc;
            ca.dfdn = dfdn;
            ca.div = dv;
        }
    }
    if (mask & OFDFT_VW) {
        real *tmp, *lap;
        if (int rc = real_ws(c, "t0", &tmp)) return rc;
        if (int rc = real_ws(c, "lap", &lap)) return rc;
        OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_SQRT>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, 0.0);
        if (int rc = fwd_zy(c, tmp, s[0], st)) return rc;
        XfIo io{};
        io.in[0] = s[0];
        io.out[0] = s[0];
        if (int rc = xfused<1, 1>(c, io, MixScale<SPEC_LAPLACE>{c->kg, 0.0, 0.0}, st, "xfused_lap")) return rc;
        if (int rc = inv_yz(c, s[0], lap, inv_n, st)) return rc;
        ca.lap_s = lap;
    }
    if (mask & OFDFT_WT_NL) {
        const double al = c->params[OFDFT_P_WT_ALPHA], be = c->params[OFDFT_P_WT_BETA];
        const double nbar = nel / c->vol;
        const double kf = std::cbrt(3.0 * kPi * kPi * nbar);
        const double pref = 5.0 / (9.0 * al * be * std::pow(nbar, al + be - kFiveThirds));
        real *tmp, *cb;
        if (int rc = real_ws(c, "t0", &tmp)) return rc;
        if (int rc = real_ws(c, "conv_b", &cb)) return rc;
        const MixScale<SPEC_LINDHARD> lind{c->kg, (real)pref, (real)(1.0 / (2.0 * kf))};
        XfIo io{};
        io.in[0] = s[0];
        io.out[0] = s[0];
        OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, be);
        if (int rc = fwd_zy(c, tmp, s[0], st)) return rc;
        if (int rc = xfused<1, 1>(c, io, lind, st, "xfused_lind")) return rc;
        if (int rc = inv_yz(c, s[0], cb, inv_n, st)) return rc;
        ca.conv_b = cb;
        ca.conv_a = nullptr;
        if (al != be) {
            real* cva;
            if (int rc = real_ws(c, "conv_a", &cva)) return rc;
            OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_POW>), dim3(pw_grid), dim3(256), 0, den, tmp, npts, al);
            if (int rc = fwd_zy(c, tmp, s[0], st)) return rc;
            if (int rc = xfused<1, 1>(c, io, lind, st, "xfused_lind")) return rc;
            if (int rc = inv_yz(c, s[0], cva, inv_n, st)) return rc;
            ca.conv_a = cva;
        }
        ca.wt_alpha = al;
        ca.wt_beta = be;
        ca.wt_nbar_pa = std::pow(nbar, al);
        ca.wt_is_56 = (al == kFiveSixths && be == kFiveSixths) ? 1 : 0;
    }
    if (mask & OFDFT_WGC99_NL) {
        const double al = c->params[OFDFT_P_WGC_ALPHA], be = c->params[OFDFT_P_WGC_BETA];
        const long long nel_r = std::llround(nel);
        double nref;
        if (int rc = ensure_wgc_tables(c, nel_r, st, &nref)) return rc;
        real *t[3], *o[6];
        const char* tn[3] = {"t0", "t1", "t2"};
        const char* names[6] = {"u0", "u1", "u2", "gA", "gB", "gC"};
        for (int i = 0; i < 3; ++i)
            if (int rc = real_ws(c, tn[i], &t[i])) return rc;
        for (int i = 0; i < 6; ++i)
            if (int rc = real_ws(c, names[i], &o[i])) return rc;
        for (int i = 1; i < 3; ++i)
            if (int rc = spec_ws(c, sn[i], &s[i])) return rc;
        const MixWgc mix = wgc_tab(c);
        XfIo io{};
        for (int i = 0; i < 3; ++i) {
            io.in[i] = s[i];
            io.out[i] = s[i];
        }
        for (int pass = 0; pass < 2; ++pass) {
            OFDFT_LAUNCH(c, st, "wgc_prep", wgc_prep_kernel, dim3(pw_grid), dim3(256), 0, den, t[0], t[1], t[2], npts,
                         pass == 0 ? be : al, nref);
            for (int i = 0; i < 3; ++i)
                if (int rc = fwd_zy(c, t[i], s[i], st)) return rc;
            if (int rc = xfused_wgc(c, io, mix, st, "xfused_wgc")) return rc;
            for (int i = 0; i < 3; ++i)
                if (int rc = inv_yz(c, s[i], o[3 * pass + i], inv_n, st)) return rc;
        }
        ca.u0 = o[0]; ca.u1 = o[1]; ca.u2 = o[2];
        ca.gA = o[3]; ca.gB = o[4]; ca.gC = o[5];
        ca.wgc_alpha = al;
        ca.wgc_beta = be;
        ca.nref = nref;
        ca.wgc_sum_53 = (std::fabs(al + be - kFiveThirds) < 4e-16) ? 1 : 0;
    }
    return finish_terms(c, ca, pbe_sums, E_terms, vn_int, st);
}

// *deferred (out): the unfused pipeline ran in the host-free form (finish_terms: defer) -- E_terms / vn_int are NOT filled yet
int run_terms(ofdft_ctx* c, const real* den, const real* vext, double* E_terms, real* v_out, double* vn_int,
              hipStream_t st, double nel_known = 0.0, bool* deferred = nullptr) {
    if (deferred) *deferred = false;
    if (c->fast && !c->force_unfused && !gga_needs_laplacian(c)) return run_terms_fast(c, den, vext, E_terms, v_out, vn_int, st);
    const bool defer = deferred && !wts_active(c);
    if (deferred) *deferred = defer;
    return run_terms_unfused(c, den, vext, E_terms, v_out, vn_int, st, nel_known, defer);
}


#include "engine_zfused.inc.h"

// ====================================================================================== C ABI
extern "C" {

int ofdft_create_dist(ofdft_ctx** out, int n0g, int n1g, int n2, int dtype, int device_id, int nranks, int rank) {
    if (!out) return OFDFT_EINVAL;
    *out = nullptr;
    if (n0g < 2 || n1g < 2 || n2 < 2)
        return fail(nullptr, OFDFT_EINVAL, "grid extents must be >= 2 (got %d %d %d)", n0g, n1g, n2);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, OFDFT_EINVAL, "bad rank %d of %d", rank, nranks);
    if (nranks > 1 && (n0g % nranks || n1g % nranks))
        return fail(nullptr, OFDFT_EINVAL, "slab decomposition needs n0 and n1 divisible by the rank count %d", nranks);
    const int n0 = n0g / nranks, n1 = n1g;
#ifdef OFDFT_REAL_F32
    if (dtype != OFDFT_F32)
        return fail(nullptr, OFDFT_EINVAL, "this is the fp32 build (libofdft_hip_f32.so): create OFDFT_F64 contexts with libofdft_hip.so");
#else
    if (dtype != OFDFT_F64)
        return fail(nullptr, OFDFT_EINVAL, "this is the fp64 build (libofdft_hip.so): create OFDFT_F32 contexts with libofdft_hip_f32.so");
#endif
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev <= 0)
        return fail(nullptr, OFDFT_EHIP, "no HIP device available (hipGetDeviceCount: %s, count %d)",
                    hipGetErrorString(e0), ndev);
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, OFDFT_EINVAL, "bad device id %d", device_id);
    ofdft_ctx* c = new (std::nothrow) ofdft_ctx();
    if (!c) return fail(nullptr, OFDFT_ENOMEM, "out of host memory");
    c->n0 = n0; c->n1 = n1; c->n2 = n2; c->device = device_id;
    c->n0g = n0g; c->n1g = n1g; c->nranks = nranks; c->rank = rank;
    c->npts = (long long)n0 * n1 * n2;
    c->npts_g = (long long)n0g * n1g * n2;
    SpecGeom& g = c->g;
    g.n0 = n0; g.n1 = n1; g.n2 = n2;
    g.nzc = n2 / 2 + 1;
    g.nzm = (g.nzc / 8) * 8;
    g.nrows = (long long)n0 * n1;
    g.main_count = (long long)g.nzm * g.nrows;
    g.total = (long long)g.nzc * g.nrows;
    c->gx = g;
    c->gx.n0 = n0g;
    c->gx.n1 = n1g / nranks;        // same nrows / totals as g
    // register / LDS line transforms (and with them the fused pipelines) for powers of two and the listed 2^a 3^b 5^c extents
    c->fast = line_extent_ok(n0g) && line_extent_ok(n1g) && row_extent_ok(n2) && is_pow2(nranks);
    if (nranks > 1 && !(c->fast && n2 / 2 <= 512 && c->gx.n1 >= 1)) {
        delete c;
        return fail(nullptr, OFDFT_EINVAL, "the slab-decomposed path needs extents with a line-transform plan (powers of two; fp64: also the listed 2^a 3^b 5^c extents), n2 <= 1024");
    }
    c->xg.nxl = n0; c->xg.nyl = c->gx.n1; c->xg.nb = g.nzm / 8; c->xg.nrem = g.nzc - g.nzm;
    c->xg.log_nyl = 0;
    while ((1 << c->xg.log_nyl) < c->xg.nyl) c->xg.log_nyl++;
    c->xg.arr_sz = (long long)g.nzc * c->gx.n1;
    xchg_chunks_set(c, 0);
    const double s5 = std::sqrt(5.0);
    const double defaults[OFDFT_NPARAMS] = {kFiveSixths, kFiveSixths, (5.0 + s5) / 6.0, (5.0 - s5) / 6.0, (double)27 / 10, 1.0, 0.0, (double)40 / 27,
                                            0.0, 0.0, 0.0, 1.0, 0.0};
    std::memcpy(c->params, defaults, sizeof(defaults));
    DeviceScope device_scope_(device_id);
    hipError_t e = device_scope_.err;
    // partial-sum rows: the pointwise kernels use <= kRedBlocks blocks, the z kernels one block per row group
    c->partial_rows = kRedBlocks;
    if (c->fast && c->n2 / 2 <= 512) {
        const long long zb = (g.nrows + 3) / 4;      // >= blocks of any z kernel (RPB >= 4)
        if (zb > c->partial_rows) c->partial_rows = zb;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_partial, sizeof(double) * (c->partial_rows + kRedMidRows) * kMaxScalars);   // + the first-level sums of OFDFT_REDUCE
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_reduced, sizeof(double) * kMaxScalars);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_scal, sizeof(double) * 8);    // [0] closure scale, [2] split WGC99 energy, [4..6] WT-style weights
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_partial, sizeof(double) * kRedBlocks * kMaxScalars);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_c, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side_stream2, hipStreamNonBlocking);
    if (e != hipSuccess) {
        fail(nullptr, OFDFT_EHIP, "context allocation failed: %s", hipGetErrorString(e));
        ofdft_destroy(c);
        return OFDFT_EHIP;
    }
    *out = c;
    return OFDFT_OK;
}

int ofdft_create(ofdft_ctx** out, int n0, int n1, int n2, int dtype, int device_id) {
    return ofdft_create_dist(out, n0, n1, n2, dtype, device_id, 1, 0);
}

void ofdft_destroy(ofdft_ctx* c) {
    if (!c) return;
    DeviceScope device_scope_(c->device);
    graph_drop(c);
    ipc_release(c);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    for (auto& kv : c->ws)
        if (kv.second.p && !kv.second.borrowed) (void)hipFree(kv.second.p);
    if (c->res_sync) (void)hipFree(c->res_sync);
    if (c->res_done) (void)hipHostFree(c->res_done);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_reduced) (void)hipFree(c->d_reduced);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->h_partial) (void)hipHostFree(c->h_partial);
    if (c->d_wgc_coef) (void)hipFree(c->d_wgc_coef);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    delete c->zr;
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_c) (void)hipEventDestroy(c->ev_c);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->side_stream2) (void)hipStreamDestroy(c->side_stream2);
    delete c;
}

const char* ofdft_last_error(const ofdft_ctx* c) { return c ? c->err : g_create_error; }

int ofdft_set_cell(ofdft_ctx* c, const double box[9]) {
    if (!c || !box) return OFDFT_EINVAL;
    c->version++;
    // det and inverse-transpose of the 3x3 (rows = lattice vectors)
    const double* a = box;
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                       a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det))
        return fail(c, OFDFT_EINVAL, "Lattice vector matrix is not invertible.");   // functional_tools.py:150
    // inv(box^T)[i][j] = cofactor(box)[i][j] / det   (inverse of the transpose = transpose of the inverse)
    double cof[9];
    cof[0] = a[4] * a[8] - a[5] * a[7];
    cof[1] = -(a[3] * a[8] - a[5] * a[6]);
    cof[2] = a[3] * a[7] - a[4] * a[6];
    cof[3] = -(a[1] * a[8] - a[2] * a[7]);
    cof[4] = a[0] * a[8] - a[2] * a[6];
    cof[5] = -(a[0] * a[7] - a[1] * a[6]);
    cof[6] = a[1] * a[5] - a[2] * a[4];
    cof[7] = -(a[0] * a[5] - a[2] * a[3]);
    cof[8] = a[0] * a[4] - a[1] * a[3];
    for (int i = 0; i < 9; ++i) {
        c->box[i] = box[i];
        c->kg.b[i] = 2.0 * kPi * cof[i] / det;                                       // functional_tools.py:149
    }
    c->kg.g = c->gx;
    c->kg.y0 = c->rank * c->gx.n1;
    c->kg.n1g = c->n1g;
    c->vol = std::fabs(det);
    c->dV = c->vol / (double)c->npts_g;
    c->cell_set = true;
    c->wgc_valid = false;
    return OFDFT_OK;
}

int ofdft_set_terms(ofdft_ctx* c, uint32_t mask, const double* params, int nparams) {
    if (!c) return OFDFT_EINVAL;
    if (mask == 0 || (mask >> OFDFT_NTERMS)) return fail(c, OFDFT_EINVAL, "bad term mask 0x%x", mask);
    if (nparams < 0 || nparams > OFDFT_NPARAMS || (nparams > 0 && !params)) return fail(c, OFDFT_EINVAL, "bad params");
    for (int i = 0; i < nparams; ++i) {
        if (i >= OFDFT_P_WGC_ALPHA && i <= OFDFT_P_WGC_KAPPA && c->params[i] != params[i]) c->wgc_valid = false;
        c->params[i] = params[i];
    }
    c->mask = mask;
    c->version++;
    return OFDFT_OK;
}

int ofdft_energy_potential(ofdft_ctx* c, const void* den, const void* vext, double* E_terms, void* dEdn, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    if (int rc = begin_call(c, st)) return rc;
    if (!den || !E_terms) return fail(c, OFDFT_EINVAL, "null argument");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    if (c->nranks > 1) return fail(c, OFDFT_ESTATE, "slab-decomposed context: use the staged ofdft_dist_* calls");
    double vn;
    if (zfused_serves(c)) {
        double nel = 0.0;
        if (c->mask & (OFDFT_WT_NL | OFDFT_WGC99_NL | OFDFT_VWGTF)) {
            double nsum;
            if (int rc = device_sum(c, (const real*)den, false, &nsum, st)) return rc;
            nel = nsum / (double)c->npts * c->vol;
        }
        if (resident_serves(c)) {
            // grids that fit on chip: one persistent kernel (resident.hip), here on the density itself
            const int rrc = resident_closure(c, (const real*)den, (const real*)vext, nel > 0.0 ? nel : 1.0, (real*)dEdn, nullptr, st, true);
            if (rrc < 0) return rrc;
            if (rrc == 0) {
                if (int rc = end_call(c, st)) return rc;
                if (c->h_partial[13] == 0.0) {
                    double sums[kNSums];
                    for (int i = 0; i < kNSums; ++i) sums[i] = c->h_partial[i];
                    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
                    energies_from_sums(c, sums, sums + kCombineScalars, E_terms, &vn);
                    c->resident_evals++;
                    return OFDFT_OK;
                }
                resident_give_up(c);          // a grid barrier ran into its time limit: the staged pipeline below redoes the evaluation
                if (int rc = begin_call(c, st)) return rc;
            }
        }
        const DenSrc ds{(const real*)den, 1.0, 0, nullptr};
        if (int rc = run_terms_zfused(c, ds, nel, (const real*)vext, E_terms, (real*)dEdn, &vn, st)) return rc;
        return end_call(c, st);
    }
    // unfused / chirp-z pipelines: the sums are reduced on the device and read from the pinned mirror after the ONE wait at the end
    // (finish_terms: defer; the GGA terms used to wait for their own sums in the middle of the call)
    bool deferred = false;
    if (int rc = run_terms(c, (const real*)den, (const real*)vext, E_terms, (real*)dEdn, &vn, st, 0.0, &deferred)) return rc;
    if (int rc = end_call(c, st)) return rc;
    if (deferred) {
        double sums[kNSums];
        zfused_collect(c, (c->mask & kGgaAny) ? 0 : kCollectNoGga, sums);
        for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
        energies_from_sums(c, sums, sums + kCombineScalars, E_terms, &vn);
    }
    return OFDFT_OK;
}

// ---- the closure evaluation chi -> (sums, grad) as ONE host-free enqueue, and its hipGraph replay --------------------
}  // extern "C"
namespace {

// a grid barrier of the persistent kernel timed out (its workgroups were not co-resident: CU mask, partition mode, another
// stream holding CUs): clear its counters, stop using it on this context; the caller re-runs the evaluation staged
void resident_give_up(ofdft_ctx* c) {
    (void)hipMemset(c->res_sync, 0, 64);
    c->res_epoch = 0;
    c->res_done_target = 0;
    if (c->res_done) *c->res_done = 0;
    c->resident = 0;
    c->resident_fallbacks++;
    c->version++;
}

int closure_enqueue(ofdft_ctx* c, const real* chi, const real* vext, double nel, real* v, real* grad, hipStream_t st) {
    const int blocks = grid_for(c->npts / 2 + 1, kRedThreads, kRedBlocks);
    OFDFT_LAUNCH(c, st, "sum", (sum_kernel<true>), dim3(blocks), dim3(kRedThreads), 0, chi, c->npts, c->d_partial);
    OFDFT_LAUNCH(c, st, "reduce", closure_scale_reduce_kernel, dim3(1), dim3(kRedThreads), 0, (const acc_t*)c->d_partial, blocks,
                 c->d_reduced + kSumsqSlot, c->d_scal, nel, c->vol / (double)c->npts);
    const DenSrc ds{chi, 0.0, 1, c->d_scal};
    if (int rc = zfused_enqueue(c, ds, nel, vext, v, c->h_partial /* any non-null: host copy wanted */, st, true)) return rc;
    if (grad) {
        const ZRun& zr = zrun(c);
        OFDFT_LAUNCH(c, st, "chi_grad", chi_grad_kernel, dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, chi, (const real*)v, grad,
                     c->npts, 0.0, (const acc_t*)c->d_scal, 2.0 * c->dV, 0.0, (const acc_t*)(c->d_reduced + 8), c->dV, nel,
                     zr.vpart_deferred ? zr.za.v_part : (const real*)nullptr,
                     zr.late_join ? (const acc_t*)(c->d_scal + 3) : (const acc_t*)nullptr);
    }
    return 0;
}

void graph_drop(ofdft_ctx* c) {
    for (auto& g : c->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

// the unfused / chirp-z closure can run without touching the host (finish_terms: defer): not the two-pass WT-style functional,
// and not the configurations run_terms hands to the x-fused pipeline
bool unfused_deferrable(const ofdft_ctx* c) {
    return !wts_active(c) && !(c->fast && !c->force_unfused && !gga_needs_laplacian(c));
}

// chi -> (sums in the pinned mirror, grad) for extents without a plan, enqueued on ONE stream without a host synchronisation when
// *deferred comes back true: sum chi^2 -> c = N_e / (mean(chi^2) vol) stays on the device (system.py:833-834), n = c chi^2 is formed
// from it, the sums are reduced on the device in a fixed order, chi_grad forms mu on the device (system.py:851).  An evaluation
// used to wait four times (sum chi^2, the GGA sums, the combine sums, the end): 53^3 WT + PBE 0.247 -> 0.186 ms.
int closure_enqueue_unfused(ofdft_ctx* c, const real* chi, const real* vext, double nel, real* v, real* grad, hipStream_t st,
                            double* E_terms, double* mu_host, bool* deferred) {
    real* den;
    {
        const int blocks = grid_for(c->npts / 2 + 1, kRedThreads, kRedBlocks);
        OFDFT_LAUNCH(c, st, "sum", (sum_kernel<true>), dim3(blocks), dim3(kRedThreads), 0, chi, c->npts, c->d_partial);
        OFDFT_LAUNCH(c, st, "reduce", closure_scale_reduce_kernel, dim3(1), dim3(kRedThreads), 0, (const acc_t*)c->d_partial, blocks,
                     c->d_reduced + kSumsqSlot, c->d_scal, nel, c->vol / (double)c->npts);
    }
    if (int rc = real_ws(c, "den", &den)) return rc;
    OFDFT_LAUNCH(c, st, "map", (map_kernel<MAP_SCALE_SQ>), dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, chi, den, c->npts, 0.0,
                 (const acc_t*)c->d_scal);
    double vn = 0.0;
    double E_tmp[OFDFT_NTERMS];
    if (int rc = run_terms(c, den, vext, E_terms ? E_terms : E_tmp, v, &vn, st, nel, deferred)) return rc;
    if (!*deferred) {          // (the x-fused pipeline with the z-fused stage off, or the two-pass WT-style functional: sums via the host)
        double cfac;
        HIP_TRY(c, hipMemcpyAsync(&cfac, c->d_scal, sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        const double mu = vn / nel;                                                   // system.py:851
        if (mu_host) *mu_host = mu;
        if (grad)
            OFDFT_LAUNCH(c, st, "chi_grad", chi_grad_kernel, dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, chi, (const real*)v, grad, c->npts,
                         cfac * 2.0 * c->dV, (const acc_t*)nullptr, 0.0, mu);
        return 0;
    }
    if (grad)
        OFDFT_LAUNCH(c, st, "chi_grad", chi_grad_kernel, dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, chi, (const real*)v, grad, c->npts, 0.0,
                     (const acc_t*)c->d_scal, 2.0 * c->dV, 0.0, (const acc_t*)(c->d_reduced + 8), c->dV, nel);
    return 0;
}

// Serve the call from a captured graph when one exists for exactly these arguments; capture one on the second call
// with the same arguments (the first call runs kernel by kernel: workspaces and tables are allocated there, which a
// capture may not do).  *done = false -> the caller runs the ordinary path.  Any capture failure disables the feature
// for this context (results never depend on it).
int closure_graph(ofdft_ctx* c, const real* chi, const real* vext, double nel, real* v, real* grad, hipStream_t st, double* sums,
                  bool* done, bool unfused = false) {
    *done = false;
    if (!graph_eligible(c) || c->profiling) return 0;
    ofdft_ctx::GraphEntry* ge = nullptr;
    for (auto& g : c->graphs)
        if (g.chi == chi && g.vext == vext && g.grad == grad && g.nel == nel && g.version == c->version) ge = &g;
    if (!ge) {
        if (c->graphs.size() >= 8 || (!c->graphs.empty() && c->graphs.front().version != c->version)) graph_drop(c);
        ofdft_ctx::GraphEntry g;
        g.chi = chi; g.vext = vext; g.grad = grad; g.nel = nel; g.version = c->version;
        c->graphs.push_back(g);
        ge = &c->graphs.back();
    }
    if (++ge->seen == 1) return 0;
    if (!ge->exec) {
        if (!c->cap_stream && hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) {
            c->use_graph = false;
            (void)hipGetLastError();
            return 0;
        }
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            bool deferred = true;
            const int rc = unfused ? closure_enqueue_unfused(c, chi, vext, nel, v, grad, c->cap_stream, nullptr, nullptr, &deferred)
                                   : closure_enqueue(c, chi, vext, nel, v, grad, c->cap_stream);
            ok = hipStreamEndCapture(c->cap_stream, &graph) == hipSuccess && rc == 0 && graph && deferred;
        }
        if (ok) ok = hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {
            (void)hipGetLastError();
            ge->exec = nullptr;
            c->use_graph = false;
            graph_drop(c);
            return 0;
        }
        ge->collect = unfused ? ((c->mask & kGgaAny) ? 0 : kCollectNoGga) : collect_flags(zrun(c), zrun(c).late_join);
        ge->fft_count = c->fft_count;
        ge->launch_count = c->launch_count;
        ge->ypass_count = c->ypass_count;
        ge->yfwd_fused = c->yfwd_fused;
    }
    c->fft_count = ge->fft_count;
    c->launch_count = ge->launch_count;
    c->ypass_count = ge->ypass_count;
    c->yfwd_fused = ge->yfwd_fused;
    HIP_TRY(c, hipGraphLaunch(ge->exec, st));
    if (int rc = end_call(c, st)) return rc;
    zfused_collect(c, ge->collect, sums);
    c->graph_replays++;
    *done = true;
    return 0;
}

}  // namespace
extern "C" {

int ofdft_energy_grad_chi(ofdft_ctx* c, const void* chi, const void* vext, double n_electrons, double* E_terms,
                          double* mu_host, void* grad, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    const bool timed = !(c->resident == 2 && resident_serves(c) && !c->profiling);
    if (int rc = begin_call(c, st, timed)) return rc;
    if (!chi || !E_terms) return fail(c, OFDFT_EINVAL, "null argument");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    if (c->nranks > 1) return fail(c, OFDFT_ESTATE, "slab-decomposed context: use the staged ofdft_dist_* calls");
    if (!(n_electrons > 0.0)) return fail(c, OFDFT_EINVAL, "n_electrons must be positive");
    real* v;
    if (int rc = real_ws(c, "v", &v)) return rc;
    if (zfused_serves(c)) {
        // sum chi^2 -> c = N_e / (mean(chi^2) vol) stays on the device; n = c chi^2 is formed on the fly inside the
        // z kernels; mean(n) vol = N_e by construction; mu is formed on the device too.  One host sync per evaluation
        // (the final sums), nothing in between: the sequence is graph-capturable (closure_graph).
        double sums[kNSums];
        bool done = false;
        if (resident_serves(c)) {
            // grids that fit on chip: the whole evaluation is one persistent kernel (resident.hip)
            const int rrc = resident_closure(c, (const real*)chi, (const real*)vext, n_electrons, v, (real*)grad, st);
            if (rrc < 0) return rrc;
            // (the kernel wrote the sums into the pinned host mirror itself)
            bool arrived = false;
            if (rrc == 0 && !timed) {
                // untimed form: the host watches the word the workgroups count themselves out on (each behind a system-scope
                // release of everything it wrote) instead of waiting for the stream -- the runtime's wake-up costs more than
                // the last phase of the kernel; anything unexpected falls back to the stream wait, which reports errors
                HIP_TRY(c, hipGetLastError());
                volatile unsigned* w = c->res_done;
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spins = 0; !arrived; ++spins) {
                    arrived = (int)(*w - c->res_done_target) >= 0;
                    if (!arrived && (spins & 1023u) == 1023u &&
                        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (rrc == 0) {
                if (arrived) {
                    c->last_ms = 0.f;
                    c->ms_pending = false;
                } else {
                    if (int rc = end_call(c, st, timed)) return rc;
                }
                if (c->h_partial[13] == 0.0) {
                    for (int i = 0; i < kNSums; ++i) sums[i] = c->h_partial[i];
                    c->resident_evals++;
                    done = true;
                } else {
                    resident_give_up(c);      // barrier time-out: the graph / staged path below redoes the evaluation
                }
            }
            if (!done)                        // (declined or given up: the fallback is an ordinary timed call)
                if (int rc = begin_call(c, st)) return rc;
        }
        if (!done)
            if (int rc = closure_graph(c, (const real*)chi, (const real*)vext, n_electrons, v, (real*)grad, st, sums, &done)) return rc;
        if (!done) {
            if (int rc = closure_enqueue(c, (const real*)chi, (const real*)vext, n_electrons, v, (real*)grad, st)) return rc;
            if (int rc = end_call(c, st)) return rc;
            zfused_collect(c, collect_flags(zrun(c), zrun(c).late_join), sums);
        }
        double vn;
        for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
        energies_from_sums(c, sums, sums + kCombineScalars, E_terms, &vn);
        if (mu_host) *mu_host = vn / n_electrons;
        return OFDFT_OK;
    }
    // Unfused / chirp-z pipelines (extents without a plan -- the reference's own odd grids): host-free like the fused form,
    // and replayed from a captured graph on small grids (closure_enqueue_unfused, closure_graph)
    {
        double sums[kNSums];
        bool done = false;
        if (unfused_deferrable(c))
            if (int rc = closure_graph(c, (const real*)chi, (const real*)vext, n_electrons, v, (real*)grad, st, sums, &done, true)) return rc;
        if (!done) {
            bool deferred = false;
            if (int rc = closure_enqueue_unfused(c, (const real*)chi, (const real*)vext, n_electrons, v, (real*)grad, st, E_terms,
                                                  mu_host, &deferred))
                return rc;
            if (int rc = end_call(c, st)) return rc;
            if (!deferred) return OFDFT_OK;        // (E_terms / mu were filled on the way)
            zfused_collect(c, (c->mask & kGgaAny) ? 0 : kCollectNoGga, sums);
        }
        double vn;
        for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
        energies_from_sums(c, sums, sums + kCombineScalars, E_terms, &vn);
        if (mu_host) *mu_host = vn / n_electrons;
    }
    return OFDFT_OK;
}

int ofdft_rfftn(ofdft_ctx* c, const void* real_dev, void* spec_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !real_dev || !spec_dev) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    cplx* s;
    if (int rc = spec_ws(c, "io", &s)) return rc;
    if (int rc = rfftn_internal(c, (const real*)real_dev, s, st)) return rc;
    OFDFT_LAUNCH(c, st, "spec_to_standard", spec_to_standard_kernel, dim3((unsigned)((c->g.total + 255) / 256)), dim3(256), 0, s,
                       (cplx*)spec_dev, c->g);
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

int ofdft_irfftn(ofdft_ctx* c, const void* spec_dev, void* real_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !real_dev || !spec_dev) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    cplx* s;
    if (int rc = spec_ws(c, "io", &s)) return rc;
    OFDFT_LAUNCH(c, st, "spec_to_internal", spec_to_internal_kernel, dim3((unsigned)((c->g.total + 255) / 256)), dim3(256), 0,
                 (const cplx*)spec_dev, s, c->g);
    if (int rc = irfftn_internal(c, s, (real*)real_dev, 1.0 / (double)c->npts, st)) return rc;
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    if (c->profiling) prof_collect(c);
    return OFDFT_OK;
}

}  // extern "C"
namespace {
__global__ void debug_math_kernel(int kind, const real* __restrict__ in, real* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const real x = in[i];
        real y;
        switch (kind) {
            case 0: y = fm::rcp(x); break;
            case 1: y = fm::log(x); break;
            case 2: y = fm::exp(x); break;
            case 3: y = fm::roots(x).y; break;
            case 4: y = fm::roots(x).n13; break;
            default: y = fm::roots(x).inv_n; break;
        }
        out[i] = y;
    }
}
}  // namespace
extern "C" {

int ofdft_debug_math(ofdft_ctx* c, int kind, const void* in_dev, void* out_dev, long long n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !in_dev || !out_dev || n < 1 || kind < 0 || kind > 5) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    OFDFT_LAUNCH(c, st, "debug_math", debug_math_kernel, dim3(grid_for(n)), dim3(256), 0, kind, (const real*)in_dev, (real*)out_dev, n);
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipGetLastError());
    return OFDFT_OK;
}

#include "ipc_exchange.inc.h"

// ------------------------------------------------------------------------------ slab-decomposed (multi-GPU) API
int ofdft_dist_sumsq(ofdft_ctx* c, const void* x_local, int square, double* local_sum, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !x_local) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    if (local_sum) return device_sum(c, (const real*)x_local, square != 0, local_sum, st);
    // device-resident form: the local sum goes to scalars[15] (ofdft_dist_scalars), no host synchronisation
    const int blocks = grid_for(c->npts / 2 + 1, kRedThreads, kRedBlocks);
    if (square)
        OFDFT_LAUNCH(c, st, "sum", (sum_kernel<true>), dim3(blocks), dim3(kRedThreads), 0, (const real*)x_local, c->npts,
                     c->d_partial);
    else
        OFDFT_LAUNCH(c, st, "sum", (sum_kernel<false>), dim3(blocks), dim3(kRedThreads), 0, (const real*)x_local, c->npts,
                     c->d_partial);
    OFDFT_REDUCE(c, st, c->d_partial, blocks, 1, c->d_reduced + kSumsqSlot);
    HIP_TRY(c, hipGetLastError());
    return OFDFT_OK;
}

int ofdft_dist_scalars(ofdft_ctx* c, void** scalars_dev) {
    if (!c || !scalars_dev) return OFDFT_EINVAL;
    *scalars_dev = c->d_reduced;
    return OFDFT_OK;
}

int ofdft_dist_begin(ofdft_ctx* c, const void* src_local, int from_chi, double cscale, double nel_global,
                     const void* vext_local, void* v_out_local, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    if (int rc = begin_call(c, st)) return rc;
    if (!src_local) return fail(c, OFDFT_EINVAL, "null argument");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    if ((c->mask & OFDFT_ION_ELECTRON) && !vext_local) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");
    if (!(c->fast && c->n2 / 2 <= 512)) return fail(c, OFDFT_EINVAL, "staged path needs the power-of-two fast path");
    if (gga_needs_laplacian(c) && !c->gga_split)
        return fail(c, OFDFT_EINVAL, "Laplacian-dependent Pauli-Gaussian members need the split-derivative GGA chain (OFDFT_OPT_GGA_SPLIT = 1)");
    if (wts_active(c) && c->nranks > 1)
        return fail(c, OFDFT_EINVAL, "the stabilised Wang-Teter style functional (OFDFT_P_WTS_KIND) is served by single-GPU contexts");
    ZRun& r = zrun(c);
    if (from_chi == 2) {      // closure scale from the (all-reduced) sum of chi^2 in scalars[15]; it never visits the host
        OFDFT_LAUNCH(c, st, "reduce", closure_scale_kernel, dim3(1), dim3(64), 0, c->d_reduced + kSumsqSlot, c->d_scal, nel_global,
                     c->vol / (double)c->npts_g);
        r.ds = DenSrc{(const real*)src_local, 0.0, 1, c->d_scal};
    } else {
        r.ds = DenSrc{(const real*)src_local, (real)cscale, from_chi, nullptr};
    }
    r.nel = nel_global;
    r.vext = (const real*)vext_local;
    r.v_out = (real*)v_out_local;
    r.setup_done = false;          // every evaluation sets up afresh (a failed earlier call must not leave its state behind)
    r.stage[0] = r.stage[1] = 0;
    c->recv_parity[0] = c->recv_parity[1] = 0;
    r.step[0] = r.step[1] = 0;
    r.step_chunk[0] = r.step_chunk[1] = 0;
    r.deferred.clear();
    r.forked = false;
    r.closure = false;
    r.vpart_deferred = false;
    r.za.v_part_deferred = 0;
    r.xlist[0].clear();
    r.xlist[1].clear();
    return OFDFT_OK;
}

// Runs stage `stage` (1..4) of chain `chain` (0: density / Hartree / vW / PBE; 1: nonlocal KEDF).  Its kernels read
// the chain's receive buffer of the previous exchange and write its send buffer directly (exchange layout, see
// XchgGeom); there are no pack / un-pack copies.  On return *bytes_per_peer is the all-to-all message size (0:
// nothing to exchange) and the buffers to use.  The chains are independent until ofdft_dist_finish, so the host
// can keep one chain's all-to-all in flight while the other chain computes.
int ofdft_dist_stage(ofdft_ctx* c, int stage, int chain, void* stream, unsigned long long* bytes_per_peer, void** sendbuf,
                     void** recvbuf) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !bytes_per_peer || !sendbuf || !recvbuf || chain < 0 || chain > 1) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    ZRun& r = zrun(c);
    if (stage != r.stage[chain] + 1 || stage < 1 || stage > 4 || (chain == 1 && r.stage[0] < 1))
        return fail(c, OFDFT_ESTATE, "stage %d of chain %d out of order", stage, chain);
    if (c->xc.n > 1)
        return fail(c, OFDFT_ESTATE, "the exchange buffers are cut into %d kz chunks (OFDFT_OPT_XCHG_CHUNKS): sequence the evaluation with ofdft_dist_step", c->xc.n);
    int rc;
    switch (stage) {
        case 1: rc = zstage1(c, st, chain); break;
        case 2: rc = zstage2(c, st, chain); break;
        case 3: rc = zstage3(c, st, chain); break;
        default: rc = zstage4(c, st, chain); break;
    }
    if (rc) return rc;
    *bytes_per_peer = 0;
    *sendbuf = *recvbuf = nullptr;
    if (c->nranks > 1 && !r.xlist[chain].empty()) {
        cplx *send, *recv;
        if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
        if ((rc = dist_recv_next(c, chain, &recv))) return rc;
        *bytes_per_peer = (unsigned long long)(sizeof(cplx) * (size_t)c->xg.nxl * c->xg.arr_sz * r.xlist[chain].size());
        *sendbuf = send;
        *recvbuf = recv;
        c->recv_parity[chain] ^= 1;         // the next stage reads what this exchange delivers
    }
    HIP_TRY(c, hipGetLastError());
    return OFDFT_OK;
}

// The same evaluation cut finer, for exchanges that overlap the chain's own kernels ("overlap by z-chunks").  The
// exchange buffers are chunk-major, [chunk][peer][xl][array][...] over K = ofdft_query(OFDFT_Q_XCHG_CHUNKS) ranges of kz blocks, so chunk
// k of an exchange is one contiguous equal-split all-to-all message.  Per chain the steps are, each for chunk = 0 .. K-1 in order:
//   1  [chunk 0: z kernels]  y-forward of the chunk into the send buffer                         -> exchange (chunk)
//   2  fused x passes on the chunk: receive buffer -> send buffer                                 -> exchange (chunk)
//   3  y-inverse of the chunk out of the receive buffer
//   4  [chunk 0: PBE mid stage / WGC99 combine on whole rows]  y-forward of the flux chunk        -> exchange (chunk; chain 0 with a GGA term)
//   5  fused x pass of the divergence on the chunk                                                -> exchange (chunk)
//   6  y-inverse of the divergence chunk
// then ofdft_dist_finish.  A step's kernels for chunk k need only chunk k of the exchange before them, so the host may keep
// the all-to-all of chunk k in flight while it enqueues chunk k + 1.  Results are bitwise those of the unchunked sequence.
// On return *bytes_per_peer (0: nothing to exchange) and the chunk's regions of the send / receive buffers.
int ofdft_dist_step(ofdft_ctx* c, int step, int chain, int chunk, void* stream, unsigned long long* bytes_per_peer, void** sendbuf,
                    void** recvbuf) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !bytes_per_peer || !sendbuf || !recvbuf || chain < 0 || chain > 1) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    ZRun& r = zrun(c);
    const int K = c->xc.n;
    const bool next_chunk = step == r.step[chain] && chunk == r.step_chunk[chain] + 1 && chunk < K;
    const bool next_step = step == r.step[chain] + 1 && chunk == 0 && (r.step[chain] == 0 || r.step_chunk[chain] == K - 1);
    if (step < 1 || step > 6 || !(next_chunk || next_step) || (chain == 1 && r.step[0] < 1))
        return fail(c, OFDFT_ESTATE, "step %d chunk %d of chain %d out of order (last: step %d chunk %d of %d)", step, chunk, chain,
                    r.step[chain], r.step_chunk[chain], K);
    int rc;
    switch (step) {
        case 1: rc = zstage1(c, st, chain, chunk); break;
        case 2: rc = zstage2(c, st, chain, chunk); break;
        case 3: rc = zstage3(c, st, chain, 1, chunk); break;
        case 4: rc = zstage3(c, st, chain, 2, chunk); break;
        case 5: rc = zstage4(c, st, chain, chunk); break;
        default: rc = chain == 0 ? zstage5(c, nullptr, st, false, 1, chunk) : 0; break;     // (the divergence belongs to chain 0)
    }
    if (rc) return rc;
    r.step[chain] = step;
    r.step_chunk[chain] = chunk;
    *bytes_per_peer = 0;
    *sendbuf = *recvbuf = nullptr;
    const bool sends = step == 1 || step == 2 || step == 4 || step == 5;
    if (sends && c->nranks > 1 && !r.xlist[chain].empty()) {
        cplx *send, *recv;
        if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
        if ((rc = dist_recv_next(c, chain, &recv))) return rc;
        const XcView v = xc_view(c, chunk);
        const long long narr = (long long)r.xlist[chain].size();
        *bytes_per_peer = (unsigned long long)(sizeof(cplx) * (size_t)c->xg.nxl * v.arr_sz * narr);
        *sendbuf = send + narr * v.base1;
        *recvbuf = recv + narr * v.base1;
        if (chunk == K - 1) c->recv_parity[chain] ^= 1;      // the next step reads what this exchange delivers
    }
    if (step == 5 && chunk == K - 1) r.stage[chain] = 4;
    HIP_TRY(c, hipGetLastError());
    return OFDFT_OK;
}

// Stage 5: y-inverse of the last exchange, combine; local_sums[12] = 9 combine scalars + GGA sums (PBE x, c, kinetic) (to be summed
// over ranks by the caller, then turned into energies by ofdft_dist_energies).
int ofdft_dist_finish(ofdft_ctx* c, double* local_sums, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    ZRun& r = zrun(c);
    if (r.stage[0] != 4 || r.stage[1] != 4) return fail(c, OFDFT_ESTATE, "ofdft_dist_finish called before stage 4 of both chains");
    const bool stepped = r.step[0] > 0;           // sequenced with ofdft_dist_step: the divergence has been y-inverted chunk by chunk
    if (stepped && !(r.step[0] == 6 && r.step_chunk[0] == c->xc.n - 1 && r.step[1] == 6 && r.step_chunk[1] == c->xc.n - 1))
        return fail(c, OFDFT_ESTATE, "ofdft_dist_finish called before step 6 of both chains");
    int rc;
    if ((rc = zstage5(c, local_sums, st, false, stepped ? 2 : 0))) return rc;
    if (!local_sums) {        // device-resident form: scalars[0..10] hold the local sums, nothing waits here
        HIP_TRY(c, hipEventRecord(c->ev1, st));
        c->ms_pending = true;
        HIP_TRY(c, hipGetLastError());
        if (c->profiling) {   // profiling pass only: the event times are read back, which needs the stream drained
            HIP_TRY(c, hipStreamSynchronize(st));
            prof_collect(c);
        }
        return OFDFT_OK;
    }
    return end_call(c, st);
}

int ofdft_dist_energies(ofdft_ctx* c, const double* global_sums, double* E_terms, double* vn_int) {
    if (!c || !global_sums || !E_terms || !vn_int) return OFDFT_EINVAL;
    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
    energies_from_sums(c, global_sums, global_sums + kCombineScalars, E_terms, vn_int);
    return OFDFT_OK;
}

int ofdft_dist_chi_grad(ofdft_ctx* c, const void* chi_local, const void* v_local, void* grad_local, double cscale,
                        double mu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || !chi_local || !v_local || !grad_local) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    OFDFT_LAUNCH(c, st, "chi_grad", chi_grad_kernel, dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, (const real*)chi_local,
                 (const real*)v_local, (real*)grad_local, c->npts, cscale * 2.0 * c->dV,
                 cscale > 0.0 ? (const acc_t*)nullptr : (const acc_t*)c->d_scal, 2.0 * c->dV, mu);
    HIP_TRY(c, hipGetLastError());
    return OFDFT_OK;
}

#ifndef OFDFT_REAL_F32
#include "engine_ions_stress.inc.h"
#else
// The fp32 build serves the density-optimisation hot path only.  The once-per-geometry-step quantities (ionic potential,
// forces, stress, ion-ion sum) are fp64 work: callers run them on the fp64 library (professad_amd.ions does).
#define OFDFT_F64_ONLY(c) fail(c, OFDFT_EINVAL, "%s is served by the fp64 library (libofdft_hip.so) only", __func__)
extern "C" {
int ofdft_ionic_potential(ofdft_ctx* c, const double*, int, const double*, const double*, int, double, int, void*, int, void*) {
    return OFDFT_F64_ONLY(c);
}
int ofdft_ion_electron_forces(ofdft_ctx* c, const void*, const double*, int, const double*, const double*, int, double, int,
                              double*, void*) {
    return OFDFT_F64_ONLY(c);
}
int ofdft_stress(ofdft_ctx* c, const void*, double*, void*) { return OFDFT_F64_ONLY(c); }
int ofdft_ion_electron_stress(ofdft_ctx* c, const void*, const double*, int, const double*, const double*, int, double, int,
                              double*, void*) {
    return OFDFT_F64_ONLY(c);
}
int ofdft_ion_ion(ofdft_ctx* c, const double*, const double*, int, double, double*, double*, double*, void*) {
    return OFDFT_F64_ONLY(c);
}
}
#endif

int ofdft_set_collectives(ofdft_ctx* c, ofdft_all_to_all_fn all_to_all, ofdft_all_reduce_fn all_reduce, void* user) {
    if (!c) return OFDFT_EINVAL;
    c->a2a = all_to_all;
    c->allreduce = all_reduce;
    c->coll_user = user;
    return OFDFT_OK;
}

int ofdft_set_option(ofdft_ctx* c, int option, double value) {
    if (!c) return OFDFT_EINVAL;
    c->version++;
    switch (option) {
        case OFDFT_OPT_GRAPH:
            c->use_graph = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_PIPELINE:
            c->force_unfused = value == 1.0;
            c->pipeline = (int)value;
            return OFDFT_OK;
        case OFDFT_OPT_SIDE_STREAM:
            c->use_side_stream = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_XCHUNKS:
            if (value < 0.0 || value > 64.0) return fail(c, OFDFT_EINVAL, "x chunks must be 0 (automatic) or 1..64");
            c->xchunks = (int)value;
            return OFDFT_OK;
        case OFDFT_OPT_BLUESTEIN:
            c->use_bluestein = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_BS_FUSED:
            c->bs_fused = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_GGA_SPLIT:
            c->gga_split = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_SPLIT_COMBINE:
            c->split_combine = value != 0.0;
            c->defer_vpart = value != 1.0;
            return OFDFT_OK;
        case OFDFT_OPT_XCHUNK_MASK:
            c->xchunk_mask = (int)value & 31;
            return OFDFT_OK;
        case OFDFT_OPT_MIXED_RADIX:
            if (c->nranks > 1 && value == 0.0 && !all_pow2(c))
                return fail(c, OFDFT_EINVAL, "a slab-decomposed context on extents with factors 3 / 5 has no other path than the mixed-radix plans");
            c->fast = line_extent_ok(c->n0g) && line_extent_ok(c->n1g) && row_extent_ok(c->n2) && is_pow2(c->nranks) &&
                      (value != 0.0 || all_pow2(c));
            return OFDFT_OK;
        case OFDFT_OPT_XWAVE:
            c->use_xwave = (int)value;
            return OFDFT_OK;
        case OFDFT_OPT_RESIDENT:
            if (value != 0.0 && value != 1.0 && value != 2.0) return fail(c, OFDFT_EINVAL, "OFDFT_OPT_RESIDENT takes 0, 1 or 2");
            c->resident = (int)value;
            return OFDFT_OK;
        case OFDFT_OPT_XCHG_CHUNKS:
            if (value < 0.0 || value > 16.0) return fail(c, OFDFT_EINVAL, "OFDFT_OPT_XCHG_CHUNKS takes 0 (automatic) or 1..16");
            xchg_chunks_set(c, (int)value);
            c->wgc_valid = false;             // the kernel tables are laid out like the buffers
            return OFDFT_OK;
        case OFDFT_OPT_YBATCH:
            c->ybatch = (int)value;
            return OFDFT_OK;
        case OFDFT_OPT_WGC_FOLD:
            c->wgc_fold = value != 0.0;
            return OFDFT_OK;
        case OFDFT_OPT_IPC_WAIT_MS:
            if (!(value >= 1.0 && value <= 3.6e6)) return fail(c, OFDFT_EINVAL, "OFDFT_OPT_IPC_WAIT_MS takes 1 .. 3 600 000 ms");
            c->ipc_wait_ms = value;
            return OFDFT_OK;
        case OFDFT_OPT_TEST_FAULT:
            c->test_fault = (int)value;
            if (c->test_fault == 2) c->res_fits = 0;
            return OFDFT_OK;
    }
    return fail(c, OFDFT_EINVAL, "unknown option %d", option);
}

int ofdft_set_profiling(ofdft_ctx* c, int on) {
    if (!c) return OFDFT_EINVAL;
    c->profiling = on != 0;
    c->prof.clear();
    c->pending.clear();
    c->ev_used = 0;
    return OFDFT_OK;
}

int ofdft_profile_count(ofdft_ctx* c) { return c ? (int)c->prof.size() : OFDFT_EINVAL; }

int ofdft_profile_get(ofdft_ctx* c, int idx, char* name_buf, int buflen, double* total_ms, long long* launches) {
    if (!c || idx < 0 || idx >= (int)c->prof.size() || !name_buf || buflen < 2) return OFDFT_EINVAL;
    auto it = c->prof.begin();
    std::advance(it, idx);
    std::snprintf(name_buf, (size_t)buflen, "%s", it->first.c_str());
    if (total_ms) *total_ms = it->second.ms;
    if (launches) *launches = it->second.launches;
    return OFDFT_OK;
}

int ofdft_query(ofdft_ctx* c, int what, double* out) {
    if (!c || !out) return OFDFT_EINVAL;
    switch (what) {
        case OFDFT_Q_FFT_COUNT: *out = c->fft_count; return OFDFT_OK;
        case OFDFT_Q_WORKSPACE_BYTES: *out = (double)c->ws_bytes; return OFDFT_OK;
        case OFDFT_Q_FAST_PATH: *out = c->fast ? 1.0 : 0.0; return OFDFT_OK;
        case OFDFT_Q_KERNEL_MS:
            if (c->ms_pending) {       // begin .. finish of the last slab-decomposed evaluation (this rank's stream)
                OFDFT_ON_DEVICE(c, c->device);
                HIP_TRY(c, hipEventSynchronize(c->ev1));
                HIP_TRY(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
                c->ms_pending = false;
            }
            *out = c->last_ms;
            return OFDFT_OK;
        case OFDFT_Q_LAUNCH_COUNT: *out = c->launch_count; return OFDFT_OK;
        case OFDFT_Q_YPASS_COUNT: *out = c->ypass_count; return OFDFT_OK;
        case OFDFT_Q_GRAPH_REPLAYS: *out = (double)c->graph_replays; return OFDFT_OK;
        case OFDFT_Q_RESIDENT_EVALS: *out = (double)c->resident_evals; return OFDFT_OK;
        case OFDFT_Q_RESIDENT_FALLBACKS: *out = (double)c->resident_fallbacks; return OFDFT_OK;
        case OFDFT_Q_XCHG_CHUNKS: *out = (double)c->xc.n; return OFDFT_OK;
        case OFDFT_Q_YFWD_FUSED: *out = (double)c->yfwd_fused; return OFDFT_OK;
        case 16: case 17: case 18: case 19: case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27:      // phase clock of the last persistent-kernel evaluation (microseconds)
            *out = c->h_partial[what] * 0.01;
            return OFDFT_OK;
    }
    return fail(c, OFDFT_EINVAL, "unknown query %d", what);
}

}  // extern "C"

#include "lbfgs_abi.h"
