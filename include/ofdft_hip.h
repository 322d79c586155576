/*
 * ofdft_hip.h -- C ABI of the MI355X-native orbital-free DFT energy/gradient engine.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference (profess-ad) has no FFI: its
 * plugin points are Python callables.  Each entry point below names the reference interface it
 * stands behind (paths relative to the reference repo root):
 *
 *   ofdft_energy_potential  <->  one or several `terms` callables f(box_vecs, den) -> E evaluated
 *                                together, plus their functional derivative:
 *                                System.__compute_energy            src/professad/system.py:759-772
 *                                get_functional_derivative          src/professad/functional_tools.py:9-31
 *                                System.functional_derivative       src/professad/system.py:414-447
 *                                the `potentials=` hook             src/professad/system.py:842-854
 *   ofdft_energy_grad_chi   <->  the optimize_density closure chi -> (E, chi.grad)
 *                                                                   src/professad/system.py:830-838
 *   ofdft_set_cell          <->  wavevecs(box_vecs, shape)          src/professad/functional_tools.py:135-162
 *   ofdft_set_terms         <->  the `terms=[...]` list of System   src/professad/system.py:35-36,66
 *   ofdft_rfftn/ofdft_irfftn<->  torch.fft.rfftn / irfftn call sites src/professad/functionals.py:65,71,650,976-981
 *                                                                   src/professad/functional_tools.py:183,227
 *
 * Conventions
 *   - every function returns 0 on success and a negative OFDFT_E* code on failure; nothing throws or
 *     aborts across the ABI; ofdft_last_error() returns a human-readable message.
 *   - all grid pointers are DEVICE pointers to C-contiguous arrays owned by the caller
 *     (real grids [n0][n1][n2]; half spectra [n0][n1][n2/2+1] interleaved re,im); the engine never
 *     retains them past the call.  Plans, twiddles, kernel tables and workspaces live in the ctx.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Work is enqueued on it and
 *     the call returns after the scalar results are on the host (one stream sync).
 *   - a ctx is bound to one device and is not re-entrant.
 *   - precision: the ABI is built once per precision from the same sources -- libofdft_hip.so (OFDFT_F64, the
 *     reference's precision) and libofdft_hip_f32.so (OFDFT_F32, same symbols; grid arrays are float / float2, every
 *     host-visible scalar stays double).  ofdft_create refuses the dtype of the other build.  The fp32 library serves
 *     the evaluation path and ofdft_lbfgs_*; the ion / stress entry points are fp64-only and return OFDFT_EINVAL there.
 *   - energies are Hartree; lengths bohr; box_vecs rows are the lattice vectors (reference layout).
 */
#ifndef OFDFT_HIP_H
#define OFDFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ofdft_ctx ofdft_ctx;

/* error codes */
#define OFDFT_OK            0
#define OFDFT_EINVAL       -1   /* bad argument / shape / unsupported option */
#define OFDFT_EHIP         -2   /* a HIP runtime call failed */
#define OFDFT_ESTATE       -3   /* call sequence error (e.g. set_cell not called) */
#define OFDFT_ENOMEM       -4

/* dtype */
#define OFDFT_F64 0
#define OFDFT_F32 1             /* served by libofdft_hip_f32.so: the same sources built with -DOFDFT_REAL_F32 */

/* term bits; index of a term in E_terms[] is its bit position */
#define OFDFT_ION_ELECTRON  (1u << 0)   /* functionals.py:31-46   (needs vext)                    */
#define OFDFT_HARTREE       (1u << 1)   /* functionals.py:49-72                                  */
#define OFDFT_TF            (1u << 2)   /* functionals.py:207-224                                */
#define OFDFT_VW            (1u << 3)   /* functionals.py:227-246                                */
#define OFDFT_WT_NL         (1u << 4)   /* non_local_KEF functionals.py:644-652, params alpha,beta */
#define OFDFT_WGC99_NL      (1u << 5)   /* WGC99 nonlocal part functionals.py:941-983            */
#define OFDFT_LDA_X         (1u << 6)   /* functionals.py:1510-1512                              */
#define OFDFT_PZ_C          (1u << 7)   /* functionals.py:1515-1521                              */
#define OFDFT_PW_C          (1u << 8)   /* functionals.py:1524-1530                              */
#define OFDFT_CHACHIYO_C    (1u << 9)   /* functionals.py:1533-1537                              */
#define OFDFT_PBE_X         (1u << 10)  /* functionals.py:1597-1603                              */
#define OFDFT_PBE_C         (1u << 11)  /* functionals.py:1606-1618                              */
#define OFDFT_GGA_K         (1u << 12)  /* Pauli part of a GGA kinetic functional, int tau_TF F(s): LuoKarasievTrickey
                                           functionals.py:309-333 (F = 1/cosh(1.3 s)) or PauliGaussian :336-403 with
                                           beta = lambda = sigma = 0 (F = exp(-mu s^2)); the vW part is OFDFT_VW */
#define OFDFT_VWGTF         (1u << 13)  /* local Pauli term of vWGTF1 / vWGTF2, int tau_TF G(n/n0), functionals.py:251-306;
                                           n0 = round(N_e)/vol; the vW part is OFDFT_VW */
#define OFDFT_NTERMS        14

/* params[] slots for ofdft_set_terms (missing trailing slots keep their defaults) */
#define OFDFT_P_WT_ALPHA    0   /* default 5/6 */
#define OFDFT_P_WT_BETA     1   /* default 5/6 */
#define OFDFT_P_WGC_ALPHA   2   /* default (5+sqrt5)/6 */
#define OFDFT_P_WGC_BETA    3   /* default (5-sqrt5)/6 */
#define OFDFT_P_WGC_GAMMA   4   /* default 2.7 */
#define OFDFT_P_WGC_KAPPA   5   /* default 1.0 */
#define OFDFT_P_GGAK_KIND   6   /* 0 = LKT (default), 1 = Pauli-Gaussian exp(-mu s^2) */
#define OFDFT_P_GGAK_MU     7   /* default 40/27 (PGS); 1.0 = PG1 */
#define OFDFT_P_GGAK_BETA   8   /* Pauli-Gaussian coefficients of q^2, -q s^2, s^4 (functionals.py:336-403); any non-zero one */
#define OFDFT_P_GGAK_LAMBDA 9   /* makes the term depend on the reduced Laplacian q (PGSL0.25 = the reference's default,    */
#define OFDFT_P_GGAK_SIGMA  10  /* PGSLr): one more spectrum each way in the GGA chain; every pipeline, slabs and stress     */
#define OFDFT_P_VWGTF_KIND  11  /* 1 = vWGTF1 (default), 2 = vWGTF2 */
#define OFDFT_P_WTS_KIND    12  /* Pauli-positivity stabilisation of WangTeterStyleFunctional (functionals.py:728-782) for a term
                                   set with OFDFT_TF and OFDFT_WT_NL: T = T_TF f(X), X = T_NL / (f'(0) T_TF).  0 (default): f(x) = 1 + x,
                                   i.e. the plain sum T_TF + T_NL; 1: f(x) = exp(x).  Then E_terms[TF] reports T_TF f(X), E_terms[WT_NL]
                                   zero, the potential and the stress carry the weights f - f' X and f'(X) / f'(0).  Single-GPU contexts. */
#define OFDFT_NPARAMS       13

/* ofdft_query selectors */
#define OFDFT_Q_FFT_COUNT        0  /* 3-D FFTs executed by the last energy call              */
#define OFDFT_Q_WORKSPACE_BYTES  1  /* device bytes held by the ctx                            */
#define OFDFT_Q_FAST_PATH        2  /* 1 if the LDS radix FFT path is used, 0 = generic DFT    */
#define OFDFT_Q_KERNEL_MS        3  /* HIP-event time of the last energy call's device work    */
#define OFDFT_Q_LAUNCH_COUNT     4  /* kernel launches of the last energy call                 */
#define OFDFT_Q_YPASS_COUNT      5  /* whole-spectrum y line passes of the last energy call    */
#define OFDFT_Q_GRAPH_REPLAYS     6  /* ofdft_energy_grad_chi calls served by a hipGraph replay so far */
#define OFDFT_Q_RESIDENT_EVALS    7  /* ofdft_energy_grad_chi calls served by the persistent small-grid kernel so far */
#define OFDFT_Q_XCHG_CHUNKS      9  /* effective number of kz chunks of the slab exchange (OFDFT_OPT_XCHG_CHUNKS)               */
#define OFDFT_Q_YFWD_FUSED      10  /* y-forward transforms of the last energy call that rode inside a yderiv launch (1C -> 2C in one pass) */
#define OFDFT_Q_RESIDENT_FALLBACKS 8 /* evaluations re-run on the staged path because a grid barrier of the persistent kernel timed out
                                        (the kernel is switched off for the context after the first one) */

int  ofdft_create(ofdft_ctx** out, int n0, int n1, int n2, int dtype, int device_id);
void ofdft_destroy(ofdft_ctx* ctx);
const char* ofdft_last_error(const ofdft_ctx* ctx);   /* ctx may be NULL: last create() error */

int  ofdft_set_cell(ofdft_ctx* ctx, const double box_vecs[9]);
int  ofdft_set_terms(ofdft_ctx* ctx, uint32_t term_mask, const double* params, int nparams);

/* E_terms_host[OFDFT_NTERMS] (entries of unset terms are 0); dEdn_dev may be NULL (energy only).
 * vext_dev is required iff OFDFT_ION_ELECTRON is set. */
int  ofdft_energy_potential(ofdft_ctx* ctx, const void* den_dev, const void* vext_dev,
                            double* E_terms_host, void* dEdn_dev, void* stream);

/* The density-optimisation closure: n = N_e chi^2 / int chi^2; returns per-term energies,
 * the chemical potential mu = int (dE/dn) n / N_e, and grad_dev = dE/dchi * dV, i.e. exactly what
 * the reference leaves in chi.grad (system.py:836-837,853).  grad_dev may be NULL. */
int  ofdft_energy_grad_chi(ofdft_ctx* ctx, const void* chi_dev, const void* vext_dev, double n_electrons,
                           double* E_terms_host, double* mu_host, void* grad_dev, void* stream);

/* Validation / building-block entry points: same semantics as torch.fft.rfftn / irfftn(s=shape). */
int  ofdft_rfftn(ofdft_ctx* ctx, const void* real_dev, void* spec_dev, void* stream);
int  ofdft_irfftn(ofdft_ctx* ctx, const void* spec_dev, void* real_dev, void* stream);
/* Validation of the lean transcendentals the fused kernels use (csrc/fastmath.h): out[i] = f(in[i]) for n elements of
 * the context's precision; kind 0: 1/x, 1: log x, 2: exp x, 3: x^(-1/6), 4: x^(1/3) and 5: 1/x derived from x^(-1/6). */
int  ofdft_debug_math(ofdft_ctx* ctx, int kind, const void* in_dev, void* out_dev, long long n, void* stream);

int  ofdft_query(ofdft_ctx* ctx, int what, double* out);

/* ---- direct peer-store transport of the slab-decomposed hot path ------------------------------------------------------
 * ofdft_dist_closure runs the whole closure evaluation of a slab-decomposed context in ONE call, the library moving the
 * spectra itself: every rank maps every peer's receive buffers and mailbox through hipIpc, copies a stage's spectra device
 * to device into chunk [rank] of each peer's buffer on the chain's stream (one direct xGMI link per peer pair), stamps an
 * epoch behind them and waits (bounded, ~2 s) for the peers' stamps in a one-wave kernel; the two small reductions use
 * the same mailboxes and add the ranks' numbers in rank order.  Set-up, once per (context, term set): every rank exports
 * the handle of its arena -- ONE allocation that holds the two receive buffers of each chain and the mailbox -- with the
 * five objects' byte offsets, the host passes them around (64 + 40 bytes, any transport) and every rank attaches every peer's.  No torch / RCCL call per evaluation; results are
 * those of the staged path (same kernels in the same order).  v_work_local_dev: a slab-sized work array for dE/dn. */
int  ofdft_ipc_export(ofdft_ctx* ctx, void* handle64, unsigned long long* offsets5);
int  ofdft_ipc_attach(ofdft_ctx* ctx, int peer_rank, const void* handle64, const unsigned long long* offsets5);
/* Close every mapping of a peer's arena this rank holds: the first step of tearing a slab job down -- every rank calls it,
 * the host's ranks meet (a barrier), and only then is any context destroyed: ofdft_destroy frees the arena, and device memory
 * must not be freed while a peer process still has it open through hipIpc.  (While a job runs nothing exported is freed and nothing
 * opened is closed: a term set that needs bigger buffers gets a second arena, csrc/ipc_exchange.inc.h.)  No-op on a context that never attached. */
int  ofdft_ipc_detach(ofdft_ctx* ctx);
int  ofdft_dist_closure(ofdft_ctx* ctx, const void* chi_local_dev, const void* vext_local_dev, double n_electrons_global,
                        double* E_terms_host, double* mu_host, void* grad_local_dev, void* v_work_local_dev, void* stream);

/* ---- collectives for the slab-decomposed per-geometry-step routines --------------------------------------------
 * The hot path (ofdft_dist_*) is staged so that the HOST issues its all-to-alls between stages.  The routines that run
 * once per geometry step (ofdft_stress, ofdft_ionic_potential, ofdft_ion_electron_forces, ofdft_ion_electron_stress)
 * need several distributed transforms and reductions in one call; on a slab-decomposed context they call back into the
 * host's communication layer (torch.distributed over RCCL in professad_amd.distributed; anything with these semantics):
 *   all_to_all(user, send_dev, recv_dev, bytes_per_peer, stream): equal-split all-to-all of device buffers (peer p's chunk
 *       at offset p * bytes_per_peer), ordered after the work already enqueued on `stream`; work enqueued on `stream`
 *       after the call returns is ordered after the exchange.
 *   all_reduce(user, host_buf, count): in-place sum over the ranks of `count` doubles in HOST memory.
 * Both return 0 on success.  Every rank must make the same sequence of calls (they are collective). */
typedef int (*ofdft_all_to_all_fn)(void* user, void* send_dev, void* recv_dev, unsigned long long bytes_per_peer, void* stream);
typedef int (*ofdft_all_reduce_fn)(void* user, double* host_buf, int count);
int  ofdft_set_collectives(ofdft_ctx* ctx, ofdft_all_to_all_fn all_to_all, ofdft_all_reduce_fn all_reduce, void* user);

/* ---- slab-decomposed evaluation over several GPUs (one process per GPU) --------------------
 * Rank r of P owns the real-space x-slab [n0/P][n1][n2].  The evaluation is cut at the four points where the
 * spectra change between the x-slab geometry (z, y passes) and the y-slab geometry (x pass); at each one the
 * engine's kernels leave the spectra in sendbuf in the exchange layout, the HOST does one equal-split all-to-all
 * (RCCL through torch.distributed) from sendbuf to recvbuf, and the next stage's kernels read recvbuf directly.
 * The work is two independent chains (0: density / Hartree / vW / PBE, 1: nonlocal KEDF) with their own buffers, so
 * one chain's all-to-all can be in flight while the other chain computes.  Sequence per evaluation:
 *     ofdft_dist_sumsq (closure form only) -> all-reduce -> c = N_e / (mean chi^2 vol)
 *     ofdft_dist_begin; for stage in 1..4, for chain in 0..1: [wait for the chain's previous all-to-all]
 *         ofdft_dist_stage(stage, chain) + all_to_all(bytes_per_peer)  (asynchronous if the transport allows);
 *     [wait for both] ofdft_dist_finish -> all-reduce of 13 local sums -> ofdft_dist_energies; ofdft_dist_chi_grad.
 * Device-resident scalars (no host round trip before the final sums): pass local_sum_host = NULL to
 * ofdft_dist_sumsq, all-reduce scalars[15] in place (ofdft_dist_scalars), call ofdft_dist_begin with from_chi = 2
 * (the closure scale is then formed on the device), ofdft_dist_finish with local_sums_host = NULL, all-reduce
 * scalars[0..12] in place and copy them to the host once; ofdft_dist_chi_grad with cscale = 0 uses the device scale.
 * With nranks == 1 the same calls work and every bytes_per_peer is 0.  These stand behind the same reference
 * interfaces as ofdft_energy_potential / ofdft_energy_grad_chi (system.py:830-838, functional_tools.py:9-31). */
int  ofdft_create_dist(ofdft_ctx** out, int n0_global, int n1_global, int n2, int dtype, int device_id, int nranks, int rank);
int  ofdft_dist_sumsq(ofdft_ctx* ctx, const void* x_local_dev, int square, double* local_sum_host, void* stream);
int  ofdft_dist_begin(ofdft_ctx* ctx, const void* src_local_dev, int from_chi, double cscale, double n_electrons_global,
                      const void* vext_local_dev, void* v_out_local_dev, void* stream);
int  ofdft_dist_stage(ofdft_ctx* ctx, int stage, int chain, void* stream, unsigned long long* bytes_per_peer, void** sendbuf_dev,
                      void** recvbuf_dev);
/* The same evaluation cut finer so that an exchange overlaps the kernels of its OWN chain ("overlap by z-chunks";
 * the reference is single-device, _optimizers/lbfgs/lbfgsnew.py:31-33, so this has no counterpart there).  The exchange buffers
 * are chunk-major -- [chunk][peer][xl][array][...] over K = ofdft_query(OFDFT_Q_XCHG_CHUNKS) ranges of kz blocks -- so chunk k of an
 * exchange is one contiguous equal-split all-to-all message.  Per chain, each step for chunk = 0 .. K-1 in order:
 *     1  [chunk 0: z kernels]  y-forward of the chunk into the send buffer                  -> exchange of the chunk
 *     2  fused x passes of the chunk, receive buffer -> send buffer                          -> exchange of the chunk
 *     3  y-inverse of the chunk out of the receive buffer
 *     4  [chunk 0: whole-row kernels (GGA mid stage, WGC99 combine)]  y-forward of the flux  -> exchange (chain 0 with a GGA term)
 *     5  fused x pass of the divergence                                                      -> exchange of the chunk
 *     6  y-inverse of the divergence chunk
 * then ofdft_dist_finish.  The kernels of (step, chunk k) need only chunk k of the preceding exchange: the host keeps chunk k's
 * all-to-all in flight while it enqueues chunk k + 1.  *bytes_per_peer = 0: nothing to exchange after this call; otherwise
 * sendbuf / recvbuf address the chunk's region.  Results are bitwise those of the unchunked sequence (same kernels per line).
 * ofdft_dist_stage serves K == 1 only. */
int  ofdft_dist_step(ofdft_ctx* ctx, int step, int chain, int chunk, void* stream, unsigned long long* bytes_per_peer,
                     void** sendbuf_dev, void** recvbuf_dev);
int  ofdft_dist_finish(ofdft_ctx* ctx, double* local_sums_host /*[13] or NULL*/, void* stream);
int  ofdft_dist_scalars(ofdft_ctx* ctx, void** scalars_dev /* 16 doubles owned by the context */);
int  ofdft_dist_energies(ofdft_ctx* ctx, const double* global_sums /*[13]*/, double* E_terms_host, double* vn_integral);
int  ofdft_dist_chi_grad(ofdft_ctx* ctx, const void* chi_local_dev, const void* v_local_dev, void* grad_local_dev,
                         double cscale, double mu, void* stream);

/* Ionic (external) potential of one species on the grid: v(r) = irfftn(S(k) v~(|k|), norm='forward') / vol with the
 * exact structure factor (pme_order = 0) or its particle-mesh-Ewald approximation of even order >= 2, and v~ the
 * cubic-Hermite interpolation of a reciprocal-space pseudopotential table (uniform k grid from 0; the Coulomb tail
 * 4 pi z / k^2 already ADDED to the table for k > 0, exactly as the reference prepares it).  Stands behind
 * System.__potential_from_ions (system.py:183-194) = lattice_sum + structure_factor[_spline] + interpolate_recpot
 * (ion_utils.py:49-286).  frac_coords_host: [nions][3] fractional coordinates (host); vext_dev: [n0][n1][n2] device
 * array that receives (accumulate = 0) or is incremented by (accumulate = 1) the potential. */
int  ofdft_ionic_potential(ofdft_ctx* ctx, const double* frac_coords_host, int nions, const double* table_k_host,
                           const double* table_v_host, int ntable, double z_ion, int pme_order, void* vext_dev,
                           int accumulate, void* stream);

/* Ion-electron forces of one species for a given density: F_a = -d/dR_a int n v_ext (Ha/bohr, [nions][3] on the host),
 * with the same structure-factor options as ofdft_ionic_potential.  Stands behind the IonElectron part of
 * System.__compute_forces (system.py:913-923), which the reference obtains by autograd through the potential build. */
int  ofdft_ion_electron_forces(ofdft_ctx* ctx, const void* den_dev, const double* frac_coords_host, int nions,
                               const double* table_k_host, const double* table_v_host, int ntable, double z_ion,
                               int pme_order, double* forces_host, void* stream);

/* ---- stress -----------------------------------------------------------------------------
 * sigma_ij = (1/Omega) dE/d eps_ij at fixed electron number, per term: what get_stress (functional_tools.py:73-101)
 * and System.__compute_stress (system.py:925-935) obtain by autograd through every FFT.  Closed forms of the
 * reference's own analytic tests (tests/tools_for_tests.py:212-307, 367-472) for Hartree, TF, Wang-Teter, LDA, PBE; the
 * exact discrete forms for vW, the density-dependent WGC99 kernel and the ion-electron term are derived in
 * oracle/stress.py.  sigma_terms_host[OFDFT_NTERMS][9]: row-major symmetric 3x3 per term bit, Ha/bohr^3; the
 * ion-electron entry stays zero (it needs the ions: ofdft_ion_electron_stress, same arguments as the forces). */
int  ofdft_stress(ofdft_ctx* ctx, const void* den_dev, double* sigma_terms_host, void* stream);
int  ofdft_ion_electron_stress(ofdft_ctx* ctx, const void* den_dev, const double* frac_coords_host, int nions,
                               const double* table_k_host, const double* table_v_host, int ntable, double z_ion,
                               int pme_order, double* sigma_host /*[9]*/, void* stream);

/* Ion-ion interaction: the real-space damped pair sum in a neutralising background of
 * ion_interaction_sum (ion_utils.py:293-333) with System.__ion_ion_interaction's parameters (system.py:733-754);
 * Rc <= 0 selects its default Rd = 2 h_max, Rc = 3 Rd^2 / h_max.  The pair list (torch-nl's neighbour list in the
 * reference) is every (i, j, lattice shift) with 0 < r <= Rc.  forces_host [nions][3] = -dE/dR and stress_host [9] =
 * (1/vol) dE/d eps are what autograd gives the reference (system.py:913-935); either may be NULL. */
int  ofdft_ion_ion(ofdft_ctx* ctx, const double* frac_coords_host, const double* charges_host, int nions, double Rc,
                   double* E_host, double* forces_host, double* stress_host, void* stream);

/* ---- limited-memory BFGS building blocks (the consumer of the closure) ----------------
 * The reference's fixed-step optimiser (_optimizers/lbfgs/lbfgsnew.py:594-663) forms y = g - g_prev and s = t d, keeps
 * the pair if y.s > 1e-10 |s|^2, and gets the direction from the two-loop recursion: 2m dependent dot / axpy pairs over
 * full-grid vectors.  Here the history lives on the device and each inner iteration is two sweeps:
 *   ofdft_lbfgs_dots    forms the candidate pair from g, the previous gradient and the previous step, and returns every
 *                       inner product the recursion needs: for each stored pair j (oldest first) (s.S_j, y.S_j, g.S_j), then
 *                       (s.Y_j, y.Y_j, g.Y_j) for each j, then s.s, s.y, y.y, g.s, g.y, g.g, |g|_1  ->  6*npairs + 7 numbers
 *                       (LOCAL sums: all-reduce them over ranks when the vectors are slabs);
 *   ofdft_lbfgs_commit  stores (push = 1) or discards the candidate pair (the caller applies the curvature test);
 *   ofdft_lbfgs_update  d = coef_g g + sum_j coef_s[j] S_j + coef_y[j] Y_j over the stored pairs (oldest first, the
 *                       just-committed pair last), x += t d in place, g_prev = g; returns the local sum |t d|_1.
 *   ofdft_lbfgs_direction  the host side between the two sweeps, natively: curvature test + commit, the Gram blocks of the stored
 *                       pairs, the two-loop recursion on the COEFFICIENTS of d in {S_j, Y_j, g} (so the result is the reference's
 *                       direction up to round-off) -> coef_s, coef_y, coef_g for ofdft_lbfgs_update, and g.d.  A host that wants
 *                       its own recursion calls ofdft_lbfgs_commit itself instead (professad_amd/optimize.py keeps that form for
 *                       backends without this entry).
 * ofdft_lbfgs_update with abs_step_sum == NULL does not wait for the stream; ofdft_lbfgs_abs_step returns the sum after the
 * stream's next synchronisation (the closure evaluation that follows an update does one). */
typedef struct ofdft_lbfgs ofdft_lbfgs;
int  ofdft_lbfgs_create(ofdft_lbfgs** out, long long n_local, int history /* 1..8 */, int device_id);
void ofdft_lbfgs_destroy(ofdft_lbfgs* h);
const char* ofdft_lbfgs_last_error(const ofdft_lbfgs* h);
int  ofdft_lbfgs_reset(ofdft_lbfgs* h);
int  ofdft_lbfgs_direction(ofdft_lbfgs* h, const double* dots /* of ofdft_lbfgs_dots, summed over ranks */, int npairs, int first_iteration,
                           double* coef_s /* [history] */, double* coef_y /* [history] */, double* coef_g, double* g_dot_d,
                           int* npairs_out, int* pushed_out);
int  ofdft_lbfgs_abs_step(ofdft_lbfgs* h, double* abs_step_sum);
int  ofdft_lbfgs_dots(ofdft_lbfgs* h, const void* g_dev, double* dots_host /* [6*8+7] */, int* npairs, void* stream);
int  ofdft_lbfgs_commit(ofdft_lbfgs* h, int push);
int  ofdft_lbfgs_update(ofdft_lbfgs* h, const double* coef_s, const double* coef_y, double coef_g, double t, void* x_dev,
                        const void* g_dev, double* abs_step_sum_host, void* stream);

/* Tuning / validation switches.  OFDFT_OPT_PIPELINE: 0 = automatic (power-of-two grids: fused x passes and z
 * passes that keep every real-space intermediate on chip), 1 = force the unfused pipeline (separate forward,
 * multiply, inverse and pointwise passes; the only one for other grids), 2 = fused x passes only. */
#define OFDFT_OPT_PIPELINE 0
/* OFDFT_OPT_SIDE_STREAM: 1 (default) = run the nonlocal-KEDF chain of the z-fused pipeline on a second HIP stream so
 * that it overlaps the Hartree/vW/PBE chain (they only meet in the combine kernel); 0 = everything on the caller's stream. */
#define OFDFT_OPT_XCHUNKS     2   /* z kernels + the y passes next to them walk the grid in x chunks: 1 = off (default), 0 = automatic (~100 MB of spectra per chunk: the round-1 default, -4 % then, +0.7 % with the round-2 kernels), 2..64 = count for six spectra */
#define OFDFT_OPT_XCHUNK_MASK 3   /* which stage pairs are chunked (bits): 1 density forward, 2 nonlocal-KEDF forward (default: measured -4 %), 4 PBE loop, 8 combine loop, 16 WGC99 x pass + y-inverse by kz blocks (all three measured neutral or slower at 256^3) */
#define OFDFT_OPT_SPLIT_COMBINE 4 /* with side streams, the WGC99 part of the combine runs as its own kernel on the nonlocal chain's stream; 2 (default): and in
                                     closure evaluations (ofdft_energy_grad_chi) the combine kernel does not wait for it -- the potential stays in two
                                     arrays that the chi.grad kernel adds, the part's share of sum(v n) joins the combine's; 1: the combine adds it; 0: off */
#define OFDFT_OPT_BLUESTEIN 5     /* 1 (default): extents that are not powers of two (<= 512) use chirp-z line transforms; 0: plain DFT kernels */
#define OFDFT_OPT_GGA_SPLIT 6     /* 1 (default): split-derivative GGA chain -- only the index derivative along x visits the x pass (and the exchange); 0: three Cartesian components */
#define OFDFT_OPT_GRAPH 7         /* 1 (default): ofdft_energy_grad_chi replays a hipGraph captured on the second call with the same
                                     arguments (device pointers, electron number, cell, terms, options): one graph launch instead
                                     of ~25-50 kernel launches, which is what bounds grids up to ~64^3 (used by single-GPU contexts up to 2^19 points, where
                                     OFDFT_OPT_SPLIT_COMBINE is then ignored); 0: always launch kernel by kernel */
#define OFDFT_OPT_SIDE_STREAM 1
#define OFDFT_OPT_MIXED_RADIX 9   /* 1 (default): extents with factors 3 and 5 that have a line-transform plan (48, 96, 120, 144, 160, 192, 240, 250, 270,
                                     288, 320, 384, 480) run the register / LDS transforms and the fused pipelines like the powers of two;
                                     0: they take the chirp-z transforms + the unfused pipeline like any other extent (validation, A/B) */
#define OFDFT_OPT_RESIDENT 10     /* ofdft_energy_grad_chi on cubic 16^3 / 32^3 / 64^3 grids with local, Hartree, von Weizsaecker and Wang-Teter terms (up to
                                     32^3 also PBE / LKT / mu-only Pauli-Gaussian and WGC99)
                                     runs as ONE persistent kernel (four phases, three grid barriers; csrc/resident.hip) instead of the staged
                                     pipeline.  2 (default): the call is not bracketed by the HIP event pair that times it (OFDFT_Q_KERNEL_MS reads
                                     0 for it) and the host watches a pinned word the last workgroup writes instead of waiting for the stream --
                                     together ~8 microseconds of a ~45-microsecond call; 1: events + stream wait as everywhere else; 0: off */
#define OFDFT_OPT_XCHG_CHUNKS 12  /* slab-decomposed contexts: the exchange buffers (and the k-point tables laid out like them) are cut into this many
                                     ranges of kz blocks, chunk-major, so that ofdft_dist_step / ofdft_dist_closure move chunk k across the fabric
                                     while chunk k + 1 is in its y pass and chunk k - 1 in its x pass.  0 (default): automatic -- 4 chunks
                                     from 8 M points per rank, 2 from 1 M, else 1 (and only when the slab extents n0 / P and n1 / P are multiples
                                     of 32, with >= 4 kz blocks per chunk); 1: off.  Every
                                     rank must use the same value; the ipc transport needs a new ofdft_ipc_export / attach round after a change. */
#define OFDFT_OPT_YBATCH 14       /* 1 (default): the y passes of the three spectra of each WGC99 half run as one launch (grid.y = 3); 0: three launches */
#define OFDFT_OPT_IPC_WAIT_MS 13  /* ipc transport: how long a delivery wait of ofdft_dist_closure stays patient before it aborts the evaluation ON ALL
                                     RANKS (default 30 000 ms: a rank may be late by a module load, a page-in, a garbage collection) */
#define OFDFT_OPT_TEST_FAULT 11   /* test hook, never set in production: 1 = the NEXT persistent-kernel launch is one workgroup short (its grid
                                     barriers time out -> the call must still return the right numbers through the staged path and switch the
                                     kernel off); 2 = the co-residency check of the persistent kernel reports "does not fit" */
#define OFDFT_OPT_BS_FUSED 15     /* chirp-z path (extents without a line-transform plan, i.e. what the reference's System.ecut2shape, system.py:74-89, gives):
                                     1 (default) = forward-x, spectral multiply and inverse-x of every convolution in ONE kernel (x extents up to 256);
                                     0 = three passes per transform and separate multiply kernels */
#define OFDFT_OPT_WGC_FOLD 28     /* 1 (default): on cells with orthogonal axes the cross-wave x pass reads the WGC99 table entry of x > n0 / 2 at
                                     n0 - x (|k| is even along a line there; functionals.py:968-972 depends on |k| only): both uses of an entry
                                     fall into one tile, the second is a cache hit, the pass' table traffic halves.  0: every k-point its own entry */
#define OFDFT_OPT_XWAVE 8         /* fused x passes: 1 (default) = the cross-wave kernel (whole runs of memory-adjacent lines per access, the
                                     radix-4 / -8 step of the line transform across the waves of a workgroup) for 256- and 512-point lines
                                     and, for passes over three or more spectra, 1024-point lines; elsewhere the wave-local kernel (a line of
                                     every spectrum in the lanes of one wavefront, mixing in registers; x extents up to 512) for passes over
                                     three or more spectra and the group-parallel kernel (spectra traded through LDS) for the rest.
                                     0 = always the group-parallel kernel, 2 = the wave-local kernel for every pass, 5 = the cross-wave
                                     kernel wherever it exists (128..1024 points), 6 = ... for passes over >= 3 spectra only, 7 = the
                                     round-3 choice (no cross-wave kernel) */
int  ofdft_set_option(ofdft_ctx* ctx, int option, double value);

/* Measurement support (bench.py): when on, every kernel launch of the energy calls is bracketed by HIP
 * events on the caller's stream and the durations are accumulated per kernel class.  Adds launch overhead:
 * never on inside a timed region. */
int  ofdft_set_profiling(ofdft_ctx* ctx, int on);
int  ofdft_profile_count(ofdft_ctx* ctx);
int  ofdft_profile_get(ofdft_ctx* ctx, int idx, char* name_buf, int buflen, double* total_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* OFDFT_HIP_H */
