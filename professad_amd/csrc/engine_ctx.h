// Shared declarations of the engine's translation units (gfx950 only).
//
// The library is built from several separately compiled sources so that a change recompiles only its own kernels:
//   engine.hip    C ABI, pipelines (unfused, x-fused, z-fused stages), ion / stress / L-BFGS entry points and their kernels
//   lines.hip     line-transform drivers: y / x passes, plain z passes, chirp-z and plain-DFT paths, the slab-decomposed
//                 3-D transforms of the per-geometry-step routines
//   xpass_a.hip   fused x passes (forward-x, k-space mix, inverse-x): table-driven and single-spectrum mixes
//   xpass_b.hip   fused x passes: density / gradient / divergence mixes
//   zfused.hip    launchers of the wave-local z kernels with fused real-space physics (zpass.h)
// Kernels are templates in the headers (instantiated where they are launched) or `static` (one copy per source that uses them).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ofdft_hip.h"
#include "bluestein.h"
#include "zpass.h"
#include "xwave.h"

using namespace ofdft;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool borrowed = false;      // a window of a larger allocation (the ipc arena): never freed on its own
};

// kz-block chunks of the slab exchange ("overlap by z-chunks"): every exchange buffer -- and the k-point tables
// laid out like one -- is chunk-major, chunk k holding the kz blocks [kb[k], kb[k + 1]) (the remainder planes ride with the
// last chunk) of every (peer, x, array): [chunk][peer][xl][array][(b, yl, kin) | planes].  A chunk is then a contiguous
// equal-split all-to-all message (or one peer scatter), so chunk k can cross the fabric while chunk k + 1 is still in its y
// pass and chunk k - 1 already in its x pass.  n == 1 is the plain layout.
struct XchgChunks {
    int n = 1;
    int kb[17] = {0};
};

struct ofdft_ctx {
    int n0 = 0, n1 = 0, n2 = 0, device = 0;     // LOCAL real-space extents (x-slab: n0 = n0g / nranks)
    int n0g = 0, n1g = 0, nranks = 1, rank = 0; // global extents and slab decomposition
    SpecGeom g{};      // spectrum geometry of the z and y passes (x-slab: n0 local, n1 global)
    SpecGeom gx{};     // spectrum geometry of the x pass (y-slab: n0 global, n1 local); == g on one GPU
    KGeom kg{};        // k-vectors in the x-pass geometry
    XchgGeom xg{};     // exchange-buffer layout of the slab-decomposed path (rec / chunk filled per stage)
    XchgChunks xc{};   // ... cut into kz-block chunks (OFDFT_OPT_XCHG_CHUNKS)
    int xchg_chunks_req = 0;     // requested chunk count: 0 = automatic (xchg_chunks_for)
    long long npts = 0;      // local points
    long long npts_g = 0;    // global points (normalisation, dV)
    bool fast = false, cell_set = false, force_unfused = false;
    int pipeline = 0;   // 0 = z-fused (default on power-of-two grids), 1 = unfused, 2 = x-fused only
    double box[9] = {0}, vol = 0.0, dV = 0.0;
    unsigned mask = 0;
    double params[OFDFT_NPARAMS];
    // twiddle tables by length
    std::map<int, cplx*> tw;
    // named workspaces
    std::map<std::string, DevBuf> ws;
    size_t ws_bytes = 0;
    // reduction partials (device) + pinned host mirror
    double* d_partial = nullptr;
    double* d_reduced = nullptr;     // second-level sums [kMaxScalars]
    double* d_scal = nullptr;        // device-resident scalars: [0] = closure scale c
    double* h_partial = nullptr;
    long long partial_rows = 0;
    // WGC tables
    double* d_wgc_coef = nullptr;   // ca[nt], cb[nt]
    long long wgc_key_nel = -1;
    int fft_passes_fused = 0;      // y-forward passes that rode inside a yderiv launch, so far (diagnostics)
    int yfwd_fused = 0;            // ... of the last energy call (OFDFT_Q_YFWD_FUSED: the byte model of bench.py)
    bool wgc_valid = false;
    double wgc_ck = 0.0;           // K3 = K2 + wgc_ck K1 for the tables in "t:wgc" ((3 - gamma) / (3 n_ref))
    bool wgc_fold = true;          // orthogonal cells: the cross-wave x pass reads the table entry of x > n0 / 2 at n0 - x (OFDFT_OPT_WGC_FOLD)
    // stats
    int fft_count = 0, launch_count = 0;
    double ypass_count = 0.0;   // whole-spectrum y passes executed (fractions for x- / kz-range launches)
    float last_ms = 0.f;
    bool ms_pending = false;    // ev1 recorded without a host wait (device-resident dist finish): elapsed time read on demand
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    hipStream_t side_stream = nullptr, side_stream2 = nullptr;
    bool use_side_stream = true;
    bool bs_fused = true;        // chirp-z path: forward-x, spectral multiply and inverse-x in one kernel (OFDFT_OPT_BS_FUSED)
    bool use_bluestein = true;   // non power-of-two extents <= 512: chirp-z line transforms (else the plain O(N^2) DFT kernels)
    bool gga_split = true;       // GGA chain in split-derivative form: only the x index-derivative visits the x pass
    bool split_combine = true;   // WGC99 part of the combine as its own kernel on the nonlocal chain's stream (forked runs)
    bool defer_vpart = true;     // ... and, in closure evaluations, merged into the potential by chi_grad (the combine kernel does not wait for it)
    int ybatch = 1;         // OFDFT_OPT_YBATCH: the y passes of the three spectra of a WGC99 half as ONE launch (grid.y = 3).  A 256^3 y pass is
                            // 8 256 waves -- exactly what the chip holds at once -- so alone every workgroup loads, transforms and stores in lock
                            // step; three spectra per launch stagger: the batched passes 50 -> 46.5 us (fp64), 34 -> 27 us (fp32) per spectrum,
                            // evaluation neutral in fp64 (three alternations within 0.3 %), +0.9 % in fp32
    int xchunk_mask = 2;    // which stage pairs are chunked: 1 density forward, 2 nonlocal forward, 4 PBE loop, 8 combine loop
    int use_xwave = 1;   // fused x pass: 1 = wave-local kernel (xwave.h) where it measured faster (passes over >= 3 spectra, x extents <= 512), 2 = wherever it exists, 0 = group-parallel kernel only
    int xchunks = 1;    // 1 (default since the round-2 kernels: -0.7 % at 256^3, neutral at 128^3 / 512^3): off; 0: automatic (about 100 MB of
                        // spectra per chunk); > 1: z kernels and the y passes next to them walk the grid in x chunks (Infinity-Cache reuse)
    // optional per-kernel-class profiling (HIP events around every launch)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Pending { const char* name; size_t a, b; };
    std::vector<Pending> pending;
    struct Acc { double ms = 0.0; long long launches = 0; };
    std::map<std::string, Acc> prof;
    struct ofdft_zrun_holder* zr = nullptr;
    // hipGraph replay of the closure evaluation (ofdft_energy_grad_chi): one entry per argument set
    struct GraphEntry {
        const void *chi = nullptr, *vext = nullptr, *grad = nullptr;
        double nel = 0.0;
        unsigned long long version = 0;      // configuration the graph was captured under
        int seen = 0;                        // calls with these arguments so far (the first one runs uncaptured: it allocates)
        hipGraphExec_t exec = nullptr;
        int collect = 0;                     // zfused_collect flags of the captured evaluation
        int fft_count = 0, launch_count = 0, yfwd_fused = 0;
        double ypass_count = 0.0;
    };
    std::vector<GraphEntry> graphs;
    unsigned long long version = 1;          // bumped by set_cell / set_terms / set_option
    bool use_graph = true;
    // persistent small-grid kernel (resident.hip): 1 = serve eligible closure evaluations with it, 2 = the same without the
    // event pair that times a call (OFDFT_Q_KERNEL_MS reads 0), 0 = off
    int resident = 2;
    unsigned* res_sync = nullptr;            // grid-barrier counter
    unsigned res_epoch = 0;                  // its value after the launches so far
    long long resident_evals = 0;
    int test_fault = 0;                      // OFDFT_OPT_TEST_FAULT
    int res_fits = -1;                       // occupancy check of the persistent kernel: -1 not made yet, 0 / 1 its verdict
    long long resident_fallbacks = 0;        // evaluations re-run on the staged path after a barrier time-out
    unsigned* res_done = nullptr;            // pinned host word the kernel's workgroups count themselves out on
    unsigned res_done_target = 0;
    long long graph_replays = 0;
    hipStream_t cap_stream = nullptr;        // capture happens here: the caller's stream may be the (uncapturable) null stream
    // host collectives of the slab-decomposed per-geometry-step routines (ofdft_set_collectives)
    // direct peer-store exchange (ofdft_ipc_*): peers' receive buffers / mailboxes mapped through hipIpc, no host in the loop
    struct ofdft_ipc_state* ipc = nullptr;
    int recv_parity[2] = {0, 0};       // which of a chain's two receive buffers the next stage reads (engine.hip: dist_buffers)
    double ipc_wait_ms = 30000.0;      // OFDFT_OPT_IPC_WAIT_MS: patience of the ipc transport's delivery waits
    ofdft_all_to_all_fn a2a = nullptr;
    ofdft_all_reduce_fn allreduce = nullptr;
    void* coll_user = nullptr;
    char err[512] = "";
};


namespace eng {

constexpr unsigned kGgaAny = OFDFT_PBE_X | OFDFT_PBE_C | OFDFT_GGA_K;   // terms served by the gradient / divergence machinery
constexpr int kNSums = kCombineScalars + kPbeScalars;                  // local sums of an evaluation (12)
constexpr int kSumsqSlot = 15;                                         // d_reduced slot of sum chi^2 (closure form)

inline GgaSel gga_sel(const ofdft_ctx* c) {
    return GgaSel{(c->mask & OFDFT_PBE_X) ? 1 : 0, (c->mask & OFDFT_PBE_C) ? 1 : 0, (c->mask & OFDFT_GGA_K) ? 1 : 0,
                  (int)c->params[OFDFT_P_GGAK_KIND], (real)c->params[OFDFT_P_GGAK_MU], (real)c->params[OFDFT_P_GGAK_BETA],
                  (real)c->params[OFDFT_P_GGAK_LAMBDA], (real)c->params[OFDFT_P_GGAK_SIGMA]};
}

// Pauli-Gaussian member with Laplacian-dependent terms (PGSL0.25 -- the reference's default --, PGSLr): one more spectrum
// each way in the split-derivative GGA chain of the z-fused pipeline (lap n in, lap(df/dL) out); the x-fused-only and
// the unsplit forms fall back to the unfused pipeline
inline bool gga_needs_laplacian(const ofdft_ctx* c) {
    return (c->mask & OFDFT_GGA_K) && (int)c->params[OFDFT_P_GGAK_KIND] == 1 &&
           (c->params[OFDFT_P_GGAK_BETA] != 0.0 || c->params[OFDFT_P_GGAK_LAMBDA] != 0.0 || c->params[OFDFT_P_GGAK_SIGMA] != 0.0);
}
// Pauli-positivity stabilised Wang-Teter style functional (functionals.py:728-782) with f = exp: two combine passes (energies
// first, then the potential with the weights f - f' X, f' they determine)
inline bool wts_active(const ofdft_ctx* c) {
    return (int)c->params[OFDFT_P_WTS_KIND] == 1 && (c->mask & OFDFT_TF) && (c->mask & OFDFT_WT_NL);
}
inline bool zfused_serves(const ofdft_ctx* c) {
    return c->fast && c->pipeline == 0 && c->n2 / 2 <= 512 && (!gga_needs_laplacian(c) || c->gga_split);
}

int fail(ofdft_ctx* c, int code, const char* fmt, ...);

#define HIP_TRY(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, OFDFT_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)


// run the rest of the enclosing ABI function on the context's device, restoring the caller's device on return
#define OFDFT_ON_DEVICE(ctx, dev)                                                                   \
    DeviceScope device_scope_(dev);                                                                 \
    if (device_scope_.err != hipSuccess)                                                            \
        return fail(ctx, OFDFT_EHIP, "cannot select device %d: %s", (dev), hipGetErrorString(device_scope_.err))

// hipGraph replay serves the launch-bound regime only (single-GPU contexts up to 2^19 points, e.g. 64 x 64 x 128; above
// that launches are hidden behind the kernels and the measured gain is nil).  There the WGC99 part of the combine stays inside the combine kernel: the forked + split stream topology
// crashes this ROCm's stream capture, and fewer launches is the better trade on small grids anyway.
inline bool graph_eligible(const ofdft_ctx* c) { return c->use_graph && c->nranks == 1 && c->npts <= (1LL << 19); }
void prof_begin(ofdft_ctx* c, hipStream_t st, const char* name);
void prof_end(ofdft_ctx* c, hipStream_t st);
void prof_collect(ofdft_ctx* c);

#define OFDFT_LAUNCH(c, st, name, kern, grid, block, lds, ...)            \
    do {                                                                  \
        prof_begin(c, st, name);                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);      \
        prof_end(c, st);                                                  \
        (c)->launch_count++;                                              \
    } while (0)

inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Extents served by the register / LDS line transforms (fft_radix.h plans) -- and with them by the fused pipelines:
// powers of two, and the 2^a 3^b 5^c extents listed here (any other extent: chirp-z line transforms + the unfused pipeline).
// OFDFT_MIXED_LINES: x / y extents; OFDFT_MIXED_ROWS: the half lengths n2 / 2 of the z rows for the same extents.
// (both precisions since round 3; OFDFT_NO_MIXED_F32 restores the round-2 fp32 build, whose other extents took the chirp-z path)
#if !defined(OFDFT_REAL_F32) || !defined(OFDFT_NO_MIXED_F32)
#define OFDFT_MIXED_LINES(X) X(48) X(96) X(120) X(144) X(160) X(192) X(240) X(250) X(270) X(288) X(320) X(384) X(480)
#define OFDFT_MIXED_ROWS(X) X(24) X(48) X(60) X(72) X(80) X(96) X(120) X(125) X(135) X(144) X(160) X(192) X(240)
#else
#define OFDFT_MIXED_LINES(X)
#define OFDFT_MIXED_ROWS(X)
#endif
inline bool mixed_line(int n) {
#define X(L) if (n == L) return true;
    OFDFT_MIXED_LINES(X)
#undef X
    return false;
}
inline bool line_extent_ok(int n) { return (is_pow2(n) && n >= 8 && n <= 1024) || mixed_line(n); }
inline bool row_extent_ok(int n2) { return (is_pow2(n2) && n2 >= 16 && n2 <= 2048) || mixed_line(n2); }
inline bool all_pow2(const ofdft_ctx* c) { return is_pow2(c->n0g) && is_pow2(c->n1g) && is_pow2(c->n2); }

inline int grid_for(long long n, int tpb = 256, int cap = 2048) {
    long long b = (n + tpb - 1) / tpb;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// partial[rows][ns] -> out[ns] (+ the pinned host mirror): one launch for few rows, two (reduce_rows_kernel, pointwise_kernels.h) for
// many.  `mid` = room for kRedMidRows x ns doubles that nobody else uses while the launch runs: by default the tail of `partial`
// itself (every partial buffer is allocated with kRedMidRows x kMaxScalars doubles beyond its rows).
#define OFDFT_REDUCE(c, st, partial, rows, ns, ...)                                                                          \
    do {                                                                                                                     \
        const acc_t* part_ = (partial);                                                                                      \
        const int rows_ = (rows), ns_ = (ns);                                                                                \
        if (rows_ > kRedTwoLevelRows) {                                                                                      \
            acc_t* mid_ = const_cast<acc_t*>(part_) + (size_t)rows_ * ns_;                                                   \
            int g_ = rows_ / 512;                                                                                            \
            g_ = g_ < 1 ? 1 : (g_ > kRedMidRows ? kRedMidRows : g_);                                                         \
            const int per_ = (rows_ + g_ - 1) / g_;                                                                          \
            g_ = (rows_ + per_ - 1) / per_;                                                                                  \
            OFDFT_LAUNCH(c, st, "reduce", reduce_rows_kernel, dim3(g_), dim3(kRedThreads), 0, part_, rows_, ns_, per_, mid_); \
            OFDFT_LAUNCH(c, st, "reduce", reduce_partials_kernel, dim3(ns_), dim3(kRedThreads), 0, (const acc_t*)mid_, g_, ns_, \
                         __VA_ARGS__);                                                                                       \
        } else {                                                                                                             \
            OFDFT_LAUNCH(c, st, "reduce", reduce_partials_kernel, dim3(ns_), dim3(kRedThreads), 0, part_, rows_, ns_, __VA_ARGS__); \
        }                                                                                                                    \
    } while (0)


// ---- kz-block chunks of the exchange layout
// view of chunk k (k = kWholeXchg: ONE chunk over all kz blocks -- the layout of the per-geometry-step transforms)
constexpr int kWholeXchg = -2;
struct XcView {
    int kb0, kb1, nb, nrem;      // kz blocks [kb0, kb1), their count, remainder planes carried by this chunk
    long long arr_sz;            // elements of one array in one x-plane record of this chunk
    long long base1;             // element offset of the chunk in a ONE-array buffer / table (x narr for a buffer of narr arrays)
};
inline XcView xc_view(const ofdft_ctx* c, int k) {
    XcView v{};
    const bool whole = k == kWholeXchg || c->xc.n <= 1;
    v.kb0 = whole ? 0 : c->xc.kb[k];
    v.kb1 = whole ? c->xg.nb : c->xc.kb[k + 1];
    v.nb = v.kb1 - v.kb0;
    v.nrem = (whole || k == c->xc.n - 1) ? c->xg.nrem : 0;
    v.arr_sz = ((long long)v.nb * 8 + v.nrem) * c->xg.nyl;
    v.base1 = (long long)c->n0g * c->xg.nyl * 8 * v.kb0;          // all peers' x planes of the earlier chunks (which carry no planes)
    return v;
}
// effective chunk count for a request (0 = automatic): a chunk must be whole workgroups of the y pass (nxl * 8 lines per kz
// block) and of the fused x passes (nyl * 8 lines per block), whose tiles are at most 256 lines -> nxl, nyl multiples of 32.
// Automatic: by the size of the rank's slab -- every chunk costs its share of host work (a step call, two events, a scatter
// or a collective, a wait: ~30 us through the library's own transport, more through Python + RCCL), which has to stay small
// beside the chunk's kernels: 4 chunks from 8 M points per rank (512^3 on 8 ranks: 202-MB messages of the nonlocal chain),
// 2 from 1 M (256^3 on 8), else 1; never fewer than 4 kz blocks per chunk.
inline int xchg_chunks_for(const ofdft_ctx* c, int req) {
    if (c->nranks < 2 || c->xg.nb < 2 || c->xg.nxl % 32 || c->xg.nyl % 32) return 1;
    int k = req;
    if (k <= 0) {
        k = c->npts >= (8LL << 20) ? 4 : (c->npts >= (1LL << 20) ? 2 : 1);
        k = std::min(k, c->xg.nb / 4);
    }
    k = std::max(1, std::min(k, std::min(16, c->xg.nb)));
    return k;
}
inline void xchg_chunks_set(ofdft_ctx* c, int req) {
    c->xchg_chunks_req = req;
    c->xc = XchgChunks{};
    c->xc.n = xchg_chunks_for(c, req);
    for (int k = 0; k <= c->xc.n; ++k) c->xc.kb[k] = (int)((long long)c->xg.nb * k / c->xc.n);
}

// ---- workspaces and tables (engine.hip)
int get_twiddle(ofdft_ctx* c, int n, cplx** out);
int get_ws(ofdft_ctx* c, const std::string& name, size_t bytes, void** out);
int real_ws(ofdft_ctx* c, const char* name, real** out);
int spec_ws(ofdft_ctx* c, const char* name, cplx** out);
size_t dist_buffer_bytes(ofdft_ctx* c, int chain);
int dist_buffers(ofdft_ctx* c, int chain, cplx** send, cplx** recv);
int dist_recv_next(ofdft_ctx* c, int chain, cplx** recv);
int dist_exchange(ofdft_ctx* c, cplx* send, cplx* recv, hipStream_t st);
int global_sums(ofdft_ctx* c, double* v, int n);

// ---- line-transform drivers (lines.hip)
void pass_maps(const ofdft_ctx* c, int axis, LineMap& main, LineMap& rem);
template <bool INV>
int fast_axis_pass_multi(ofdft_ctx* c, int axis, cplx* const* specs, int narr, hipStream_t st, int x0 = 0, int cx = 0,
                         int kb0 = 0, int kb1 = 0);
template <bool INV> int fast_axis_pass(ofdft_ctx* c, int axis, cplx* spec, hipStream_t st);
// xk: chunk of the exchange layout (-1: every chunk, one launch each; kWholeXchg: the unchunked layout)
template <bool INV> int ypass_xchg(ofdft_ctx* c, const std::vector<cplx*>& list, cplx* buf, hipStream_t st, int xk = -1);
// (fwd: also store the y-forward transform of `in` there -- in place when fwd == in; fft_kernels.h: yderiv_kernel)
int yderiv(ofdft_ctx* c, const cplx* in, cplx* out, double scale, hipStream_t st, cplx* fwd = nullptr);
int rfftn_internal(ofdft_ctx* c, const real* in, cplx* spec, hipStream_t st);
int rfftn_internal_multi(ofdft_ctx* c, const real* const* in, cplx* const* spec, int n, hipStream_t st);
// chirp-z path with the fused x pass (lines.hip): z + y passes | forward-x, mix, inverse-x | y + z passes
bool bluestein_xmix_ok(const ofdft_ctx* c);
template <int NIN, int NOUT, class Mix>
int bluestein_xmix(ofdft_ctx* c, const cplx* const* in, cplx* const* out, const Mix& mix, hipStream_t st);
// (prep: pointwise pre-operations of the r2c pass, bluestein.h: BsIo -- array a is transformed as f_a(in[a]))
struct BsPrep { int kind[kBsBatch] = {0, 0, 0, 0}; double e[kBsBatch] = {0, 0, 0, 0}; double nref = 0.0; };
int bluestein_fwd_zy_multi(ofdft_ctx* c, const real* const* in, cplx* const* spec, int n, hipStream_t st, const BsPrep* prep = nullptr);
int bluestein_inv_yz_multi(ofdft_ctx* c, cplx* const* spec, real* const* out, int n, double scale, hipStream_t st);
int irfftn_internal_multi(ofdft_ctx* c, cplx* const* spec, real* const* out, int n, double scale, hipStream_t st);
int irfftn_internal(ofdft_ctx* c, cplx* spec, real* out, double scale, hipStream_t st);
int fwd_zy(ofdft_ctx* c, const real* in, cplx* spec, hipStream_t st);
int inv_yz(ofdft_ctx* c, cplx* spec, real* out, double scale, hipStream_t st);

// ---- fused x passes (xpass_a.hip, xpass_b.hip; xpass_impl.h).  Where the x pass finds its spectra: {} = y-slab arrays in
// the block-8 layout (one GPU); otherwise the exchange buffers of the slab-decomposed path (x-major records, see XchgGeom):
// element strides along x of the inputs, the outputs and the k-point tables
struct XfLayout {
    long long se_in = 0, se_out = 0, tse = 0;
    int kb0 = 0, kb1 = 0;          // one GPU: kb1 > kb0 = kz blocks [kb0, kb1) of the full arrays only
    int xnb = -1, xnrem = 0;       // exchange buffers: kz blocks / remainder planes of the chunk the pointers address (xnb < 0: all of c->xg)
    int kz0 = 0;                   // ... and the kz of its first block
};
template <int NIN, int NOUT, class Mix>
int xfused(ofdft_ctx* c, const XfIo& io, const Mix& mix, hipStream_t st, const char* nm, const XfLayout& lay = XfLayout{});
// the WGC99 pass (3 -> 3): with folded table reads (MixWgcFold) where the cell's axes are orthogonal and the cross-wave kernel serves
// the pass, the plain form everywhere else (xpass_b.hip)
int xfused_wgc(ofdft_ctx* c, const XfIo& io, const MixWgc& mix, hipStream_t st, const char* nm, const XfLayout& lay = XfLayout{});

// ---- launchers of the fused z kernels (zfused.hip).  (chunk, nchunks): the launch covers that share of the rows, i.e. the
// x planes [chunk, chunk + 1) * n0 / nchunks (x-chunked pipeline); partial sums land where a full launch would put them.
int launch_zf_density(ofdft_ctx* c, const DenSrc& ds, cplx* out_n, cplx* out_s, hipStream_t st, int chunk = 0, int nchunks = 1,
                      real* dzn = nullptr);
// WGC99 kernel tables (w0, K1, K2, K3 interleaved per k-point, spectrum order) for round(N_e) = nel_rounded; *nref_out = kappa n0
int ensure_wgc_tables(ofdft_ctx* c, long long nel_rounded, hipStream_t st, double* nref_out);
// the tables as the mix functor of the fused x passes sees them, from element `off` (a kz chunk of a chunk-major table)
inline MixWgc wgc_tab(ofdft_ctx* c, long long off = 0) {
    const cplx* t01 = (const cplx*)c->ws["t:wgc"].p;
    return MixWgc{t01 + off, reinterpret_cast<const real*>(t01 + c->g.total) + off, (real)c->wgc_ck};
}
bool resident_serves(const ofdft_ctx* c);
constexpr int kResidentDeclined = 1;       // resident_closure: not an error -- the caller takes the graph / staged path instead
// chi -> (sums, v, chi.grad) -- or, with from_den, density -> (sums, v) -- by the persistent small-grid kernel (resident.hip)
int resident_closure(ofdft_ctx* c, const real* chi, const real* vext, double nel, real* v, real* grad, hipStream_t st, bool from_den = false);
int launch_zf_powers(ofdft_ctx* c, const DenSrc& ds, const PowersArgs& pa, hipStream_t st, int chunk = 0, int nchunks = 1);
int launch_zpbe(ofdft_ctx* c, const DenSrc& ds, cplx* gx, cplx* gy, cplx* gz, real* dfdn, double inv_n, int* blocks_out,
                hipStream_t st, int chunk = 0, int nchunks = 1);
int launch_zpbe2(ofdft_ctx* c, const DenSrc& ds, cplx* A, cplx* B, const real* dzn, real* dfdn, double inv_n, int* blocks_out,
                 hipStream_t st, cplx* L = nullptr);
int launch_zi_combine(ofdft_ctx* c, const ZCombineArgs& a, int* blocks_out, hipStream_t st, int chunk = 0, int nchunks = 1);
int launch_zi_wgc(ofdft_ctx* c, const ZCombineArgs& a, real* v_part, double* partial, int* blocks_out, hipStream_t st);

}  // namespace eng
