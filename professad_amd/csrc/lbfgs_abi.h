// C ABI of the device-side limited-memory BFGS building blocks (see lbfgs_kernels.h and include/ofdft_hip.h).
// Included once, at the end of engine.hip (one translation unit: the kernels in the headers are not inline).
#pragma once
#include <new>

#include "lbfgs_kernels.h"

struct ofdft_lbfgs {
    long long n = 0;
    int hist = 0, device = 0;
    int count = 0;                 // stored pairs
    int order[kLbfgsMaxHist + 1];  // physical slot of the logical pair j (oldest first); order[count] = free slot
    real* S[kLbfgsMaxHist + 1] = {nullptr};
    real* Y[kLbfgsMaxHist + 1] = {nullptr};
    real *d = nullptr, *g_prev = nullptr;
    double *d_partial = nullptr, *d_out = nullptr, *h_out = nullptr;
    bool have_prev = false;        // a step (d, t) and the gradient before it exist
    bool pending = false;          // a candidate pair sits in the free slot
    double t_prev = 0.0;
    // host side of the recursion (ofdft_lbfgs_direction): Gram blocks of the stored pairs, oldest first
    int gk = 0;                    // pairs in the Gram blocks
    double SS[kLbfgsMaxHist][kLbfgsMaxHist], SY[kLbfgsMaxHist][kLbfgsMaxHist], YY[kLbfgsMaxHist][kLbfgsMaxHist];
    double gamma = 1.0;
    bool abs_pending = false;      // an update's |step| sum is on its way to h_out[kAbsSlot]
    char err[256] = "";
};

namespace {

static_assert(lbfgs_nscal(kLbfgsMaxHist) < 60, "scalar block");
constexpr int kAbsSlot = 60;       // h_out slot of the last update's sum |t d| (behind the sums of a dots sweep)

int lfail(ofdft_lbfgs* o, int code, const char* msg) {
    if (o) std::snprintf(o->err, sizeof(o->err), "%s", msg);
    return code;
}
#define L_TRY(o, expr)                                                     \
    do {                                                                   \
        hipError_t e_ = (expr);                                            \
        if (e_ != hipSuccess) return lfail(o, OFDFT_EHIP, hipGetErrorString(e_)); \
    } while (0)

LbfgsVecs logical(const ofdft_lbfgs* o, int count) {
    LbfgsVecs v{};
    for (int j = 0; j < count; ++j) {
        v.S[j] = o->S[o->order[j]];
        v.Y[j] = o->Y[o->order[j]];
    }
    return v;
}

template <int K>
void launch_dots(ofdft_lbfgs* o, const LbfgsVecs& v, const real* g, int blocks, hipStream_t st) {
    const int slot = o->order[o->count];
    hipLaunchKernelGGL((lbfgs_dots_kernel<K>), dim3(blocks), dim3(kRedThreads), 0, st, v, g, o->g_prev, o->d, o->t_prev,
                       o->have_prev ? 1 : 0, o->S[slot], o->Y[slot], o->n, o->d_partial);
}
template <int K>
void launch_update(ofdft_lbfgs* o, const LbfgsVecs& v, const LbfgsCoef& c, const real* g, double t, real* x, int blocks,
                   hipStream_t st) {
    hipLaunchKernelGGL((lbfgs_update_kernel<K>), dim3(blocks), dim3(kRedThreads), 0, st, v, c, g, t, o->d, x, o->g_prev, o->n,
                       o->d_partial);
}

}  // namespace

extern "C" {

int ofdft_lbfgs_create(ofdft_lbfgs** out, long long n_local, int history, int device_id) {
    if (!out || n_local < 1 || history < 1 || history > kLbfgsMaxHist) return OFDFT_EINVAL;
    *out = nullptr;
    ofdft_lbfgs* o = new (std::nothrow) ofdft_lbfgs();
    if (!o) return OFDFT_ENOMEM;
    o->n = n_local;
    o->hist = history;
    o->device = device_id;
    DeviceScope device_scope_(device_id);
    hipError_t e = device_scope_.err;
    const size_t vb = sizeof(real) * (size_t)n_local;
    for (int i = 0; i <= history && e == hipSuccess; ++i) {
        e = hipMalloc((void**)&o->S[i], vb);
        if (e == hipSuccess) e = hipMalloc((void**)&o->Y[i], vb);
        o->order[i] = i;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&o->d, vb);
    if (e == hipSuccess) e = hipMalloc((void**)&o->g_prev, vb);
    if (e == hipSuccess) e = hipMalloc((void**)&o->d_partial, sizeof(double) * kRedBlocks * lbfgs_nscal(kLbfgsMaxHist));
    if (e == hipSuccess) e = hipMalloc((void**)&o->d_out, sizeof(double) * 64);           // the sums of a dots sweep (55) + the update's slot
    if (e == hipSuccess) e = hipHostMalloc((void**)&o->h_out, sizeof(double) * 64);
    if (e != hipSuccess) {
        ofdft_lbfgs_destroy(o);
        return e == hipErrorOutOfMemory ? OFDFT_ENOMEM : OFDFT_EHIP;
    }
    *out = o;
    return OFDFT_OK;
}

void ofdft_lbfgs_destroy(ofdft_lbfgs* o) {
    if (!o) return;
    DeviceScope device_scope_(o->device);
    for (int i = 0; i <= kLbfgsMaxHist; ++i) {
        if (o->S[i]) (void)hipFree(o->S[i]);
        if (o->Y[i]) (void)hipFree(o->Y[i]);
    }
    if (o->d) (void)hipFree(o->d);
    if (o->g_prev) (void)hipFree(o->g_prev);
    if (o->d_partial) (void)hipFree(o->d_partial);
    if (o->d_out) (void)hipFree(o->d_out);
    if (o->h_out) (void)hipHostFree(o->h_out);
    delete o;
}

const char* ofdft_lbfgs_last_error(const ofdft_lbfgs* o) { return o ? o->err : "null handle"; }

int ofdft_lbfgs_dots(ofdft_lbfgs* o, const void* g_dev, double* dots_host, int* npairs, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!o || !g_dev || !dots_host || !npairs) return OFDFT_EINVAL;
    DeviceScope device_scope_(o->device);
    L_TRY(o, device_scope_.err);
    const int K = o->count;
    const int ns = lbfgs_nscal(K);
    const LbfgsVecs v = logical(o, K);
    long long want = (o->n / 2 + kRedThreads) / kRedThreads;
    const int blocks = (int)(want < 1 ? 1 : (want > kRedBlocks ? kRedBlocks : want));
    const real* g = (const real*)g_dev;
    switch (K) {
        case 0: launch_dots<0>(o, v, g, blocks, st); break;
        case 1: launch_dots<1>(o, v, g, blocks, st); break;
        case 2: launch_dots<2>(o, v, g, blocks, st); break;
        case 3: launch_dots<3>(o, v, g, blocks, st); break;
        case 4: launch_dots<4>(o, v, g, blocks, st); break;
        case 5: launch_dots<5>(o, v, g, blocks, st); break;
        case 6: launch_dots<6>(o, v, g, blocks, st); break;
        case 7: launch_dots<7>(o, v, g, blocks, st); break;
        default: launch_dots<8>(o, v, g, blocks, st); break;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ns), dim3(kRedThreads), 0, st, (const double*)o->d_partial, blocks, ns,
                       o->d_out);
    L_TRY(o, hipMemcpyAsync(o->h_out, o->d_out, sizeof(double) * ns, hipMemcpyDeviceToHost, st));
    L_TRY(o, hipStreamSynchronize(st));
    L_TRY(o, hipGetLastError());
    std::memcpy(dots_host, o->h_out, sizeof(double) * ns);
    *npairs = K;
    o->pending = o->have_prev;
    return OFDFT_OK;
}

int ofdft_lbfgs_commit(ofdft_lbfgs* o, int push) {
    if (!o) return OFDFT_EINVAL;
    if (push && !o->pending) return lfail(o, OFDFT_ESTATE, "no candidate pair to store (call ofdft_lbfgs_dots after a step)");
    if (push) {
        if (o->count == o->hist) {          // drop the oldest: its slot becomes the free one
            const int freed = o->order[0];
            for (int j = 0; j < o->hist; ++j) o->order[j] = o->order[j + 1];
            o->order[o->hist] = freed;
        } else {
            o->count++;
        }
    }
    o->pending = false;
    return OFDFT_OK;
}

int ofdft_lbfgs_update(ofdft_lbfgs* o, const double* coef_s, const double* coef_y, double coef_g, double t, void* x_dev,
                       const void* g_dev, double* abs_step_sum, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!o || !x_dev || !g_dev || (o->count > 0 && (!coef_s || !coef_y))) return OFDFT_EINVAL;
    if (o->pending) return lfail(o, OFDFT_ESTATE, "ofdft_lbfgs_commit must follow ofdft_lbfgs_dots");
    DeviceScope device_scope_(o->device);
    L_TRY(o, device_scope_.err);
    const int K = o->count;
    const LbfgsVecs v = logical(o, K);
    LbfgsCoef c{};
    for (int j = 0; j < K; ++j) {
        c.cs[j] = coef_s[j];
        c.cy[j] = coef_y[j];
    }
    c.cg = coef_g;
    long long want = (o->n / 2 + kRedThreads) / kRedThreads;
    const int blocks = (int)(want < 1 ? 1 : (want > kRedBlocks ? kRedBlocks : want));
    const real* g = (const real*)g_dev;
    real* x = (real*)x_dev;
    switch (K) {
        case 0: launch_update<0>(o, v, c, g, t, x, blocks, st); break;
        case 1: launch_update<1>(o, v, c, g, t, x, blocks, st); break;
        case 2: launch_update<2>(o, v, c, g, t, x, blocks, st); break;
        case 3: launch_update<3>(o, v, c, g, t, x, blocks, st); break;
        case 4: launch_update<4>(o, v, c, g, t, x, blocks, st); break;
        case 5: launch_update<5>(o, v, c, g, t, x, blocks, st); break;
        case 6: launch_update<6>(o, v, c, g, t, x, blocks, st); break;
        case 7: launch_update<7>(o, v, c, g, t, x, blocks, st); break;
        default: launch_update<8>(o, v, c, g, t, x, blocks, st); break;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kRedThreads), 0, st, (const double*)o->d_partial, blocks, 1,
                       o->d_out + kAbsSlot);
    L_TRY(o, hipMemcpyAsync(o->h_out + kAbsSlot, o->d_out + kAbsSlot, sizeof(double), hipMemcpyDeviceToHost, st));
    o->abs_pending = true;
    if (abs_step_sum) {            // NULL: nobody waits here -- ofdft_lbfgs_abs_step reads the sum after the stream's next synchronisation
        L_TRY(o, hipStreamSynchronize(st));
        L_TRY(o, hipGetLastError());
        *abs_step_sum = o->h_out[kAbsSlot];
    }
    o->have_prev = true;
    o->t_prev = t;
    return OFDFT_OK;
}

// sum |t d| of the last ofdft_lbfgs_update that was called without an output pointer; valid once the stream of that call has
// been synchronised (the closure evaluation and the dots sweep that follow an update both do)
int ofdft_lbfgs_abs_step(ofdft_lbfgs* o, double* abs_step_sum) {
    if (!o || !abs_step_sum) return OFDFT_EINVAL;
    if (!o->abs_pending) return lfail(o, OFDFT_ESTATE, "no update has run");
    *abs_step_sum = o->h_out[kAbsSlot];
    return OFDFT_OK;
}

// Host side of one inner iteration (lbfgsnew.py:594-663 on coefficients): from the sums of the last ofdft_lbfgs_dots (all-reduced
// by the caller on several ranks; layout [S|Y][j][s, y, g] then s.s, s.y, y.y, g.s, g.y, g.g, |g|_1) decide whether the candidate
// pair enters the history (s.y > 1e-10 s.s, :622; `first` != 0: the very first iteration, history cleared), commit, keep the Gram
// blocks S_i.S_j, S_i.Y_j, Y_i.Y_j of the stored pairs, and run the two-loop recursion on the coefficients of the direction
// d = sum_j coef_s[j] S_j + coef_y[j] Y_j + coef_g g.  Outputs sized for `history` pairs; *npairs_out = pairs in use.
int ofdft_lbfgs_direction(ofdft_lbfgs* o, const double* dots, int k, int first, double* coef_s, double* coef_y, double* coef_g,
                          double* g_dot_d, int* npairs_out, int* pushed_out) {
    if (!o || !dots || !coef_s || !coef_y || !coef_g || !g_dot_d || k < 0 || k > kLbfgsMaxHist) return OFDFT_EINVAL;
    const double* tail = dots + 6 * k;
    const double ss = tail[0], sy = tail[1], yy = tail[2], gs = tail[3], gy = tail[4], gg = tail[5];
    double gS[kLbfgsMaxHist], gY[kLbfgsMaxHist];
    int push = 0;
    if (first) {
        if (int rc = ofdft_lbfgs_commit(o, 0)) return rc;
        o->gk = 0;
        o->gamma = 1.0;
    } else {
        if (k != o->gk) return lfail(o, OFDFT_ESTATE, "ofdft_lbfgs_direction: the sums do not belong to the stored history");
        push = sy > 1e-10 * ss;
        if (int rc = ofdft_lbfgs_commit(o, push)) return rc;
        const double *sS = dots, *sY = dots + 3 * k;        // [j][s, y, g]
        int off = 0, kk = k;
        if (push) {
            if (k == o->hist) {                              // drop the oldest pair
                for (int i = 1; i < k; ++i)
                    for (int j = 1; j < k; ++j) {
                        o->SS[i - 1][j - 1] = o->SS[i][j];
                        o->SY[i - 1][j - 1] = o->SY[i][j];
                        o->YY[i - 1][j - 1] = o->YY[i][j];
                    }
                off = 1;
                kk = k - 1;
            }
            for (int j = 0; j < kk; ++j) {
                const double* a = sS + 3 * (j + off);        // S_j . (s, y, g)
                const double* b = sY + 3 * (j + off);        // Y_j . (s, y, g)
                o->SS[kk][j] = o->SS[j][kk] = a[0];
                o->YY[kk][j] = o->YY[j][kk] = b[1];
                o->SY[kk][j] = b[0];                         // s_new . Y_j
                o->SY[j][kk] = a[1];                         // S_j . y_new
                gS[j] = a[2];
                gY[j] = b[2];
            }
            o->SS[kk][kk] = ss;
            o->YY[kk][kk] = yy;
            o->SY[kk][kk] = sy;
            gS[kk] = gs;
            gY[kk] = gy;
            o->gk = kk + 1;
            o->gamma = sy / yy;
        } else {
            for (int j = 0; j < k; ++j) {
                gS[j] = sS[3 * j + 2];
                gY[j] = sY[3 * j + 2];
            }
        }
    }
    const int K = o->gk;
    double dS[kLbfgsMaxHist], dY[kLbfgsMaxHist], al[kLbfgsMaxHist], rho[kLbfgsMaxHist], dg = -1.0;
    for (int i = 0; i < K; ++i) {
        dS[i] = dY[i] = 0.0;
        rho[i] = 1.0 / o->SY[i][i];
    }
    for (int i = K - 1; i >= 0; --i) {
        double a = dg * gS[i];
        for (int j = 0; j < K; ++j) a += dS[j] * o->SS[j][i] + dY[j] * o->SY[i][j];
        al[i] = a * rho[i];
        dY[i] -= al[i];
    }
    for (int i = 0; i < K; ++i) {
        dS[i] *= o->gamma;
        dY[i] *= o->gamma;
    }
    dg *= o->gamma;
    for (int i = 0; i < K; ++i) {
        double b = dg * gY[i];
        for (int j = 0; j < K; ++j) b += dS[j] * o->SY[j][i] + dY[j] * o->YY[j][i];
        dS[i] += al[i] - b * rho[i];
    }
    double gtd = dg * gg;
    for (int j = 0; j < K; ++j) gtd += dS[j] * gS[j] + dY[j] * gY[j];
    for (int j = 0; j < K; ++j) {
        coef_s[j] = dS[j];
        coef_y[j] = dY[j];
    }
    *coef_g = dg;
    *g_dot_d = gtd;
    if (npairs_out) *npairs_out = K;
    if (pushed_out) *pushed_out = push;
    return OFDFT_OK;
}

int ofdft_lbfgs_reset(ofdft_lbfgs* o) {
    if (!o) return OFDFT_EINVAL;
    o->gk = 0;
    o->gamma = 1.0;
    o->abs_pending = false;
    o->count = 0;
    o->have_prev = false;
    o->pending = false;
    for (int i = 0; i <= kLbfgsMaxHist; ++i) o->order[i] = i;
    return OFDFT_OK;
}

}  // extern "C"
