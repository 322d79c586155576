// Launchers of the wave-local z kernels with fused real-space physics (zpass.h).  gfx950 only.
// (fp32 build: this unit's kernels -- the wave-local z kernels -- use the packed complex arithmetic of fft_radix.h: zf_density -9 %,
// zi_combine -5 %, zpbe -6 %, zi_wgc +8 % at their sizes of BASELINE configs 3 / 5, profiles/r05_ab_f32_packed.jsonl; every other
// unit keeps the component-wise forms, which the chirp-z kernels of lines.hip need)
#if defined(OFDFT_REAL_F32) && !defined(OFDFT_F32_PK)
#define OFDFT_F32_PK 1
#endif
#include "engine_ctx.h"

namespace eng {

#define OFDFT_ZCASES(X) X(8) X(16) X(32) X(64) X(128) X(256) X(512) OFDFT_MIXED_ROWS(X)
#ifndef OFDFT_EZ
#define OFDFT_EZ 4
#endif
constexpr int EZ = OFDFT_EZ;   // points per lane wanted by the register-hungry fused z kernels
#ifndef OFDFT_EZ_POWERS
#define OFDFT_EZ_POWERS OFDFT_EZ
#endif
constexpr int EZP = OFDFT_EZ_POWERS;   // ... and by zf_powers (one input row, up to six output spectra)

int z_tables(ofdft_ctx* c, cplx** twM, cplx** twN) {
    if (int rc = get_twiddle(c, c->n2 / 2, twM)) return rc;
    return get_twiddle(c, c->n2, twN);
}
template <int M, int E> int z_blocks(const ofdft_ctx* c) { return (int)((c->g.nrows + ZW<M, E>::RPB - 1) / ZW<M, E>::RPB); }

// The z launchers take (chunk, nchunks): the launch covers that share of the rows, i.e. the x planes
// [chunk, chunk + 1) * n0 / nchunks (x-chunked pipeline); partial sums land where a full launch would put them.
int launch_zf_density(ofdft_ctx* c, const DenSrc& ds, cplx* out_n, cplx* out_s, hipStream_t st, int chunk,
                      int nchunks, real* dzn) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
    if (chunk == 0) c->fft_count += (out_n ? 1 : 0) + (out_s ? 1 : 0);
    SpecGeom gz = c->g;
#define X(M_)                                                                                                       \
    case M_: {                                                                                                      \
        const int nb = z_blocks<M_, ZPick<M_, 8>::E>(c) / nchunks;                                                  \
        gz.blk0 = chunk * nb;                                                                                       \
        OFDFT_LAUNCH(c, st, "zf_density", (zf_density_kernel<M_, ZPick<M_, 8>::E>), dim3(nb), dim3(256),            \
                     (ZW<M_, ZPick<M_, 8>::E>::LDS), ds, out_n, out_s, gz, twM, twN, dzn);                          \
        return 0;                                                                                                   \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}

int launch_zf_powers(ofdft_ctx* c, const DenSrc& ds, const PowersArgs& pa, hipStream_t st, int chunk, int nchunks) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
    if (chunk == 0)
        for (int i = 0; i < 6; ++i) c->fft_count += pa.out[i] ? 1 : 0;
    SpecGeom gz = c->g;
#define X(M_)                                                                                                      \
    case M_: {                                                                                                     \
        const int nb = z_blocks<M_, ZPick<M_, EZP>::E>(c) / nchunks;                                                \
        gz.blk0 = chunk * nb;                                                                                      \
        if (pa.out[0] && !pa.out[1] && !pa.out[2] && !pa.out[3] && !pa.out[4] && !pa.out[5])                       \
            OFDFT_LAUNCH(c, st, "zf_powers", (zf_powers_kernel<M_, ZPick<M_, EZP>::E, true>), dim3(nb), dim3(256), \
                         (ZW<M_, ZPick<M_, EZP>::E>::LDS), ds, pa, gz, twM, twN);                                   \
        else                                                                                                       \
            OFDFT_LAUNCH(c, st, "zf_powers", (zf_powers_kernel<M_, ZPick<M_, EZP>::E>), dim3(nb), dim3(256),       \
                         (ZW<M_, ZPick<M_, EZP>::E>::LDS), ds, pa, gz, twM, twN);                                   \
        return 0;                                                                                                  \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}

int launch_zpbe(ofdft_ctx* c, const DenSrc& ds, cplx* gx, cplx* gy, cplx* gz, real* dfdn, double inv_n,
                int* blocks_out, hipStream_t st, int chunk, int nchunks) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
    if (chunk == 0) c->fft_count += 6;    // three c2r finished + three r2c started on chip
    SpecGeom gq = c->g;
#define X(M_)                                                                                                   \
    case M_: {                                                                                                  \
        *blocks_out = z_blocks<M_, ZPick<M_, EZ>::E>(c);                                                        \
        const int nb = *blocks_out / nchunks;                                                                   \
        gq.blk0 = chunk * nb;                                                                                   \
        OFDFT_LAUNCH(c, st, "zpbe", (zpbe_kernel<M_, ZPick<M_, EZ>::E>), dim3(nb), dim3(256),                   \
                     (ZW<M_, ZPick<M_, EZ>::E>::LDS), ds, gx, gy, gz, dfdn, inv_n, gga_sel(c), gq, twM, twN,   \
                     c->d_partial);                                                                             \
        return 0;                                                                                               \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}

// split-derivative GGA mid stage (zpass.h: zpbe2_kernel); L != nullptr: the Laplacian-dependent Pauli-Gaussian form
int launch_zpbe2(ofdft_ctx* c, const DenSrc& ds, cplx* A, cplx* B, const real* dzn, real* dfdn, double inv_n,
                 int* blocks_out, hipStream_t st, cplx* L) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
    c->fft_count += L ? 8 : 6;    // same six 3-D transforms as the plain form (three c2r finished, three r2c started); + lap n, df/dL
    Bmat bm{};
    std::memcpy(bm.b, c->kg.b, sizeof(bm.b));
#define X(M_)                                                                                                   \
    case M_: {                                                                                                  \
        *blocks_out = z_blocks<M_, ZPick<M_, EZ>::E>(c);                                                        \
        if (L)                                                                                                  \
            OFDFT_LAUNCH(c, st, "zpbe", (zpbe2_kernel<M_, ZPick<M_, EZ>::E, true>), dim3(*blocks_out), dim3(256), \
                         (ZW<M_, ZPick<M_, EZ>::E>::LDS), ds, A, B, dzn, dfdn, inv_n, 1.0 / (double)c->n2, gga_sel(c), bm, \
                         c->g, twM, twN, c->d_partial, L);                                                      \
        else                                                                                                    \
            OFDFT_LAUNCH(c, st, "zpbe", (zpbe2_kernel<M_, ZPick<M_, EZ>::E, false>), dim3(*blocks_out), dim3(256), \
                         (ZW<M_, ZPick<M_, EZ>::E>::LDS), ds, A, B, dzn, dfdn, inv_n, 1.0 / (double)c->n2, gga_sel(c), bm, \
                         c->g, twM, twN, c->d_partial, L);                                                      \
        return 0;                                                                                               \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}

int launch_zi_combine(ofdft_ctx* c, const ZCombineArgs& a, int* blocks_out, hipStream_t st, int chunk, int nchunks) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
    SpecGeom gq = c->g;
    const size_t park = sizeof(real) * 256 * kParkSlots;
#define X(M_)                                                                                                     \
    case M_: {                                                                                                    \
        using W = ZW<M_, ZPick<M_, EZ>::E>;                                                                       \
        *blocks_out = z_blocks<M_, W::E>(c);                                                                      \
        const int nb = *blocks_out / nchunks;                                                                     \
        gq.blk0 = chunk * nb;                                                                                     \
        if (a.v_part || !(a.mask & OFDFT_WGC99_NL))   /* no inline WGC99 section needed: the lean instantiation */  \
            OFDFT_LAUNCH(c, st, "zi_combine", (zi_combine_kernel<M_, W::E, false>), dim3(nb), dim3(256),          \
                         (W::LDS + park), a, gq, twM, twN, c->d_partial);                                         \
        else                                                                                                      \
            OFDFT_LAUNCH(c, st, "zi_combine", (zi_combine_kernel<M_, W::E, true>), dim3(nb), dim3(256),           \
                         (W::LDS + park), a, gq, twM, twN, c->d_partial);                                         \
        return 0;                                                                                                 \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}

// split form: the WGC99 part of the combine on the nonlocal chain's stream -> v_part rows + one energy partial per block
int launch_zi_wgc(ofdft_ctx* c, const ZCombineArgs& a, real* v_part, double* partial, int* blocks_out, hipStream_t st) {
    cplx *twM, *twN;
    if (int rc = z_tables(c, &twM, &twN)) return rc;
#define X(M_)                                                                                                     \
    case M_: {                                                                                                    \
        using W = ZW<M_, ZPick<M_, EZ>::E>;                                                                       \
        *blocks_out = z_blocks<M_, W::E>(c);                                                                      \
        OFDFT_LAUNCH(c, st, "zi_wgc", (zi_wgc_kernel<M_, W::E>), dim3(*blocks_out), dim3(256), (W::LDS), a, v_part, \
                     c->g, twM, twN, partial);                                                                    \
        return 0;                                                                                                 \
    }
    switch (c->n2 / 2) { OFDFT_ZCASES(X) }
#undef X
    return fail(c, OFDFT_EINVAL, "bad n2");
}


}  // namespace eng
