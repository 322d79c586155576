// Fused x passes, part B: density / gradient / Laplacian mixes.  gfx950 only.
#include "xpass_impl.h"

namespace eng {
template int xfused<1, 1, MixDensity<true, false>>(ofdft_ctx*, const XfIo&, const MixDensity<true, false>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 3, MixDensity<false, true>>(ofdft_ctx*, const XfIo&, const MixDensity<false, true>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 4, MixDensity<true, true>>(ofdft_ctx*, const XfIo&, const MixDensity<true, true>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 1, MixDensityA<false, false>>(ofdft_ctx*, const XfIo&, const MixDensityA<false, false>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 2, MixDensityA<true, false>>(ofdft_ctx*, const XfIo&, const MixDensityA<true, false>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 2, MixDensityA<false, true>>(ofdft_ctx*, const XfIo&, const MixDensityA<false, true>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 3, MixDensityA<true, true>>(ofdft_ctx*, const XfIo&, const MixDensityA<true, true>&, hipStream_t, const char*, const XfLayout&);

extern template int xfused<3, 3, MixWgc>(ofdft_ctx*, const XfIo&, const MixWgc&, hipStream_t, const char*, const XfLayout&);      // xpass_a.hip
int xfused_wgc(ofdft_ctx* c, const XfIo& io, const MixWgc& mix, hipStream_t st, const char* nm, const XfLayout& lay) {
    const real* bb = c->kg.b;
    const bool ortho = bb[1] == 0.0 && bb[2] == 0.0 && bb[3] == 0.0 && bb[5] == 0.0 && bb[6] == 0.0 && bb[7] == 0.0;
    if (ortho && c->wgc_fold && xc_serves<3, 3>(c)) {
        MixWgcFold f;
        static_cast<MixWgc&>(f) = mix;
        f.fold_n0 = c->n0g;
        switch (c->n0g) {
            case 128: return launch_xc_t<128, 3, 3, MixWgcFold>(c, io, f, lay, st, nm);
            case 256: return launch_xc_t<256, 3, 3, MixWgcFold>(c, io, f, lay, st, nm);
            case 512: return launch_xc_t<512, 3, 3, MixWgcFold>(c, io, f, lay, st, nm);
            case 1024: return launch_xc_t<1024, 3, 3, MixWgcFold>(c, io, f, lay, st, nm);
        }
    }
    return xfused<3, 3, MixWgc>(c, io, mix, st, nm, lay);
}
}  // namespace eng
