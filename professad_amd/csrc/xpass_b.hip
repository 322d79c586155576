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
}  // namespace eng
