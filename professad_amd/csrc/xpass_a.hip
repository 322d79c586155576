// Fused x passes, part A: table-driven (WGC99) and single-spectrum mixes.  gfx950 only.
#include "xpass_impl.h"

namespace eng {
template int xfused<3, 3, MixWgc>(ofdft_ctx*, const XfIo&, const MixWgc&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 1, MixScale<SPEC_LAPLACE>>(ofdft_ctx*, const XfIo&, const MixScale<SPEC_LAPLACE>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 1, MixScale<SPEC_LINDHARD>>(ofdft_ctx*, const XfIo&, const MixScale<SPEC_LINDHARD>&, hipStream_t, const char*, const XfLayout&);
template int xfused<1, 1, MixDerivA>(ofdft_ctx*, const XfIo&, const MixDerivA&, hipStream_t, const char*, const XfLayout&);
template int xfused<2, 1, MixDerivAL>(ofdft_ctx*, const XfIo&, const MixDerivAL&, hipStream_t, const char*, const XfLayout&);
template int xfused<3, 1, MixDiv>(ofdft_ctx*, const XfIo&, const MixDiv&, hipStream_t, const char*, const XfLayout&);
}  // namespace eng
