// The two-sweep kernels (fs_sweep_rows2.hpp) in a translation unit of their own: they are the kernels that run
// faster with the compiler's SLP vectorizer on (3-4 % at 256^3 and 512^3), while the one-pass sweeps and the
// advection are faster - and the wide one-pass sweeps spill no register - without it.  The library asks for a
// kernel by its template arguments and launches it with hipLaunchKernel.
#include "fs_sweep_rows2.hpp"

namespace fsamd {

template <typename T, int NSEG, int CP, int LPR>
const void* rows2_kernel() { return reinterpret_cast<const void*>(&k_rb_rows2<T, NSEG, CP, LPR>); }

#define FSAMD_ROWS2_SHAPES(T, NSEG, CP)                     \
    template const void* rows2_kernel<T, NSEG, CP, 64>();   \
    template const void* rows2_kernel<T, NSEG, CP, 32>();   \
    template const void* rows2_kernel<T, NSEG, CP, 16>();
FSAMD_ROWS2_SHAPES(float, 1, 2)
FSAMD_ROWS2_SHAPES(float, 1, 1)
FSAMD_ROWS2_SHAPES(double, 1, 1)

}  // namespace fsamd
