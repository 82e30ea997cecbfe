// gemm_kernel instantiations for k-contiguous A, n-contiguous B (dx = dy W and the soft-max backward epilogue); see gemm_impl.h
#include "gemm_impl.h"

int gemm_launch_tn(const GemmArgs& a, hipStream_t st) {
    if (a.epi_mode == 3) return launch_cfg3<128, 128, true, false, true, 0, 3>(a, st);
    return launch_tr<true, false>(a, st);
}
