// gemm_kernel instantiations for k-contiguous A and B (y = x W^T and the vocabulary heads with soft-max epilogues); see gemm_impl.h
#include "gemm_impl.h"

int gemm_launch_tt(const GemmArgs& a, hipStream_t st) {
    if (a.epi_mode == 2) return launch_cfg3<128, 128, true, true, true, 0, 2>(a, st);
    if (a.epi_mode == 1) return launch_cfg3<128, 128, true, true, true, 0, 1>(a, st);
    return launch_tr<true, true>(a, st);
}
