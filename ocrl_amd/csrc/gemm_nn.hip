// gemm_kernel instantiations for m-contiguous A (weight gradients dW = dy^T x, split-k); see gemm_impl.h
#include "gemm_impl.h"

int gemm_launch_nn(const GemmArgs& a, hipStream_t st) {
    if (!a.bkc) return launch_tr<false, false>(a, st);
    return launch_tr<false, true>(a, st);
}
