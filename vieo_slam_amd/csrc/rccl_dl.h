// rccl_dl.h -- in-library RCCL (dlopen'ed): in-place sum all-reduce of doubles on a stream, no host synchronisation
#pragma once
#include <hip/hip_runtime.h>

namespace vieo {
int rccl_allreduce_sum_f64(void* comm, double* d_buf, size_t n, hipStream_t st);
}
