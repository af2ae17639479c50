// Tuned bf16 NT GEMM for the LLaMA-sized projections (placeholder: not yet applicable to any shape).
#include "common.h"
int egomi_gemm_fast_try(const egomi_gemm_desc* d, hipStream_t s) { (void)d; (void)s; return 1; }
