// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>

namespace srwn {
int set_error(int code, const char* fmt, ...);  // records a thread-local message, returns code
int check_launch(const char* what);             // hipGetLastError -> 0 or the hipError_t (message recorded)
// srwn_gemm.hip: row-streaming GEMM for cout_pad == 256; returns 1 if it took the call (*rc = result)
int rowgemm_dispatch(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int chunk_len, int Cin,
                     const void* wpack, const float* bias, void* y, int64_t y_row_stride, int cout_pad,
                     int cout_valid, int64_t rows, const void* aux, int64_t aux_row_stride, const int32_t* targets,
                     float* loss_partials, float* logits_out, float grad_scale, int pro, int epi, int dtype,
                     hipStream_t st, int* rc);
// srwn_group.hip: the buffer registered with srwn_debug_stamp_buffer (nullptr = production kernels)
unsigned long long* debug_stamps();
// Diagnostic build: `python sr-wavenet_amd/build.py --diag` compiles every source with -DSRWN_DIAG into libsrwn_diag.so
// (loaded with SRWN_LIB_PATH).  Only that library holds the stamped kernel instantiations and the timing-ablation switch
// SRWN_WT_DEBUG (under which results are wrong by design); the shipped libsrwn.so has neither, and its
// srwn_debug_stamp_buffer refuses a buffer.
#ifdef SRWN_DIAG
#define SRWN_DIAG_ONLY(...) __VA_ARGS__
#else
#define SRWN_DIAG_ONLY(...)
#endif
// SRWN_SAFE_WAIT=1: every hand-counted s_waitcnt vmcnt(N) of the kernels becomes vmcnt(0) (a test runs both and compares
// bits: a count that a compiler change had made too large would show there)
int safe_wait();
}  // namespace srwn
