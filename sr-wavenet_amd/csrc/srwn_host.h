// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstdarg>
#include <cstdio>

namespace srwn {
int set_error(int code, const char* fmt, ...);  // records a thread-local message, returns code
int check_launch(const char* what);             // hipGetLastError -> 0 or the hipError_t (message recorded)
}  // namespace srwn
