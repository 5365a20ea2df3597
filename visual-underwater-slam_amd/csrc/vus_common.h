// vus_common.h -- shared host-side helpers of libvus_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../include/vus.h"

namespace vus {

// Thread-local text of the last failure, returned by vus_last_error().
char* last_error_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace vus

#define VUS_REQUIRE(cond, ...)                                    \
  do {                                                            \
    if (!(cond)) return vus::fail(VUS_E_INVALID, __VA_ARGS__);    \
  } while (0)

// Check the launch itself (configuration errors); execution errors surface at the next sync.
#define VUS_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) return vus::fail(VUS_E_HIP, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

#define VUS_CHECK_HIP(expr)                                                           \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) return vus::fail(VUS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
