// vus_common.hip -- error reporting and version entry points of libvus_hip.so.
#include "vus_common.h"
#include <cstdarg>

namespace vus {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace vus

extern "C" int vus_abi_version(void) { return VUS_ABI_VERSION; }
extern "C" const char* vus_last_error(void) { return vus::last_error_buf(); }
#ifndef VUS_OFFLOAD_TARGET
#define VUS_OFFLOAD_TARGET "unknown"
#endif
extern "C" const char* vus_build_target(void) { return VUS_OFFLOAD_TARGET; }
