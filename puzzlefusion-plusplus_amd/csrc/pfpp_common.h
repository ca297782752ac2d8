// Shared host/device helpers for libpfpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "pfpp.h"

#define PFPP_WAVE 64

namespace pfpp {

// thread-local "last error" text behind pfpp_last_error()
void set_error(const char* fmt, ...);

// Every entry point converts its stream right before launching: also drop any sticky error an
// earlier, unrelated HIP call of the process left behind, so check_launch() reports only ours.
inline hipStream_t as_stream(pfpp_stream_t s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

// after a launch: turn a launch failure into PFPP_EHIP
int check_launch(const char* what);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pfpp

#define PFPP_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      pfpp::set_error("%s: %s", __func__, msg);                   \
      return PFPP_EINVAL;                                         \
    }                                                             \
  } while (0)

#define PFPP_SUPPORTED(cond, msg)                                 \
  do {                                                            \
    if (!(cond)) {                                                \
      pfpp::set_error("%s: unsupported: %s", __func__, msg);      \
      return PFPP_EUNSUPPORTED;                                   \
    }                                                             \
  } while (0)
