// Private to segmif_amd/csrc: per-device one-time state for host-side launch code.  hipFuncSetAttribute (raising the
// dynamic-LDS limit) and hipGetSymbolAddress act on the CURRENT device, so a flag or address cached in a plain static
// would be wrong for a second GPU driven by the same process.  One slot per device ordinal; races are benign (the
// guarded calls are idempotent).
#pragma once
#include <hip/hip_runtime.h>

namespace segmif {

constexpr int kMaxDevices = 64;

inline int current_device_slot() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < kMaxDevices ? dev : 0;
}

struct PerDeviceFlag {
  bool done[kMaxDevices] = {};
  bool& here() { return done[current_device_slot()]; }
};

template <typename T>
struct PerDeviceValue {
  T v[kMaxDevices] = {};
  T& here() { return v[current_device_slot()]; }
};

}  // namespace segmif
