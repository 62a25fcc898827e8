// Library identification helpers for libsegmif_hip.so.
#include <hip/hip_runtime.h>
#include <string.h>

#include "segmif_hip.h"

extern "C" int segmif_abi_version(void) { return SEGMIF_ABI_VERSION; }

extern "C" int segmif_device_name(char* buf, int len) {
  if (!buf || len <= 0) return SEGMIF_EINVAL;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}
