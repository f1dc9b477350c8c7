// library / device probe
#include <string.h>

#include "cnr_common.h"

extern "C" int cnr_version(void) { return 100; }  // 0.1.0

extern "C" int cnr_device_info(int* n_cu, int* lds_bytes, int* gcn_arch_is_gfx950) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (gcn_arch_is_gfx950) *gcn_arch_is_gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  return CNR_OK;
}
