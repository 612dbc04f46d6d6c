// Library identification and error strings of libammc_hip.so.
#include "ammc_common.h"

extern "C" int ammc_abi_version(void) { return 22; }

extern "C" const char* ammc_build_info(void) {
  return "libammc_hip gfx950 (CDNA4) fp32-MFMA build, HIP " __VERSION__;
}

extern "C" const char* ammc_error_string(int code) {
  switch (code) {
    case AMMC_OK: return "ok";
    case AMMC_EINVAL: return "invalid argument (null pointer, bad shape, stride or alignment)";
    case AMMC_EUNSUP: return "shape not supported by the gfx950 kernels";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown ammc error";
  }
}
