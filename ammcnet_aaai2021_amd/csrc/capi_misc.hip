// Library identification and error strings of libammc_hip.so.
#include "ammc_common.h"

#include <stdlib.h>
#include <string.h>

extern "C" int ammc_abi_version(void) { return 36; }

// dispatch options (ammc_common.h): initialised from the environment once, changed by ammc_set_option
int g_ammc_s16_mf = -2;              // -2 = not read yet; -1 = auto (per variant); 0 / 1 = forced

int g_ammc_outc_stream = -2;        // -2 = not read yet; 1 = the streaming output-layer kernel (default), 0 = halo-patch kernel

int ammc_opt_outc_stream() {
  if (g_ammc_outc_stream == -2) {
    const char* e = getenv("AMMC_OUTC_STREAM");
    g_ammc_outc_stream = e ? atoi(e) != 0 : 1;
  }
  return g_ammc_outc_stream;
}

int g_ammc_memory_rt = -2;          // -2 = not read yet; 0 = by size (default), 1 / 2 = 32- / 64-row workgroups of memory_topk_s16

int ammc_opt_memory_rt() {
  if (g_ammc_memory_rt == -2) {
    const char* e = getenv("AMMC_MEMORY_RT");
    const int v = e ? atoi(e) : 0;
    g_ammc_memory_rt = v == 1 || v == 2 ? v : 0;
  }
  return g_ammc_memory_rt;
}

int g_ammc_memory_split = -2;      // -2 = not read yet; -1 / 0 = fused launch (default), 1 = split contraction / gather

int ammc_opt_memory_split() {
  if (g_ammc_memory_split == -2) {
    const char* e = getenv("AMMC_MEMORY_SPLIT");
    const int v = e ? atoi(e) : -1;
    g_ammc_memory_split = v == 0 || v == 1 ? v : -1;
  }
  return g_ammc_memory_split;
}

int ammc_opt_s16_mf() {
  if (g_ammc_s16_mf == -2) {
    const char* e = getenv("AMMC_S16_MF");
    g_ammc_s16_mf = e ? (atoi(e) < 0 ? -1 : atoi(e) != 0) : -1;
  }
  return g_ammc_s16_mf;
}

extern "C" int ammc_set_option(const char* key, int32_t value) {
  if (!key) return AMMC_EINVAL;
  if (!strcmp(key, "s16_mf")) {
    if (value < -1 || value > 1) return AMMC_EINVAL;
    g_ammc_s16_mf = value;
    return AMMC_OK;
  }
  if (!strcmp(key, "outc_stream")) {
    if (value < 0 || value > 1) return AMMC_EINVAL;
    g_ammc_outc_stream = value;
    return AMMC_OK;
  }
  if (!strcmp(key, "memory_rt")) {
    if (value < 0 || value > 2) return AMMC_EINVAL;
    g_ammc_memory_rt = value;
    return AMMC_OK;
  }
  if (!strcmp(key, "memory_split")) {
    if (value < -1 || value > 1) return AMMC_EINVAL;
    g_ammc_memory_split = value;
    return AMMC_OK;
  }
  return AMMC_EUNSUP;
}

#ifndef AMMC_SRC_DIGESTS
#define AMMC_SRC_DIGESTS ""
#endif
// "file=sha256[:12],..." of the sources this library was compiled from (ammcnet_aaai2021_amd/build.py: file_digests)
extern "C" const char* ammc_source_digests(void) { return AMMC_SRC_DIGESTS; }

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
