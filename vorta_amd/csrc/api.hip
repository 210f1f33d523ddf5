// Introspection entry points and the error slot of libvorta_hip.
#include <hip/hip_runtime.h>

#include "vorta_hip.h"
#include "common.h"

static thread_local int g_last_hip_error = 0;

int vorta_set_hip_error(hipError_t e) {
  g_last_hip_error = (int)e;
  return VORTA_ELAUNCH;
}

extern "C" int vorta_abi_version(void) { return VORTA_ABI_VERSION; }
// a variant build (experimental -D flags: vorta_amd/build.py and tools/dbg/build_variant.sh pass the same flags to every source,
// and build.py refuses them for the product library) says so in its build info: the builders hand the flag list over as
// VORTA_VARIANT_FLAGS
static const char kBuildInfo[] = "libvorta_hip gfx950 (CDNA4) hipcc " __VERSION__
#ifdef VORTA_VARIANT_FLAGS
    " " VORTA_VARIANT_FLAGS
#endif
    ;
extern "C" const char* vorta_build_info(void) { return kBuildInfo; }
extern "C" int vorta_last_hip_error(void) { return g_last_hip_error; }

// struct sizes, so a binding can verify its layout before the first call
extern "C" int vorta_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(vorta_tensor);
    case 1: return (int)sizeof(vorta_attn_args);
    case 2: return (int)sizeof(vorta_coreset_args);
    case 3: return (int)sizeof(vorta_sta_args);
    case 4: return (int)sizeof(vorta_router_args);
    case 5: return (int)sizeof(vorta_norm_rope_args);
    case 6: return (int)sizeof(vorta_mix_args);
    case 7: return (int)sizeof(vorta_fp8_quant_args);
    case 8: return (int)sizeof(vorta_attn_fp8_ext);
    case 9: return (int)sizeof(vorta_permute_args);
    case 10: return (int)sizeof(vorta_fp8_v_args);
    case 11: return (int)sizeof(vorta_i8_quant_args);
    case 12: return (int)sizeof(vorta_attn_i8_ext);
    default: return -1;
  }
}
