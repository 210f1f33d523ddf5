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
// every compile-time knob of the kernels that was given on the command line (vorta_amd/build.py passes the same extra flags
// to every source, and refuses them for the product library): a variant build says so in its build info
#define VORTA_STR2_(x) #x
#define VORTA_STR_(x) VORTA_STR2_(x)
static const char kBuildInfo[] = "libvorta_hip gfx950 (CDNA4) hipcc " __VERSION__
#ifdef VORTA_FP8_DIAG
    " -DVORTA_FP8_DIAG"
#endif
#ifdef VORTA_TRACE8
    " -DVORTA_TRACE8=" VORTA_STR_(VORTA_TRACE8)
#endif
#ifdef VORTA_TRACE
    " -DVORTA_TRACE"
#endif
#ifdef VORTA_DIAG_NOMFMA
    " -DVORTA_DIAG_NOMFMA"
#endif
#ifdef VORTA_DIAG_NOLDSRD
    " -DVORTA_DIAG_NOLDSRD"
#endif
#ifdef VORTA_DIAG_NODEP
    " -DVORTA_DIAG_NODEP"
#endif
#ifdef VORTA_DIAG_NOEXP
    " -DVORTA_DIAG_NOEXP"
#endif
#ifdef VORTA_DIAG_NOCVT
    " -DVORTA_DIAG_NOCVT"
#endif
#ifdef VORTA_DIAG_NOMAX
    " -DVORTA_DIAG_NOMAX"
#endif
#ifdef VORTA_DIAG_NODMA
    " -DVORTA_DIAG_NODMA"
#endif
#ifdef VORTA_DIAG_NOBAR
    " -DVORTA_DIAG_NOBAR"
#endif
#ifdef VORTA_DIAG_NOSYNC
    " -DVORTA_DIAG_NOSYNC"
#endif
#ifdef VORTA_DIAG_ALLX
    " -DVORTA_DIAG_ALLX"
#endif
#ifdef VORTA_RING
    " -DVORTA_RING=" VORTA_STR_(VORTA_RING)
#endif
#ifdef VORTA_KPRE
    " -DVORTA_KPRE=" VORTA_STR_(VORTA_KPRE)
#endif
#ifdef VORTA_SCHED
    " -DVORTA_SCHED=" VORTA_STR_(VORTA_SCHED)
#endif
#ifdef VORTA_MX_KPRE
    " -DVORTA_MX_KPRE=" VORTA_STR_(VORTA_MX_KPRE)
#endif
#ifdef VORTA_MX_SCHED
    " -DVORTA_MX_SCHED=" VORTA_STR_(VORTA_MX_SCHED)
#endif
#ifdef VORTA_MX_SC_VALU
    " -DVORTA_MX_SC_VALU=" VORTA_STR_(VORTA_MX_SC_VALU)
#endif
#ifdef VORTA_MX_PV_VALU
    " -DVORTA_MX_PV_VALU=" VORTA_STR_(VORTA_MX_PV_VALU)
#endif
#ifdef VORTA_I8_SCHED
    " -DVORTA_I8_SCHED=" VORTA_STR_(VORTA_I8_SCHED)
#endif
#ifdef VORTA_I8_SC_VALU
    " -DVORTA_I8_SC_VALU=" VORTA_STR_(VORTA_I8_SC_VALU)
#endif
#ifdef VORTA_I8_PV_VALU
    " -DVORTA_I8_PV_VALU=" VORTA_STR_(VORTA_I8_PV_VALU)
#endif
#ifdef VORTA_SCHED8
    " -DVORTA_SCHED8=" VORTA_STR_(VORTA_SCHED8)
#endif
#ifdef VORTA_PRIO8
    " -DVORTA_PRIO8=" VORTA_STR_(VORTA_PRIO8)
#endif
#ifdef VORTA_DMA_SPLIT
    " -DVORTA_DMA_SPLIT=" VORTA_STR_(VORTA_DMA_SPLIT)
#endif
#ifdef VORTA_MULTI_SWAP
    " -DVORTA_MULTI_SWAP=" VORTA_STR_(VORTA_MULTI_SWAP)
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
