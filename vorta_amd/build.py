"""Build libvorta_hip.so (hand-written HIP kernels + C ABI) for gfx950, in tree.

    python -m vorta_amd.build          # or  __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libvorta_hip.so")
SOURCES = ["api.hip", "attn_fwd.hip", "attn_fwd_fp8.hip", "attn_fwd_mx.hip", "attn_fwd_i8.hip", "fp8_quant.hip", "i8_quant.hip", "coreset.hip",
           "sta_tables.hip", "router.hip", "qk_norm_rope.hip", "mix.hip", "permute.hip"]
# -fno-slp-vectorize: the SLP vectoriser packs adjacent fp32 adds of the softmax row sum into v_pk_add_f32 plus the
# v_mov pairs to feed them -- more instructions on the VALU issue port that bounds the attention loop (+2 % without)
# -enable-post-misched=0: the post-RA scheduler re-orders the hand-interleaved MFMA / VALU / LDS stream of the attention
# loop for the worse (+1.7 % on the fused layer kernel without it; measured A/B in one session)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0",
         "-I" + INCLUDE, "-I" + CSRC, "-Wno-unused-result"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True, extra_flags=None, suffix: str = "") -> str:
    """Compile the sources and link the library.  Experimental flags (`extra_flags`, or VORTA_EXTRA_FLAGS in the
    environment) never touch the product library: they require a `suffix` (VORTA_BUILD_SUFFIX) and produce
    csrc/libvorta_hip<suffix>.so from objects of their own (load it with VORTA_HIP_LIB for a same-session A/B)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra_flags) if extra_flags is not None else os.environ.get("VORTA_EXTRA_FLAGS", "").split()
    suffix = suffix or os.environ.get("VORTA_BUILD_SUFFIX", "")
    if extra and not suffix:
        raise SystemExit("VORTA_EXTRA_FLAGS needs VORTA_BUILD_SUFFIX: experimental flags are never built into "
                         "libvorta_hip.so itself")
    lib = LIB if not suffix else os.path.join(CSRC, f"libvorta_hip{suffix}.so")
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(CSRC, "attn_fwd_fp8_diag.inc"),
            os.path.join(CSRC, "attn_fwd_i8_diag.inc"), os.path.join(CSRC, "attn_fwd_diag.inc"), os.path.join(INCLUDE, "vorta_hip.h")]
    if extra:  # vorta_build_info() of a variant library names its flags
        extra = extra + ['-DVORTA_VARIANT_FLAGS="%s"' % " ".join(extra).replace('"', "'")]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", f"{suffix}.o"))
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print("[vorta_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or any(_newer(o, lib) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print("[vorta_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


SP_LIB = os.path.join(CSRC, "libvorta_sp.so")


def build_sp(force: bool = False, verbose: bool = True) -> str:
    """libvorta_sp.so (include/vorta_sp.h): the Ulysses exchange on RCCL behind a C ABI -- host code, linked against librccl; a
    library of its own, so that libvorta_hip.so (kernels) keeps depending on the HIP runtime alone"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "sp_rccl.hip")
    if force or _newer(src, SP_LIB) or _newer(os.path.join(INCLUDE, "vorta_sp.h"), SP_LIB):
        rocm = os.path.dirname(os.path.dirname(hipcc))
        cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, src, "-o", SP_LIB,
               "-L" + os.path.join(rocm, "lib"), "-lrccl"]
        if verbose:
            print("[vorta_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SP_LIB


def kernel_resources(source: str):
    """Compile one source to gfx950 assembly with the product flags and return, per kernel symbol, the register /
    spill / scratch / LDS figures of its code-object metadata (works without a GPU)."""
    import re
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, source), "-o", out])
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels"):]
    res = {}
    for block in meta.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda key: int(re.search(key + r":\s+(\d+)", block).group(1))
        res[name] = dict(vgpr=get(r"\.vgpr_count"), vgpr_spill=get(r"\.vgpr_spill_count"), sgpr_spill=get(r"\.sgpr_spill_count"),
                         scratch=get(r"\.private_segment_fixed_size"), lds=get(r"\.group_segment_fixed_size"))
    return res


if __name__ == "__main__":
    build(force="--force" in sys.argv or bool(os.environ.get("VORTA_BUILD_SUFFIX")))
    build_sp(force="--force" in sys.argv)
