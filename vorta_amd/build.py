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
SOURCES = ["api.hip", "attn_fwd.hip", "coreset.hip", "sta_tables.hip", "router.hip", "qk_norm_rope.hip", "mix.hip"]
# -fno-slp-vectorize: the SLP vectoriser packs adjacent fp32 adds of the softmax row sum into v_pk_add_f32 plus the
# v_mov pairs to feed them -- more instructions on the VALU issue port that bounds the attention loop (+2 % without)
# -enable-post-misched=0: the post-RA scheduler re-orders the hand-interleaved MFMA / VALU / LDS stream of the attention
# loop for the worse (+1.7 % on the fused layer kernel without it; measured A/B in one session)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0",
         "-I" + INCLUDE, "-I" + CSRC, "-Wno-unused-result"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("VORTA_EXTRA_FLAGS", "").split()  # experiments only (e.g. -DVORTA_SCHED=1)
    force = force or bool(extra)
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(INCLUDE, "vorta_hip.h")]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print("[vorta_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[vorta_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
