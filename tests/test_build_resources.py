"""CPU-side check of the compiled attention kernels (hipcc cross-compiles gfx950 without a GPU): no kernel of the hot
path may spill vector registers or use scratch -- a spill is a silent 10-20 % on a loop that runs at the register
budget (attn_fwd_pipe_kernel<T,4,true> spilled 5 VGPRs)."""
import shutil

import pytest

from vorta_amd import build

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not __import__("os").path.exists("/opt/rocm/bin/hipcc"),
                                reason="needs hipcc")


@pytest.mark.parametrize("source", ["attn_fwd.hip", "attn_fwd_fp8.hip"])
def test_attention_kernels_do_not_spill(source):
    res = build.kernel_resources(source)
    assert len(res) >= 10
    for name, r in res.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, r)
        assert r["vgpr"] <= 256, (name, r)  # two waves per SIMD
        assert r["lds"] <= 80 * 1024, (name, r)
