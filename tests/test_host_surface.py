"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares,
struct layouts agree, pure-host queries work, and the Python mirror of the reference's surface
(group info tables, pixel->token map, utils) matches the golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _t(a):
    return tuple(int(x) for x in a)


def test_library_exports_every_declared_symbol():
    from vorta_amd import _C
    lib = _C.lib()
    header = open(os.path.join(ROOT, "include", "vorta_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vorta_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_C.SYMBOLS), declared ^ set(_C.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vorta_abi_version() == _C.ABI_VERSION
    # the product library carries no experiment knob (variant builds list their -D flags here, vorta_amd/build.py
    # refuses extra flags without a suffix)
    assert "-DVORTA" not in lib.vorta_build_info().decode()
    assert b"gfx950" in lib.vorta_build_info()
    for which, st in enumerate((_C.Tensor, _C.AttnArgs, _C.CoresetArgs, _C.StaArgs, _C.RouterArgs, _C.NormRopeArgs)):
        assert lib.vorta_sizeof(which) == ctypes.sizeof(st)
    assert lib.vorta_sizeof(99) == -1


def test_argument_validation_happens_before_any_launch():
    """EINVAL / EUNSUPPORTED paths return on the host (no GPU needed)."""
    from vorta_amd import _C
    lib = _C.lib()
    a = _C.AttnArgs()
    a.struct_size = 7
    assert lib.vorta_attn_fwd(ctypes.byref(a), None) == _C.VORTA_EINVAL
    a.struct_size = ctypes.sizeof(_C.AttnArgs)
    a.dtype, a.head_dim, a.n_heads, a.n_q, a.n_kv, a.n_splits = 0, 64, 1, 64, 64, 1
    assert lib.vorta_attn_fwd(ctypes.byref(a), None) == _C.VORTA_EUNSUPPORTED
    a.head_dim = 128  # null tensors
    assert lib.vorta_attn_fwd(ctypes.byref(a), None) == _C.VORTA_EINVAL
    a.n_heads = 0  # nothing to do is not an error
    assert lib.vorta_attn_fwd(ctypes.byref(a), None) == _C.VORTA_OK
    with pytest.raises(ValueError):
        _C.check(_C.VORTA_EINVAL, "x")
    with pytest.raises(_C.VortaHipError):
        _C.check(_C.VORTA_EUNSUPPORTED, "x")


def test_ops_refuse_cpu_tensors():
    from vorta_amd import _C, ops
    q = torch.zeros((1, 64, 128), dtype=torch.bfloat16)
    with pytest.raises(_C.VortaHipError):
        ops.attn_fwd(q, q, q, q.clone(), n_q=64, n_kv=64)
    with pytest.raises(_C.VortaHipError):
        ops.coreset_select(q, (4, 4, 4), (2, 2, 2), 3)


@pytest.mark.parametrize("latent,tile,window,te", [((8, 6, 8), (2, 3, 4), (3, 3, 3), 11), ((4, 6, 8), (2, 3, 4), (3, 3, 3), 5),
                                                   ((10, 6, 8), (2, 3, 4), (5, 3, 1), 0), ((33, 45, 80), (11, 9, 8), (3, 3, 3), 96)])
def test_sta_table_sizes_host_query(latent, tile, window, te):
    from vorta_amd import ops
    n_tiles, tok, n_kv = ops.sta_table_sizes(latent, tile, window, te)
    vis = O.sta_window_tiles(latent, tile, window)
    assert n_tiles == vis.shape[0] and tok == tile[0] * tile[1] * tile[2]
    assert np.all(vis.sum(1) * tok + te == n_kv)  # every q tile sees the same number of keys
    with pytest.raises(ValueError):
        ops.sta_table_sizes((7, 6, 8), tile, window, 0)


@pytest.mark.parametrize("tag", ["4x6x4", "8x6x8", "9x6x8_g18", "9x7x9_crop", "8x6x8_r075"])
def test_get_group_info_golden(golden, tag):
    from vorta_amd.attention import get_group_info
    g = golden("g1_group_info")
    gi = get_group_info(_t(g[f"{tag}_latent"]), _t(g[f"{tag}_window"]), float(g[f"{tag}_rate"]))
    assert np.array_equal(gi.center_indices.numpy(), g[f"{tag}_center"])
    assert np.array_equal(gi.margin_indices.numpy(), g[f"{tag}_margin"])
    assert gi.num_unpooled_tokens_per_group == int(g[f"{tag}_num_unpooled"])


def test_pixel2token_golden(golden):
    from vorta_amd.patch import hunyuan_pixel2token, wan_pixel2token
    g = golden("g10_pixel2token")
    for size, tok in zip(g["sizes"], g["tokens_hunyuan"]):
        assert hunyuan_pixel2token(_t(size)) == _t(tok) == wan_pixel2token(_t(size))
    for size, raises in zip(g["bad_sizes"], g["bad_raises"]):
        assert raises
        with pytest.raises(ValueError):
            hunyuan_pixel2token(_t(size))


def test_prepare_kwargs_contract():
    from vorta_amd.attention import LowresGroupInfo, SlidingTileDescriptor
    from vorta_amd.patch import prepare_hunyuan_self_attn_kwargs, prepare_wan_self_attn_kwargs
    base = dict(latent_shape=(8, 6, 8), window_size=(3, 3, 3), tile_size=(2, 3, 4), lowres_window_size=(2, 3, 2),
                lowres_reduction_rate=0.5)
    kw = prepare_hunyuan_self_attn_kwargs(dict(base), torch.device("cpu"), tau_sparse=0.3)
    assert set(kw) == {"latent_shape", "window_size", "tile_size", "lowres_group_info", "tau_sparse"}
    assert isinstance(kw["lowres_group_info"], LowresGroupInfo) and kw["lowres_group_info"].num_unpooled_tokens_per_group == 5
    kw = prepare_wan_self_attn_kwargs(dict(base), torch.device("cpu"))
    assert isinstance(kw["flex_attn_mask_func"], SlidingTileDescriptor) and "tau_sparse" not in kw
    with pytest.raises(ValueError):
        prepare_wan_self_attn_kwargs(dict(base, tile_size=(3, 3, 4)), torch.device("cpu"))


def test_utils(tmp_path):
    import argparse
    from pathlib import Path
    from vorta_amd import utils as U
    assert U.str_to_dtype("BF16") is torch.bfloat16 and U.dtype_to_str(torch.float16) == "fp16"
    with pytest.raises(ValueError):
        U.str_to_dtype("int8")
    assert U.prompt_to_file_name("A cat, on a mat -- sleeping!", prefix=7, suffix=2) == "007-a-cat-on-a-mat-sleep-02"
    assert U.parent_to_ckpt_dir(None, tmp_path) == (None, 0)
    assert U.parent_to_ckpt_dir("latest", tmp_path) == (None, 0)
    for s in (10, 200, 30):
        (tmp_path / f"step-{s:06d}").mkdir()
    assert U.parent_to_ckpt_dir("latest", tmp_path) == (tmp_path / "step-000200", 200)
    assert U.parent_to_ckpt_dir("step-000030", tmp_path) == (tmp_path / "step-000030", 30)
    with pytest.raises(FileNotFoundError):
        U.parent_to_ckpt_dir("step-000031", tmp_path)
    js = U.arg_to_json(argparse.Namespace(b=Path("/x/y"), a=1))
    assert js.index('"a"') < js.index('"b"') and '"/x/y"' in js


def test_router_module_state_dict_keys():
    from vorta_amd.patch import Router
    r = Router(48, 6)
    assert set(r.state_dict()) == {"linear.weight", "linear.bias"}
    assert r.linear.weight.shape == (18, 48)


def test_prepare_kwargs_validates_geometry_early():
    """the published (6,9,8)/(2,3,2) geometry fits the 117-frame latent, not the 129-frame one"""
    import torch
    from vorta.patch.utils import hunyuan_pixel2token, prepare_hunyuan_self_attn_kwargs, prepare_wan_self_attn_kwargs
    cfg = dict(window_size=(3, 3, 3), tile_size=(6, 9, 8), lowres_window_size=(2, 3, 2), lowres_reduction_rate=0.5)
    ok = prepare_hunyuan_self_attn_kwargs(dict(cfg, latent_shape=hunyuan_pixel2token((117, 720, 1280))), torch.device("cpu"))
    assert ok["lowres_group_info"].window_size == (2, 3, 2)
    with pytest.raises(ValueError, match="does not divide latent shape"):
        prepare_hunyuan_self_attn_kwargs(dict(cfg, latent_shape=hunyuan_pixel2token((129, 720, 1280))), torch.device("cpu"))
    with pytest.raises(ValueError, match="Low-res window"):
        prepare_wan_self_attn_kwargs(dict(cfg, tile_size=(3, 9, 8), latent_shape=(21, 45, 80)), torch.device("cpu"))


@pytest.mark.parametrize("sp", [1, 2])
def test_tile_layout_module_matches_reference_permutation(sp):
    """vorta.attention.tile.{tile_layout, untile_layout} (tile.py:7-78) against golden G4, both head layouts"""
    from vorta.attention.tile import tile_layout, untile_layout
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4_tile_perm.npz"))
    latent, tile = (8, 6, 8), (2, 3, 4)
    n = latent[0] * latent[1] * latent[2]
    x = torch.arange(n, dtype=torch.float32).view(1, 1, n, 1).expand(1, 2, n, 3)  # value = source position
    tiled = tile_layout(x, sp, tile, latent, head_dim=1)
    assert np.array_equal(tiled[0, 0, :, 0].long().numpy(), g[f"sp{sp}_tiled_src"])
    assert torch.equal(untile_layout(tiled, sp, tile, latent, head_dim=1), x)
    x2 = x.transpose(1, 2)  # (B,S,H,D), the default head_dim=2
    assert torch.equal(tile_layout(x2, sp, tile, latent).transpose(1, 2), tiled)
    assert torch.equal(untile_layout(tile_layout(x2, sp, tile, latent), sp, tile, latent), x2)
    with pytest.raises(ValueError):
        tile_layout(x, 1, (3, 3, 4), latent, head_dim=1)


def test_integration_md_stub_matches_the_library():
    """the ctypes struct a reference maintainer would paste has the layout of vorta_attn_args"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = text[text.index("_lib = None"):text.index("def dense_attention")]
    ns = {"C": ctypes}
    exec(code, ns)
    from vorta_amd import _C
    assert ctypes.sizeof(ns["_AttnArgs"]) == _C.lib().vorta_sizeof(1) == ctypes.sizeof(_C.AttnArgs)
    assert [f[0] for f in ns["_AttnArgs"]._fields_] == [f[0] for f in _C.AttnArgs._fields_]


def test_torch_custom_ops_are_registered_with_shape_functions():
    """torch.ops.vorta.*: schemas with declared mutations, fake (shape) functions usable without a GPU"""
    import vorta_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    names = {"attn_fwd", "coreset_select", "sta_build_tables", "route_scores", "router_route", "qk_norm_rope", "mix_experts"}
    assert names <= set(dir(torch.ops.vorta))
    schema = str(torch.ops.vorta.attn_fwd.default._schema)
    assert "Tensor(a3!) out" in schema and "Tensor? kv_rows=None" in schema and schema.endswith("-> ()")
    assert "Tensor(a0!) x" in str(torch.ops.vorta.qk_norm_rope.default._schema)
    with FakeTensorMode():
        x = torch.empty((3, 8 * 6 * 8 + 5, 128), dtype=torch.bfloat16, device="cuda")
        keep, drop = torch.ops.vorta.coreset_select(x, [8, 6, 8], [2, 3, 2], 5, tail_first=384, n_tail=5)
        assert keep.shape == (3, 32 * 6 + 5) and drop.shape == (3, 32, 6) and keep.dtype == torch.int32
        q_rows, kv_rows = torch.ops.vorta.sta_build_tables(x, [8, 6, 8], [2, 3, 4], [3, 3, 3], 4)
        assert q_rows.shape == (384,) and kv_rows.shape[0] == 16
        e, lists, counts = torch.ops.vorta.route_scores(torch.empty((1, 6, 3), device="cuda"), 0.3)
        assert e.shape == (6,) and lists.shape == (3, 6) and counts.shape == (3,)
        assert torch.ops.vorta.attn_fwd(x, x, x, torch.empty_like(x), 10, 10) is None


def test_bench_window_tile_matrix_is_the_oracles_table():
    """bench.py restates the clamped tile window for its borrowed-kernel context measurement (compiled flex_attention
    under that mask); the restatement has to be the mask the oracle pins against the reference's (G3)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for lat, tile, win in [((33, 45, 80), (11, 9, 8), (3, 3, 3)), ((21, 45, 80), (7, 9, 8), (3, 3, 3)), ((8, 6, 8), (2, 3, 4), (3, 3, 3)),
                           ((30, 45, 80), (6, 9, 8), (3, 3, 3)), ((4, 6, 4), (2, 3, 2), (3, 3, 3)), ((21, 30, 52), (7, 6, 4), (3, 3, 1))]:
        got = bench.window_tile_matrix(lat, tile, win, "cpu").numpy()
        assert np.array_equal(got, O.sta_window_tiles(lat, tile, win)), (lat, tile, win)


def test_all_e4m3_scores_are_not_a_product_precision():
    """"fp8" (e4m3 q k^T) reaches 20.8-35.7 dB on structured inputs -- no product switch may select it: the
    setter refuses it without `measurement_only=True`, the environment variable refuses it at import."""
    import subprocess
    import sys

    import vorta_amd
    from vorta_amd import routed
    before = routed.DEFAULT_FP8
    try:
        with pytest.raises(ValueError, match="not a product precision"):
            vorta_amd.set_attention_precision("fp8")
        assert routed.DEFAULT_FP8 == before
        for p in routed.PRODUCT_PRECISIONS:
            vorta_amd.set_attention_precision(p)
            assert routed.DEFAULT_FP8 == (False if p == "native" else p)
        vorta_amd.set_attention_precision("fp8", measurement_only=True)
        assert routed.DEFAULT_FP8 is True
        with pytest.raises(ValueError):
            vorta_amd.set_attention_precision("fp4")
    finally:
        routed.DEFAULT_FP8 = before
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import vorta_amd.routed"], cwd=root, capture_output=True, text=True,
                       env=dict(os.environ, VORTA_ATTENTION_PRECISION="fp8"))
    assert r.returncode != 0 and "not a product precision" in r.stderr
    r = subprocess.run([sys.executable, "-c", "import vorta_amd.routed as r; print(r.DEFAULT_FP8)"], cwd=root, capture_output=True,
                       text=True, env=dict(os.environ, VORTA_ATTENTION_PRECISION="auto8"))
    assert r.returncode == 0 and r.stdout.strip() == "auto8", r.stderr[-500:]
