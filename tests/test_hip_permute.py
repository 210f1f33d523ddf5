"""GPU parity of vorta_permute_heads (the staging passes of the Ulysses exchange) against the torch
index ops the engine uses on CPU tensors -- a copy: bit-exact."""
import pytest
import torch

from _util import dev

pytestmark = pytest.mark.gpu


def _order(H, seed):
    return torch.randperm(H, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_send_side_staging_of_three_projection_views(dtype):
    """q, k, v as the (rows, H*D) projection outputs viewed (H, rows, D) -> head-ordered contiguous blocks, one launch"""
    from vorta_amd import ops
    H, N = 24, 777
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn((N + 5, H, 128), generator=g).to(dtype).to(dev()) for _ in range(3)]
    views = [x[:N].transpose(0, 1) for x in xs]  # (H, N, D), strides (D, H*D, 1)
    order = _order(H, 1)
    outs = [torch.full((H, N, 128), 7.0, dtype=dtype, device=dev()) for _ in range(3)]
    ops.permute_heads(views, outs, src_map=order.to(torch.int32).to(dev()))
    for v, o in zip(views, outs):
        assert torch.equal(o, torch.index_select(v, 0, order.to(dev())))


def test_receive_side_unpermute_into_a_strided_result():
    """head-ordered contiguous blocks -> the (rows, H, D) buffer the output projection reads (`index_copy_`)"""
    from vorta_amd import ops
    H, N = 12, 1000
    g = torch.Generator().manual_seed(2)
    src = torch.randn((H, N, 128), generator=g).to(torch.bfloat16).to(dev())
    order = _order(H, 3)
    res = torch.zeros((N + 3, H, 128), dtype=torch.bfloat16, device=dev())
    ops.permute_heads([src], [res[:N].transpose(0, 1)], dst_map=order.to(torch.int32).to(dev()))
    ref = torch.zeros_like(res)
    ref[:N].transpose(0, 1).index_copy_(0, order.to(dev()), src)
    assert torch.equal(res, ref)


def test_text_rows_behind_each_local_head_slot_and_e4m3_bytes():
    """the replicated text rows of the rank's own heads land behind each head slot's video rows (destination head stride
    = Sl rows); also 1-byte elements, four tensors, a map that selects Hl of the source's H heads"""
    from vorta_amd import ops
    H, Hl, T, Sl = 24, 3, 19, 50
    g = torch.Generator().manual_seed(4)
    mine = torch.tensor([17, 2, 9], dtype=torch.int32)
    for dtype in (torch.float16, torch.uint8):
        texts = [(torch.randn((H, T, 128), generator=g) * 40).to(dtype).to(dev()) for _ in range(4)]
        bufs = [torch.zeros((Hl * Sl + 64, 128), dtype=dtype, device=dev()) for _ in range(4)]
        rows_video = 7 if dtype == torch.float16 else 16
        dsts = [b[rows_video:].as_strided((Hl, T, 128), (Sl * 128, 128, 1)) for b in bufs]
        ops.permute_heads(texts, dsts, src_map=mine.to(dev()))
        for t, b in zip(texts, bufs):
            ref = torch.zeros_like(b)
            for i in range(Hl):
                ref[rows_video + i * Sl:rows_video + i * Sl + T] = t[int(mine[i])]
            assert torch.equal(b, ref)


def test_bad_arguments():
    from vorta_amd import ops
    x = torch.zeros((4, 8, 128), dtype=torch.bfloat16, device=dev())
    y = torch.zeros((4, 8, 128), dtype=torch.bfloat16, device=dev())
    with pytest.raises(ValueError):
        ops.permute_heads([x], [y], src_map=torch.arange(4, device=dev()))  # int64 map
    with pytest.raises(ValueError):
        ops.permute_heads([x, x, x, x, x], [y, y, y, y, y])
    with pytest.raises(ValueError):
        ops.permute_heads([x.to(torch.float16)], [y])
    with pytest.raises(ValueError):
        ops.permute_heads([x[:2]], [y])  # fewer source heads and no map
    ops.permute_heads([x[:, :0]], [y[:, :0]])  # empty: accepted, no launch


def test_bandwidth_at_a_rank_of_eight():
    """one rank's q, k, v staging at Hunyuan-129f / P = 8 (3 x 91 MB in, 3 x 91 MB out): HBM-rate, not index-kernel rate"""
    from vorta_amd import ops
    H, Sl = 24, 14850
    xs = [torch.randn((Sl, H, 128), device=dev(), dtype=torch.bfloat16) for _ in range(3)]
    views = [x.transpose(0, 1) for x in xs]
    outs = [torch.empty((H, Sl, 128), dtype=torch.bfloat16, device=dev()) for _ in range(3)]
    m = _order(H, 5).to(torch.int32).to(dev())
    ops.permute_heads(views, outs, src_map=m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.permute_heads(views, outs, src_map=m)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tbs = 6 * H * Sl * 256 / ms / 1e9
    print(f"permute_heads: {ms * 1e3:.0f} us, {tbs:.2f} TB/s")
    assert tbs > 0.5, tbs  # ~5.5 TB/s on a quiet box; the gate only catches a launch that is not HBM-streaming at all


@pytest.mark.parametrize("groups", [1, 2])
def test_exchange_engine_on_device_equals_the_cpu_engine(groups):
    """ulysses/engine.py on device tensors takes the vorta_permute_heads passes (send staging, text rows, output
    un-permute, text scatter); on CPU tensors the torch index ops.  Same data, one rank (P = 1, every pass still runs
    because the head order is a permutation): identical buffers and outputs."""
    from vorta_amd.ulysses import UlyssesLayout, exchange_and_attend, slot_groups
    H, S, T, D = 6, 192, 7, 128
    order = [3, 0, 5, 1, 4, 2]
    g = torch.Generator().manual_seed(0)
    shards_c = [torch.randn((S, H, D), generator=g).to(torch.bfloat16).transpose(0, 1) for _ in range(3)]
    texts_c = [torch.randn((H, T, D), generator=g).to(torch.bfloat16) for _ in range(3)]
    res = {}
    for device in ("cpu", dev()):
        lay = UlyssesLayout(H, S, T, D, 1, 0, device, torch.bfloat16)
        shards = [x.transpose(0, 1).contiguous().to(device).transpose(0, 1) for x in shards_c]  # keep the (S,H,D) storage
        texts = [x.to(device) for x in texts_c]
        b = [lay.new_buffer() for _ in range(4)]

        sgs = lay.grouping(groups)[0]  # (with P = 1 a slot group's rows are the slots' own: same buffers either way)

        def attend(g0, g1, gi):
            sgs[gi].head_view(b[3]).copy_(sgs[gi].head_view(b[0]))

        o = torch.zeros((S, H, D), dtype=torch.bfloat16, device=device).transpose(0, 1)
        t = torch.zeros((H, T, D), dtype=torch.bfloat16, device=device)
        exchange_and_attend(lay, shards, b, order, texts, slot_groups(lay.Hl, groups), attend, o, t)
        res[str(device)] = [x.cpu() for x in (*b[:3], o, t)]
    for x, y in zip(*res.values()):
        assert torch.equal(x, y)
    assert torch.equal(res["cpu"][3], shards_c[0]) and torch.equal(res["cpu"][4], texts_c[0])
