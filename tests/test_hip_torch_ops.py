"""GPU: the launchers as `torch.ops.vorta.*` custom ops -- same bits as ops.py in eager mode, and traceable: a function
that mixes torch code with the HIP operators goes through `torch.compile(fullgraph=True)` (aot_eager: traced with
fake tensors through the registered shape functions, no code generation involved)."""
import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O
from _util import dev, rel_fro

pytestmark = pytest.mark.gpu


def test_custom_ops_match_the_direct_launchers():
    import vorta_amd.torch_ops  # noqa: F401  (registers torch.ops.vorta.*)
    from vorta_amd import ops
    torch.manual_seed(0)
    q, k, v = (torch.randn((3, 500, 128), device=dev()).to(torch.bfloat16) for _ in range(3))
    a, b = torch.empty_like(q), torch.empty_like(q)
    ops.attn_fwd(q, k, v, a, n_q=500, n_kv=430, q_valid=470)
    torch.ops.vorta.attn_fwd(q, k, v, b, 500, 430, q_valid=470)
    assert torch.equal(a, b)
    latent, group = [8, 6, 8], [2, 3, 2]
    x = torch.randn((2, 8 * 6 * 8 + 5, 128), device=dev()).to(torch.bfloat16)
    k1, d1 = ops.coreset_select(x, latent, group, 5, tail_first=384, n_tail=5)
    k2, d2 = torch.ops.vorta.coreset_select(x, latent, group, 5, tail_first=384, n_tail=5)
    assert torch.equal(k1, k2) and torch.equal(d1, d2)
    t1 = ops.sta_build_tables(latent, (2, 3, 4), (3, 3, 3), 4, dev())
    t2 = torch.ops.vorta.sta_build_tables(x, latent, [2, 3, 4], [3, 3, 3], 4)
    assert torch.equal(t1[0], t2[0]) and torch.equal(t1[1], t2[1])
    sc = torch.softmax(torch.randn((1, 6, 3), device=dev()), -1)
    for r1, r2 in zip(ops.route_scores(sc, 0.4), torch.ops.vorta.route_scores(sc, 0.4)):
        assert torch.equal(r1, r2)


def test_custom_ops_trace_under_torch_compile():
    import vorta_amd.torch_ops  # noqa: F401
    torch.manual_seed(1)
    H, S = 4, 384
    lin = torch.nn.Linear(64, H * 128).to(dev()).to(torch.bfloat16)
    w = torch.rand(128, device=dev()).to(torch.bfloat16) + 0.5

    def layer(x, scores):
        # torch code around the operators: a projection, a head split, the HIP norm + dense attention, a HIP mix
        qkv = lin(x)[0].view(S, H, 128).permute(1, 0, 2)          # (H,S,D) strided view
        q = qkv.clone()
        torch.ops.vorta.qk_norm_rope(q, w, 1e-6)
        outs = [torch.empty_like(q) for _ in range(3)]
        for i, n_kv in enumerate((S, S // 2, S // 4)):
            torch.ops.vorta.attn_fwd(q, qkv, qkv, outs[i], S, n_kv)
        mixed = torch.empty_like(q)
        torch.ops.vorta.mix_experts(outs[0], outs[1], outs[2], scores, mixed)
        return mixed.float().sum(dim=0) * 0.5

    x = torch.randn((1, S, 64), device=dev()).to(torch.bfloat16)
    scores = torch.softmax(torch.randn((1, H, 3), device=dev()), -1)
    with torch.no_grad():
        eager = layer(x, scores)
        compiled = torch.compile(layer, backend="aot_eager", fullgraph=True)(x, scores)
    assert torch.equal(eager, compiled)
    # and the operators did what they say (oracle on the same inputs)
    with torch.no_grad():
        qkv = lin(x)[0].view(S, H, 128).permute(1, 0, 2)
    f = qkv.double().cpu().numpy()
    qn = f / np.sqrt((f * f).mean(-1, keepdims=True) + 1e-6) * w.double().cpu().numpy()
    qn = torch.tensor(qn).to(torch.bfloat16).double().numpy()
    ref = sum(scores[0, :, i, None, None].double().cpu().numpy() * O.dense_attention(qn, f[:, :n], f[:, :n])
              for i, n in enumerate((S, S // 2, S // 4)))
    assert rel_fro(eager.cpu().numpy(), ref.sum(0) * 0.5) < 1e-2
