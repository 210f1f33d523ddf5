"""Router gate (vorta/patch/router.py:17-43): softmax(Linear(SiLU(temb)).view(B, H, 3)).

Same module layout as the reference (`linear.weight (3H,E)`, `linear.bias (3H)`), so the published `router.pt`
files load by key (`...blocks.{i}.router.linear.{weight,bias}`, vorta/train/checkpoint.py:63-73).  The forward
runs in libvorta_hip.so (vorta_router_route) -- inference only, no autograd.
"""
import torch
from torch import nn

from .. import ops


class Router(nn.Module):
    def __init__(self, embedding_dim: int, heads: int, num_experts: int = 3):
        super().__init__()
        self.heads = heads
        self.num_experts = num_experts
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, heads * num_experts, bias=True)
        self.softmax = nn.Softmax(dim=-1)

    @torch.no_grad()
    def forward(self, temb: torch.Tensor) -> torch.Tensor:
        """temb (B, E) timestep embedding -> routing scores (B, H, num_experts) in the module's dtype."""
        w, b = self.linear.weight, self.linear.bias
        if w.dtype not in (torch.bfloat16, torch.float16):
            # the reference runs the router in bf16 (router_dtype, scripts/hunyuan/inference.py:117)
            w, b = w.to(torch.bfloat16), b.to(torch.bfloat16)
        scores, _, _, _ = ops.router_route(temb.to(w.dtype), w, b, self.heads, 0.0, self.num_experts)
        return scores


def load_router_checkpoint(ckpt_file, transformer: nn.Module):
    """Merge a router-only checkpoint into the model (vorta/train/checkpoint.py:63-73)."""
    state = torch.load(ckpt_file, weights_only=True)
    full = transformer.state_dict()
    full.update({k: v for k, v in state.items() if k in full})
    transformer.load_state_dict(full)
    return transformer
