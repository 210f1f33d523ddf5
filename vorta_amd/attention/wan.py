"""Wan-2.1 attention processors (video-only self attention + separate cross attention).

Same classes, call protocol and keyword names as vorta/attention/wan.py.  Only self attention is routed
(modeling_wan.py:215-229); cross attention and `use_original_attn` go through the dense kernel.
"""
from typing import List, Optional, Tuple

import torch

from .. import ops
from .. import torch_ops as _torch_ops  # registers torch.ops.vorta.*: what the processors launch through
from ..routed import HeadRouting, dense_attention
from ..ulysses import SP_STATE, shrink_dim
from .coreset_select import LowresGroupInfo
from .sliding_tile import SlidingTileDescriptor


def apply_rotary_emb(hidden_states: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """complex rotation in float64, as the reference does it (wan.py:34-37); before the attention boundary."""
    x = torch.view_as_complex(hidden_states.to(torch.float64).unflatten(3, (-1, 2)))
    return torch.view_as_real(x * freqs).flatten(3, 4).type_as(hidden_states)


_ROPE_CACHE = {}


def _cos_sin(freqs: torch.Tensor):
    """complex frequencies (1,1,S,D/2) -> fp32 (cos, sin) of shape (S,D) with each value repeated for its pair.
    One entry, keyed on the frequency tensor OBJECT (held by the entry: a freed tensor's address is routinely reused
    for the next forward's table, e.g. 480x832 then 832x480 give equal shapes and different contents) and its version:
    the 30-40 blocks of one forward share the tensor (modeling_wan.py:242-262), the next forward builds a new one."""
    hit = _ROPE_CACHE.get("entry")
    if hit is not None and hit[0] is freqs and hit[1] == freqs._version:
        return hit[2]
    f = freqs.reshape(-1, freqs.shape[-1])
    cs = (f.real.float().repeat_interleave(2, dim=1).contiguous(), f.imag.float().repeat_interleave(2, dim=1).contiguous())
    _ROPE_CACHE["entry"] = (freqs, freqs._version, cs)
    return cs


class WanAttnProcessor2_0:
    """Dense attention (wan.py:40-160): self, text cross (Sq != Skv) and the optional I2V image branch."""

    def __init__(self):
        ops._C.lib()  # no fallback: fail now if the HIP library is missing

    def _input_proj(self, attn, hidden_states, encoder_hidden_states=None, rotary_emb=None):
        """wan.py:64-101: q/k/v projections, RMSNorm across heads (before the head split), RoPE."""
        enc_img = None
        if attn.add_k_proj is not None:
            enc_img = encoder_hidden_states[:, :257]
            encoder_hidden_states = encoder_hidden_states[:, 257:]
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        q = attn.to_q(hidden_states)
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        H = attn.heads
        rope = shrink_dim(rotary_emb, dim=2) if rotary_emb is not None else None
        from .hunyuan import _fusable_norm
        fuse = (q.shape[0] == 1 and q.shape[-1] == H * 128
                and (rope is None or (rope.shape[-2] == q.shape[1] == k.shape[1]))
                and _fusable_norm(attn.norm_q, q.view(1, q.shape[1], H, 128), H * 128)
                and _fusable_norm(attn.norm_k, k.view(1, k.shape[1], H, 128), H * 128))
        if fuse:
            # RMSNorm over all H*D channels of a token + rotation, one in-place HIP pass per tensor
            cos, sin = _cos_sin(rope) if rope is not None else (None, None)
            q, k, v = (x.unflatten(2, (H, -1)).transpose(1, 2) for x in (q, k, v))
            torch.ops.vorta.qk_norm_rope(q[0], attn.norm_q.weight, float(attn.norm_q.eps), cos, sin, -1, True)
            torch.ops.vorta.qk_norm_rope(k[0], attn.norm_k.weight, float(attn.norm_k.eps), cos, sin, -1, True)
        else:
            if attn.norm_q is not None:
                q = attn.norm_q(q)
            if attn.norm_k is not None:
                k = attn.norm_k(k)
            q, k, v = (x.unflatten(2, (H, -1)).transpose(1, 2) for x in (q, k, v))
            if rope is not None:
                q, k = apply_rotary_emb(q, rope), apply_rotary_emb(k, rope)
        return q, k, v, enc_img

    @staticmethod
    def _new_out(q: torch.Tensor):
        B, H, N, D = q.shape
        buf = torch.empty((B, N, H, D), dtype=q.dtype, device=q.device)
        return buf, buf.permute(0, 2, 1, 3)

    def _attn(self, attn, q, k, v, enc_img, is_cross_attn: bool):
        """wan.py:103-149 without SP (the SP branch lives in _sp.py). Returns (B,S,H,D) buffers."""
        buf_img = None
        if enc_img is not None:
            k_img = attn.norm_added_k(attn.add_k_proj(enc_img)).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            v_img = attn.add_v_proj(enc_img).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            buf_img, out_img = self._new_out(q)
            for b in range(q.shape[0]):
                dense_attention(q[b:b + 1], k_img[b:b + 1], v_img[b:b + 1], out=out_img[b:b + 1])
        buf, out = self._new_out(q)
        for b in range(q.shape[0]):
            dense_attention(q[b:b + 1], k[b:b + 1], v[b:b + 1], out=out[b:b + 1])
        return buf, buf_img

    @staticmethod
    def _output_proj(attn, buf, buf_img=None):
        hidden = buf.flatten(2, 3)
        if buf_img is not None:
            hidden = hidden + buf_img.flatten(2, 3)
        return attn.to_out[1](attn.to_out[0](hidden))

    @torch.no_grad()  # forward only: the HIP ops have no backward (training is out of scope)
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is always None on this path (wan.py:141)")
        is_cross = encoder_hidden_states is not None
        q, k, v, enc_img = self._input_proj(attn, hidden_states, encoder_hidden_states, rotary_emb)
        if SP_STATE.enabled:
            from ._sp import sp_wan_dense
            return self._output_proj(attn, *sp_wan_dense(self, attn, q, k, v, enc_img, is_cross))
        return self._output_proj(attn, *self._attn(attn, q, k, v, enc_img, is_cross))


class WanAttnProcessorTripleEval(WanAttnProcessor2_0):
    """Inference-time routed self attention (wan.py:303-437)."""

    def __init__(self, check_input: bool = False):
        super().__init__()
        self.check_input = check_input

    def _check_input(self, hidden_states, lowres_group_info, latent_shape, window_size, tile_size):
        """wan.py:168-193."""
        if not self.check_input:
            return
        seq_length = hidden_states.shape[1] * SP_STATE.sp_size
        num_groups = lowres_group_info.center_indices.shape[0]
        group_size = lowres_group_info.center_indices.shape[1] + lowres_group_info.margin_indices.shape[1]
        if seq_length != latent_shape[0] * latent_shape[1] * latent_shape[2]:
            raise ValueError(f"Input sequence length {seq_length} does not match latent shape {latent_shape}.")
        for t_size, l_size in zip(tile_size, latent_shape):
            if l_size % t_size != 0:
                raise ValueError(
                    f"Tile size {tile_size} (dim={t_size}) does not divide latent shape {latent_shape} (dim={l_size}).")
        if seq_length != num_groups * group_size:
            raise ValueError(f"Input sequence length {seq_length} does not match low-res info {num_groups}x{group_size}.")

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, rotary_emb,
                 tau_sparse: float, routing_score: torch.Tensor,
                 lowres_group_info: Optional[LowresGroupInfo] = None,
                 flex_attn_mask_func: Optional[SlidingTileDescriptor] = None,
                 window_size: Tuple[int, int, int] = (3, 3, 3), tile_size: Tuple[int, int, int] = (6, 8, 8),
                 latent_shape: Tuple[int, int, int] = (20, 30, 52), use_original_attn: bool = False,
                 head_routing: Optional[HeadRouting] = None, experts_host: Optional[List[int]] = None):
        """Keyword names as wan.py:308-328; `head_routing` / `experts_host`: see the Hunyuan processor."""
        if encoder_hidden_states is not None or use_original_attn:
            return WanAttnProcessor2_0.__call__(self, attn, hidden_states, encoder_hidden_states, attention_mask,
                                                rotary_emb)
        self._check_input(hidden_states, lowres_group_info, latent_shape, window_size, tile_size)
        q, k, v, _ = self._input_proj(attn, hidden_states, None, rotary_emb)
        B = q.shape[0]
        # the reference routes EVERY batch item by item 0's scores (wan.py:388-416: `routing_score[0].topk(1)`); the kernels
        # take one item per launch, so a batch is a loop over its items with the same routes (the scripts run B = 1:
        # pipeline_wan.py:322-344 does CFG as two forwards)
        if SP_STATE.enabled:
            from ._sp import sp_attention
            bufs = [sp_attention(q[b:b + 1], k[b:b + 1], v[b:b + 1], 0, routing_score, tau_sparse, model="wan",
                                 lowres_group_info=lowres_group_info, window_size=window_size, tile_size=tile_size,
                                 latent_shape=latent_shape, experts_host=experts_host) for b in range(B)]
            return self._output_proj(attn, bufs[0] if B == 1 else torch.cat(bufs, dim=0))
        if head_routing is None:
            _, lists, counts = torch.ops.vorta.route_scores(routing_score, float(tau_sparse))
            head_routing = HeadRouting.from_device(lists, counts)
        buf, out = self._new_out(q)
        for b in range(B):
            torch.ops.vorta.routed_attention(q[b:b + 1], k[b:b + 1], v[b:b + 1], out[b:b + 1],
                                             **_torch_ops.routing_args(head_routing),
                                             **_torch_ops.geometry_args(lowres_group_info, window_size, tile_size,
                                                                        latent_shape), model="wan")
        return self._output_proj(attn, buf)


class WanAttnProcessorTripleTrain(WanAttnProcessorTripleEval):
    """Soft-mixture training forward (wan.py:163-300), FORWARD only (no backward kernels); the
    dense teacher (`use_original_attn=True`) and cross attention are the dense processor."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, rotary_emb=None,
                 use_original_attn: bool = False, routing_score: Optional[torch.Tensor] = None,
                 lowres_group_info: Optional[LowresGroupInfo] = None,
                 flex_attn_mask_func: Optional[SlidingTileDescriptor] = None,
                 window_size: Tuple[int, int, int] = (3, 3, 3), tile_size: Tuple[int, int, int] = (6, 8, 8),
                 latent_shape: Tuple[int, int, int] = (20, 30, 52)):
        if encoder_hidden_states is not None or use_original_attn:
            return WanAttnProcessor2_0.__call__(self, attn, hidden_states, encoder_hidden_states, attention_mask,
                                                rotary_emb)
        if torch.is_grad_enabled() and (hidden_states.requires_grad or routing_score.requires_grad):
            raise NotImplementedError("the soft-mixture forward of this build has no backward: call it under "
                                      "torch.no_grad() (router training is outside the inference hot path)")
        if SP_STATE.enabled:
            raise NotImplementedError("the soft-mixture forward is not sequence-parallel in this build")
        with torch.no_grad():
            self._check_input(hidden_states, lowres_group_info, latent_shape, window_size, tile_size)
            q, k, v, _ = self._input_proj(attn, hidden_states, None, rotary_emb)
            assert q.shape[0] == 1, "the soft mixture runs one batch item per call"
            buf, out = self._new_out(q)
            torch.ops.vorta.soft_mixture_attention(q, k, v, routing_score, out,
                                                   **_torch_ops.geometry_args(lowres_group_info, window_size, tile_size,
                                                                              latent_shape), model="wan")
            return self._output_proj(attn, buf)
