"""HunyuanVideo (MMDiT: video + text tokens in one sequence) attention processors.

Same classes, call protocol and keyword names as vorta/attention/hunyuan.py, so diffusers' `Attention.forward`
and the patched block forwards (modeling_hunyuan.py:492-499,556-563) can call them unchanged.  Projections, qk
RMSNorm and RoPE stay in torch (rocBLAS); everything between post-RoPE q,k,v and the output projection runs in
libvorta_hip.so (vorta_amd/routed.py).
"""
from typing import List, Optional, Tuple

import torch

from .. import ops
from .. import torch_ops as _torch_ops  # registers torch.ops.vorta.*: what the processors launch through
from ..routed import HeadRouting
from ..ulysses import SP_STATE, shrink_dim
from .coreset_select import LowresGroupInfo
from .sliding_tile import SlidingTileDescriptor


def apply_rotary_emb(x: torch.Tensor, freqs_cis: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Rotary embedding on (B,H,S,D) with (cos, sin) of shape (S,D), interleaved real/imag pairs.
    [ext] restates diffusers==0.33.1 `apply_rotary_emb(use_real=True, use_real_unbind_dim=-1)`, which the
    reference imports (hunyuan.py:18,97-98); it sits before the attention boundary."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


def _fusable_norm(norm, x: torch.Tensor, numel: int) -> bool:
    """can vorta_qk_norm_rope stand in for this module?  (RMSNorm with an `eps` and an optional `weight` of the
    expected size, half-precision CUDA input)"""
    if norm is None or not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or x.shape[-1] != 128:
        return False
    if "RMSNorm" not in type(norm).__name__ or getattr(norm, "eps", None) is None:
        return False
    w = getattr(norm, "weight", None)
    return w is None or w.numel() == numel


def fused_norm_rope(x: torch.Tensor, norm, rope: Optional[Tuple[torch.Tensor, torch.Tensor]], rope_tokens: int = 0):
    """RMSNorm (+ RoPE on the first `rope_tokens` tokens) of a (1,H,N,D) projection view, in place, in one HIP
    kernel (hunyuan.py:62-104 are two module calls and ~10 elementwise passes)."""
    w = getattr(norm, "weight", None)
    cos, sin = rope if rope is not None else (None, None)
    torch.ops.vorta.qk_norm_rope(x[0], w, float(norm.eps), cos, sin, rope_tokens if rope is not None else 0)
    return x


# project video and text tokens straight into ONE (1, S+T, H*D) buffer per tensor: the `torch.cat([q, eq], dim=2)` of a
# dual-stream block (hunyuan.py:106-134: a read and a write of all of q, k and v) and the input concat of a
# single-stream block (hunyuan.py:47-48) disappear (SURVEY.md §8f N1 "+ text concat").  VORTA_DEBUG=joint_projection=0: the
# reference's route (A/B, tests).
JOINT_PROJECTION = __import__("vorta_amd._debug", fromlist=["flag"]).flag("joint_projection", "1") != "0"


def _linear_into(lin: torch.nn.Linear, x2d: torch.Tensor, dst: torch.Tensor) -> None:
    """dst (rows, out_features; contiguous rows of a larger buffer) = lin(x2d), written by the GEMM itself"""
    if lin.bias is not None:
        torch.addmm(lin.bias, x2d, lin.weight.t(), out=dst)
    else:
        torch.mm(x2d, lin.weight.t(), out=dst)


def _joint_projectable(x: torch.Tensor, e: torch.Tensor, lins) -> bool:
    if not JOINT_PROJECTION or x.shape[0] != 1 or not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    if e.dtype != x.dtype or not all(type(m) is torch.nn.Linear and m.weight.dtype == x.dtype for m in lins):
        return False
    # writing through lin.weight bypasses Linear.forward: only when nothing hangs on that forward (accelerate's offload
    # hooks keep the weights on meta / CPU until it runs; user hooks would be skipped)
    if any(m.weight.device != x.device or hasattr(m, "_hf_hook") or m._forward_hooks or m._forward_pre_hooks for m in lins):
        return False
    # `out=` is an inference-only form: autograd refuses it as soon as an argument requires grad
    return not (torch.is_grad_enabled() and (x.requires_grad or e.requires_grad or
                                             any(m.weight.requires_grad for m in lins)))


def _valid_keys(attention_mask: torch.Tensor) -> torch.Tensor:
    """Number of valid keys L (int32, shape (1,), on the device) from the reference's attention mask
    (hunyuan.py:169 `attention_mask.squeeze().sum()` on the diffusers 0.33 [B,1,1,N] key mask).  Other diffusers versions
    hand a [B,N,N] / [B,1,N,N] mask: one query row is reduced then, anything else is refused instead of silently
    attending padded text."""
    N = attention_mask.shape[-1]
    rows = attention_mask.numel() // (attention_mask.shape[0] * N)
    if rows not in (1, N):
        raise ValueError(f"unsupported attention_mask shape {tuple(attention_mask.shape)}: expected [B,1,1,N] or [B,(1,)N,N]")
    row = attention_mask.reshape(attention_mask.shape[0], rows, N)[0, 0]
    return row.sum(dtype=torch.int32).reshape(1)


class HunyuanVideoFlashAttnProcessor:
    """Dense attention for every head: the --native_attention path (hunyuan.py:35-238)."""

    def __init__(self):
        ops._C.lib()  # fail loudly now if libvorta_hip.so is missing: there is no fallback path

    # -- steps 1-4 of hunyuan.py:42-134: everything before the attention boundary ----------------------
    def _project(self, attn, hidden_states, encoder_hidden_states, image_rotary_emb):
        single_stream = attn.add_q_proj is None  # single blocks carry text inside hidden_states
        T = encoder_hidden_states.shape[1]
        rope = None
        if image_rotary_emb is not None:
            rope = (shrink_dim(image_rotary_emb[0], dim=0), shrink_dim(image_rotary_emb[1], dim=0))
        joint = self._project_joint(attn, hidden_states, encoder_hidden_states, rope, single_stream)
        if joint is not None:
            return (*joint, T)
        if single_stream:
            hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        q = attn.to_q(hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
        k = attn.to_k(hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
        v = attn.to_v(hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
        n_video = q.shape[2] - (T if single_stream else 0)
        if q.shape[0] == 1 and _fusable_norm(attn.norm_q, q, 128) and _fusable_norm(attn.norm_k, k, 128):
            # one in-place HIP pass per tensor: norm everywhere, rotation on the video tokens only
            fused_norm_rope(q, attn.norm_q, rope, n_video)
            fused_norm_rope(k, attn.norm_k, rope, n_video)
        else:
            if attn.norm_q is not None:
                q = attn.norm_q(q)
            if attn.norm_k is not None:
                k = attn.norm_k(k)
            if rope is not None:
                if single_stream:
                    q = torch.cat([apply_rotary_emb(q[:, :, :-T], rope), q[:, :, -T:]], dim=2)
                    k = torch.cat([apply_rotary_emb(k[:, :, :-T], rope), k[:, :, -T:]], dim=2)
                else:
                    q, k = apply_rotary_emb(q, rope), apply_rotary_emb(k, rope)
        if not single_stream:
            eq = attn.add_q_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            ek = attn.add_k_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            ev = attn.add_v_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            if eq.shape[0] == 1 and _fusable_norm(attn.norm_added_q, eq, 128) and _fusable_norm(attn.norm_added_k, ek, 128):
                fused_norm_rope(eq, attn.norm_added_q, None)
                fused_norm_rope(ek, attn.norm_added_k, None)
            else:
                if attn.norm_added_q is not None:
                    eq = attn.norm_added_q(eq)
                if attn.norm_added_k is not None:
                    ek = attn.norm_added_k(ek)
            q, k, v = torch.cat([q, eq], dim=2), torch.cat([k, ek], dim=2), torch.cat([v, ev], dim=2)
        return q, k, v, T

    @staticmethod
    def _project_joint(attn, hidden_states, encoder_hidden_states, rope, single_stream: bool):
        """Fused-norm case: the video and text projections land in ONE (1, S+T, H*D) buffer per tensor, written by the
        GEMMs -- no `torch.cat([q, eq], dim=2)` pass over q, k, v in a dual-stream block (hunyuan.py:106-134), no concat
        of the block inputs in a single-stream one (hunyuan.py:47-48); qk-norm + RoPE run in place on the row ranges.
        None when the modules do not allow it (the caller then takes the reference's route)."""
        vid = (attn.to_q, attn.to_k, attn.to_v)
        txt = vid if single_stream else (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj)
        if not _joint_projectable(hidden_states, encoder_hidden_states, vid + txt):
            return None
        S_, T, H = hidden_states.shape[1], encoder_hidden_states.shape[1], attn.heads
        out_f = attn.to_q.out_features
        probe = hidden_states.new_empty((1, H, 1, out_f // H))
        norms = (attn.norm_q, attn.norm_k) if single_stream else (attn.norm_q, attn.norm_k, attn.norm_added_q,
                                                                   attn.norm_added_k)
        if out_f // H != 128 or not all(_fusable_norm(n, probe, 128) for n in norms):
            return None
        if any(m.out_features != out_f for m in vid + txt):
            return None
        x2d, e2d = hidden_states[0], encoder_hidden_states[0]
        qkv = []
        for lin, elin in zip(vid, txt):
            buf = hidden_states.new_empty((1, S_ + T, out_f))
            _linear_into(lin, x2d, buf[0, :S_])
            _linear_into(elin, e2d, buf[0, S_:])
            qkv.append(buf.unflatten(2, (H, -1)).transpose(1, 2))  # (1, H, S+T, D) view
        q, k, v = qkv
        if single_stream:  # one norm for both row ranges, rotation on the video rows
            fused_norm_rope(q, attn.norm_q, rope, S_)
            fused_norm_rope(k, attn.norm_k, rope, S_)
        else:
            fused_norm_rope(q[:, :, :S_], attn.norm_q, rope, S_)
            fused_norm_rope(k[:, :, :S_], attn.norm_k, rope, S_)
            fused_norm_rope(q[:, :, S_:], attn.norm_added_q, None)
            fused_norm_rope(k[:, :, S_:], attn.norm_added_k, None)
        return q, k, v

    # -- step 6 (hunyuan.py:191-208) ---------------------------------------------------------------------
    @staticmethod
    def _output(attn, out_bshd: torch.Tensor, T: int):
        """out_bshd: (B, S+T, H, D) -- the kernels wrote it in this layout, so the head merge is a view."""
        hidden = out_bshd[:, :-T].flatten(2, 3)
        enc = out_bshd[:, -T:].flatten(2, 3)
        if getattr(attn, "to_out", None) is not None:
            hidden = attn.to_out[1](attn.to_out[0](hidden))
        if getattr(attn, "to_add_out", None) is not None:
            enc = attn.to_add_out(enc)
        return hidden, enc

    @staticmethod
    def _new_out(q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        B, H, N, D = q.shape
        buf = torch.empty((B, N, H, D), dtype=q.dtype, device=q.device)
        return buf, buf.permute(0, 2, 1, 3)  # (B,H,N,D) view for the kernels

    @staticmethod
    def _text_valid(attention_mask: torch.Tensor, T: int, descriptor=None) -> int:
        if descriptor is not None:
            return descriptor.text_seq_length_no_pad
        # the reference reads this on the host as well (hunyuan.py:169 `attention_mask.squeeze().sum()`).  The mask
        # spans [video | text]: the global video under the reference's patched forward (modeling_hunyuan.py:86-88),
        # the local shard under the stock diffusers forward -- either way its video part is all ones.
        return int(_valid_keys(attention_mask).item()) - (attention_mask.shape[-1] - T)

    def _dense(self, q, k, v, attention_mask, T):
        B = q.shape[0]
        assert B == 1, f"Batch size {B} is not supported for {self.__class__.__name__}."  # hunyuan.py:168
        if SP_STATE.enabled:
            from ._sp import sp_attention
            return sp_attention(q, k, v, T, None, None, model="hunyuan", attention_mask=attention_mask)
        buf, out = self._new_out(q)
        # L = attention_mask.sum() stays on the device: no host sync (the reference syncs at hunyuan.py:169)
        L = _valid_keys(attention_mask)
        N = q.shape[2]
        torch.ops.vorta.attn_fwd(q[0], k[0], v[0], out[0], N, N, q_valid=N, n_kv_dev=L, q_valid_dev=L)
        return buf

    @torch.no_grad()  # forward only: the HIP ops have no backward (training is out of scope)
    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb):
        q, k, v, T = self._project(attn, hidden_states, encoder_hidden_states, image_rotary_emb)
        return self._output(attn, self._dense(q, k, v, attention_mask, T), T)


class HunyuanVideoFlashAttnProcessorTripleEval(HunyuanVideoFlashAttnProcessor):
    """Inference-time routed attention: hard top-1 expert per head (hunyuan.py:516-661)."""

    def __init__(self, check_input: bool = False):
        super().__init__()
        self.check_input = check_input

    def _check_input(self, hidden_states, lowres_group_info, latent_shape, window_size, tile_size):
        """hunyuan.py:247-272 (same conditions, same messages)."""
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
    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb,
                 routing_score: torch.Tensor, tau_sparse: float,
                 lowres_group_info: Optional[LowresGroupInfo] = None,
                 flex_attn_mask_func: Optional[SlidingTileDescriptor] = None,
                 window_size: Tuple[int, int, int] = (3, 3, 3), tile_size: Tuple[int, int, int] = (6, 8, 8),
                 latent_shape: Tuple[int, int, int] = (30, 48, 80),
                 head_routing: Optional[HeadRouting] = None, experts_host: Optional[List[int]] = None):
        """Keyword names as hunyuan.py:521-539.  `head_routing` / `experts_host` are this build's additions: the
        routes of this layer already dispatched by the step's route plan (vorta_amd/patch/_engine.py), on the device /
        on the host; without them the dispatch runs here from `routing_score`."""
        self._check_input(hidden_states, lowres_group_info, latent_shape, window_size, tile_size)
        q, k, v, T = self._project(attn, hidden_states, encoder_hidden_states, image_rotary_emb)
        assert q.shape[0] == 1, f"Batch size {q.shape[0]} is not supported for {self.__class__.__name__}."
        te = self._text_valid(attention_mask, T, flex_attn_mask_func)
        if SP_STATE.enabled:
            from ._sp import sp_attention
            buf = sp_attention(q, k, v, T, routing_score, tau_sparse, model="hunyuan", text_valid=te,
                               lowres_group_info=lowres_group_info, window_size=window_size, tile_size=tile_size,
                               latent_shape=latent_shape, experts_host=experts_host)
            return self._output(attn, buf, T)
        # top-1 / tau dispatch on the device: no torch.nonzero host sync (hunyuan.py:612-640)
        if head_routing is None:
            _, lists, counts = torch.ops.vorta.route_scores(routing_score, float(tau_sparse))
            head_routing = HeadRouting.from_device(lists, counts)
        buf, out = self._new_out(q)
        torch.ops.vorta.routed_attention(q, k, v, out, **_torch_ops.routing_args(head_routing),
                                         **_torch_ops.geometry_args(lowres_group_info, window_size, tile_size, latent_shape),
                                         model="hunyuan", text_len=T, text_valid=te)
        return self._output(attn, buf, T)


class HunyuanVideoFlashAttnProcessorTripleTrain(HunyuanVideoFlashAttnProcessorTripleEval):
    """Training-time soft mixture of the three experts (hunyuan.py:241-513), FORWARD only: every head runs all
    three experts and the outputs are summed with the routing scores.  There are no backward
    kernels: a call that would need gradients raises instead of silently returning a detached result.
    `use_original_attn=True` (the dense teacher, hunyuan.py:312-321) is the dense processor."""

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb,
                 use_original_attn: bool = False, routing_score: Optional[torch.Tensor] = None,
                 lowres_group_info: Optional[LowresGroupInfo] = None,
                 flex_attn_mask_func: Optional[SlidingTileDescriptor] = None,
                 window_size: Tuple[int, int, int] = (3, 3, 3), tile_size: Tuple[int, int, int] = (6, 8, 8),
                 latent_shape: Tuple[int, int, int] = (30, 48, 80)):
        if use_original_attn:
            return HunyuanVideoFlashAttnProcessor.__call__(self, attn, hidden_states, encoder_hidden_states,
                                                           attention_mask, image_rotary_emb)
        if torch.is_grad_enabled() and (hidden_states.requires_grad or routing_score.requires_grad):
            raise NotImplementedError("the soft-mixture forward of this build has no backward: call it under "
                                      "torch.no_grad() (router training is outside the inference hot path)")
        if SP_STATE.enabled:
            raise NotImplementedError("the soft-mixture forward is not sequence-parallel in this build")
        with torch.no_grad():
            self._check_input(hidden_states, lowres_group_info, latent_shape, window_size, tile_size)
            q, k, v, T = self._project(attn, hidden_states, encoder_hidden_states, image_rotary_emb)
            assert q.shape[0] == 1, f"Batch size {q.shape[0]} is not supported for {self.__class__.__name__}."
            te = self._text_valid(attention_mask, T, flex_attn_mask_func)
            buf, out = self._new_out(q)
            torch.ops.vorta.soft_mixture_attention(q, k, v, routing_score, out,
                                                   **_torch_ops.geometry_args(lowres_group_info, window_size, tile_size,
                                                                              latent_shape),
                                                   model="hunyuan", text_len=T, text_valid=te)
            return self._output(attn, buf, T)
