"""Small structural stand-ins for the diffusers classes the reference patches (diffusers is not installed here).

Written for these tests only: they keep the attribute names and the CALL STRUCTURE of diffusers 0.33.1 that the
reference's patches rely on (vorta/patch/modeling_hunyuan.py, modeling_wan.py, pipeline_*.py) -- who calls whom,
with which positional / keyword arguments, what `Attention.forward` hands to its processor -- with toy widths
(4 heads x 128) and simplified modulation.  They say nothing about diffusers' numerics.
"""
import inspect
import math
from types import SimpleNamespace

import torch
from torch import nn

H, D = 4, 128
INNER = H * D


# ----------------------------------------------------------------------------------------------------- attention
class MiniAttention(nn.Module):
    """`diffusers.models.attention_processor.Attention` protocol: projections + a pluggable processor, extra
    keywords filtered by the processor's signature."""

    def __init__(self, query_dim=INNER, cross_dim=None, added_kv=False, pre_only=False, norm="per_head", bias=True):
        super().__init__()
        self.heads = H
        kv_dim = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, INNER, bias=bias)
        self.to_k = nn.Linear(kv_dim, INNER, bias=bias)
        self.to_v = nn.Linear(kv_dim, INNER, bias=bias)
        n = D if norm == "per_head" else INNER
        self.norm_q = nn.RMSNorm(n, eps=1e-6)
        self.norm_k = nn.RMSNorm(n, eps=1e-6)
        self.add_q_proj = self.add_k_proj = self.add_v_proj = None
        self.norm_added_q = self.norm_added_k = None
        self.to_add_out = None
        if added_kv:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (nn.Linear(query_dim, INNER) for _ in range(3))
            self.norm_added_q, self.norm_added_k = nn.RMSNorm(D, eps=1e-6), nn.RMSNorm(D, eps=1e-6)
            self.to_add_out = nn.Linear(INNER, query_dim)
        self.to_out = None if pre_only else nn.ModuleList([nn.Linear(INNER, query_dim), nn.Dropout(0.0)])
        self.processor = None

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        accepted = set(inspect.signature(self.processor.__call__).parameters)
        kwargs = {k: v for k, v in kwargs.items() if k in accepted}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kwargs)


class _AdaNorm(nn.Module):
    """AdaLayerNormZero-like: `.linear` maps the conditioning to modulation terms (its in_features is what the
    reference reads as the router width, modeling_hunyuan.py:674,692)."""

    def __init__(self, dim, n_terms):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, n_terms * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.n_terms = n_terms

    def forward(self, x, emb):
        terms = self.linear(self.silu(emb)).chunk(self.n_terms, dim=1)
        x = self.norm(x) * (1 + terms[1][:, None]) + terms[0][:, None]
        return (x,) + tuple(terms[2:])


# ------------------------------------------------------------------------------------------------------- Hunyuan
class MiniHunyuanDualBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = _AdaNorm(dim, 3)
        self.norm1_context = _AdaNorm(dim, 3)
        self.attn = MiniAttention(dim, added_kv=True)
        self.ff = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.ff_context = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))

    def forward(self, hidden_states, encoder_hidden_states, temb, attention_mask=None, freqs_cis=None, *args, **kwargs):
        nh, gate = self.norm1(hidden_states, emb=temb)
        ne, c_gate = self.norm1_context(encoder_hidden_states, emb=temb)
        attn_output, context_attn_output = self.attn(hidden_states=nh, encoder_hidden_states=ne,
                                                     attention_mask=attention_mask, image_rotary_emb=freqs_cis)
        hidden_states = hidden_states + attn_output * gate.unsqueeze(1)
        encoder_hidden_states = encoder_hidden_states + context_attn_output * c_gate.unsqueeze(1)
        hidden_states = hidden_states + 0.1 * self.ff(hidden_states)
        encoder_hidden_states = encoder_hidden_states + 0.1 * self.ff_context(encoder_hidden_states)
        return hidden_states, encoder_hidden_states


class MiniHunyuanSingleBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = _AdaNorm(dim, 3)
        self.attn = MiniAttention(dim, pre_only=True)
        self.proj_mlp = nn.Linear(dim, dim)
        self.act_mlp = nn.GELU()
        self.proj_out = nn.Linear(INNER + dim, dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, attention_mask=None, image_rotary_emb=None, *args,
                **kwargs):
        text_seq_length = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        residual = hidden_states
        norm_hidden_states, gate = self.norm(hidden_states, emb=temb)
        mlp_hidden_states = self.act_mlp(self.proj_mlp(norm_hidden_states))
        nh, ne = norm_hidden_states[:, :-text_seq_length], norm_hidden_states[:, -text_seq_length:]
        attn_output, context_attn_output = self.attn(hidden_states=nh, encoder_hidden_states=ne,
                                                     attention_mask=attention_mask, image_rotary_emb=image_rotary_emb)
        attn_output = torch.cat([attn_output, context_attn_output], dim=1)
        hidden_states = gate.unsqueeze(1) * self.proj_out(torch.cat([attn_output, mlp_hidden_states], dim=2)) + residual
        return hidden_states[:, :-text_seq_length], hidden_states[:, -text_seq_length:]


class _TimestepEmbedder(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear_1, self.act, self.linear_2 = nn.Linear(32, dim), nn.SiLU(), nn.Linear(dim, dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


def _sinusoid(t, n=32):
    f = torch.exp(-math.log(10000.0) * torch.arange(n // 2, device=t.device, dtype=torch.float32) / (n // 2))
    a = t.float()[:, None] * f[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)


class MiniHunyuanConditionEmbedding(nn.Module):
    """timestep + pooled text (+ guidance): returns `(conditioning, token_replace_emb)` like 0.33.1."""

    def __init__(self, dim, token_replace=False):
        super().__init__()
        self.time_proj = _sinusoid
        self.timestep_embedder = _TimestepEmbedder(dim)
        self.text_embedder = nn.Linear(16, dim)
        self.guidance_embedder = _TimestepEmbedder(dim)
        self.image_condition_type = "token_replace" if token_replace else None

    def forward(self, timestep, pooled_projection, guidance=None):
        timesteps_emb = self.timestep_embedder(self.time_proj(timestep).to(dtype=pooled_projection.dtype))
        pooled = self.text_embedder(pooled_projection)
        conditioning = timesteps_emb + pooled
        token_replace_emb = None
        if self.image_condition_type == "token_replace":
            zero = self.time_proj(torch.zeros_like(timestep)).to(dtype=pooled_projection.dtype)
            token_replace_emb = self.timestep_embedder(zero) + pooled
        if guidance is not None:
            conditioning = conditioning + self.guidance_embedder(self.time_proj(guidance).to(dtype=pooled_projection.dtype))
        return conditioning, token_replace_emb


class MiniHunyuanRope(nn.Module):
    """reads only shape and device of its input; returns (cos, sin) of shape (S, D)."""

    def __init__(self):
        super().__init__()
        self.rope_dim = (16, 56, 56)

    def forward(self, hidden_states):
        _, _, f, h, w = hidden_states.shape
        grids = torch.meshgrid(*[torch.arange(n, device=hidden_states.device, dtype=torch.float32) for n in (f, h, w)],
                               indexing="ij")
        cos, sin = [], []
        for g, d in zip(grids, self.rope_dim):
            freq = 1.0 / (256.0 ** (torch.arange(0, d, 2, device=g.device, dtype=torch.float32) / d))
            ang = g.reshape(-1)[:, None] * freq[None]
            cos.append(ang.cos().repeat_interleave(2, dim=1))
            sin.append(ang.sin().repeat_interleave(2, dim=1))
        return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


class MiniHunyuanTransformer(nn.Module):
    """`HunyuanVideoTransformer3DModel.forward(hidden_states, timestep, encoder_hidden_states,
    encoder_attention_mask, pooled_projections, guidance=None, attention_kwargs=None, return_dict=True)`."""

    def __init__(self, dim=INNER, n_dual=2, n_single=2, in_channels=4, token_replace=False):
        super().__init__()
        self.config = SimpleNamespace(patch_size=1, patch_size_t=1, in_channels=in_channels)
        self.rope = MiniHunyuanRope()
        self.time_text_embed = MiniHunyuanConditionEmbedding(dim, token_replace)
        self.x_embedder = nn.Linear(in_channels, dim)
        self.context_embedder = nn.Linear(24, dim)
        self.transformer_blocks = nn.ModuleList([MiniHunyuanDualBlock(dim) for _ in range(n_dual)])
        self.single_transformer_blocks = nn.ModuleList([MiniHunyuanSingleBlock(dim) for _ in range(n_single)])
        self.norm_out = nn.LayerNorm(dim, elementwise_affine=False)
        self.proj_out = nn.Linear(dim, in_channels)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections,
                guidance=None, attention_kwargs=None, return_dict=True):
        b, c, f, h, w = hidden_states.shape
        image_rotary_emb = self.rope(hidden_states)
        temb, token_replace_emb = self.time_text_embed(timestep, pooled_projections, guidance)
        hidden_states = self.x_embedder(hidden_states.flatten(2).transpose(1, 2))
        encoder_hidden_states = self.context_embedder(encoder_hidden_states)
        latent_len, cond_len = hidden_states.shape[1], encoder_hidden_states.shape[1]
        attention_mask = torch.zeros(b, latent_len + cond_len, device=hidden_states.device, dtype=torch.bool)
        eff = latent_len + encoder_attention_mask.sum(dim=1, dtype=torch.int)
        for i in range(b):
            attention_mask[i, : eff[i]] = True
        attention_mask = attention_mask.unsqueeze(1).unsqueeze(1)
        for block in self.transformer_blocks:
            hidden_states, encoder_hidden_states = block(hidden_states, encoder_hidden_states, temb, attention_mask,
                                                         image_rotary_emb, token_replace_emb, h * w)
        for block in self.single_transformer_blocks:
            hidden_states, encoder_hidden_states = block(hidden_states, encoder_hidden_states, temb, attention_mask,
                                                         image_rotary_emb, token_replace_emb, h * w)
        hidden_states = self.proj_out(self.norm_out(hidden_states))
        hidden_states = hidden_states.transpose(1, 2).reshape(b, c, f, h, w)
        if not return_dict:
            return (hidden_states,)
        return SimpleNamespace(sample=hidden_states)


# ----------------------------------------------------------------------------------------------------------- Wan
class MiniWanRope(nn.Module):
    def forward(self, hidden_states):
        _, _, f, h, w = hidden_states.shape
        parts = []
        for n, d, shape in ((f, 44, (f, 1, 1)), (h, 42, (1, h, 1)), (w, 42, (1, 1, w))):
            freq = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64, device=hidden_states.device) / d))
            ang = torch.arange(n, dtype=torch.float64, device=hidden_states.device)[:, None] * freq[None]
            parts.append(torch.polar(torch.ones_like(ang), ang).view(*shape, -1).expand(f, h, w, -1))
        return torch.cat(parts, dim=-1).reshape(1, 1, f * h * w, -1)  # complex128 (1,1,S,D/2)


class MiniWanConditionEmbedder(nn.Module):
    def __init__(self, dim, text_dim=24):
        super().__init__()
        self.timesteps_proj = _sinusoid
        self.time_embedder = _TimestepEmbedder(dim)
        self.act_fn = nn.SiLU()
        self.time_proj = nn.Linear(dim, 6 * dim)
        self.text_embedder = nn.Linear(text_dim, dim)

    def forward(self, timestep, encoder_hidden_states, encoder_hidden_states_image=None):
        temb = self.time_embedder(self.timesteps_proj(timestep).to(encoder_hidden_states.dtype)).type_as(encoder_hidden_states)
        timestep_proj = self.time_proj(self.act_fn(temb))
        return temb, timestep_proj, self.text_embedder(encoder_hidden_states), encoder_hidden_states_image


class MiniWanBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.attn1 = MiniAttention(dim, norm="across_heads")
        self.attn2 = MiniAttention(dim, cross_dim=dim, norm="across_heads")
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ffn = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, hidden_states, encoder_hidden_states, temb, rotary_emb):
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (self.scale_shift_table + temb.float()).chunk(6, dim=1)
        nh = (self.norm1(hidden_states.float()) * (1 + scale_msa) + shift_msa).type_as(hidden_states)
        attn_output = self.attn1(hidden_states=nh, rotary_emb=rotary_emb)
        hidden_states = (hidden_states.float() + attn_output * gate_msa).type_as(hidden_states)
        nh = self.norm2(hidden_states.float()).type_as(hidden_states)
        hidden_states = hidden_states + self.attn2(hidden_states=nh, encoder_hidden_states=encoder_hidden_states)
        nh = (self.norm1(hidden_states.float()) * (1 + c_scale) + c_shift).type_as(hidden_states)
        return (hidden_states.float() + 0.1 * self.ffn(nh).float() * c_gate).type_as(hidden_states)


class MiniWanTransformer(nn.Module):
    """`WanTransformer3DModel.forward(hidden_states, timestep, encoder_hidden_states,
    encoder_hidden_states_image=None, return_dict=True, attention_kwargs=None)`."""

    def __init__(self, dim=INNER, n_blocks=3, in_channels=4):
        super().__init__()
        self.config = SimpleNamespace(patch_size=(1, 1, 1), in_channels=in_channels)
        self.rope = MiniWanRope()
        self.patch_embedding = nn.Conv3d(in_channels, dim, kernel_size=1)
        self.condition_embedder = MiniWanConditionEmbedder(dim)
        self.blocks = nn.ModuleList([MiniWanBlock(dim) for _ in range(n_blocks)])
        self.norm_out = nn.LayerNorm(dim, elementwise_affine=False)
        self.proj_out = nn.Linear(dim, in_channels)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
                return_dict=True, attention_kwargs=None):
        b, c, f, h, w = hidden_states.shape
        rotary_emb = self.rope(hidden_states)
        hidden_states = self.patch_embedding(hidden_states).flatten(2).transpose(1, 2)
        temb, timestep_proj, encoder_hidden_states, _ = self.condition_embedder(timestep, encoder_hidden_states, None)
        timestep_proj = timestep_proj.unflatten(1, (6, -1))
        for block in self.blocks:
            hidden_states = block(hidden_states, encoder_hidden_states, timestep_proj, rotary_emb)
        hidden_states = self.proj_out(self.norm_out(hidden_states)).transpose(1, 2).reshape(b, c, f, h, w)
        if not return_dict:
            return (hidden_states,)
        return SimpleNamespace(sample=hidden_states)


# ----------------------------------------------------------------------------------------------------- pipelines
class _MiniVae(nn.Module):
    def __init__(self, wan=False):
        super().__init__()
        self.config = SimpleNamespace(scaling_factor=0.5, z_dim=4, latents_mean=[0.1, 0.2, 0.3, 0.4],
                                      latents_std=[1.0, 2.0, 0.5, 1.5])
        self.scale = nn.Parameter(torch.tensor(2.0))

    @property
    def dtype(self):
        return self.scale.dtype

    def decode(self, latents, return_dict=True):
        return (latents * self.scale,)


class _MiniVideoProcessor:
    @staticmethod
    def postprocess_video(video, output_type="np"):
        return video.float().cpu().numpy() if output_type == "np" else video


class _MiniPipelineBase:
    """The parts of `DiffusionPipeline` the patched calls touch."""

    def __init__(self, transformer, device):
        self.transformer = transformer
        self.vae = _MiniVae().to(device)
        self.video_processor = _MiniVideoProcessor()
        self.device = self._execution_device = torch.device(device)
        self._current_timestep = None
        self.freed = 0

    def maybe_free_model_hooks(self):
        self.freed += 1

    def prepare_latents(self, batch, channels, height, width, num_frames, dtype, device, generator, latents=None):
        if latents is not None:
            return latents.to(device=device, dtype=dtype)
        shape = (batch, channels, num_frames, height, width)  # toy: one latent per pixel/frame
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class MiniHunyuanPipeline(_MiniPipelineBase):
    """Stock call skeleton of `HunyuanVideoPipeline.__call__` (encode, prepare latents, loop: transformer(...,
    return_dict=False)[0], Euler step, decode)."""

    @torch.no_grad()
    def __call__(self, prompt=None, height=6, width=8, num_frames=4, num_inference_steps=2, guidance_scale=6.0,
                 generator=None, latents=None, prompt_embeds=None, pooled_prompt_embeds=None,
                 prompt_attention_mask=None, output_type="np", return_dict=True, attention_kwargs=None):
        device, tdtype = self._execution_device, self.transformer.dtype
        prompt_embeds = prompt_embeds.to(tdtype)
        prompt_attention_mask = prompt_attention_mask.to(tdtype)
        pooled_prompt_embeds = pooled_prompt_embeds.to(tdtype)
        sigmas = torch.linspace(1.0, 0.0, num_inference_steps + 1, device=device)
        latents = self.prepare_latents(1, self.transformer.config.in_channels, height, width, num_frames, torch.float32,
                                       device, generator, latents)
        guidance = torch.tensor([guidance_scale] * latents.shape[0], dtype=tdtype, device=device) * 1000.0
        for i in range(num_inference_steps):
            t = sigmas[i] * 1000.0
            self._current_timestep = t
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            noise_pred = self.transformer(hidden_states=latents.to(tdtype), timestep=timestep,
                                          encoder_hidden_states=prompt_embeds,
                                          encoder_attention_mask=prompt_attention_mask,
                                          pooled_projections=pooled_prompt_embeds, guidance=guidance,
                                          attention_kwargs=attention_kwargs, return_dict=False)[0]
            latents = latents + (sigmas[i + 1] - sigmas[i]) * noise_pred.float()
        self._current_timestep = None
        if output_type != "latent":
            video = self.vae.decode(latents.to(self.vae.dtype) / self.vae.config.scaling_factor, return_dict=False)[0]
            video = self.video_processor.postprocess_video(video, output_type=output_type)
        else:
            video = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return SimpleNamespace(frames=video)


class MiniWanPipeline(_MiniPipelineBase):
    """Stock call skeleton of `WanPipeline.__call__`: two batch-1 forwards per step under guidance."""

    @torch.no_grad()
    def __call__(self, prompt=None, height=6, width=8, num_frames=4, num_inference_steps=2, guidance_scale=5.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="np",
                 return_dict=True, attention_kwargs=None):
        device, tdtype = self._execution_device, self.transformer.dtype
        prompt_embeds = prompt_embeds.to(tdtype)
        cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None
        sigmas = torch.linspace(1.0, 0.0, num_inference_steps + 1, device=device)
        latents = self.prepare_latents(1, self.transformer.config.in_channels, height, width, num_frames, torch.float32,
                                       device, generator, latents)
        for i in range(num_inference_steps):
            t = sigmas[i] * 1000.0
            self._current_timestep = t
            timestep = t.expand(latents.shape[0])
            noise_pred = self.transformer(hidden_states=latents.to(tdtype), timestep=timestep,
                                          encoder_hidden_states=prompt_embeds, attention_kwargs=attention_kwargs,
                                          return_dict=False)[0]
            if cfg:
                noise_uncond = self.transformer(hidden_states=latents.to(tdtype), timestep=timestep,
                                                encoder_hidden_states=negative_prompt_embeds.to(tdtype),
                                                attention_kwargs=attention_kwargs, return_dict=False)[0]
                noise_pred = noise_uncond + guidance_scale * (noise_pred - noise_uncond)
            latents = latents + (sigmas[i + 1] - sigmas[i]) * noise_pred.float()
        self._current_timestep = None
        if output_type != "latent":
            c = self.vae.config
            lat = latents.to(self.vae.dtype)
            mean = torch.tensor(c.latents_mean).view(1, c.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
            inv_std = 1.0 / torch.tensor(c.latents_std).view(1, c.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
            video = self.vae.decode(lat / inv_std + mean, return_dict=False)[0]
            video = self.video_processor.postprocess_video(video, output_type=output_type)
        else:
            video = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return SimpleNamespace(frames=video)
