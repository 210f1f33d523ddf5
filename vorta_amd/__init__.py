"""vorta_amd: VORTA's routed sparse-attention denoising path on MI355X (gfx950) -- hand-written HIP kernels behind the
reference's `vorta.attention` / `vorta.patch` / `vorta.ulysses` surface (the `vorta` package is an import alias)."""


def set_attention_precision(precision: str) -> None:
    """"native" (contractions in the dtype of q,k,v, as the reference) or "fp8" (e4m3 contractions on the fp8 MFMA,
    16-bit output; BASELINE.json configs[4]).  Also: VORTA_ATTENTION_PRECISION=fp8 in the environment."""
    from .routed import set_attention_precision as _set
    _set(precision)
