"""vorta_amd: VORTA's routed sparse-attention denoising path on MI355X (gfx950) -- hand-written HIP kernels behind the
reference's `vorta.attention` / `vorta.patch` / `vorta.ulysses` surface (the `vorta` package is an import alias)."""


def set_attention_precision(precision: str, *, measurement_only: bool = False) -> None:
    """"native" (contractions in the dtype of q,k,v, as the reference), "auto8" / "i8pv" / "fp8pv" (8-bit contractions on the
    int8 / fp8 MFMA with 16-bit output: BASELINE.json configs[4]; every one holds 40 dB against native on every input family
    tested).  Also: VORTA_ATTENTION_PRECISION in the environment.  "fp8" (e4m3 scores too) is measurement-only
    (vorta_amd/routed.py)."""
    from .routed import set_attention_precision as _set
    _set(precision, measurement_only=measurement_only)
