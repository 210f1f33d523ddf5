"""Shared helpers of the GPU parity tests (tolerances, rounding, zero padding of the golden vectors)."""
import numpy as np
import torch

# Stated tolerances: bf16/fp16 I/O, fp32 accumulation.
#   vs the oracle evaluated on the SAME rounded inputs: P/O rounding plus the kernel's one extra rounding of
#   Q * (scale*log2 e) to the I/O type (the scores leave the MFMA in the exp2 domain); that extra term only shows
#   with keys of large norm (test_online_softmax_rescale_branch uses |k| = 4|q|) and stays inside BASELINE.md's bound
ATOL_SAME = {torch.bfloat16: 2e-2, torch.float16: 2.5e-3}
RELF_SAME = {torch.bfloat16: 6e-3, torch.float16: 1.2e-3}
#   vs golden vectors computed by the reference in fp32 from unrounded inputs
ATOL_GOLD = 2e-2
RELF_GOLD = 1e-2


def dev():
    return torch.device("cuda:0")


def rel_fro(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def to_dev(x, dtype):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dtype).to(dev())


def rounded(x, dtype):
    """fp64 numpy copy of x after rounding to `dtype` (what the kernel actually sees)."""
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dtype).to(torch.float64).numpy()


def pad128(x):
    """zero-pad the last dim to 128: q.k and cosines are unchanged, extra output columns are zero."""
    x = np.asarray(x)
    out = np.zeros(x.shape[:-1] + (128,), dtype=x.dtype)
    out[..., : x.shape[-1]] = x
    return out


def check(out, ref, dtype, gold=False):
    out = out.float().cpu().numpy()
    atol = ATOL_GOLD if gold else ATOL_SAME[dtype]
    relf = RELF_GOLD if gold else RELF_SAME[dtype]
    err = np.abs(out - ref).max()
    rf = rel_fro(out, ref)
    assert err <= atol and rf <= relf, f"max|d|={err:.3e} (tol {atol}), relF={rf:.3e} (tol {relf})"


