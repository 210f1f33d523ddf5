"""ONE gate for the A/B switches that lost their measurements: `VORTA_DEBUG="key=value,key=value"`.

The defaults are the product; every key below selects a path that exists only so that a same-box A/B (tools/measure/,
profiles/) can be re-run, and none is a tuning knob:
    coreset_kv_order=group   key-side coreset lists in group-major order (measured neutral, profiles/r03_coreset_kv_order.txt)
    fused_text_splits=0      the sliding expert's text-query segment last and key-split (round 1's order)
    sta_merge=0              one query group per sliding tile instead of merged groups of equal key lists
    fp8_center_k=0           no key centring in the e4m3 conversion
    joint_projection=0       separate q / k / v projections + torch.cat (the reference's route) in the Hunyuan processors
    attn_variant=1           the first, register-staged 16-bit kernel
    no_xcd_remap=1           workgroups in launch order (no XCD-aware live order)
    sp_staging=torch         index ops instead of vorta_permute_heads in the sequence-parallel staging passes
    sp_group_streams=0       slot groups attend on the current stream instead of alternating streams
    sp_v_wire=0              e4m3 attention under sequence parallelism exchanges v in 16 bits
    sp_emulate_link_gbps=x   one-GPU emulation of a rank: every exchange holds a side stream for bytes / x (an ASSUMED wire)
The product switches (README "Environment switches") are not here: VORTA_ATTENTION_PRECISION, VORTA_SP_PLACEMENT, VORTA_SP_GROUPS,
VORTA_SP_KV_SPLITS, VORTA_SP_BUFFER_SETS, VORTA_I8_TAIL_MIN_RMS, VORTA_HIP_LIB."""
import os

KEYS = ("coreset_kv_order", "fused_text_splits", "sta_merge", "fp8_center_k", "joint_projection", "attn_variant", "no_xcd_remap",
        "sp_staging", "sp_group_streams", "sp_v_wire", "sp_emulate_link_gbps")


def _parse(text: str) -> dict:
    out = {}
    for item in filter(None, (x.strip() for x in text.split(","))):
        key, _, value = item.partition("=")
        if key not in KEYS:
            raise ValueError(f"VORTA_DEBUG: unknown key {key!r} (one of {KEYS})")
        out[key] = value if value else "1"
    return out


FLAGS = _parse(os.environ.get("VORTA_DEBUG", ""))


def flag(key: str, default: str) -> str:
    if key not in KEYS:
        raise KeyError(key)
    return FLAGS.get(key, default)
