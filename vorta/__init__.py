"""Import alias: `import vorta.attention`, `vorta.patch.utils`, `vorta.ulysses`, `vorta.utils` resolve to the
MI355X-native implementation in `vorta_amd`, so code written against the reference's package name
(scripts/hunyuan/inference.py:27-39) imports unchanged.  The model / pipeline monkey-patch modules
(`vorta.patch.modeling_*`, `vorta.patch.pipeline_*`) are not re-stated yet (they need `diffusers`)."""
import importlib
import sys

import vorta_amd

for _name in ("attention", "attention.coreset_select", "attention.hunyuan", "attention.wan", "patch", "patch.router",
              "patch.utils", "ulysses", "utils"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"vorta_amd.{_name}")
attention = sys.modules[f"{__name__}.attention"]
patch = sys.modules[f"{__name__}.patch"]
ulysses = sys.modules[f"{__name__}.ulysses"]
utils = sys.modules[f"{__name__}.utils"]
