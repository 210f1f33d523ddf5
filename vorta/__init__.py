"""Import alias: every `vorta.X` module IS the `vorta_amd.X` module (same object), so code written against the
reference's package name -- `vorta.attention`, `vorta.patch.modeling_hunyuan`, `vorta.patch.pipeline_wan`,
`vorta.patch.utils`, `vorta.ulysses`, `vorta.utils` (scripts/hunyuan/inference.py:27-44, scripts/wan/inference.py:32-49)
-- imports unchanged and shares all state (SP_STATE, pipeline registry) with the native package."""
import importlib
import importlib.abc
import importlib.util
import sys

import vorta_amd


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)

    def exec_module(self, module):  # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(__name__ + "."):
            return None
        real = "vorta_amd" + fullname[len(__name__):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

attention = importlib.import_module(__name__ + ".attention")
patch = importlib.import_module(__name__ + ".patch")
ulysses = importlib.import_module(__name__ + ".ulysses")
utils = importlib.import_module(__name__ + ".utils")
