"""`afldm` import-path alias: `afldm.X` resolves to the SAME module object as `afldm_amd.X`, so
code written against the reference package (scripts/shift_ldm_ffhq.py: `from
afldm.af_modules.af_api import make_af_unet`, ...) runs on the MI355X implementation unchanged.
No code lives here."""
import importlib
import importlib.abc
import importlib.util
import sys

import afldm_amd  # noqa: F401

_PREFIX = "afldm."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_PREFIX):
            try:
                if importlib.util.find_spec("afldm_amd." + fullname[len(_PREFIX):]) is None:
                    return None
            except ModuleNotFoundError:
                return None
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("afldm_amd." + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
