"""`music_style_transfer` — the reference's package name, kept so that
`python -m music_style_transfer.VarAutoEncoder.main <flags>` (scripts/train-vae.sh:5) runs unchanged.

Every submodule is the corresponding module of `musicstyletransfer_amd` (the MI355X-native
implementation): `music_style_transfer.VarAutoEncoder.model` IS `musicstyletransfer_amd.VarAutoEncoder.model`.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_IMPL = "musicstyletransfer_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == __name__ or not fullname.startswith(__name__ + "."):
            return None
        real = _IMPL + fullname[len(__name__):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=True)

    def create_module(self, spec):
        real = _IMPL + spec.name[len(__name__):]
        return importlib.import_module(real)

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
