"""Importable alias of the hot-path package.

The package directory carries the name the build contract prescribes
(``automated-brain-mri-...-assistance_amd``), which is not a Python identifier; this shim
loads it once and aliases it (and its submodules) as ``brats_amd``.
"""
import importlib
import sys

_LONG = "automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd"
_pkg = importlib.import_module(_LONG)
for _k, _v in list(sys.modules.items()):
    if _k == _LONG or _k.startswith(_LONG + "."):
        sys.modules["brats_amd" + _k[len(_LONG):]] = _v
sys.modules[__name__] = _pkg
