"""Importable alias of the package `3d-gaussian-splatting-for-novel-view-synthesis_amd` (whose directory name is
not a Python identifier):  `import gsplat_amd as gs; gs.render(...)`."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
