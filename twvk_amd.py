"""Import shim: registers the directory `tacotron-wavenet-vocoder-korean_amd/` (hyphenated, so not importable
by name) as the package `twvk_amd`.  `import twvk_amd` then `twvk_amd.WaveNetModel`, `twvk_amd.hparams`, ..."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tacotron-wavenet-vocoder-korean_amd")
_spec = importlib.util.spec_from_file_location("twvk_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["twvk_amd"] = _mod
_spec.loader.exec_module(_mod)
