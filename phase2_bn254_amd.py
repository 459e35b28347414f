"""Import shim: `import phase2_bn254_amd` -> the package in ./phase2-bn254_amd/ (hyphenated directory)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("phase2-bn254_amd")
