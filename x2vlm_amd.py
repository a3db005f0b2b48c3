"""Import alias: `import x2vlm_amd` -> the package directory `x2-vlm_amd/` (hyphenated name)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("x2-vlm_amd")
