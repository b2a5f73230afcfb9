"""Import alias: `import fast_srgan_amd` -> the package in ./fast-srgan_amd/ (hyphenated directory)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("fast-srgan_amd")
