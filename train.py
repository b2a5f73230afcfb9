"""CLI shim: `python train.py [a.b=c ...]` (see fast-srgan_amd/train.py)."""
import importlib

if __name__ == "__main__":
    importlib.import_module("fast-srgan_amd.train").main()
