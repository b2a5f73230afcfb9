"""CLI shim: `python train.py [a.b=c ...]` (see fast-srgan_amd/train.py)."""
import importlib

if __name__ == "__main__":
    import os
    # hydra resolves config_path="configs" next to the script (train.py:46), not in the working directory
    importlib.import_module("fast-srgan_amd.train").main(config_dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"))
