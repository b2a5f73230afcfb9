"""CLI shim: `python inference.py --image_dir D --output_dir O` (see fast-srgan_amd/inference.py)."""
import importlib

if __name__ == "__main__":
    importlib.import_module("fast-srgan_amd.inference").main()
