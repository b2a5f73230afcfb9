"""`python inference.py --image_dir D --output_dir O`: drop-in for /root/reference/inference.py.

Reads configs/config.yaml and models/model.pt relative to the working directory (inference.py:26-27), strips
the `_orig_mod.` prefix torch.compile left in the shipped checkpoint (:31-33), accepts .png/.jpg/jpeg in any
case (:37-45), writes each result under the same basename (:57) and converts with the reference's
TRUNCATING uint8 cast (:53-56).  The generator forward runs on the MI355X kernels.
"""
import os
from argparse import ArgumentParser

import numpy as np
import torch

from .config import load_config
from .model import Generator

parser = ArgumentParser("Real Time Image Super Resolution")
parser.add_argument("--image_dir", default=None, required=True, type=str)
parser.add_argument("--output_dir", default=None, required=True, type=str)
parser.add_argument("--compute_dtype", default=None, choices=["bf16", "f32"], help="extension: kernel precision")


def load_generator(config, model_path, device="cuda", compute_dtype=None):
    model = Generator(config.generator, compute_dtype=compute_dtype or getattr(config.training, "compute_dtype", "bf16"))
    weights = torch.load(model_path, map_location="cpu")
    model.load_state_dict({k.replace("_orig_mod.", ""): v for k, v in weights.items()})
    return model.to(device).eval()


@torch.no_grad()
def super_resolve(model, lr_u8_hwc, device="cuda"):
    """uint8 (H,W,3) -> uint8 (4H,4W,3), inference.py:48-56."""
    lr_image = (torch.from_numpy(lr_u8_hwc) / 127.5) - 1.0
    lr_image = lr_image.permute(2, 0, 1).unsqueeze(dim=0).to(device)
    sr_image = model(lr_image).cpu()
    sr_image = (sr_image + 1.0) / 2.0
    sr_image = sr_image.permute(0, 2, 3, 1).squeeze()
    return (sr_image * 255).numpy().astype(np.uint8)


def main(argv=None):
    from PIL import Image
    args = parser.parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("fast-srgan_amd runs on an MI355X only: no GPU is visible")
    device = "cuda"
    print(f"Using device: {device}")
    config = load_config("configs/config.yaml")
    model = load_generator(config, "models/model.pt", device, args.compute_dtype)
    image_paths = sorted(x for x in os.listdir(args.image_dir)
                         if x.lower().endswith(".png") or x.lower().endswith(".jpg") or x.lower().endswith("jpeg"))
    print(f"Found {len(image_paths)} to super resolve, starting...")
    for image_path in image_paths:
        lr_image = np.array(Image.open(os.path.join(args.image_dir, image_path)).convert("RGB"))
        Image.fromarray(super_resolve(model, lr_image, device)).save(os.path.join(args.output_dir, os.path.basename(image_path)))


if __name__ == "__main__":
    main()
