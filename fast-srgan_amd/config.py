"""configs/config.yaml loader with hydra-style `a.b=c` overrides (/root/reference/train.py:46, README.md:45).

hydra / omegaconf are not dependencies; PyYAML reads the same file and the result is an attribute tree with
the reference's 20 keys (configs/config.yaml:1-25) plus `generator.n_upsample` (default 2),
`training.compute_dtype` (bf16 | f16 | f32, default bf16), `training.loss_scale` (static, default 16384 for f16 else 1), `training.vgg19_weights` (path of torchvision's vgg19 checkpoint) and
`training.allow_random_vgg` (tests / benchmarks only).
"""
import os
import types

import yaml

DEFAULTS = {
    "experiment": {"name": "SRGAN", "seed": 1234},
    "data": {"image_dir": "", "numpy_dir": "", "lr_image_size": 24, "scale_factor": 4},
    "generator": {"n_filters": 64, "n_layers": 8, "n_upsample": 2},
    "discriminator": {"n_filters": 64, "n_layers": 7},
    "training": {"compiled": False, "pretrain_iterations": 100, "iterations": 100, "device": "cuda", "log_iter": 5000,
                 "checkpoint_iter": 5000, "batch_size": 24, "num_workers": 16, "generator_lr": 1e-4,
                 "discriminator_lr": 1e-4, "compute_dtype": "bf16", "vgg19_weights": "", "allow_random_vgg": False,
                 "hip_graph": True},
}


class Node(types.SimpleNamespace):
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else v) for k, v in vars(self).items()}


def _to_node(d):
    return Node(**{k: (_to_node(v) if isinstance(v, dict) else v) for k, v in d.items()})


def _parse_scalar(text):
    return yaml.safe_load(text)


def load_config(path=None, overrides=()):
    cfg = {k: dict(v) for k, v in DEFAULTS.items()}
    if path is not None and os.path.exists(path):
        with open(path) as f:
            loaded = yaml.safe_load(f) or {}
        for group, vals in loaded.items():
            cfg.setdefault(group, {}).update(vals or {})
    for ov in overrides:
        if "=" not in ov:
            raise ValueError("override %r is not of the form a.b=c" % ov)
        key, val = ov.split("=", 1)
        parts = key.split(".")
        node = cfg
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = _parse_scalar(val)
    for k in ("generator_lr", "discriminator_lr"):      # YAML 1.1 reads "1e-4" as a string
        cfg["training"][k] = float(cfg["training"][k])
    if cfg["training"]["device"] in ("mps", "cpu"):     # the shipped config names mps (configs/config.yaml:19)
        import warnings
        warnings.warn("training.device=%s: this framework runs on the MI355X only, using 'cuda'" % cfg["training"]["device"])
        cfg["training"]["device"] = "cuda"
    return _to_node(cfg)
