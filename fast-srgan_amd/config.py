"""configs/config.yaml loader with hydra-style `a.b=c` overrides (/root/reference/train.py:46, README.md:45).

hydra / omegaconf are not dependencies; PyYAML reads the same file and the result is an attribute tree with
the reference's 20 keys (configs/config.yaml:1-25) plus `generator.n_upsample` (default 2),
`training.loss_scale` / `dynamic_loss_scale` / `loss_scale_growth_interval` / `loss_scale_growth` (fp16 and x3v: DESIGN 2c, 5), `training.compute_dtype` (bf16 | f16 | x3 | x3v | f32; x3v = x3 Generator / Discriminator + fp16 frozen perceptual network: every loss and output still within 1e-3 of fp32, 31 % faster than x3; default f16 -- the 16-bit mode that tracks fp32 training best, profiles/r05_convergence.txt; x3 = split-bf16 operands, the fast mode whose forward outputs and losses sit inside the reference's 1e-3 fp32 tolerance -- its parameter gradients are 1.3x (G) / 2.8x (D) as far from float64 as the float32 reference's own), `training.vgg19_weights` (path of torchvision's vgg19
checkpoint), `training.allow_random_vgg` (tests / benchmarks only), `training.hip_graph` (replay the iteration as hipGraphs)
and the fp16 loss scaler:

* `training.loss_scale` -- the INITIAL scale (default 2**20 for f16, 1 otherwise; must be > 0),
* `training.dynamic_loss_scale` (default true for f16): the scale is adapted on the device -- a non-finite gradient arena
  skips BOTH optimizer updates of the iteration and halves the scale; `training.loss_scale_growth_interval` (default 1000,
  >= 1) clean iterations double it.  With `false` the scale is static.

hydra run directory (/root/reference/train.py:46 is `@hydra.main(version_base="1.1", ...)`, under which hydra changes the
working directory to `outputs/<date>/<time>` before `main` runs, so `runs/...` checkpoints land there): `enter_run_dir`
reproduces that, including hydra's own override keys `hydra.run.dir=<path>` and `hydra.job.chdir=false`.
"""
import datetime
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
                 "discriminator_lr": 1e-4, "compute_dtype": "f16", "vgg19_weights": "", "allow_random_vgg": False,
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
        if ov.startswith("hydra."):                     # hydra's own keys: enter_run_dir reads the two it knows
            if ov.split("=", 1)[0] not in ("hydra.run.dir", "hydra.job.chdir"):
                # (a warning, not an error: a reference command line carrying e.g. hydra.verbose=true must keep working)
                import warnings
                warnings.warn("override %r ignored: the only hydra.* keys reproduced here are hydra.run.dir and hydra.job.chdir" % ov)
            continue
        if ov[:1] in "+~":                              # hydra's append / delete syntax is not a config key
            raise ValueError("override %r: hydra's +key / ~key syntax is not supported (use a.b=c)" % ov)
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


def hydra_run_settings(overrides=(), now=None):
    """(chdir, run_dir) as hydra 1.1 would choose them: chdir on, `outputs/%Y-%m-%d/%H-%M-%S` unless overridden."""
    chdir, run_dir = True, None
    for ov in overrides:
        if ov.startswith("hydra.job.chdir="):
            chdir = bool(_parse_scalar(ov.split("=", 1)[1]))
        elif ov.startswith("hydra.run.dir="):
            run_dir = str(ov.split("=", 1)[1])
    if run_dir is None:
        now = now or datetime.datetime.now()
        run_dir = os.path.join("outputs", now.strftime("%Y-%m-%d"), now.strftime("%H-%M-%S"))
    return chdir, run_dir


def enter_run_dir(cfg, overrides=(), run_dir=None, create=True):
    """What `@hydra.main(version_base="1.1")` does around the reference's `main` (/root/reference/train.py:46): make the run
    directory, record the composed config and the overrides under `.hydra/`, and change into it.  Relative `data.*` paths are
    made absolute first (hydra users call `to_absolute_path` for that; the reference's config holds absolute paths).
    `run_dir`: the directory rank 0 chose (every rank must enter the same one).  Returns the directory, or None when
    `hydra.job.chdir=false`."""
    chdir, chosen = hydra_run_settings(overrides)
    run_dir = run_dir or chosen
    for key in ("image_dir", "numpy_dir"):
        val = getattr(cfg.data, key, "")
        if val:
            setattr(cfg.data, key, os.path.abspath(val))
    if getattr(cfg.training, "vgg19_weights", ""):
        cfg.training.vgg19_weights = os.path.abspath(cfg.training.vgg19_weights)
    if not chdir:
        return None
    run_dir = os.path.abspath(run_dir)
    if create:
        os.makedirs(os.path.join(run_dir, ".hydra"), exist_ok=True)
        with open(os.path.join(run_dir, ".hydra", "config.yaml"), "w") as f:
            yaml.safe_dump(cfg.to_dict(), f, sort_keys=False)
        with open(os.path.join(run_dir, ".hydra", "overrides.yaml"), "w") as f:
            yaml.safe_dump([o for o in overrides if not o.startswith("hydra.")], f)
    os.chdir(run_dir)
    return run_dir
