"""Does training in the 16-bit modes converge like training at the reference's precision?  (trainer.py:89-141 + :158-196.)

tools/convergence.py trains the BASELINE cfg #1-size networks from one initialisation on identical batches: three exact-f32
runs that differ only in their label noise (the fp32 run-to-run band) and three bf16 / f16 runs with the same three noise
seeds; the gated quantities are the MEDIAN of a mode's three runs and, per checkpoint, the NUMBER of single runs beyond the
same gate (at most one of three; the table also lists, per seed, how many checkpoints a run spends outside the band).  Gates, in half-widths of the band (floored at +-2 % of the
value): every smoothed loss from iteration 100 on within 6, the first 50 iterations -- where the three f32 runs have not
spread yet and the band is a hair -- within 12, final PSNR / SSIM of the generator on a held-out batch within 4.5; everything
finite.  Why the median: GAN training is chaotic.  With ONE run per mode (round 3's first form of this test) a re-ordering of
the weight gradient's summation (3e-7 relative, tests/test_ops.py::test_conv_wgrad_forms_agree_at_training_shapes) moved bf16 /
seed 0 from 0.7 to 11.4 half-widths on the content loss while its seeds 1-4 stayed inside the band -- deterministically, run
after run (profiles/r03_convergence.txt).  What the gate still excludes is the failure round 3 found and fixed: fp16 with the
former static loss scale of 2^14 ended EVERY run with a content loss 5-10x the fp32 band's and 6 dB PSNR."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LATE, EARLY, QUALITY = 6.0, 12.0, 4.5      # half-widths of the fp32 band: iterations >= 100 / the first checkpoints / PSNR, SSIM


def _tool():
    spec = importlib.util.spec_from_file_location("convergence", os.path.join(ROOT, "tools", "convergence.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_16bit_training_tracks_fp32_training(pkg):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    conv = _tool()
    lines = []
    # x3 (split bf16, three MFMAs per product: the fast mode inside the fp32 tolerance) runs beside the 16-bit modes, same gates
    # (bf16, the 16-bit mode of rounds 1-4, ran here too until round 6: 56 s of a suite with a time limit; its 8-seed table is
    # profiles/r05_convergence.txt, `python tools/convergence.py 300 out.json --modes bf16,f16,x3` reproduces it)
    results, rows, worst = conv.main(300, modes=("f16", "x3", "x3v"), log=lines.append)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "convergence.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    for r in results["f32"] + results["f16"] + results["x3"] + results["x3v"]:
        assert r["finite"]
    # the f32 runs themselves learn: pre-training lowers the pixel loss, the discriminator separates real from fake
    pre = results["f32"][0]["curves"]["pretrain_loss"]
    assert sum(pre[-10:]) < 0.5 * sum(pre[:10]), (pre[:3], pre[-3:])
    bad = []
    for row in rows:
        mode, k, t, lo, hi, v, dist, _runs = row
        gate = QUALITY if k in ("psnr", "ssim") else (LATE if t >= 100 and k != "pretrain_loss" else EARLY)
        if not dist < gate:
            bad.append((mode, k, t, round(lo, 5), round(hi, 5), round(v, 5), round(dist, 2)))
        # per seed, not only the median (round-3 verdict): at most ONE of a mode's three runs may be beyond the gate at any
        # checkpoint -- a mode where two trajectories leave is not tracking fp32, whatever its median does
        far = [round(d, 2) for d in conv.row_distances(row) if not d < gate]
        if len(far) > 1:
            bad.append((mode, k, t, "single runs beyond the gate", far))
    assert not bad, bad
    # and at most one of a mode's runs may spend most of its checkpoints outside the band proper (distance > 1; a band of three
    # f32 runs is narrow -- round 3: bf16 seeds 0 / 1 / 2 were outside at 18 / 8 / 5 of 28 checkpoints, f16 at 11 / 11 / 10)
    for mode, per_seed in conv.single_run_exits(rows).items():
        mostly_out = [seed for seed, (n_out, n_rows, _far) in enumerate(per_seed) if n_out > 0.6 * n_rows]
        assert len(mostly_out) <= 1, (mode, per_seed)
