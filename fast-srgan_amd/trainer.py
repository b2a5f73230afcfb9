"""GAN training loop on the MI355X kernels: drop-in for /root/reference/trainer.py.

Same surface (Trainer(config), .pretrain, .train, .save_checkpoints, attributes generator / discriminator /
perceptual_network / optim_generator / optim_discriminator / loss_fn / l1_loss / writer, checkpoint file
names) and the same arithmetic per iteration as trainer.py:171-196.  Differences, all results-neutral:
  * G(lr) is evaluated once per iteration and shared by the D step (detached) and the G step -- the
    reference evaluates it twice with identical parameters (trainer.py:173 and :185);
  * the discriminator's weight gradients of the G step, which the reference computes and then discards
    at the next zero_grad (trainer.py:171), are not computed;
  * D(hr) and D(G(lr).detach()) of the D step (trainer.py:172,174) run as one batch of 2B images (same weights,
    per-sample ops), and the frozen perceptual branch runs on a second HIP stream;
  * one process per GPU: gradients are averaged across ranks with one RCCL all-reduce per optimizer.
"""
import os
import os.path as osp

import torch

from . import distributed as D
from . import ops
from .model import VGG19, Discriminator, Generator
from .optim import ArenaAdamW


class _NullWriter:
    """Stand-in for torch.utils.tensorboard.SummaryWriter (tensorboard is optional): keeps the last scalars."""

    def __init__(self, *a, **k):
        self.scalars = {}

    def add_scalar(self, tag, value, global_step=None):
        self.scalars[tag] = (float(value), global_step)

    def add_images(self, *a, **k):
        pass

    def flush(self):
        pass


def _make_writer(log_dir):
    try:
        from torch.utils.tensorboard.writer import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return _NullWriter()


class Trainer:
    fixed_lr_images = torch.tensor([])
    fixed_hr_images = torch.tensor([])

    def __init__(self, config, vgg_weights=None, perceptual_network=None):
        self.config = config
        dev = self.config.training.device
        cdt = getattr(config.training, "compute_dtype", "bf16")
        self.is_main = D.rank() == 0
        self.writer = _make_writer(osp.join("runs", config.experiment.name)) if self.is_main else _NullWriter()
        self.generator = Generator(config=config.generator, compute_dtype=cdt).to(dev)
        self.discriminator = Discriminator(config=config.discriminator, compute_dtype=cdt).to(dev)
        self.perceptual_network = (perceptual_network or VGG19(weights=vgg_weights, compute_dtype=cdt)).to(dev)
        # training.compiled (trainer.py:23-26) selects torch.compile/Triton in the reference; here every op is
        # already a hand-written HIP kernel, so the key is accepted and changes nothing.
        self.perceptual_network.eval()
        for p in self.perceptual_network.parameters():
            p.requires_grad = False
        self.optim_generator = ArenaAdamW(self.generator.parameters(), lr=self.config.training.generator_lr)
        self.optim_discriminator = ArenaAdamW(self.discriminator.parameters(), lr=self.config.training.discriminator_lr)
        D.broadcast_parameters(self.optim_generator)
        D.broadcast_parameters(self.optim_discriminator)
        self._sync_g = D.GradSync(self.optim_generator)
        self._sync_d = D.GradSync(self.optim_discriminator)
        self._streams = {}
        self.use_side_stream = os.environ.get("FSR_SIDE_STREAM", "1") != "0"
        self.loss_fn = ops.bce_with_logits      # torch.nn.BCEWithLogitsLoss(), trainer.py:41
        self.l1_loss = ops.smooth_l1            # torch.nn.SmoothL1Loss(), trainer.py:43

    def _side_stream(self, device):
        st = self._streams.get(str(device))
        if st is None:
            st = self._streams[str(device)] = torch.cuda.Stream(device=device)
        return st

    # ------------------------------------------------------------------ one GAN iteration, trainer.py:171-196
    def train_step(self, lr_images, hr_images, noise=None):
        """noise: optional (n0, n1, n2) replacing the torch.rand_like draws of trainer.py:175,176,187."""
        ops.zero_pool_reset(lr_images.device)   # one memset for all statistics / reduction scratch of the iteration
        ops.wgrad_stream_begin(lr_images.device)   # weight gradients run beside the data-gradient chain (ops.py)
        try:
            return self._train_step(lr_images, hr_images, noise)
        finally:
            ops.wgrad_stream_end()
            ops.zero_pool_end(lr_images.device)

    def _train_step(self, lr_images, hr_images, noise):
        G, Dm, V = self.generator, self.discriminator, self.perceptual_network
        # The frozen perceptual branch (VGG(hr), VGG(sr) and its backward: ~45 % of the kernel time) runs on a second
        # HIP stream: it only meets the rest of the iteration at `sr_images` and at the loss sum, so its kernels
        # fill the gaps the discriminator / generator kernels leave (tails, 1-workgroup-per-CU weight gradients,
        # bandwidth-bound elementwise passes).  Captured hipGraphs keep the two branches as parallel graph paths.
        main = torch.cuda.current_stream() if lr_images.is_cuda else None
        side = self._side_stream(lr_images.device) if (main is not None and self.use_side_stream) else None

        def on_side(fn):
            if side is None:
                return fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                return fn()

        def features_no_grad():
            with torch.no_grad():                                               # :191 (no gradient needed: target)
                return V.features_nhwc(hr_images)

        real_features = on_side(features_no_grad)
        # ---- discriminator step
        self.optim_discriminator.zero_grad()                                    # :171
        sr_images = G(lr_images)                                                # :173 / :185 (shared)
        # :172 and :174 use the same discriminator weights and every op is per-sample, so real and fake images go
        # through D as ONE batch of 2B: half the launches, one weight-gradient pass instead of two
        y_both = Dm(torch.cat([hr_images, sr_images.detach()], dim=0))
        y_real, y_fake = y_both[:hr_images.shape[0]], y_both[hr_images.shape[0]:]
        n0 = torch.rand_like(y_real) if noise is None else noise[0]
        n1 = torch.rand_like(y_fake) if noise is None else noise[1]
        real_labels = 0.3 * n0 + 0.8                                            # :175
        fake_labels = 0.3 * n1                                                  # :176
        loss_real = self.loss_fn(y_real, real_labels)                           # :177
        loss_fake = self.loss_fn(y_fake, fake_labels)                           # :178
        discriminator_loss = 0.5 * loss_real + 0.5 * loss_fake                  # :179
        discriminator_loss.backward()                                           # :180
        ops.wgrad_stream_join()
        self._sync_d.start()
        # content branch of the generator step (:190, :192) starts as soon as sr_images exists
        content_loss = on_side(lambda: self.l1_loss(V.features_nhwc(sr_images), real_features))
        self._sync_d.wait()
        self.optim_discriminator.step()                                         # :181
        # ---- generator step
        self.optim_generator.zero_grad()                                        # :184
        for p in Dm.parameters():
            p.requires_grad_(False)      # D's weight gradients of this pass are discarded by the reference (:171)
        y_fake = Dm(sr_images)                                                  # :186 (updated D)
        n2 = torch.rand_like(y_fake) if noise is None else noise[2]
        real_labels = 0.3 * n2 + 0.7                                            # :187
        adv_loss = 1e-1 * self.loss_fn(y_fake, real_labels)                     # :188
        if side is not None:
            main.wait_stream(side)
        generator_loss = 0.5 * adv_loss + 0.5 * content_loss                    # :194
        generator_loss.backward()                                               # :195
        for p in Dm.parameters():
            p.requires_grad_(True)
        ops.wgrad_stream_join()
        self._sync_g.start()
        self._sync_g.wait()
        self.optim_generator.step()                                             # :196
        # the loss scalars live in the per-iteration scratch arena: copy them out (one launch) so they survive the next reset
        vals = torch.stack([loss_real.detach(), loss_fake.detach(), adv_loss.detach(), content_loss.detach()])
        return dict(zip(("loss_real", "loss_fake", "adv_loss", "content_loss"), vals.unbind(0)))

    # ------------------------------------------------------------------ hipGraph replay of the whole iteration
    def capture_train_step(self, lr_images, hr_images, warmup=2):
        """Captures one full iteration (both optimizer steps included) into a hipGraph; `graphed_train_step`
        then replays it with new batch contents: ~500 kernel launches become one graph launch, which removes the
        host-side launch gaps of the eager loop.  Single-process only (the RCCL exchange stays eager).  Label noise
        comes from torch's graph-safe Philox generator, so every replay draws fresh numbers."""
        if D.world_size() > 1:
            raise RuntimeError("capture_train_step is single-process; data-parallel runs use train_step")
        self._g_lr, self._g_hr = lr_images.clone(), hr_images.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm every lazily created buffer / cache on a side stream
            for _ in range(warmup):
                self.train_step(self._g_lr, self._g_hr)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_out = self.train_step(self._g_lr, self._g_hr)
        return self._graph

    def graphed_train_step(self, lr_images, hr_images):
        self._g_lr.copy_(lr_images, non_blocking=True)
        self._g_hr.copy_(hr_images, non_blocking=True)
        self._graph.replay()
        # the replay stepped both optimizers on the device without running any Python: advance the host-side epochs so
        # that eager code running afterwards (evaluation, checkpoint-time inference) re-packs the filters it caches
        for opt in (self.optim_generator, self.optim_discriminator):
            opt.mark_updated()
        return self._graph_out

    def pretrain_step(self, lr_images, hr_images):
        """trainer.py:107-111."""
        ops.zero_pool_reset(lr_images.device)
        try:
            self.optim_generator.zero_grad()
            fake_hr_images = self.generator(lr_images)
            gen_loss = self.l1_loss(fake_hr_images, hr_images)
            gen_loss.backward()
            self._sync_g.start()
            self._sync_g.wait()
            self.optim_generator.step()
            return gen_loss.detach().clone()
        finally:
            ops.zero_pool_end(lr_images.device)

    # ------------------------------------------------------------------ cold paths
    @staticmethod
    def _ssim(a, b, data_range=1.0):
        """torchmetrics StructuralSimilarityIndexMeasure defaults (trainer.py:46-48): gaussian 11x11, sigma 1.5,
        k1 0.01, k2 0.03, per-image mean over the valid (unpadded) region.  Eval-only cold path: plain torch ops."""
        k = torch.arange(11, dtype=torch.float32, device=a.device) - 5
        g = torch.exp(-(k ** 2) / (2 * 1.5 ** 2))
        g = (g / g.sum()).outer(g / g.sum())
        w = g.expand(a.shape[1], 1, 11, 11).contiguous()
        c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        pad = lambda t: torch.nn.functional.pad(t, (5, 5, 5, 5), mode="reflect")          # torchmetrics pads, then crops
        f = lambda t: torch.nn.functional.conv2d(pad(t), w, groups=a.shape[1])
        mu_a, mu_b = f(a), f(b)
        s_aa, s_bb, s_ab = f(a * a) - mu_a ** 2, f(b * b) - mu_b ** 2, f(a * b) - mu_a * mu_b
        ssim = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2))
        return ssim[..., 5:-5, 5:-5].reshape(a.shape[0], -1).mean(-1)

    @torch.no_grad()
    def _calculate_metrics_over_dataset(self, dataloader, phase, step):
        """SSIM and PSNR over the loader (data_range 1.0), trainer.py:53-69 (torchmetrics restated; eval-only)."""
        self.generator.eval()
        psnr, ssim, count = 0.0, 0.0, 0
        for lr_images, hr_images in dataloader:
            lr_images = lr_images.to(self.config.training.device, non_blocking=True)
            hr_images = hr_images.to(self.config.training.device, non_blocking=True)
            sr = ((1.0 + self.generator(lr_images)) / 2.0).contiguous()
            hr = (1.0 + hr_images) / 2.0
            mse = ((sr - hr) ** 2).mean(dim=(1, 2, 3))
            psnr += float((10.0 * torch.log10(1.0 / mse)).sum())
            ssim += float(self._ssim(sr, hr).sum())
            count += mse.numel()
        if count:
            self.writer.add_scalar(f"{phase}/SSIM", ssim / count, global_step=step)
            self.writer.add_scalar(f"{phase}/PSNR", psnr / count, global_step=step)
        self.writer.flush()

    @classmethod
    def _pre_train_setup(cls, dataloader):
        if cls.fixed_lr_images.ndim == 1:
            for fixed_lr_images, fixed_hr_images in dataloader:
                cls.fixed_lr_images = (fixed_lr_images + 1.0) / 2.0
                cls.fixed_hr_images = (fixed_hr_images + 1.0) / 2.0
                break

    def pretrain(self, train_dataloader, val_dataloader):
        # The reference looks for runs/pretrain.pt but writes runs/pretrain_generator.pt (trainer.py:90 vs :133),
        # so its resume never triggers; both names are honoured here.
        for name in ("runs/pretrain.pt", "runs/pretrain_generator.pt"):
            if osp.exists(name):
                print("Pretrained model found, skipping pretraining")
                ckpt = torch.load(name, map_location="cpu")
                self.generator.load_state_dict(ckpt["model"])
                self.optim_generator.load_state_dict(ckpt["optimizer"])
                return
        self._calculate_metrics_over_dataset(val_dataloader, "Pretrain", step=0)
        self._pre_train_setup(val_dataloader)
        dev = self.config.training.device
        for step, (lr_images, hr_images) in enumerate(train_dataloader, start=1):
            lr_images, hr_images = lr_images.to(dev, non_blocking=True), hr_images.to(dev, non_blocking=True)
            gen_loss = self.pretrain_step(lr_images, hr_images)
            if step % self.config.training.log_iter == 0:
                self.writer.add_scalar("Pretrain/Generator/Loss", gen_loss, global_step=step)
            if step % self.config.training.checkpoint_iter == 0:
                self._calculate_metrics_over_dataset(val_dataloader, "Pretrain", step)
                self.generator.train()
        if self.is_main:
            os.makedirs("runs", exist_ok=True)
            torch.save({"model": self.generator.state_dict(), "optimizer": self.optim_generator.state_dict()},
                       "runs/pretrain_generator.pt")
            torch.save({"model": self.discriminator.state_dict(), "optimizer": self.optim_discriminator.state_dict()},
                       "runs/pretrain_discriminator.pt")

    def save_checkpoints(self, step):
        """trainer.py:143-156 file names; rank 0 only under data parallelism."""
        if not self.is_main:
            return
        save_dir = osp.join("runs", self.config.experiment.name)
        os.makedirs(save_dir, exist_ok=True)
        torch.save(self.generator.state_dict(), osp.join(save_dir, f"generator_epoch_{step}.pt"))
        torch.save(self.discriminator.state_dict(), osp.join(save_dir, f"discriminator_epoch_{step}.pt"))
        torch.save(self.optim_generator.state_dict(), osp.join(save_dir, f"generator_optim_epoch_{step}.pt"))
        torch.save(self.optim_discriminator.state_dict(), osp.join(save_dir, f"discriminator_optim_epoch_{step}.pt"))

    def train(self, train_dataloader, val_dataloader):
        self._calculate_metrics_over_dataset(val_dataloader, "GAN", step=0)
        self.generator.train()
        self.discriminator.train()
        dev = self.config.training.device
        for step, (lr_images, hr_images) in enumerate(train_dataloader, start=1):
            lr_images, hr_images = lr_images.to(dev, non_blocking=True), hr_images.to(dev, non_blocking=True)
            losses = self.train_step(lr_images, hr_images)
            if step % self.config.training.log_iter == 0:       # the only host<->device syncs of the loop
                self.writer.add_scalar("Loss/Discriminator/Real", losses["loss_real"], global_step=step)
                self.writer.add_scalar("Loss/Discriminator/Fake", losses["loss_fake"], global_step=step)
                self.writer.add_scalar("Loss/Generator/Adversarial", losses["adv_loss"], global_step=step)
                self.writer.add_scalar("Loss/Generator/Content", losses["content_loss"], global_step=step)
            if step % self.config.training.checkpoint_iter == 0:
                self.generator.eval()
                self._calculate_metrics_over_dataset(val_dataloader, "GAN", step=step)
                self.save_checkpoints(step)
                self.generator.train()
