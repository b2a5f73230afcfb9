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
        cdt = getattr(config.training, "compute_dtype", "f16")
        self.is_main = D.rank() == 0
        self.writer = _make_writer(osp.join("runs", config.experiment.name)) if self.is_main else _NullWriter()
        self.generator = Generator(config=config.generator, compute_dtype=cdt).to(dev)
        self.discriminator = Discriminator(config=config.discriminator, compute_dtype=cdt).to(dev)
        if perceptual_network is None:      # extension keys: training.vgg19_weights (path), training.allow_random_vgg
            vgg_weights = vgg_weights or getattr(config.training, "vgg19_weights", None) or None
            perceptual_network = VGG19(weights=vgg_weights, compute_dtype=cdt,
                                       allow_random=bool(getattr(config.training, "allow_random_vgg", False)))
        self.perceptual_network = perceptual_network.to(dev)
        # training.compiled (trainer.py:23-26) selects torch.compile/Triton in the reference; here every op is
        # already a hand-written HIP kernel, so the key is accepted and changes nothing.
        self.perceptual_network.eval()
        for p in self.perceptual_network.parameters():
            p.requires_grad = False
        self.optim_generator = ArenaAdamW(self.generator.parameters(), lr=self.config.training.generator_lr)
        self.optim_discriminator = ArenaAdamW(self.discriminator.parameters(), lr=self.config.training.discriminator_lr)
        D.broadcast_parameters(self.optim_generator)
        D.broadcast_parameters(self.optim_discriminator)
        self._sync_g = D.GradSync(self.optim_generator, "generator")
        self._sync_d = D.GradSync(self.optim_discriminator, "discriminator")
        # Loss scaling (extension keys training.loss_scale, training.dynamic_loss_scale; defaults 1 / off, and 2^20 / on in
        # the fp16 mode): both backward passes are seeded with S and AdamW divides the gradients by S again.  fp16 activation
        # gradients need it: the content loss is a mean over N x 512 x 24 x 24 values, its per-element gradient (~4e-8) lies
        # below the smallest fp16 subnormal.  bf16 and f32 share float32's exponent range and run unscaled.
        # Dynamic (the fp16 default): S lives on the device; every optimizer step checks its gradient arena for inf / NaN
        # and skips the update when it finds one, the iteration then halves S, and 1000 clean iterations double it
        # (fsr_grad_nonfinite / fsr_adamw_step_scaled / fsr_loss_scale_update: decided on the device, so hipGraph replays adapt).
        # COUPLED skip, unlike torch.amp.GradScaler's per-optimizer found_inf: both optimizers share one non-finite flag, so a
        # discriminator overflow also skips the generator update of that iteration (its gradients went through the same
        # too-large scale).  The flag is cleared at the START of every iteration as well as by the scale update, so an
        # exception between a flagged step and the update cannot leave it up for the next iteration.
        # The initial 2^20 is measured (profiles/r03_f16_loss_scale.txt): at 2^14 nothing overflows
        # but part of the perceptual gradient underflows and 300 iterations end with a content loss 5-10x the fp32 runs';
        # 2^20 .. 2^22 track fp32; 2^26 overflows, is halved four times in the first iterations and then tracks fp32 too.
        # (x3v: the perceptual network alone runs in fp16 -- its backward needs the scale all the same; the x3 networks share
        # float32's exponent range and are indifferent to it)
        scaled = "f16" in (self.generator.compute.name, self.perceptual_network.compute.name)
        self.loss_scale = float(getattr(config.training, "loss_scale", 1048576.0 if scaled else 1.0))
        if not (self.loss_scale > 0.0 and self.loss_scale == self.loss_scale and self.loss_scale != float("inf")):
            raise ValueError("training.loss_scale must be a positive finite number, got %r" % (self.loss_scale,))
        self.dynamic_loss_scale = bool(getattr(config.training, "dynamic_loss_scale", scaled))
        self.loss_scale_growth_interval = float(getattr(config.training, "loss_scale_growth_interval", 1000))
        # Growth factor after `growth_interval` clean iterations: 2 when the TRAINED networks are fp16 (their own gradient arenas
        # overflow first and bring the scale back down), 1 -- no growth -- in x3v, where only the frozen perceptual network is
        # fp16: the x3 arenas share float32's range and would never report that the scale has outgrown fp16, so a growing scale
        # would end in a perceptual backward that overflows every iteration (found in round 6).  An overflow that does reach an
        # arena still skips the iteration and halves the scale.
        self.loss_scale_growth = float(getattr(config.training, "loss_scale_growth", 2.0 if self.generator.compute.name == "f16" else 1.0))
        if not self.loss_scale_growth >= 1.0:
            raise ValueError("training.loss_scale_growth must be >= 1, got %r" % (self.loss_scale_growth,))
        if not self.loss_scale_growth_interval >= 1.0:
            raise ValueError("training.loss_scale_growth_interval must be >= 1, got %r" % (self.loss_scale_growth_interval,))
        self._scale_state = None
        if self.dynamic_loss_scale:
            self._scale_state = torch.tensor([self.loss_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
            self._seed = self._scale_state[0]          # a 0-dim VIEW: the next backward is seeded with whatever S has become
            for opt in (self.optim_generator, self.optim_discriminator):
                opt.scale_state = self._scale_state
        else:
            self._seed = torch.tensor(self.loss_scale, dtype=torch.float32, device=dev) if self.loss_scale != 1.0 else None
            for opt in (self.optim_generator, self.optim_discriminator):
                opt.grad_scale = opt.grad_scale / self.loss_scale
        self._streams = {}
        self._seed_consts = {}
        self.use_side_stream = os.environ.get("FSR_SIDE_STREAM", "1") != "0"
        self.loss_fn = ops.bce_with_logits      # torch.nn.BCEWithLogitsLoss(), trainer.py:41
        self.l1_loss = ops.smooth_l1            # torch.nn.SmoothL1Loss(), trainer.py:43

    def _side_stream(self, device):
        st = self._streams.get(str(device))
        if st is None:
            st = self._streams[str(device)] = torch.cuda.Stream(device=device)
        return st

    # ------------------------------------------------------------------ one GAN iteration, trainer.py:171-196
    # The iteration is three phases separated by the two gradient exchanges of data parallelism:
    #   D phase : G(lr), VGG(hr), VGG(sr) + content loss (forward only), D(hr | sr.detach()), BCE, D backward
    #             -- all-reduce of the discriminator gradient arena --
    #   G phase : AdamW(D), updated D(sr), adversarial loss, backward through D / VGG / G
    #             -- all-reduce of the generator gradient arena --
    #   end     : AdamW(G), loss scalars
    # Eager execution runs them back to back; capture_train_step turns each phase into a hipGraph (ONE graph when
    # there is no exchange, i.e. a single process).
    def train_step(self, lr_images, hr_images, noise=None):
        """noise: optional (n0, n1, n2) replacing the torch.rand_like draws of trainer.py:175,176,187."""
        try:
            st = self._phase_d(lr_images, hr_images, noise)
            self._sync_d.start()        # RCCL all-reduce of D's gradient arena on the collective's stream; nothing waits yet
            self._phase_g(st, noise)    # ... waits for it right before optim_discriminator.step(); the side stream's VGG branch
            self._sync_g.start()        #     and the generator's zero_grad run beside the exchange
            return self._phase_end(st)  # waits for the generator exchange right before optim_generator.step()
        finally:
            self._end_iteration(lr_images.device)

    def _end_iteration(self, device):
        for p in self.discriminator.parameters():
            p.requires_grad_(True)
        ops.wgrad_stream_end()
        ops.zero_pool_end(device)

    def _phase_d(self, lr_images, hr_images, noise, join_side=False):
        G, Dm, V = self.generator, self.discriminator, self.perceptual_network
        dev = lr_images.device
        ops.zero_pool_reset(dev)        # one memset for all statistics / reduction scratch of the iteration
        ops.wgrad_stream_begin(dev)     # weight gradients run beside the data-gradient chain (ops.py)
        self._clear_nonfinite_flag()
        # The frozen perceptual branch (VGG(hr), VGG(sr) and its backward: ~45 % of the kernel time) runs on a second
        # HIP stream: it only meets the rest of the iteration at `sr_images` and at the loss sum, so its kernels
        # fill the gaps the discriminator / generator kernels leave (tails, 1-workgroup-per-CU weight gradients,
        # bandwidth-bound elementwise passes).  Captured hipGraphs keep the two branches as parallel graph paths.
        main = torch.cuda.current_stream() if lr_images.is_cuda else None
        side = self._side_stream(dev) if (main is not None and self.use_side_stream) else None

        def on_side(fn):
            """fn's launches go to the side stream, ordered after everything queued on the main stream SO FAR."""
            if side is None:
                return fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                return fn()

        def features_no_grad():
            with torch.no_grad():                                               # :191 (no gradient needed: target)
                return V.features_nhwc(hr_images)

        real_features = on_side(features_no_grad)
        self.optim_discriminator.zero_grad()                                    # :171
        sr_images = G(lr_images)                                                # :173 / :185 (shared)
        # content branch of the generator step (:190, :192): queued behind G(lr) only -- the side stream starts it as soon
        # as sr_images exists, beside the discriminator's forward and backward
        content_loss = on_side(lambda: self.l1_loss(V.features_nhwc(sr_images), real_features, cd=V.compute))
        # :172 and :174 use the same discriminator weights and every op is per-sample, so real and fake images go
        # through D as ONE batch of 2B: half the launches, one weight-gradient pass instead of two
        y_both = Dm(torch.cat([hr_images, sr_images.detach()], dim=0))
        y_real, y_fake = y_both[:hr_images.shape[0]], y_both[hr_images.shape[0]:]
        # :175-176  0.3 * rand + 0.8 and 0.3 * rand are U[0.8, 1.1) and U[0, 0.3): ONE generator launch each instead of
        # rand + mul + add (given noise tensors -- the parity tests -- keep the reference's arithmetic)
        real_labels = torch.empty_like(y_real).uniform_(0.8, 1.1) if noise is None else 0.3 * noise[0] + 0.8
        fake_labels = torch.empty_like(y_fake).uniform_(0.0, 0.3) if noise is None else 0.3 * noise[1]
        loss_real = self.loss_fn(y_real, real_labels)                           # :177
        loss_fake = self.loss_fn(y_fake, fake_labels)                           # :178
        # :179-180  backward of 0.5 * loss_real + 0.5 * loss_fake: the weights ride in the two backward seeds (times the loss
        # scale, if any) -- no scalar-arithmetic kernels and no Mul / Add nodes in the iteration
        half = self._seeds(lr_images.device)[0]
        torch.autograd.backward([loss_real, loss_fake], [half, half])
        ops.wgrad_stream_join()
        joined = False
        if join_side and side is not None:      # a captured phase must end with every forked stream joined
            main.wait_stream(side)
            joined = True
        return dict(sr=sr_images, content=content_loss, loss_real=loss_real, loss_fake=loss_fake, main=main, side=side,
                    joined=joined)

    def _phase_g(self, st, noise):
        Dm = self.discriminator
        self.optim_generator.zero_grad()                                        # :184 (independent of the exchange in flight)
        self._sync_d.wait()             # no-op without a pending exchange (single process, graph capture)
        self.optim_discriminator.step()                                         # :181
        for p in Dm.parameters():
            p.requires_grad_(False)      # D's weight gradients of this pass are discarded by the reference (:171)
        try:
            y_fake = Dm(st["sr"])                                               # :186 (updated D)
            real_labels = torch.empty_like(y_fake).uniform_(0.7, 1.0) if noise is None else 0.3 * noise[2] + 0.7   # :187
            bce = self.loss_fn(y_fake, real_labels)
            adv_loss = bce.detach() * 1e-1                                      # :188 (the reported value)
            if st["side"] is not None and not st["joined"]:
                st["main"].wait_stream(st["side"])
            # :194-195  backward of 0.5 * (0.1 * bce) + 0.5 * content_loss, the weights in the seeds as above
            half, adv_w = self._seeds(y_fake.device)
            torch.autograd.backward([bce, st["content"]], [adv_w, half])
        finally:
            for p in Dm.parameters():
                p.requires_grad_(True)
        ops.wgrad_stream_join()
        st["adv"] = adv_loss

    def _seeds(self, dev):
        """(0.5 S, 0.05 S) as 0-dim device tensors, S = the loss scale (1 without one): the backward seeds that carry
        trainer.py:179 / :188 / :194's loss weights.  Constant scales: made once (before any graph capture: the warm-up
        iterations come first); the dynamic fp16 scale lives on the device, so its seeds are two tiny launches per phase."""
        if self._scale_state is not None:
            return self._seed * 0.5, self._seed * 0.05
        c = self._seed_consts.get(str(dev))
        if c is None:
            s = 1.0 if self._seed is None else float(self.loss_scale)
            c = self._seed_consts[str(dev)] = (torch.tensor(0.5 * s, dtype=torch.float32, device=dev),
                                               torch.tensor(0.05 * s, dtype=torch.float32, device=dev))
        return c

    def _clear_nonfinite_flag(self):
        if self._scale_state is not None:
            self._scale_state[2:3].zero_()

    def _update_loss_scale(self):
        if self._scale_state is not None:
            from . import _lib as L
            L.check(L.lib().fsr_loss_scale_update(ops._p(self._scale_state), self.loss_scale_growth_interval, self.loss_scale_growth, 0.5, ops._stream()),
                    "fsr_loss_scale_update")

    def loss_scale_state(self):
        """(current loss scale, skipped iterations) -- a host read; None when the scale is static."""
        if self._scale_state is None:
            return None
        v = self._scale_state.detach().cpu()
        return float(v[0]), int(v[3])

    def _phase_end(self, st):
        self._sync_g.wait()
        self.optim_generator.step()                                             # :196
        self._update_loss_scale()
        # the loss scalars live in the per-iteration scratch arena: copy them out (one launch) so they survive the next reset
        vals = torch.stack([st["loss_real"].detach(), st["loss_fake"].detach(), st["adv"].detach(), st["content"].detach()])
        return dict(zip(("loss_real", "loss_fake", "adv_loss", "content_loss"), vals.unbind(0)))

    # ------------------------------------------------------------------ hipGraph replay of the whole iteration
    def capture_train_step(self, lr_images, hr_images, warmup=2, noise=None):
        """Captures one full iteration (both optimizer steps included) into hipGraphs; `graphed_train_step` then replays it
        with new batch contents: ~600 kernel launches become one graph launch (single process) or three (data parallel:
        the two RCCL gradient exchanges stay eager launches between the phase graphs, which share one memory pool).
        Label noise comes from torch's graph-safe Philox generator, so every replay draws fresh numbers -- unless `noise`
        (three tensors) is given: then the replays read the static copies `graphed_train_step(..., noise=)` refreshes."""
        self._g_lr, self._g_hr = lr_images.clone(), hr_images.clone()
        self._g_noise = None if noise is None else [t.clone() for t in noise]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm every lazily created buffer / cache on a side stream
            for _ in range(warmup):
                self.train_step(self._g_lr, self._g_hr, self._g_noise)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # every filter image is re-packed (in place, ops.packed_filter) by a launch INSIDE the graph at its first use
        for opt in (self.optim_generator, self.optim_discriminator):
            opt.mark_updated()
        segmented = D.is_distributed()
        kw = dict(capture_error_mode="thread_local") if segmented else {}   # RCCL's watchdog thread polls events meanwhile
        dev = self._g_lr.device
        self._graphs = []
        try:
            if not segmented:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **kw):
                    st = self._phase_d(self._g_lr, self._g_hr, self._g_noise)
                    self._phase_g(st, self._g_noise)
                    self._graph_out = self._phase_end(st)
                self._graphs = [g]
            else:
                ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # the exchanges between the captures are REAL collectives: every rank issues exactly these two, whether or not its
                # own capture succeeds (a rank that raised inside a capture would otherwise leave its peers in an all-reduce)
                err = None
                try:
                    with torch.cuda.graph(ga, **kw):
                        st = self._phase_d(self._g_lr, self._g_hr, self._g_noise, join_side=True)
                except Exception as exc:  # noqa: BLE001
                    err = exc
                self._sync_d.run()
                if err is None:
                    try:
                        with torch.cuda.graph(gb, pool=ga.pool(), **kw):
                            self._phase_g(st, self._g_noise)
                    except Exception as exc:  # noqa: BLE001
                        err = exc
                self._sync_g.run()
                if err is None:
                    try:
                        with torch.cuda.graph(gc, pool=ga.pool(), **kw):
                            self._graph_out = self._phase_end(st)
                    except Exception as exc:  # noqa: BLE001
                        err = exc
                # all ranks replay graphs or none does: a rank replaying while another runs eager would pair different
                # numbers of collectives per step
                ok = D.all_ranks_ok(err is None, dev)
                if not ok:
                    self._graphs = []
                    raise RuntimeError("hipGraph capture failed on %s: %s" % ("this rank" if err is not None else "another rank", err))
                self._graphs = [ga, gb, gc]
        finally:
            self._end_iteration(dev)
        self._graph = self._graphs[0]
        return self._graph

    def graphed_train_step(self, lr_images, hr_images, noise=None):
        self._g_lr.copy_(lr_images, non_blocking=True)
        self._g_hr.copy_(hr_images, non_blocking=True)
        if noise is not None:
            if self._g_noise is None:
                raise RuntimeError("the step was captured without injected noise")
            for dst, src in zip(self._g_noise, noise):
                dst.copy_(src, non_blocking=True)
        if len(self._graphs) == 1:
            self._graphs[0].replay()
        else:
            self._graphs[0].replay()
            self._sync_d.start()        # the exchange is enqueued behind phase D on RCCL's stream ...
            self._sync_d.wait()         # ... and phase G's graph launch is ordered behind it (stream wait, no host block)
            self._graphs[1].replay()
            self._sync_g.start()
            self._sync_g.wait()
            self._graphs[2].replay()
        # the replay stepped both optimizers on the device without running any Python: advance the host-side epochs so
        # that eager code running afterwards (evaluation, checkpoint-time inference) re-packs the filters it caches
        for opt in (self.optim_generator, self.optim_discriminator):
            opt.mark_updated()
        return self._graph_out

    def pretrain_step(self, lr_images, hr_images):
        """trainer.py:107-111."""
        ops.zero_pool_reset(lr_images.device)
        try:
            self._clear_nonfinite_flag()
            self.optim_generator.zero_grad()
            fake_hr_images = self.generator(lr_images)
            gen_loss = self.l1_loss(fake_hr_images, hr_images)
            gen_loss.backward(self._seed)
            self._sync_g.run()
            self.optim_generator.step()
            self._update_loss_scale()
            return gen_loss.detach().clone()
        finally:
            ops.zero_pool_end(lr_images.device)

    # ------------------------------------------------------------------ validation metrics, trainer.py:46-69
    @torch.no_grad()
    def _calculate_metrics_over_dataset(self, dataloader, phase, step):
        """SSIM and PSNR over the loader as the reference's torchmetrics objects accumulate them (trainer.py:53-69):
        SSIM(reduction="none") keeps one value per image and logs their mean; PSNR(dim=None) accumulates the squared error
        and the element count of EVERYTHING it saw and logs one global 10 log10(1 / mse).  Both come from one HIP kernel
        per batch (ops.ssim_sse) reading the generator's output and the HR batch in place."""
        self.generator.eval()
        ssim_sum = sse = None
        images, elements = 0, 0
        for lr_images, hr_images in dataloader:
            lr_images = lr_images.to(self.config.training.device, non_blocking=True)
            hr_images = hr_images.to(self.config.training.device, non_blocking=True)
            sr_images = self.generator(lr_images)
            n, c, h, w = hr_images.shape
            r = ops.ssim_sse(sr_images, hr_images)
            part = torch.stack([r[:, 0].sum() / float(c * (h - 10) * (w - 10)), r[:, 1].sum()])
            ssim_sum, sse = (part[0], part[1]) if ssim_sum is None else (ssim_sum + part[0], sse + part[1])
            images += n
            elements += n * c * h * w
        if images:      # the only host<->device syncs of the evaluation
            self.writer.add_scalar(f"{phase}/SSIM", float(ssim_sum) / images, global_step=step)
            mse = float(sse) / elements
            self.writer.add_scalar(f"{phase}/PSNR", 10.0 * __import__("math").log10(1.0 / mse) if mse > 0 else float("inf"),
                                   global_step=step)
        self.writer.flush()

    @classmethod
    def _pre_train_setup(cls, dataloader):
        if cls.fixed_lr_images.ndim == 1:
            for fixed_lr_images, fixed_hr_images in dataloader:
                cls.fixed_lr_images = (fixed_lr_images + 1.0) / 2.0
                cls.fixed_hr_images = (fixed_hr_images + 1.0) / 2.0
                break

    def _log_fixed_images(self, phase):
        """trainer.py:71-79 (cold: once per phase; the bicubic yardstick is the reference's own CPU interpolate)."""
        if Trainer.fixed_lr_images.ndim == 1:
            return
        dev = self.config.training.device
        Trainer.fixed_hr_images = Trainer.fixed_hr_images.to(dev)
        Trainer.fixed_lr_images = Trainer.fixed_lr_images.to(dev)
        scale = Trainer.fixed_hr_images.shape[-1] // max(1, Trainer.fixed_lr_images.shape[-1])
        upsampled = torch.nn.functional.interpolate(Trainer.fixed_lr_images.cpu(), scale_factor=scale, mode="bicubic",
                                                    antialias=True).to(dev)
        self.writer.add_images(f"{phase}/HighRes", Trainer.fixed_hr_images, global_step=0)
        self.writer.add_images(f"{phase}/Bicubic", upsampled, global_step=0)

    @torch.no_grad()
    def _log_generated(self, tag, step):
        """trainer.py:121-127 / :221-230: the generator on the fixed LR batch."""
        if Trainer.fixed_lr_images.ndim == 1:
            return
        self.generator.eval()
        generated = (1.0 + self.generator(2.0 * Trainer.fixed_lr_images.to(self.config.training.device) - 1.0)) / 2.0
        self.writer.add_images(tag, generated, global_step=step)

    def pretrain(self, train_dataloader, val_dataloader):
        # As in the reference (trainer.py:90-94): only runs/pretrain.pt resumes.  The reference WRITES
        # runs/pretrain_generator.pt (:133), so its resume never triggers by itself; a user renames the file to opt in.
        # (Picking up pretrain_generator.pt automatically would make every second experiment in a directory silently skip
        # pre-training with the previous run's generator.)
        if osp.exists("runs/pretrain.pt"):
            print("Pretrained model found, skipping pretraining")
            ckpt = torch.load("runs/pretrain.pt", map_location="cpu")
            self.generator.load_state_dict(ckpt["model"])
            self.optim_generator.load_state_dict(ckpt["optimizer"])
            return
        self._calculate_metrics_over_dataset(val_dataloader, "Pretrain", step=0)
        self._pre_train_setup(val_dataloader)
        self._log_fixed_images("Pretrain")
        dev = self.config.training.device
        for step, (lr_images, hr_images) in enumerate(train_dataloader, start=1):
            lr_images, hr_images = lr_images.to(dev, non_blocking=True), hr_images.to(dev, non_blocking=True)
            gen_loss = self.pretrain_step(lr_images, hr_images)
            if step % self.config.training.log_iter == 0:
                self.writer.add_scalar("Pretrain/Generator/Loss", gen_loss, global_step=step)
            if step % self.config.training.checkpoint_iter == 0:
                self._log_generated("Pretrain/Generated", step)
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
        # trainer.py:160-162 tests `fixed_lr_images is None`, which never holds (the attribute starts as tensor([])), so a
        # run that skipped pre-training reaches :224 with an empty batch.  Fixed here: the fixed batch is drawn whenever it
        # is still unset.
        if Trainer.fixed_lr_images.ndim == 1:
            self._pre_train_setup(val_dataloader)
            self._log_fixed_images("GAN")
        self.generator.train()
        self.discriminator.train()
        dev = self.config.training.device
        # training.hip_graph (extension key, default true): the first two iterations run eagerly (they create every lazily
        # allocated buffer), the third is captured and from then on replayed -- the path bench.py times.  Shapes never
        # change: every loader of train.py drops the last partial batch.
        use_graph = bool(getattr(self.config.training, "hip_graph", True)) and str(dev).startswith("cuda") and torch.cuda.is_available()
        graph_shape = None
        for step, (lr_images, hr_images) in enumerate(train_dataloader, start=1):
            lr_images, hr_images = lr_images.to(dev, non_blocking=True), hr_images.to(dev, non_blocking=True)
            if use_graph and graph_shape is None and step == 3:     # steps 1-2 ran eagerly: every lazy buffer exists
                try:
                    self.capture_train_step(lr_images, hr_images, warmup=0)
                    graph_shape = (tuple(lr_images.shape), tuple(hr_images.shape))
                except Exception as exc:  # noqa: BLE001 -- eager launches are always available
                    print("Trainer.train: hipGraph capture failed (%s: %s); running eager" % (type(exc).__name__, exc))
                    torch.cuda.synchronize()
                    use_graph = False
            if graph_shape == (tuple(lr_images.shape), tuple(hr_images.shape)):
                losses = self.graphed_train_step(lr_images, hr_images)
            else:
                losses = self.train_step(lr_images, hr_images)
            if step % self.config.training.log_iter == 0:       # the only host<->device syncs of the loop
                self.writer.add_scalar("Loss/Discriminator/Real", losses["loss_real"], global_step=step)
                self.writer.add_scalar("Loss/Discriminator/Fake", losses["loss_fake"], global_step=step)
                self.writer.add_scalar("Loss/Generator/Adversarial", losses["adv_loss"], global_step=step)
                self.writer.add_scalar("Loss/Generator/Content", losses["content_loss"], global_step=step)
            if step % self.config.training.checkpoint_iter == 0:
                self._log_generated("GAN/Generated", step)
                self._calculate_metrics_over_dataset(val_dataloader, "GAN", step=step)
                self.save_checkpoints(step)
                self.generator.train()
