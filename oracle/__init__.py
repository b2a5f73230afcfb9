"""TEST INFRASTRUCTURE -- the parity oracle.  Not part of the product path.

`oracle/` holds a CPU (PyTorch fp32, single code path) restatement of the Fast-SRGAN hot path
(/root/reference/model.py, trainer.py:171-196, dataloader.py:24-38, inference.py:47-57).  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
only as the checker.  Nothing under `fast-srgan_amd/` imports this package.

Pinning: the reference ships no tests and no golden vectors.  The oracle is
pinned against the reference's OWN modules, imported from /root/reference by
`tests/golden/make_golden.py` in the authoring container; the outputs are committed under
`tests/golden/` and `tests/test_oracle.py` replays them.  Two boundaries stay unpinned because the
third-party pieces are absent offline: torchvision's ImageNet VGG19 weights
(a structural stand-in with seeded kaiming-normal weights is used) and torchvision.transforms.v2.Resize
(restated as torch.nn.functional.interpolate(bicubic, antialias=True), the torch kernel it forwards to).
"""
