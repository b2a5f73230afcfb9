import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import importlib, torch, torch.nn.functional as F
L = importlib.import_module("fast-srgan_amd._lib"); ops = importlib.import_module("fast-srgan_amd.ops")
dev = torch.device("cuda:0"); cd = ops.Compute("f16"); torch.manual_seed(0)
for (n, h, w, cout, ps) in ((4, 96, 96, 64, False), (3, 50, 70, 128, False), (2, 48, 64, 256, True)):
    x = torch.randn(n, h, w, 64, device=dev).half(); wt = (torch.randn(cout, 64, 3, 3, device=dev) * 0.05).half().float(); b = torch.randn(cout, device=dev) * 0.1
    wpk = ops.packed_filter(cd, wt, L.PACK_FWD_PS if ps else L.PACK_FWD, 64)
    y, _, st = ops.conv3x3_raw(cd, x, wpk, cout, bias=b, act=L.ACT_RELU if not ps else L.ACT_NONE, pixel_shuffle=ps, want_stats=(cout == 64))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt, b, 1, 1)
    pre = ref
    ref = F.pixel_shuffle(ref, 2) if ps else F.relu(ref)
    got = y.float().permute(0, 3, 1, 2)
    e = float((got - ref).abs().max() / ref.abs().max())
    msg = "%s n%d %dx%d cout %d: max rel err %.2e" % (L.lib().fsr_last_kernel().decode(), n, h, w, cout, e)
    if st is not None: msg += "  stats err %.2e" % float((st[..., 0] - pre.sum((2, 3))).abs().max() / pre.sum((2, 3)).abs().max())
    print(msg)
