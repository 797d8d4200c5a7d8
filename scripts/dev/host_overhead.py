import time, torch, sys, os
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo/wild_deep_mvs_amd") else sys.path.insert(0, os.getcwd())
from wild_deep_mvs_amd import _lib as L, ops
dt = torch.float16
x = torch.randn(1, 8, 8, 16, 8, device="cuda").to(dt)
w = torch.randn(8, 8, 3, 3, 3) / 15
layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=True, dtype=dt)
out = torch.empty(1, 8, 8, 16, 8, device="cuda", dtype=dt)
for _ in range(100): ops.conv3d(x, layer, out=out)
torch.cuda.synchronize()
N = 5000
t0 = time.perf_counter()
for _ in range(N): ops.conv3d(x, layer, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"ops.conv3d host time per call {(t1 - t0) / N * 1e6:.2f} us (wall incl. GPU drain {(t2 - t0) / N * 1e6:.2f})")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ops.conv3d(x, layer, out=out)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
