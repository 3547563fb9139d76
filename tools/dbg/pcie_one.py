import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, crypto_amd
eng = crypto_amd.Engine(0)
n = 1 << 20
h = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
d = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, n, d.data_ptr())
pinned = d.cpu().pin_memory()
for _ in range(6):
    t0 = time.perf_counter(); eng.msm_host_ptr(h, pinned.data_ptr(), n); print((time.perf_counter() - t0) * 1e3)
