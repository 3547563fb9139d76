import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
V = sys.argv[1]
if "dist" in V:
    import torch.distributed as dist
if "setdev" in V:
    torch.cuda.set_device(0)
import crypto_amd
eng = crypto_amd.Engine(0)
eng.set_phase_timing(True)  # last_timing() below
n = 1 << 20
if "basesfirst" in V:
    bases = eng.random_bases(0x5EED0001, n, flags=1)
scal = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, n, scal.data_ptr())
if "basesfirst" not in V:
    bases = eng.random_bases(0x5EED0001, n, flags=1)
if "zeros" in V:
    part = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
acc = []
for i in range(12):
    eng.msm_dev(bases, scal.data_ptr(), n)
    acc.append(eng.last_timing()[3])
    if "sleep" in V:
        time.sleep(0.002)
print(V, " ".join(f"{a:.3f}" for a in acc))
