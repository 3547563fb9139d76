import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch, crypto_amd
eng = crypto_amd.Engine(0)
eng.set_phase_timing(True)  # last_timing() below
n = 1 << 20
b = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
s = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, n, s.data_ptr())
def run(tag, gap):
    acc, wall = [], []
    for _ in range(60):
        t0 = time.perf_counter()
        eng.msm_dev(b, s.data_ptr(), n)
        wall.append((time.perf_counter() - t0) * 1e3)
        acc.append(eng.last_timing()[3])
        if gap: time.sleep(gap)
    acc, wall = acc[10:], wall[10:]
    print(tag, "acc mean %.3f median %.3f min %.3f max %.3f | wall median %.3f" % (statistics.mean(acc), statistics.median(acc), min(acc), max(acc), statistics.median(wall)), flush=True)
run("back-to-back", 0)
run("gap 0.2ms   ", 0.0002)
run("gap 2ms     ", 0.002)
run("back-to-back", 0)
# after an in-flight burst
for _ in range(3):
    t = [eng.msm_submit(b, s.data_ptr(), n) for _ in range(3)]
    for k in range(40):
        eng.msm_wait(t[k % 3]); t[k % 3] = eng.msm_submit(b, s.data_ptr(), n)
    for x in t: eng.msm_wait(x)
    run("after burst ", 0)
