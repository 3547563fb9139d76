import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch, crypto_amd
eng = crypto_amd.Engine(0)
for lg in (20, 24):
    n = 1 << lg
    b = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    s = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    eng.random_scalars_into(0x5EED0002, n, s.data_ptr())
    for _ in range(3): eng.msm_dev(b, s.data_ptr(), n)
    ts = []
    for _ in range(40 if lg == 20 else 6):
        t0 = time.perf_counter(); eng.msm_dev(b, s.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
    print("timing env", os.environ.get("BN256_PHASE_TIMING"), "2^%d" % lg, "blocking median %.4f ms" % statistics.median(ts), flush=True)
    if lg == 20:
        # three in flight
        t = [eng.msm_submit(b, s.data_ptr(), n) for _ in range(3)]
        t0 = time.perf_counter()
        for k in range(90):
            eng.msm_wait(t[k % 3]); t[k % 3] = eng.msm_submit(b, s.data_ptr(), n)
        dt = (time.perf_counter() - t0) / 90 * 1e3
        for x in t: eng.msm_wait(x)
        print("   three in flight %.4f ms per step" % dt, flush=True)
    b.free()
