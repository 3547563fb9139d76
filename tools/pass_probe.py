import sys, os, time, statistics
sys.path.insert(0, os.getcwd())
import crypto_amd
eng = crypto_amd.Engine(0)
eng.set_phase_timing(True)  # last_timing() below
N = (1 << 21) + 64
H = eng.random_bases(0xB9901, N, flags=crypto_amd.PRECOMPUTE)
v = eng.fr_random(0x21, N)
def t(fn, reps=6):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)
for n in (1 << 20, (1 << 20) + 9, (1 << 20) + 5000, (1 << 20) + (1 << 18), 1 << 21, (1 << 21) + 9):
    ms = t(lambda: eng.msm_dev(H, v.device_ptr(), n))
    print(n, "msm_dev %.3f ms" % ms, ["%.3f" % x for x in eng.last_timing()], flush=True)
    ms = t(lambda: eng.msm_fr_multi([(H, 0, 1, n, v, 0, 1)]))
    print(n, "fr_multi 1 seg %.3f ms" % ms, flush=True)
G = eng.random_bases(0xB9902, 1 << 20, flags=crypto_amd.PRECOMPUTE)
w = eng.fr_random(0x22, 1 << 20)
for nh in (1 << 20, (1 << 20) + 9):
    ms = t(lambda: eng.msm_fr_multi([(H, 0, 1, nh, v, 0, 1), (G, 0, 1, 1 << 20, w, 0, 1)]))
    print(nh, "two terms %.3f ms" % ms, ["%.3f" % x for x in eng.last_timing()], flush=True)
