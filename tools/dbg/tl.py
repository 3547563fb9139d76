import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/tests", ROOT + "/oracle", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import crypto_amd
from oracle_c import OracleC
orc = OracleC()
eng = crypto_amd.Engine(0)
for n in (1 << 20, 1572864 + 13, 3 << 20):
    g = eng.random_bases(0x5EED0001, n)
    v = eng.fr_random(0x5EED0002, n)
    want = orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); r = eng.msm_dev(g, v.device_ptr(), n); ts.append(time.perf_counter() - t0)
    print(n, "ok" if r == want else "MISMATCH", ["%.3f" % (t * 1e3) for t in ts], flush=True)
    g.free(); v.free()
