"""Child of tests/test_gpu_errors.py: started with BN256_FAIL_ALLOC_ABOVE in the environment (the
library reads it once per process), walks the out-of-memory paths and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import crypto_amd  # noqa: E402
from crypto_amd.engine import BN256Error  # noqa: E402
from oracle_c import OracleC  # noqa: E402

out = {}
if os.environ.get("ALLOC_CHILD_MODE") == "create":
    try:
        crypto_amd.Engine(0)
        out["create"] = 0
    except BN256Error as e:
        out["create"] = e.code
    print(json.dumps(out))
    sys.exit(0)

orc = OracleC()
eng = crypto_amd.Engine(0)
n = 4096
braw, sraw = orc.synth_bases(1, n), orc.synth_scalars(2, n)
want = orc.msm(braw, sraw, mt=True)
h = eng.upload(braw)
out["small_before"] = eng.msm(h, sraw) == want


def code_of(fn):
    try:
        fn()
        return 0
    except BN256Error as e:
        return e.code


# a handle larger than the limit (2^20 points with a table = 1 GiB)
out["big_handle"] = code_of(lambda: eng.random_bases(3, 1 << 20, flags=crypto_amd.PRECOMPUTE))
# an Fr vector larger than the limit
out["big_frvec"] = code_of(lambda: eng.fr_random(4, 1 << 23))
# a workspace larger than the limit in the middle of an MSM's reservations: 2^18 table-less points are a
# 16 MiB handle (fits) and need 72 MiB of bucket sums (does not)
hb = eng.random_bases(5, 1 << 18)
vb = eng.fr_random(6, 1 << 18)
out["big_workspace_blocking"] = code_of(lambda: eng.msm_dev(hb, vb.device_ptr(), 1 << 18))
# a submitted call reports it at the wait (the lane's thread reserves the workspaces), and frees its lane
out["big_workspace_submit"] = [code_of(lambda: eng.msm_wait(eng.msm_submit(hb, vb.device_ptr(), 1 << 18))) for _ in range(5)]
# the context, its lanes and its pool are as usable as before
out["small_after"] = eng.msm(h, sraw) == want
v = eng.fr_upload(sraw)
t = [eng.msm_submit(h, v.device_ptr(), n) for _ in range(3)]
out["lanes_after"] = all(eng.msm_wait(x) == want for x in t)
m = 1 << 16
out["mid_after"] = eng.msm_dev(hb, vb.device_ptr(), m) == orc.synth_msm_closed_form(5, 6, m)
print(json.dumps(out))
