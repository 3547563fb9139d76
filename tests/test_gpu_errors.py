"""Error paths of the C ABI: bad devices, out-of-memory in handles / workspaces / context creation
(forced with BN256_FAIL_ALLOC_ABOVE in a child process), tickets lost by the caller (bn256_ctx_drain)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

import crypto_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ctx_create_rejects_bad_devices():
    from crypto_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p(1)
    for dev in (99, -1, 1 << 20):
        assert lib.bn256_ctx_create(dev, C.byref(ctx)) == -1 and not ctx.value   # BN256_EINVAL, *out cleared
    assert lib.bn256_ctx_create(0, None) == -1
    assert lib.bn256_ctx_create_multi(None, 1, C.byref(ctx)) == -1
    arr = (C.c_int * 9)(*([0] * 9))
    assert lib.bn256_ctx_create_multi(arr, 9, C.byref(ctx)) == -1 and not ctx.value
    assert lib.bn256_ctx_create_multi(arr, 0, C.byref(ctx)) == -1
    # create / destroy in a loop: a leak of streams, events or pinned buffers would show as a failure here
    for _ in range(50):
        assert lib.bn256_ctx_create(0, C.byref(ctx)) == 0
        lib.bn256_ctx_destroy(ctx)


def _child(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_alloc_failure_child.py")], env=env,
                       capture_output=True, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def test_out_of_memory_paths_leave_the_context_usable():
    out = _child({"BN256_FAIL_ALLOC_ABOVE": str(48 << 20)})
    assert out == {"small_before": True, "big_handle": -6, "big_frvec": -6, "big_workspace_blocking": -6,
                   "big_workspace_submit": [-6] * 5, "small_after": True, "lanes_after": True, "mid_after": True}


def test_ctx_create_out_of_memory_returns_enomem():
    assert _child({"BN256_FAIL_ALLOC_ABOVE": "64", "ALLOC_CHILD_MODE": "create"}) == {"create": -6}


def test_drain_voids_lost_tickets(eng, orc):
    from crypto_amd.engine import BN256Error
    n = 5000
    h = eng.random_bases(0xD1, n)
    v = eng.fr_random(0xD2, n)
    want = orc.synth_msm_closed_form(0xD1, 0xD2, n)
    try:
        for _ in range(crypto_amd.ASYNC_LANES):  # a caller that loses its tickets
            eng.msm_submit(h, v.device_ptr(), n)
        with pytest.raises(BN256Error) as ei:
            eng.msm_submit(h, v.device_ptr(), n)
        assert ei.value.code == -8               # BN256_EBUSY
        eng.drain()
        with pytest.raises(BN256Error):
            eng.msm_wait(0)                      # the ticket is void
        t = [eng.msm_submit(h, v.device_ptr(), n) for _ in range(crypto_amd.ASYNC_LANES)]
        assert [eng.msm_wait(x) for x in t] == [want] * crypto_amd.ASYNC_LANES
        # a lost fold ticket: the handle stays valid and its contents are complete after the drain
        f, tk = eng.fold2_submit(h, 0, 16, h, 16, 16, 3, 5)
        eng.drain()
        with pytest.raises(BN256Error):
            eng.fold_wait(tk)
        g = eng.fold(h, 0, h, 16, 16, 3, 5)
        assert f.download() == g.download()
        eng.drain()                              # nothing outstanding: a no-op
    finally:
        h.free()
        v.free()


def test_first_calls_of_fresh_lanes(orc):
    """A context's lanes are made at their first use, and whatever their set-up leaves in flight must be done
    before the lane's first pass runs (the one-time clearing of a new context's counter block once ran on the
    null stream, which a lane's non-blocking stream does not wait for: the first call of a fresh lane then
    lost its group totals two runs in three).  New contexts, all four lanes submitted at once -- three of them
    are made while the first one's call is running --, every result against the closed form."""
    n = (1 << 19) + 77  # long enough that lanes 1..3 are made while lane 0's call keeps the chip busy
    seed_b, seed_s = 0xF1E5, 0xF1E6
    for rep in range(4):
        e = crypto_amd.Engine(0)
        try:
            h = e.random_bases(seed_b + rep, n, flags=crypto_amd.PRECOMPUTE)
            v = e.fr_random(seed_s + rep, n)
            ks = orc.synth_base_scalars(seed_b + rep, n)
            want = orc.g0_mul(orc.fr_dot(b"".join(x.to_bytes(32, "big") for x in v.download()), ks))
            tickets = [e.msm_submit(h, v.device_ptr(), n) for _ in range(crypto_amd.ASYNC_LANES)]
            assert [e.msm_wait(t) for t in tickets] == [want] * crypto_amd.ASYNC_LANES, f"context {rep}"
            assert e.msm_dev(h, v.device_ptr(), n) == want
            h.free()
            v.free()
        finally:
            e.close()
