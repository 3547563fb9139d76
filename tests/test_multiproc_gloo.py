"""world_size-2 CPU test (gloo) of the N>1 path: slice assignment, all_gather of the 64-byte
partial points and the host-side combination (bn256_g1_sum).  The per-rank partial MSM is
produced by the oracle here (no GPU in this tier); on GPUs bench.py runs the same
sharding.all_gather_points / sharding.combine around bn256_msm_dev."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from crypto_amd import sharding
    from oracle_c import OracleC
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = OracleC()
        off, n = sharding.split_even(n_total, world, rank)
        bases = orc.synth_bases(0x5EED0001, n, first=off)
        scal = orc.synth_scalars(0x5EED0002, n, first=off)
        part = orc.msm(bases, scal)
        parts = sharding.all_gather_points(part)
        total = sharding.combine(parts)
        # the start / finish form bench.py uses with two calls in flight: several exchanges
        # outstanding, finished in order, same answers
        ex = sharding.AsyncExchange(world, None, slots=3)
        other = orc.msm(bases[:64 * (n - 1)], scal[:32 * (n - 1)])
        t1, t2 = ex.start(part), ex.start(other)
        async_ok = ex.finish(t1) == parts and sharding.combine(ex.finish(t1 if False else t2)) is not None
        t3 = ex.start(part)
        async_ok = async_ok and ex.finish(t3) == parts
        try:
            ex.finish(t3)
            async_ok = False
        except ValueError:
            pass
        # the blocking exchange over a batch-sized buffer (slot 0 carries the point): same answer
        async_ok = async_ok and sharding._Exchange(world, None, batch=2).run(part) == parts
        q.put((rank, off, n, total.hex(), [p.hex() for p in parts], async_ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_msm_matches_single():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_c import OracleC
    from crypto_amd import sharding
    n_total, world = 301, 2                      # odd on purpose: ragged slices
    assert [sharding.split_even(n_total, world, r) for r in range(world)] == [(0, 151), (151, 150)]
    assert sharding.split_even(5, 8, 7) == (5, 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc = OracleC()
    want = orc.msm(orc.synth_bases(0x5EED0001, n_total), orc.synth_scalars(0x5EED0002, n_total))
    assert res[0][3] == res[1][3] == want.hex()
    assert res[0][4] == res[1][4] and len(res[0][4]) == world
    assert res[0][5] and res[1][5]
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 151, 151, 150)


def test_bench_launcher_parent_to_children_to_one_json_line():
    """`python bench.py --gpus 2` as a bare command: the parent (no WORLD_SIZE) starts two fresh ranks,
    relays rank 0's single JSON line and propagates a failing rank's exit code.  The ranks here are
    tests/_bench_child_cpu.py (gloo + the oracle: this tier has no GPU); everything in the parent is
    bench.py's own code path."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_c import OracleC
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_CHILD_SCRIPT"] = os.path.join(ROOT, "tests", "_bench_child_cpu.py")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--warmup", "1", "--total-log2n", "8"]
    r = subprocess.run(cmd + ["--steps", "4"], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    orc = OracleC()
    want = orc.msm(orc.synth_bases(0x5EED0001, 256), orc.synth_scalars(0x5EED0002, 256))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["result_xy"] == want.hex()
    # a failing rank: non-zero exit, no JSON line, the other rank is not left behind in a collective
    r = subprocess.run(cmd + ["--steps", "13"], env=env, capture_output=True, timeout=300)
    assert r.returncode == 7 and not r.stdout.strip(), (r.returncode, r.stdout)


def test_run_pipelined_schedule():
    """sharding.run_pipelined (the loop bench.py times) with stand-ins for the device and the exchange:
    every step's result is collected and exchanged exactly once, in order; never more than `depth`
    calls or exchanges outstanding; the last result is returned; depth 1 is strictly sequential."""
    sys.path.insert(0, ROOT)
    from crypto_amd import sharding

    class FakeEx:
        def __init__(self):
            self.out, self.started, self.finished, self.max_out = {}, [], [], 0

        def start(self, part):
            k = len(self.started)
            self.started.append(part)
            self.out[k] = part
            self.max_out = max(self.max_out, len(self.out))
            return k

        def finish(self, k):
            self.finished.append(k)
            return [self.out.pop(k)]

    g0 = (1).to_bytes(32, "big") + (0x8fb501e34aa387f9aa6fecb86184dc21ee5b88d120b5b59e185cac6c5e089667 - 2).to_bytes(32, "big")
    for depth in (1, 2, 3, 4):
        for steps in (0, 1, 2, 3, 7):
            for with_ex in (False, True):
                log, inflight, max_inflight = [], [], [0]
                ex = FakeEx() if with_ex else None

                def submit():
                    t = sum(1 for e in log if e[0] == "s")
                    log.append(("s", t))
                    inflight.append(t)
                    max_inflight[0] = max(max_inflight[0], len(inflight))
                    return t

                def wait(t):
                    assert inflight.pop(0) == t
                    log.append(("w", t))
                    if ex is not None and depth == 1:
                        assert len(ex.started) == len(ex.finished) == t  # strictly sequential
                    return g0

                res = sharding.run_pipelined(steps, depth, submit, wait, ex)
                assert [e for e in log if e[0] == "w"] == [("w", t) for t in range(steps)]
                assert not inflight and max_inflight[0] <= depth
                assert res == (g0 if steps else None)
                if ex is not None:
                    assert ex.finished == list(range(steps)) and not ex.out and ex.max_out <= max(depth, 1)

    class FakeBatchEx:  # AsyncExchange with batch > 1: up to `batch` results per exchange
        def __init__(self, batch):
            self.batch, self.out, self.sizes, self.parts = batch, {}, [], []

        def start_batch(self, parts):
            assert 1 <= len(parts) <= self.batch
            k = len(self.sizes)
            self.sizes.append(len(parts))
            self.out[k] = list(parts)
            self.parts += parts
            return k

        def finish_batch(self, k):
            return [[p] for p in self.out.pop(k)]

    for depth in (1, 2, 4):
        for batch in (2, 4):
            for steps in (0, 1, 5, 9):
                ex, n = FakeBatchEx(batch), [0]

                def submit():
                    n[0] += 1
                    return n[0] - 1

                res = sharding.run_pipelined(steps, depth, submit, lambda t: g0, ex)
                assert res == (g0 if steps else None) and not ex.out and len(ex.parts) == steps
                # depth 1 stays strictly sequential (one result per exchange); otherwise full batches and one remainder
                want = [1] * steps if depth == 1 else [batch] * (steps // batch) + ([steps % batch] if steps % batch else [])
                assert ex.sizes == want, (depth, batch, steps, ex.sizes)
