"""Stand-in rank for the CPU test of bench.py's launcher path (BENCH_CHILD_SCRIPT): the same argv and
environment contract as a bench.py rank, gloo instead of RCCL, and the oracle in place of the device
MSM (this tier has no GPU).  Rank 0 prints one JSON line; `--steps 13` makes rank 1 fail on purpose."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402  (argument parsing is bench.py's own)

args = bench.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == args.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1"
if args.steps == 13 and rank == 1:
    sys.exit(7)
import torch.distributed as dist  # noqa: E402
from crypto_amd import sharding  # noqa: E402
from oracle_c import OracleC  # noqa: E402

dist.init_process_group("gloo", rank=rank, world_size=world)
orc = OracleC()
n_total = 1 << args.total_log2n
off, n = sharding.split_even(n_total, world, rank)
part = orc.msm(orc.synth_bases(bench.SEED_BASES, n, first=off), orc.synth_scalars(bench.SEED_SCALARS, n, first=off))
ex = sharding.AsyncExchange(world, None, batch=args.depth)  # as bench.py does
steps = iter(range(args.steps))
res = sharding.run_pipelined(args.steps, args.depth, lambda: next(steps), lambda t: part, ex)
print(f"rank {rank} done", file=sys.stderr if rank == 0 else sys.stdout)  # non-zero ranks' stdout must not reach the parent's
if rank == 0:
    print(json.dumps({"n_gpus": world, "scaling": "strong", "points_total": n_total, "result_xy": res.hex()}), flush=True)
dist.destroy_process_group()
