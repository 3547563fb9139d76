"""One rank of the multi-rank GPU tests (tests/test_gpu_multirank.py), started by crypto_amd.launch:
its slice of the synthetic workload through the HIP library (bn256_msm_submit / bn256_msm_wait),
partial sums exchanged with sharding.AsyncExchange, host sum -- the loop bench.py times.
  --backend gloo --same-device   every rank on cuda:0, exchange over gloo (one-GPU box)
  --backend nccl                 rank r on cuda:r, exchange over RCCL (needs >= 2 GPUs)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--same-device", action="store_true")
ap.add_argument("--n-total", type=int, default=(1 << 17) + 1)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--precompute", type=int, default=1)
args = ap.parse_args()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = 0 if args.same_device else int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if args.backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)

import crypto_amd  # noqa: E402
from crypto_amd import sharding  # noqa: E402

eng = crypto_amd.Engine(local)
off, n = sharding.split_even(args.n_total, world, rank)
bases = eng.random_bases(0x5EED0001, n, first=off, flags=crypto_amd.PRECOMPUTE if args.precompute else 0)
v = eng.fr_random(0x5EED0002, n, first=off)
ex = sharding.AsyncExchange(world, dev if args.backend == "nccl" else None, batch=args.depth)  # as bench.py does
results = []
for depth in (args.depth, 1):
    res = sharding.run_pipelined(args.steps, depth, lambda: eng.msm_submit(bases, v.device_ptr(), n), eng.msm_wait, ex)
    results.append(res.hex())
part = eng.msm_dev(bases, v.device_ptr(), n)
parts = sharding.all_gather_points(part, dev if args.backend == "nccl" else None)
results.append(sharding.combine(parts).hex())
allr = [None] * world
dist.all_gather_object(allr, results)
if rank == 0:
    print(json.dumps({"world": world, "slices": [sharding.split_even(args.n_total, world, r) for r in range(world)],
                      "results_per_rank": allr, "partials": [p.hex() for p in parts]}), flush=True)
bases.free()
v.free()
eng.close()
dist.destroy_process_group()
