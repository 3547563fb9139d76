#!/usr/bin/env python3
"""Times sharding._Exchange (pinned staging + RCCL all_gather of 64-byte partial points) with a
1-rank nccl group on one GPU: the only part of the N>1 path that a 1-GPU box can exercise."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from crypto_amd import sharding
ex = sharding._Exchange(1, dev)
p = bytes(range(64))
assert ex.run(p) == [p]
t0 = time.perf_counter()
for i in range(200):
    q = bytes([(i + j) & 255 for j in range(64)])
    assert ex.run(q) == [q]
print("exchange us/call", (time.perf_counter() - t0) / 200 * 1e6)
dist.destroy_process_group()
