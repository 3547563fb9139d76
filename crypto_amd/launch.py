"""One process per GPU without an outer launcher: start N fresh ranks of a script, relay rank 0's
stdout, propagate failure.  The parent makes no GPU call (it never imports torch), and no rank is a
re-exec of a process that has touched a GPU: every rank is a new child started with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its environment, the same contract
`python -m torch.distributed.run --nproc-per-node N` fulfils."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script: str, argv: list[str], nproc: int, env_extra: dict | None = None,
                 timeout_s: float | None = None, json_only: bool = True) -> int:
    """Runs `python script argv...` as ranks 0..nproc-1 of one job on this node.  Rank 0's stdout is
    relayed to this process's stdout when all ranks are done; the other ranks' stdout goes to stderr.
    With json_only, only rank 0's lines that start with "{" reach stdout (the result line); whatever
    else a library prints there (gloo / RCCL connection notices) goes to stderr.
    Returns 0 when every rank exited 0; otherwise the first non-zero exit code (the remaining ranks
    are terminated -- by the exact PIDs started here -- so a failed rank cannot leave its peers hung
    in a collective)."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = free_port()
    procs = []
    for rank in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc),
                    "LOCAL_WORLD_SIZE": str(nproc), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    t0 = time.monotonic()
    rc = 0
    out0 = b""
    try:
        # rank 0's pipe is drained while waiting so that a long line cannot block it
        import threading
        buf = []
        th = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        th.start()
        pending = set(range(nproc))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = code
                    print(f"[launch] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                    break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s and pending:
                rc = 124
                print(f"[launch] ranks {sorted(pending)} still running after {timeout_s} s", file=sys.stderr, flush=True)
            if pending and rc == 0:
                time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    th.join(timeout=10)
    out0 = buf[0] if buf else b""
    for ln in out0.splitlines(keepends=True):
        dst = sys.stdout if (not json_only or ln.lstrip().startswith(b"{")) else sys.stderr
        dst.buffer.write(ln)
    sys.stdout.flush()
    sys.stderr.flush()
    return rc
