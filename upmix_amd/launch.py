#!/usr/bin/env python3
"""
One process per GPU without torch.distributed.run: starts N copies of a module (or script) with the environment the
one-process-per-GPU entries read (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), waits for them, and ends
the others as soon as one fails.

    python -m upmix_amd.launch --nproc 8 -m upmix_amd.multi_gpu long.wav --export-mode stereo_sum
    python -m upmix_amd.launch --nproc 8 -m upmix_amd.batch in/*.wav
    python -m upmix_amd.launch --nproc 2 bench.py --gpus 2 --steps 20 --warmup 5

The ranks meet over upmix_amd.rendezvous on MASTER_PORT (nothing else listens there under this launcher).  Any other
launcher that exports the same five variables works as well (torch.distributed.run, mpirun + a wrapper, srun).
The reference has no counterpart: it is a single process (main.py).
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from typing import List


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(nproc: int, command: List[str], master_addr: str = "127.0.0.1", master_port: int = 0, env=None) -> int:
    """Start `command` nproc times (rank r gets RANK = LOCAL_RANK = r); -> 0, or the first non-zero exit code."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = master_port or free_port()
    procs = []
    for rank in range(nproc):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nproc), MASTER_ADDR=master_addr,
                 MASTER_PORT=str(port))
        e.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # no launcher-side store on MASTER_PORT here
        procs.append(subprocess.Popen(command, env=e))
    rc = 0
    try:
        alive = list(procs)
        while alive and rc == 0:
            time.sleep(0.05)
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0:
                    rc = code
    finally:
        for p in procs:                                   # a failed rank takes the others down (exact PIDs only)
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="upmix_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc", type=int, required=True, help="processes = GPUs")
    ap.add_argument("--master-addr", default="127.0.0.1")
    ap.add_argument("--master-port", type=int, default=0, help="0 = pick a free port")
    ap.add_argument("-m", dest="module", default=None, help="run a module (python -m MODULE ...)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="script and / or its arguments")
    a = ap.parse_args(argv)
    rest = a.rest[1:] if a.rest[:1] == ["--"] else a.rest
    if a.module:
        command = [sys.executable, "-m", a.module] + rest
    elif rest:
        command = [sys.executable] + rest
    else:
        ap.error("nothing to launch: give -m MODULE or a script")
    return run(a.nproc, command, a.master_addr, a.master_port)


if __name__ == "__main__":
    sys.exit(main())
