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
import secrets
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


class _Terminated(Exception):
    """SIGTERM / SIGHUP reached the launcher: raised into the wait loop so that its `finally` ends the ranks."""

    def __init__(self, signum):
        super().__init__(signum)
        self.signum = signum


def exit_code(code: int) -> int:
    """A child killed by signal N has return code -N; the launcher exits 128 + N, like a shell."""
    return 128 - code if code < 0 else code


def run(nproc: int, command: List[str], master_addr: str = "127.0.0.1", master_port: int = 0, env=None,
        capture_rank0: bool = False):
    """
    Start `command` nproc times (rank r gets RANK = LOCAL_RANK = r); -> 0, or the first non-zero exit code (128 + N for a
    rank killed by signal N).  A SIGTERM or SIGHUP to the launcher (scheduler pre-emption, `timeout`) ends the ranks too.
    Every job gets its own UPX_RDZV_TOKEN (rendezvous.job_token).  `capture_rank0`: -> (code, rank 0's stdout as text).
    """
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = master_port or free_port()
    token = secrets.token_hex(8)
    procs = []
    rc = 0
    out0 = ""
    old = {}

    def on_signal(signum, _frame):
        raise _Terminated(signum)

    try:
        for sig in (signal.SIGTERM, signal.SIGHUP):
            try:
                old[sig] = signal.signal(sig, on_signal)
            except ValueError:                            # not the main thread: the caller keeps its own handlers
                pass
        for rank in range(nproc):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nproc), MASTER_ADDR=master_addr,
                     MASTER_PORT=str(port), UPX_RDZV_TOKEN=token)
            e.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # no launcher-side store on MASTER_PORT here
            procs.append(subprocess.Popen(command, env=e,
                                          stdout=subprocess.PIPE if capture_rank0 and rank == 0 else None, text=True))
        reader = None
        if capture_rank0:
            import threading
            chunks = []
            reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
            reader.start()                                # (the wait loop below keeps watching the other ranks)
        alive = list(procs)
        while alive and rc == 0:
            time.sleep(0.05)
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0:
                    rc = exit_code(code)
        if reader is not None and rc == 0:
            reader.join()
            out0 = "".join(chunks)
    except _Terminated as t:
        rc = 128 + t.signum
    finally:
        for sig, handler in old.items():
            signal.signal(sig, handler)
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
    return (rc, out0) if capture_rank0 else rc


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
