"""
Process-group plumbing of the one-process-per-GPU entries (multi_gpu, batch, bench.py) on the standard library alone.

The data path needs exactly one collective, the RCCL all-reduce of the overlap-add seam; it lives in
libupmix_hip.so.  What is left for the host side is tiny: hand rank 0's 128-byte RCCL id to the other ranks, a barrier,
the maximum of a few doubles (the global peak of main.py:85-97, the wall time of a benchmark) and "did every rank get
this far".  A star over TCP does that: rank 0 listens, ranks 1..N-1 connect, every operation is one message up and
one message down.  No torch: a process that runs the kernels maps ONE HIP runtime, the one libupmix_hip.so links.

Address: the launcher's MASTER_ADDR / MASTER_PORT (torch.distributed.run, or any launcher that exports RANK,
WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  torch.distributed.run keeps its own store on MASTER_PORT (it says so with
TORCHELASTIC_USE_AGENT_STORE=True), so the star then starts at MASTER_PORT + 1.  Should that port be taken, rank 0
listens on the next free one of the following seven and the other ranks find it by trying the same eight in turn (the
hello carries a magic word, the rank, the world size and a hash of the job token - UPX_RDZV_TOKEN, which upmix_amd.launch
generates per job, else TORCHELASTIC_RUN_ID - so a foreign service, and a rank of ANOTHER job of the same size on the
same port, is recognised and skipped; an accepted connection has 5 s to say hello, so a silent one - a port probe -
cannot hold up the accept loop).  UPX_RDZV_PORT pins one port.
The reference has no counterpart: its only parallelism is a thread pool (center_extraction.py:499-501).
"""
from __future__ import annotations

import hashlib
import math
import os
import socket
import struct
import time
from typing import List, Optional, Sequence

_MAGIC = b"UPXRDZV2"
_OK, _FAIL = 0, 1
_HELLO_TIMEOUT = 5.0      # an accepted connection that says nothing for this long is dropped (the ranks retry)


def job_token(env=os.environ) -> bytes:
    """8 bytes that identify the job: two jobs of the same world size on one MASTER_PORT must not capture each other's ranks."""
    tok = env.get("UPX_RDZV_TOKEN") or env.get("TORCHELASTIC_RUN_ID") or ""
    return hashlib.sha256(tok.encode("utf-8", "replace")).digest()[:8]


class RendezvousError(RuntimeError):
    """The process group could not be formed, a peer went away, or a peer reported a failure."""


def _send_frame(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks, got = [], 0
    while got < n:
        b = sock.recv(n - got)
        if not b:
            raise RendezvousError("a peer closed its connection")
        chunks.append(b)
        got += len(b)
    return b"".join(chunks)


def _recv_frame(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


def default_port(env=os.environ) -> int:
    """The star's port: UPX_RDZV_PORT, else MASTER_PORT (+ 1 when the launcher's own store occupies MASTER_PORT)."""
    if env.get("UPX_RDZV_PORT"):
        return int(env["UPX_RDZV_PORT"])
    port = int(env.get("MASTER_PORT", "29500"))
    return port + 1 if env.get("TORCHELASTIC_USE_AGENT_STORE") == "True" else port


class Rendezvous:
    """
    rank / world plus four operations, all collective (every rank calls them in the same order):
    ``allgather_bytes``, ``broadcast_bytes``, ``barrier``, ``allreduce_max`` and ``all_ok``.
    world == 1 needs no socket.
    """

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29500, timeout: float = 600.0,
                 n_ports: int = 1, token: Optional[bytes] = None):
        """`port` .. `port + n_ports - 1`: rank 0 listens on the first one it can bind, the others try them in turn.
        `token`: 8 bytes shared by the ranks of one job (default: job_token() of the environment)."""
        token = (job_token() if token is None else bytes(token)[:8]).ljust(8, b"\0")
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"rank {rank} of {world}")
        self.rank, self.world = int(rank), int(world)
        self._peers: List[Optional[socket.socket]] = []   # rank 0: socket of rank i at [i]
        self._up: Optional[socket.socket] = None          # other ranks: socket to rank 0
        self._listener: Optional[socket.socket] = None
        if world == 1:
            return
        deadline = time.monotonic() + timeout
        if rank == 0:
            ls, last = None, None
            for cand in range(port, port + max(1, n_ports)):
                ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    ls.bind(("127.0.0.1" if addr in ("127.0.0.1", "localhost", "") else "", cand))
                    break
                except OSError as exc:
                    ls.close()
                    ls, last = None, exc
            if ls is None:
                raise RendezvousError(f"rank 0 cannot listen on {addr}:{port}..{port + max(1, n_ports) - 1} ({last}); "
                                      "set UPX_RDZV_PORT") from last
            ls.listen(world)
            self._listener = ls
            self._peers = [None] * world
            missing = world - 1
            while missing:
                ls.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = ls.accept()
                except socket.timeout as exc:
                    self.close()
                    raise RendezvousError(f"{missing} of {world - 1} ranks did not connect within {timeout:g} s") from exc
                conn.settimeout(min(timeout, _HELLO_TIMEOUT))
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    hello = _recv_exact(conn, len(_MAGIC) + 16)
                    magic, r, w = hello[:len(_MAGIC)], *struct.unpack("<ii", hello[len(_MAGIC):len(_MAGIC) + 8])
                    theirs = hello[len(_MAGIC) + 8:]
                except (RendezvousError, OSError, struct.error):
                    conn.close()                    # silent or foreign: the accept loop moves on
                    continue
                if magic != _MAGIC or w != world or theirs != token or not 0 < r < world or self._peers[r] is not None:
                    conn.close()                    # not one of ours (or a rank of another job: it keeps trying its own ports)
                    continue
                try:
                    conn.sendall(_MAGIC)
                except OSError:
                    conn.close()
                    continue
                conn.settimeout(timeout)
                self._peers[r] = conn
                missing -= 1
        else:
            last: Optional[BaseException] = None
            attempt = 0
            while True:
                cand = port + attempt % max(1, n_ports)
                attempt += 1
                s = None
                try:
                    s = socket.create_connection((addr, cand), timeout=min(5.0, max(0.1, deadline - time.monotonic())))
                    s.settimeout(min(timeout, 10.0))          # a foreign service that never answers is given up on
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.sendall(_MAGIC + struct.pack("<ii", rank, world) + token)
                    if _recv_exact(s, len(_MAGIC)) != _MAGIC:
                        raise RendezvousError("unexpected reply")
                    s.settimeout(timeout)
                    self._up = s
                    break
                except (OSError, RendezvousError) as exc:   # rank 0 is not listening (yet, or not on this port)
                    last = exc
                    if s is not None:
                        s.close()
                    if time.monotonic() > deadline:
                        raise RendezvousError(f"rank {rank} cannot reach rank 0 at {addr}:{port}"
                                              f"{'..' + str(port + n_ports - 1) if n_ports > 1 else ''}: {last}") from exc
                    time.sleep(0.05)

    # ---- construction from the launcher's environment ------------------------------------------------------
    @classmethod
    def from_env(cls, env=os.environ, timeout: Optional[float] = None) -> "Rendezvous":
        rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
        t = float(env.get("UPX_RDZV_TIMEOUT", "600")) if timeout is None else timeout
        return cls(rank, world, env.get("MASTER_ADDR", "127.0.0.1"), default_port(env), t,
                   n_ports=1 if env.get("UPX_RDZV_PORT") else 8)

    # ---- the one primitive: everybody's bytes to everybody -----------------------------------------------------
    def allgather_bytes(self, payload: bytes) -> List[bytes]:
        if self.world == 1:
            return [bytes(payload)]
        try:
            if self.rank == 0:
                parts = [bytes(payload)] + [_recv_frame(self._peers[r]) for r in range(1, self.world)]
                blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
                for r in range(1, self.world):
                    _send_frame(self._peers[r], blob)
                return parts
            _send_frame(self._up, bytes(payload))
            blob = _recv_frame(self._up)
        except (OSError, struct.error) as exc:
            raise RendezvousError(f"rank {self.rank}: lost the process group ({exc})") from exc
        parts, pos = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<I", blob, pos)
            parts.append(blob[pos + 4:pos + 4 + n])
            pos += 4 + n
        return parts

    def broadcast_bytes(self, payload: Optional[bytes], src: int = 0) -> bytes:
        return self.allgather_bytes(payload if self.rank == src and payload is not None else b"")[src]

    def barrier(self) -> None:
        self.allgather_bytes(b"")

    def allreduce_max(self, values: Sequence[float]) -> List[float]:
        """Element-wise maximum over the ranks; a NaN on any rank gives NaN on every rank (max is not order dependent)."""
        mine = [float(v) for v in values]
        rows = [struct.unpack(f"<{len(mine)}d", p) for p in self.allgather_bytes(struct.pack(f"<{len(mine)}d", *mine))]
        out = []
        for col in zip(*rows):
            out.append(float("nan") if any(math.isnan(v) for v in col) else max(col))
        return out

    def all_ok(self, ok: bool = True, message: str = "") -> None:
        """Every rank reports whether its last step worked; if one did not, EVERY rank raises (nobody waits forever)."""
        flags = self.allgather_bytes(bytes([_OK if ok else _FAIL]) + message.encode("utf-8", "replace")[:400])
        bad = [(r, f[1:].decode("utf-8", "replace")) for r, f in enumerate(flags) if f[:1] != bytes([_OK])]
        if bad:
            raise RendezvousError("; ".join(f"rank {r}: {m or 'failed'}" for r, m in bad))

    def close(self) -> None:
        for s in self._peers:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers = []
        for s in (self._up, self._listener):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._up = self._listener = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
