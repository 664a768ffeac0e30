"""
Process rendezvous for the one-process-per-GPU model, on ONE node, with nothing but the standard
library: the 128-byte RCCL unique id has to travel from rank 0 to every rank, and a benchmark
needs a barrier and a max over ranks.  Neither justifies a second communication stack in the
process (round 1 used `torch.distributed`/gloo for it, which pulled torch's own bundled
`librccl.so` + HIP runtime into the process before liblynxhip bound RCCL -- which RCCL build the
communicator ran on then depended on Python import order).

Topology: a star.  Rank 0 listens on an ephemeral loopback port and publishes the port number in
a file under the temp directory whose name is derived from what all ranks of one launch share --
`MASTER_ADDR`, `MASTER_PORT` and the launcher's pid (`torchrun` exports the first two and is the
parent of every rank; `MASTER_PORT` itself is taken by the launcher's own store) -- or from
`LYNX_RDZV_KEY` (what `bench.py`'s own launcher and any launcher whose ranks are not direct children
of one process should set).  The file holds the port, a random nonce and rank 0's pid; it is created
exclusively with mode 0600 after any leftover of a crashed launch with the same key has been removed.
Every other rank polls for the file, connects, says who it is and checks that the answer carries the
nonce it read -- a stale file cannot lead it to somebody else's rank 0.  A collective is one message up
and one message down per rank; payloads are a few hundred bytes.

The reference has no counterpart (it is single-process).
"""

from __future__ import annotations

import os
import secrets
import socket
import struct
import tempfile
import time
from pathlib import Path

_MAGIC = b"LYNXRDZV1"


def default_key() -> str:
    key = os.environ.get("LYNX_RDZV_KEY")
    if key:
        return key
    return "{}-{}-{}".format(os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"),
                             os.getppid())


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks = []
    while n:
        part = sock.recv(n)
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        chunks.append(part)
        n -= len(part)
    return b"".join(chunks)


def _recv(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class Rendezvous:
    """Collectives among the `world` ranks of one launch on this node (see module docstring)."""

    def __init__(self, rank: int, world: int, key: str | None = None, timeout_s: float = 300.0):
        assert 0 <= rank < world
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self.peers: dict = {}      # rank 0: rank -> socket
        self.hub: socket.socket | None = None  # other ranks: socket to rank 0
        self._listener = None
        safe = "".join(c if c.isalnum() or c in "-_." else "_" for c in (key or default_key()))
        self.path = Path(tempfile.gettempdir()) / f"lynx-rdzv-{safe}.port"
        if world == 1:
            return
        if rank == 0:
            self._serve()
        else:
            self._join()

    # -- bring-up ---------------------------------------------------------------------------
    def _serve(self) -> None:
        listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        listener.bind(("127.0.0.1", 0))
        listener.listen(self.world)
        listener.settimeout(self.timeout_s)
        self._listener = listener
        port = listener.getsockname()[1]
        self._nonce = secrets.token_hex(16)
        try:
            self.path.unlink()  # a leftover of a crashed launch with the same key is not this launch's rank 0
        except FileNotFoundError:
            pass
        tmp = self.path.with_suffix(f".{os.getpid()}.tmp")
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "w") as fh:
            fh.write(f"{port} {self._nonce} {os.getpid()}\n")
        os.replace(tmp, self.path)  # atomic: a reader sees the whole line or no file
        try:
            while len(self.peers) < self.world - 1:
                conn, _ = listener.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(self.timeout_s)
                hello = _recv(conn)
                if not hello.startswith(_MAGIC):
                    conn.close()
                    continue
                (peer,) = struct.unpack("<I", hello[len(_MAGIC):])
                assert 0 < peer < self.world and peer not in self.peers, f"unexpected rank {peer}"
                _send(conn, _MAGIC + self._nonce.encode())  # the joiner checks it reached ITS rank 0
                self.peers[peer] = conn
        except socket.timeout:
            raise TimeoutError(f"rendezvous: only {len(self.peers) + 1} of {self.world} ranks showed up "
                               f"within {self.timeout_s:.0f} s at {self.path} (key from "
                               f"{'LYNX_RDZV_KEY' if os.environ.get('LYNX_RDZV_KEY') else 'MASTER_ADDR-MASTER_PORT-parent pid'}: "
                               "ranks that are not children of one launcher must share LYNX_RDZV_KEY)") from None

    def _join(self) -> None:
        deadline = time.monotonic() + self.timeout_s
        last = None
        while time.monotonic() < deadline:
            try:
                port, nonce = self.path.read_text().split()[:2]
                sock = socket.create_connection(("127.0.0.1", int(port)), timeout=5.0)
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                sock.settimeout(3.0)  # the handshake only: a stale file may lead to a listener that never answers
                _send(sock, _MAGIC + struct.pack("<I", self.rank))
                if _recv(sock) != _MAGIC + nonce.encode():  # a stale file led somewhere else
                    sock.close()
                    raise ConnectionError("not this launch's rank 0")
                sock.settimeout(self.timeout_s)
                self.hub = sock
                return
            except (FileNotFoundError, ValueError, ConnectionError, OSError) as exc:  # not published yet / stale
                last = exc
                time.sleep(0.05)
        raise TimeoutError(f"rendezvous: rank {self.rank} found no rank 0 at {self.path} within {self.timeout_s:.0f} s "
                           f"({last}); LYNX_RDZV_KEY={os.environ.get('LYNX_RDZV_KEY')!r}, computed key file above -- "
                           "ranks that are not children of one launcher must share LYNX_RDZV_KEY")

    # -- collectives ------------------------------------------------------------------------
    def all_gather(self, payload: bytes) -> list:
        """Every rank's payload, in rank order, on every rank."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(self.peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
            for r in range(1, self.world):
                _send(self.peers[r], blob)
            return parts
        _send(self.hub, payload)
        blob, parts, at = _recv(self.hub), [], 0
        while at < len(blob):
            (n,) = struct.unpack_from("<I", blob, at)
            parts.append(blob[at + 4: at + 4 + n])
            at += 4 + n
        assert len(parts) == self.world
        return parts

    def broadcast(self, payload: bytes | None) -> bytes:
        """Rank 0's payload on every rank."""
        return self.all_gather(payload if self.rank == 0 else b"")[0]

    def barrier(self) -> None:
        self.all_gather(b"")

    def max(self, value: float) -> float:
        return max(struct.unpack("<d", p)[0] for p in self.all_gather(struct.pack("<d", float(value))))

    def all_true(self, flag: bool) -> bool:
        return all(p == b"\x01" for p in self.all_gather(b"\x01" if flag else b"\x00"))

    def close(self) -> None:
        for sock in list(self.peers.values()) + [self.hub, self._listener]:
            if sock is not None:
                try:
                    sock.close()
                except OSError:
                    pass
        self.peers, self.hub, self._listener = {}, None, None
        if self.rank == 0 and self.world > 1:
            try:
                self.path.unlink()
            except OSError:
                pass
