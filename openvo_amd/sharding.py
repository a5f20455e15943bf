"""Multi-GPU host logic: independent frame-pairs shard across the GPUs of one node.

The expensive per-frame work depends on one stereo pair only; the pose chain is a cheap sequential
product.  Each rank (one process per GPU) takes a contiguous chunk of frame indices plus a one-frame halo,
runs its own StereoOdometer on it and produces the relative transforms T_k (frame k-1 -> k).  The only
exchange is one all-gather of 17 float64 per frame at the end of the batch; rank 0 prefix-composes the
trajectory.  No data-path collective exists.

Transport.  `init_from_env()` reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (whatever
launcher set them -- `python -m torch.distributed.run` only serves as the launcher, nothing of torch is
imported) and builds a `Group`:

  * ranks meet on a TCP socket of the node: rank 0 listens on an ephemeral port and publishes it in a small
    file keyed by MASTER_PORT and the launcher's pid; that socket carries the control traffic (barrier, the
    128-byte RCCL unique id) and is itself a complete, GPU-free transport (tests, ranks sharing one GPU);
  * with one GPU per rank the pose gather goes over RCCL (ncclAllGather over xGMI), bound directly behind
    the C ABI (vo_mgpu_*, include/vo355.h) -- no PyTorch.

The reference has no counterpart (stereo_odometer.py is single-process); what sharding can break is the
history the chain carries from frame to frame [stereo_odometer.py:137-160,215-220] -- `boundary_report`
says exactly where a sharded run is not guaranteed to equal the sequential one.
"""
import atexit
import json
import os
import socket
import struct
import time

import numpy as np


def shard_range(n_frames, rank, world):
    """Frames [lo, hi) whose poses rank `rank` owns; it must also read frame lo-1 (halo) to
    form the first relative transform, except for rank 0."""
    base, rem = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def relative_from_chain(c_T_w_before, c_T_w_after):
    """T with c_T_w_after = T @ c_T_w_before (what one accepted update() multiplied in)."""
    return c_T_w_after @ np.linalg.inv(c_T_w_before)


def compose(relative, accepted=None):
    """Prefix-compose relative transforms into camera poses: c_T_w(k) = T_k @ c_T_w(k-1);
    pose(k) = inv(c_T_w(k)) (reference stereo_odometer.py:137-138,225-226).  A rejected frame
    (accepted[k] False) leaves the chain unchanged."""
    c_T_w = np.eye(4)
    poses = []
    for k, T in enumerate(relative):
        if accepted is None or accepted[k]:
            c_T_w = np.asarray(T, np.float64) @ c_T_w
        poses.append(np.linalg.inv(c_T_w))
    return np.array(poses)


def boundary_report(accepted, n_local, world, used_fallback=None):
    """Where a sharded run may differ from the sequential reference.

    The reference's update() carries history across frames: a rejected frame is not saved, so the next
    frame pairs with the older `current` [stereo_odometer.py:152-155]; the motion gates widen with
    `skipped_frames` [:215-220]; and a failed pair retries against `prev` [:139-150].  A shard starts from
    its halo frame with none of that history, so its first pose steps equal the sequential ones iff the
    halo frame -- the last frame of the previous shard -- was accepted there, and (used_fallback given)
    the first step of the shard did not need the one-frame-back fallback it does not have.

    accepted: (world * n_local,) accept flags in frame order; returns a list of dicts, one per inexact
    boundary: {"shard": r, "frame": first frame of shard r, "cause": ...}.  Empty list = the sharded
    trajectory is the sequential one."""
    accepted = np.asarray(accepted).astype(bool).reshape(-1)
    out = []
    for r in range(1, int(world)):
        first = r * int(n_local)
        if first >= len(accepted):
            break
        if not accepted[first - 1]:
            out.append({"shard": r, "frame": first, "cause": "halo frame %d was rejected by shard %d: the sequential chain pairs "
                                                            "frame %d with an older frame and widens its gates" % (first - 1, r - 1, first)})
        elif not accepted[first]:
            out.append({"shard": r, "frame": first, "cause": "first step of the shard was rejected: sequentially it could still "
                                                            "succeed through the one-frame-back fallback"})
        elif used_fallback is not None and used_fallback[first]:
            out.append({"shard": r, "frame": first, "cause": "first step used the fallback"})
    return out


# ---- transport ---------------------------------------------------------------------------------------------
def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Group:
    """world processes of one node.  Control plane: a star of TCP sockets around rank 0.  Data plane for the
    pose gather: RCCL when `attach_rccl` succeeded, the sockets otherwise."""

    def __init__(self, rank, world, master_addr=None, key=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers, self._sock, self._mgpu, self._lib, self._late = [], None, None, None, None
        self.transport = "socket"
        if self.world == 1:
            return
        key = key or rendezvous_key()
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "vo355_rdzv_%s.json" % key)
        # the ranks of one node meet on the loopback interface unless told otherwise (the master_addr argument, else
        # VO_RDZV_BIND): nothing outside the node has any business connecting here, and a launcher's MASTER_ADDR may be a
        # name that does not resolve
        master_addr = master_addr or os.environ.get("VO_RDZV_BIND", "127.0.0.1")
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((master_addr, 0))                      # ephemeral port: MASTER_PORT itself belongs to the launcher
            srv.listen(self.world)
            srv.settimeout(timeout)
            tmp = path + ".%d" % os.getpid()
            with open(tmp, "w") as fh:
                json.dump({"addr": master_addr, "port": srv.getsockname()[1], "pid": os.getpid()}, fh)
            os.replace(tmp, path)
            atexit.register(lambda: os.path.exists(path) and os.remove(path))
            self._path = path
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                if not (1 <= r < self.world) or r in peers:     # not one of this job's ranks (or one that is already here):
                    try:                                        # told so before the door closes, so that it fails at once
                        conn.sendall(b"\x00")
                    except OSError:
                        pass
                    conn.close()
                    continue
                conn.sendall(b"\x01")
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    with open(path) as fh:
                        info = json.load(fh)
                    s = socket.create_connection((info["addr"], info["port"]), timeout=timeout)
                    break
                except (OSError, ValueError):
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rank %d: no rendezvous file %s" % (self.rank, path))
                    time.sleep(0.02)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<I", self.rank))
            if _recv_exact(s, 1) != b"\x01":
                s.close()
                raise ConnectionError("rank %d was turned away by the rank 0 listening at %s:%s (rendezvous file %s): another job's "
                                      "rendezvous, or this rank number is already connected -- give the jobs different "
                                      "MASTER_PORT / VO_RUN_ID values" % (self.rank, info["addr"], info["port"], path))
            self._sock = s

    # -- control plane (sockets)
    def all_gather_bytes(self, payload):
        """Every rank contributes a bytes object; every rank receives the list in rank order."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._peers]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for c in self._peers:
                _send_msg(c, blob)
            return parts
        _send_msg(self._sock, payload)
        blob, parts, o = _recv_msg(self._sock), [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, o)
            parts.append(blob[o + 8: o + 8 + n])
            o += 8 + n
        return parts

    def barrier(self):
        self.all_gather_bytes(b"")

    def broadcast_bytes(self, payload, root=0):
        return self.all_gather_bytes(payload if self.rank == root else b"")[root]

    # -- data plane
    def attach_rccl(self, device, lib=None):
        """One GPU per rank: route the pose gather through RCCL (vo_mgpu_*).  Returns True on success;
        on any failure the group stays on the socket transport (every rank takes the same decision: the vote below).
        lib: the object providing vo_mgpu_* (tests pass a stub; default: the native library)."""
        import ctypes
        if self.world == 1:
            return False
        if lib is None:
            from . import _native
            lib = _native.lib()
        L = lib
        ident = (ctypes.c_uint8 * 128)()
        ok = 1
        if self.rank == 0 and L.vo_mgpu_unique_id(ident) != 0:
            ok = 0
        blob = self.broadcast_bytes(bytes([ok]) + bytes(ident))
        if blob[0] == 0:
            return False
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(blob[1:129])
        h = ctypes.c_void_p()
        # communicator creation is a collective: if a peer never arrives it blocks inside RCCL, so it runs on a helper
        # thread with a deadline (VO_RCCL_TIMEOUT seconds, default 90) and a rank that gives up votes "no" below
        import threading
        res = {}

        def _create():
            res["rc"] = L.vo_mgpu_create(int(device), self.rank, self.world, ident, ctypes.byref(h))

        th = threading.Thread(target=_create, daemon=True)
        th.start()
        th.join(float(os.environ.get("VO_RCCL_TIMEOUT", "90")))
        rc = res.get("rc", -1)
        oks = self.all_gather_bytes(bytes([1 if rc == 0 else 0]))
        if not all(b == b"\x01" for b in oks):
            if th.is_alive():
                # still inside ncclCommInitRank: the handle belongs to that thread until it returns; close() collects it
                self._late = (th, h, L, res)
            elif rc == 0:
                L.vo_mgpu_destroy(h)
            return False
        self._mgpu, self._lib, self.transport = h, L, "rccl"
        return True

    def describe(self):
        """What the data plane really is: for RCCL the communicator's own answers (ncclCommCount / ncclCommUserRank)."""
        import ctypes
        d = {"transport": self.transport, "world": self.world, "rank": self.rank}
        if self._mgpu is not None and hasattr(self._lib, "vo_mgpu_info"):
            n, r, dev = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
            if self._lib.vo_mgpu_info(self._mgpu, ctypes.byref(n), ctypes.byref(r), ctypes.byref(dev)) == 0:
                d.update({"rccl_ranks": n.value, "rccl_rank": r.value, "device": dev.value})
        return d

    def all_gather_f64(self, local):
        """(n,) float64 per rank (same n everywhere) -> (world, n)."""
        local = np.ascontiguousarray(local, np.float64).reshape(-1)
        if self.world == 1 and self._mgpu is None:
            return local[None].copy()
        if self._mgpu is not None:
            out = np.empty((self.world, len(local)), np.float64)
            rc = self._lib.vo_mgpu_all_gather_f64(self._mgpu, local.ctypes.data, len(local), out.ctypes.data)
            if rc != 0:
                raise RuntimeError("vo_mgpu_all_gather_f64: %s" % self._lib.vo_mgpu_last_error(self._mgpu).decode())
            return out
        parts = self.all_gather_bytes(local.tobytes())
        return np.stack([np.frombuffer(p, np.float64) for p in parts])

    def all_reduce_max(self, value):
        """max of a float over the ranks (the slowest rank's time)."""
        if self.world == 1:
            return float(value)
        if self._mgpu is not None:
            v = np.array([value], np.float64)
            rc = self._lib.vo_mgpu_all_reduce_max_f64(self._mgpu, v.ctypes.data, 1)
            if rc != 0:
                raise RuntimeError("vo_mgpu_all_reduce_max_f64: %s" % self._lib.vo_mgpu_last_error(self._mgpu).decode())
            return float(v[0])
        return float(self.all_gather_f64([value]).max())

    def gather_relative(self, local_T, local_ok):
        """All-gather of this rank's (n_local, 4, 4) relative transforms and accept flags: the path's only
        exchange, 17 float64 per frame.  Every rank must pass the same n_local (weak scaling: equal chunks).
        Returns (world * n_local, 4, 4), (world * n_local,) bool."""
        local_T = np.ascontiguousarray(local_T, np.float64).reshape(-1, 4, 4)
        local_ok = np.ascontiguousarray(local_ok, np.float64).reshape(-1)
        payload = np.concatenate([local_T.reshape(len(local_T), 16), local_ok[:, None]], 1)
        allp = self.all_gather_f64(payload).reshape(-1, 17)
        return allp[:, :16].reshape(-1, 4, 4), allp[:, 16] > 0

    def close(self):
        if self._mgpu is not None:
            self._lib.vo_mgpu_destroy(self._mgpu)
            self._mgpu = None
        if self._late is not None:
            # a communicator creation that outlived its deadline: if it has returned since, give the communicator back;
            # if it never does, the daemon thread dies with the process
            th, h, L, res = self._late
            th.join(1.0)
            if not th.is_alive() and res.get("rc", -1) == 0:
                L.vo_mgpu_destroy(h)
            self._late = None
        for c in self._peers:
            c.close()
        self._peers = []
        if self._sock is not None:
            self._sock.close()
            self._sock = None
        if self.rank == 0 and getattr(self, "_path", None) and os.path.exists(self._path):
            os.remove(self._path)


def rendezvous_key():
    """Names the rendezvous file of one job on this node: the launcher's MASTER_ADDR:MASTER_PORT plus its run id when it
    exports one (torch.distributed.run: TORCHELASTIC_RUN_ID) -- the same for every rank of a job whatever started them
    (ranks need not share a parent process), different for two jobs on one node."""
    run = os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("VO_RUN_ID", ""))
    if not run and not os.environ.get("MASTER_PORT"):
        run = "ppid%d" % os.getppid()      # nothing names the job: ranks forked by one parent still find each other, two such jobs do not collide
    raw = "%s_%s_%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"), run)
    return "".join(ch if ch.isalnum() or ch in "._-" else "-" for ch in raw)


def device_count():
    """HIP devices visible to this process (does not create a context on any of them)."""
    import ctypes
    from . import _native
    n = ctypes.c_int(0)
    _native.lib().vo_device_count(ctypes.byref(n))
    return n.value


def init_from_env(want_rccl=True):
    """Group of the ranks a launcher started on this node (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR,
    MASTER_PORT); returns (group, device index for this rank).

    The device index is an index into the devices THIS process sees, which a launcher may have masked
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES):
      * at least as many visible devices as ranks on the node: rank r takes visible device LOCAL_RANK -- whatever physical GPUs
        the mask names (`HIP_VISIBLE_DEVICES=4,5,6,7` with four ranks: LOCAL_RANK 0..3 -> physical 4..7);
      * a mask that leaves each rank fewer devices than there are ranks (typically ONE device per rank, a different one for
        each): LOCAL_RANK modulo the visible count, i.e. device 0 of a one-device mask -- LOCAL_RANK is then NOT a device index;
      * fewer devices than ranks and no mask: refused, unless the ranks say they mean to share (VO_SHARE_GPU, a rehearsal).
    RCCL is attached when every rank can have a GPU of its own (enough visible devices, or a per-rank mask and no VO_SHARE_GPU);
    two ranks that do land on one GPU make the communicator's creation fail, and the creation vote then leaves ALL ranks on the
    socket transport together (Group.attach_rccl)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world)
    g = Group(rank, world)
    ndev = device_count()
    masked = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    sharing = os.environ.get("VO_SHARE_GPU", "0") not in ("", "0")
    if world > 1 and 0 < ndev < local_world and not masked and not sharing:
        # several ranks on one GPU oversubscribe its hardware queues (12 engines each: measured 96 pairs/s instead of 1400);
        # a launcher that gives every rank a device of its own masks them per rank.  Rehearsals opt in with
        # VO_SHARE_GPU=<ranks per GPU>, set BEFORE openvo_amd is imported: every rank then takes its share of the queues.
        g.close()
        raise RuntimeError("%d ranks but only %d visible GPU(s) and no per-rank device mask: refusing to share a GPU "
                           "(set VO_SHARE_GPU=<ranks per GPU> in the ranks' environment for a rehearsal)" % (local_world, ndev))
    device = local if ndev >= local_world else local % max(ndev, 1)
    own_gpu = ndev >= local_world or (masked and not sharing and ndev > 0)
    if want_rccl and world > 1 and own_gpu and os.environ.get("VO_NO_RCCL", "0") != "1":
        g.attach_rccl(device)
    return g, device


def gather_relative(local_T, local_ok, group=None):
    """Module-level convenience: group.gather_relative, or the identity for a single process."""
    if group is None or group.world == 1:
        local_T = np.ascontiguousarray(local_T, np.float64).reshape(-1, 4, 4)
        return local_T, np.ascontiguousarray(local_ok, np.float64).reshape(-1) > 0
    return group.gather_relative(local_T, local_ok)
