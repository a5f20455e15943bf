"""Multi-GPU host logic: independent frame-pairs shard across the GPUs of one node.

The expensive per-frame work depends on one stereo pair only; the pose chain is a cheap
sequential product.  Each rank (one process per GPU) therefore takes a
contiguous chunk of frame indices plus a one-frame halo, runs its own StereoOdometer on it and
produces the relative transforms T_k (frame k-1 -> k).  The only exchange is one all_gather of
16 float64 per frame at the end of the batch (RCCL when the tensors live on GPUs, gloo on CPU);
rank 0 prefix-composes the trajectory.  No data-path collective exists.
"""
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


def gather_relative(local_T, local_ok, dist=None, device=None):
    """all_gather of this rank's (n_local, 4, 4) transforms and accept flags.  `dist` is
    torch.distributed (already initialised) or None for a single process.  Every rank must pass
    the same n_local (weak scaling: equal chunks).  Returns (world*n_local, 4, 4), (world*n_local,)."""
    local_T = np.ascontiguousarray(local_T, np.float64).reshape(-1, 4, 4)
    local_ok = np.ascontiguousarray(local_ok, np.float64).reshape(-1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_T, local_ok > 0
    import torch
    world = dist.get_world_size()
    payload = np.concatenate([local_T.reshape(len(local_T), 16), local_ok[:, None]], 1)
    t = torch.from_numpy(payload)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    allp = torch.cat(out, 0).cpu().numpy()
    return allp[:, :16].reshape(-1, 4, 4), allp[:, 16] > 0
