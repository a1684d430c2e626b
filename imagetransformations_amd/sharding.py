"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed).

The hot path has no cross-frame dependency (the reference's outer loop over images,
/root/reference/transformation.py:113, is embarrassingly parallel), so a batch shards by
FRAME in contiguous blocks and the compute needs no collective at all.  The only exchange
steps are the optional scatter of a root-resident batch to the ranks and the gather of the
results: point-to-point sends from/to the root (RCCL send/recv — one direct xGMI link per
peer), batched so all 7 links of the root run concurrently.  Works with the `nccl` (= RCCL)
backend on device tensors and with `gloo` on host tensors (used by the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def shard_counts(n_frames: int, world: int) -> list[int]:
    """Contiguous block sizes; the first n % world ranks take one extra frame."""
    if n_frames < 0 or world < 1:
        raise ValueError("n_frames >= 0 and world >= 1 required")
    base, extra = divmod(n_frames, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def shard_range(n_frames: int, world: int, rank: int) -> tuple[int, int]:
    """[start, stop) of the frames owned by `rank`."""
    counts = shard_counts(n_frames, world)
    start = sum(counts[:rank])
    return start, start + counts[rank]


def _world(group=None) -> tuple[int, int]:
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _peer(group, r: int) -> int:
    """P2POp addresses peers by GLOBAL rank; `r` is a rank of `group` (identical without a sub-group)."""
    return r if group is None else dist.get_global_rank(group, r)


def scatter_frames(frames: Optional[torch.Tensor], n_frames: int, frame_shape: Sequence[int],
                   device: torch.device, root: int = 0, group=None) -> torch.Tensor:
    """Root holds `frames` [n_frames, *frame_shape] uint8; every rank returns its block."""
    rank, world = _world(group)
    start, stop = shard_range(n_frames, world, rank)
    if world == 1:
        return frames[start:stop]
    local = torch.empty((stop - start, *frame_shape), dtype=torch.uint8, device=device)
    ops = []
    if rank == root:
        for peer in range(world):
            s, e = shard_range(n_frames, world, peer)
            if peer == root:
                local.copy_(frames[s:e])
            elif e > s:
                ops.append(dist.P2POp(dist.isend, frames[s:e].contiguous(), _peer(group, peer), group))
    elif stop > start:
        ops.append(dist.P2POp(dist.irecv, local, _peer(group, root), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return local


def gather_frames(local: torch.Tensor, n_frames: int, root: int = 0, group=None) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames: the root returns [n_frames, ...], other ranks None."""
    rank, world = _world(group)
    if world == 1:
        return local
    out = None
    ops = []
    if rank == root:
        out = torch.empty((n_frames, *local.shape[1:]), dtype=local.dtype, device=local.device)
        for peer in range(world):
            s, e = shard_range(n_frames, world, peer)
            if peer == root:
                out[s:e].copy_(local)
            elif e > s:
                ops.append(dist.P2POp(dist.irecv, out[s:e], _peer(group, peer), group))
    elif local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local.contiguous(), _peer(group, root), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def _on(device: torch.device):
    """Make `device` the current device for the enclosed launches (a no-op for host tensors)."""
    import contextlib
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


def map_frames(fn: Callable[[torch.Tensor], torch.Tensor], frames: Optional[torch.Tensor], n_frames: int,
               frame_shape: Sequence[int], device: torch.device, root: int = 0, group=None):
    """scatter -> fn(local block) -> gather.  `fn` is any per-batch op from `ops` (or a
    composition); it runs on each rank's block with no communication.  `device` is made current
    for the whole call, so a rank that never called set_device still allocates and launches there."""
    with _on(device):
        local = scatter_frames(frames, n_frames, frame_shape, device, root, group)
        out = fn(local) if local.shape[0] > 0 else local
        return gather_frames(out, n_frames, root, group)


def map_frames_pipelined(fn: Callable[[torch.Tensor], torch.Tensor], frames: Optional[torch.Tensor], n_frames: int,
                         frame_shape: Sequence[int], device: torch.device, chunk: int = 8, root: int = 0, group=None,
                         out_shape: Optional[Sequence[int]] = None, out_dtype: torch.dtype = torch.uint8):
    """map_frames as a three-stage pipeline over chunks of `chunk` frames:

        step s:   transfers { input chunk s  root -> peers,  output chunk s-2  peers -> root }
                  ||  fn(chunk s-1) on every rank

    The transfers of one step are ONE grouped batch of point-to-point operations (RCCL progresses the
    sends and receives of a group concurrently, each root<->peer pair on its own xGMI link), issued
    before the step's compute and waited for after it, so a rank computes chunk s-1 while chunk s
    arrives and the result of chunk s-2 leaves.  `fn` must map [k, *frame_shape] -> [k, *out_shape]
    frame by frame (any `ops` function).  Returns the gathered result on the root (an empty
    [0, *out_shape] tensor for an empty batch, like map_frames), None elsewhere; equal to map_frames
    for every chunk size.  `out_shape` / `out_dtype` (per-frame geometry of fn's result) are only
    needed when fn changes the geometry AND the root owns no frame (fewer frames than ranks with
    root != 0); without them that root learns the geometry from rank 0's first result — fn is never
    run on a frame its rank does not own."""
    rank, world = _world(group)
    if chunk < 1:
        raise ValueError("chunk >= 1 required")
    with _on(device):
        if world == 1:
            return fn(frames) if n_frames > 0 else frames
        start, stop = shard_range(n_frames, world, rank)
        mine = stop - start
        counts = shard_counts(n_frames, world)
        nsteps = (max(counts) + chunk - 1) // chunk
        if n_frames == 0:
            shape = tuple(out_shape) if out_shape is not None else tuple(frame_shape)
            return torch.empty((0, *shape), dtype=out_dtype, device=device) if rank == root else None
        # a root without frames cannot see fn's output geometry: rank 0 (which always owns frames) tells it
        tell_root = counts[root] == 0 and out_shape is None

        def piece(r: int, k: int) -> tuple[int, int]:
            """frames [a, b) of rank r's block that form its chunk k (empty when k is out of range)"""
            a = min(k * chunk, counts[r])
            return a, min(a + chunk, counts[r])

        is_root = rank == root
        local_in = frames[start:stop] if is_root else torch.empty((mine, *frame_shape), dtype=torch.uint8, device=device)
        local_out = None            # peers: results of the own block, allocated once fn's output shape is known
        result = None               # root: the gathered batch

        for s in range(nsteps + 2):
            ops = []
            # ---- transfers of step s (posted first, waited for after the compute)
            if is_root:
                for peer in range(world):
                    if peer == root:
                        continue
                    ps, _ = shard_range(n_frames, world, peer)
                    a, b = piece(peer, s)
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, frames[ps + a:ps + b].contiguous(), _peer(group, peer), group))
                    a, b = piece(peer, s - 2)
                    if s >= 2 and b > a:
                        ops.append(dist.P2POp(dist.irecv, result[ps + a:ps + b], _peer(group, peer), group))
            else:
                a, b = piece(rank, s)
                if b > a:
                    ops.append(dist.P2POp(dist.irecv, local_in[a:b], _peer(group, root), group))
                a, b = piece(rank, s - 2)
                if s >= 2 and b > a:
                    ops.append(dist.P2POp(dist.isend, local_out[a:b], _peer(group, root), group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            # ---- compute chunk s-1 (its input arrived with step s-1's transfers)
            a, b = piece(rank, s - 1)
            if s >= 1 and b > a:
                y = fn(local_in[a:b])
                if is_root:
                    if result is None:
                        result = torch.empty((n_frames, *y.shape[1:]), dtype=y.dtype, device=y.device)
                    result[start + a:start + b].copy_(y)
                else:
                    if local_out is None:
                        local_out = torch.empty((mine, *y.shape[1:]), dtype=y.dtype, device=y.device)
                    local_out[a:b].copy_(y)
            if s == 1 and counts[root] == 0:
                # an empty root block (root != 0, fewer frames than ranks): the geometry of the receives
                # posted from step 2 on is given by the caller or sent by rank 0 after its first chunk
                if tell_root and (rank == 0 or is_root):
                    meta = torch.zeros(10, dtype=torch.int64, device=device)
                    if rank == 0:
                        dims = list(local_out.shape[1:])
                        meta[:2 + len(dims)] = torch.tensor([_DTYPES.index(local_out.dtype), len(dims), *dims])
                        dist.send(meta, _peer(group, root), group)
                    else:
                        dist.recv(meta, _peer(group, 0), group)
                        m = meta.tolist()
                        out_dtype, out_shape = _DTYPES[m[0]], tuple(m[2:2 + m[1]])
                if is_root:
                    result = torch.empty((n_frames, *out_shape), dtype=out_dtype, device=device)
            for req in reqs:
                req.wait()
        return result if is_root else None


_DTYPES = [torch.uint8, torch.float32, torch.int32, torch.int64, torch.float16, torch.float64, torch.int16, torch.int8]


def checksum(t: torch.Tensor, group=None) -> int:
    """Byte sum of a sharded batch, summed over the ranks: equal for any sharding of the same frames
    (and blind to their order — `checksum_weighted` is the one that sees misplaced frames)."""
    rank, world = _world(group)
    v = t.reshape(t.shape[0], -1).to(torch.int64).sum(dim=1) if t.shape[0] else torch.zeros(0, dtype=torch.int64, device=t.device)
    total = v.sum().reshape(1)
    if world > 1:
        if dist.get_backend(group) != "nccl":
            total = total.cpu()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return int(total.item())


def checksum_weighted(t: torch.Tensor, offset: int = 0, group=None, reduce: bool = True) -> int:
    """Checksum of a sharded batch that also sees misplaced bytes and frames: the sum over the frames
    j of this rank's block, numbered globally (`offset` = index of the block's first frame), of
    (offset + j + 1) * sum_p byte[j, p] * (p mod 65521 + 1), in wrapping int64, summed over the ranks.
    Equal for every sharding of the same batch; bench.py compares the N-rank value of a fixed-seed
    batch through Gaussian + rotate with the value the root computes alone."""
    rank, world = _world(group)
    total = torch.zeros(1, dtype=torch.int64, device=t.device)
    if t.shape[0]:
        flat = t.reshape(t.shape[0], -1)
        w = torch.arange(flat.shape[1], dtype=torch.int64, device=t.device) % 65521 + 1
        for j in range(flat.shape[0]):
            total += (offset + j + 1) * (flat[j].to(torch.int64) * w).sum()
    if world > 1 and reduce:
        if dist.get_backend(group) != "nccl":
            total = total.cpu()                      # host collectives (gloo rehearsal / CPU tests)
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return int(total.item())
