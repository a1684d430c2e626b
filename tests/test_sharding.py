"""CPU, world_size 2 over gloo: the multi-GPU frame sharding (scatter -> per-rank op ->
gather) reproduces the single-process result; partition arithmetic covers ragged batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import synth
from oracle import imgxf_oracle as O


def test_shard_ranges_are_a_partition():
    from imagetransformations_amd import sharding as S
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            ranges = [S.shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1 and sizes == S.shard_counts(n, world)
    assert S.shard_counts(1024, 8) == [128] * 8          # BASELINE configs[4]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetransformations_amd import sharding as S
        frames = None
        if rank == 0:
            frames = torch.from_numpy(np.stack([synth(50 + i, 24, 40) for i in range(n_frames)]))

        def per_shard(block):     # stand-in for an `ops.*` call: the oracle, frame by frame
            return torch.from_numpy(np.stack([O.apply_brightness(f.numpy(), 0.05) for f in block]))

        out = S.map_frames(per_shard, frames, n_frames, (24, 40, 3), torch.device("cpu"))
        local = S.scatter_frames(frames, n_frames, (24, 40, 3), torch.device("cpu"))
        csum = S.checksum(local)
        if rank == 0:
            np.savez(out_path, out=out.numpy(), csum=csum, inp=frames.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 2, 1])
def test_scatter_compute_gather_world2(tmp_path, n_frames):
    out_path = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(2, _free_port(), n_frames, out_path), nprocs=2, join=True)
    res = np.load(out_path)
    want = np.stack([O.apply_brightness(f, 0.05) for f in res["inp"]])
    assert np.array_equal(res["out"], want)
    assert int(res["csum"]) == int(res["inp"].astype(np.int64).sum())


def _worker_pipelined(rank, world, port, n_frames, chunk, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetransformations_amd import sharding as S
        frames = None
        if rank == 0:
            frames = torch.from_numpy(np.stack([synth(80 + i, 20, 28) for i in range(n_frames)])) if n_frames else \
                torch.zeros((0, 20, 28, 3), dtype=torch.uint8)
        calls = []

        def per_shard(block):     # a frame-by-frame map that also changes the frame geometry (RGB -> L)
            calls.append(int(block.shape[0]))
            return torch.from_numpy(np.stack([O.rgb2l(f.numpy()) for f in block]))

        piped = S.map_frames_pipelined(per_shard, frames, n_frames, (20, 28, 3), torch.device("cpu"), chunk=chunk)
        plain = S.map_frames(per_shard, frames, n_frames, (20, 28, 3), torch.device("cpu")) if n_frames else None
        if rank == 0:
            np.savez(out_path, piped=piped.numpy(), plain=plain.numpy() if plain is not None else np.zeros(0),
                     inp=frames.numpy(), max_call=max(calls[:-1] or [0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,chunk", [(13, 2), (7, 3), (5, 8), (2, 1), (1, 4)])
def test_pipelined_map_equals_plain_world2(tmp_path, n_frames, chunk):
    """The chunk pipeline (transfers of chunk s / s-2 around the compute of chunk s-1)
    gathers exactly what the one-shot scatter -> compute -> gather does, for ragged blocks and chunk
    sizes above and below the block size."""
    out_path = str(tmp_path / "res.npz")
    mp.spawn(_worker_pipelined, args=(2, _free_port(), n_frames, chunk, out_path), nprocs=2, join=True)
    res = np.load(out_path)
    want = np.stack([O.rgb2l(f) for f in res["inp"]])
    assert np.array_equal(res["piped"], want)
    assert np.array_equal(res["plain"], want)
