"""Multi-GPU path on CPU: world_size-2 gloo processes check the frame-range sharding that
bench.py --gpus N and glfer_amd.shard use -- ranges partition the frames, each rank's sample
window (hops + left halo) is sufficient, and the rows a rank computes from ITS window alone
are identical to the rows of a single-process run over the whole stream."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, mode, n, overlap, frames, sub_mean, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glfer_amd.shard import frame_range, halo_samples, sample_window
    from oracle import oracle as O
    from _signals import synth
    hop = O.hop(n, overlap)
    x = synth(frames * hop, seed=11)                       # every rank can see the same stream
    first, count = frame_range(frames, rank, world)
    begin, end = sample_window(first, count, hop, n)
    local = x[begin:end]                                    # what this rank would upload
    # the local run starts with zero history, exactly what rank 0 needs; other ranks prepend
    # whole warm-up hops so that local frame (first - warm) is global frame `first`
    warm = (first * hop - begin) // hop
    # the halo is the N-H history rounded up to whole hops (per-hop means need complete hops)
    assert (first * hop - begin) % hop == 0 and (rank == 0 or first * hop - begin == halo_samples(hop, n))
    assert rank == 0 or n - hop <= first * hop - begin < n
    if mode == "fft":
        rows = O.spectrogram_fft(local, n, overlap, 0, sub_mean=sub_mean)
    else:
        rows = O.spectrogram_mtm(local, n, overlap, 2.5, 4, sub_mean=sub_mean)
    mine = torch.from_numpy(rows[warm:warm + count].copy())
    # gather (first, count) and the rows on rank 0 -- test plumbing, not the data path
    meta = [None] * world
    dist.all_gather_object(meta, (first, count))
    parts = [None] * world
    dist.gather_object(mine, parts if rank == 0 else None, dst=0)
    if rank == 0:
        q.put((meta, [p.numpy() for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,n,overlap,frames,sub_mean", [
    ("fft", 1024, 0.5, 77, 0), ("fft", 1024, 0.75, 85, 1), ("mtm", 4096, 0.0, 40, 0), ("mtm", 1024, 0.5, 45, 1),
    ("fft", 1024, 0.9, 130, 1), ("fft", 1024, 0.9, 130, 0)])      # hop 102, history 922: not a whole number of hops
def test_two_rank_sharding_matches_single_process(mode, n, overlap, frames, sub_mean):
    from oracle import oracle as O
    from _signals import synth
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, n, overlap, frames, sub_mean, q)) for r in range(world)]
    for p in procs:
        p.start()
    meta, parts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the ranges partition [0, frames)
    assert meta[0][0] == 0 and meta[0][0] + meta[0][1] == meta[1][0] and meta[1][0] + meta[1][1] == frames
    hop = O.hop(n, overlap)
    x = synth(frames * hop, seed=11)
    full = O.spectrogram_fft(x, n, overlap, 0, sub_mean=sub_mean) if mode == "fft" else \
        O.spectrogram_mtm(x, n, overlap, 2.5, 4, sub_mean=sub_mean)
    got = np.concatenate(parts, axis=0)
    assert got.shape == full.shape
    assert np.array_equal(got, full)           # same arithmetic on the same samples: identical rows


def test_frame_range_and_window_properties():
    from glfer_amd.shard import frame_range, sample_window
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 4, 8):
            spans = [frame_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            # dealt out in units of FRAME_ALIGN frames: every cut is a multiple of it (the kernels
            # that work on aligned groups of frames then give bit-identical rows), balance to one unit
            assert all(f % 32 == 0 or f == total for f, _ in spans)
            assert max(c for _, c in spans) - min(c for _, c in spans) < 64      # one unit + the partial last unit
            assert [frame_range(total, r, world, align=1) for r in range(world)][-1][0] <= total
    assert sample_window(0, 10, 1024, 4096) == (0, 10240)
    assert sample_window(10, 10, 1024, 4096) == (10 * 1024 - 3072, 20 * 1024)
    assert sample_window(10, 10, 1024, 4096, history_mode=1) == (10240, 20480)
    assert sample_window(5, 0, 1024, 4096) == (0, 0)
