"""glfer_hip_spectrogram_host_workers with 1, 2, 4, 8 workers sharing ONE GPU (the host side of the multi-GPU entry: a thread, a plan
and a chunk ring per worker; the GPU and its link are shared here, so the rate should hold, not scale), pinned ends; rows compared
with the one-worker run.   python tools/workers_probe.py
Measured (round 4, one box): 5.15 / 4.31 / 3.62 / 3.09 M frames/s with 1 / 2 / 4 / 8 workers.  The fall is this rehearsal's own: a device parks ONE
chunk ring between calls (ingest_ring_take), so all but one of the workers that share the GPU allocate 2 x 256 MiB of pinned and device buffers
inside every call; with a worker per GPU -- what the entry is for -- each device has its parked ring."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import glfer_amd as G

frames = 262144
params = G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=G.SAMPLES_S16)
pcm = G.pinned_empty((frames * 4096,), np.int16)
pcm[:] = (np.random.default_rng(1).standard_normal(pcm.size) * 6000).clip(-32768, 32767).astype(np.int16)
rows = G.pinned_empty((frames, 2049), np.float32)
ref = None
for workers in (1, 2, 4, 8):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        out = G.spectrogram_host_multi(params, pcm, [0] * workers if workers > 1 else [0], out=rows)
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    assert out.shape[0] == frames
    if ref is None:
        ref = rows[::997].copy()
    else:
        assert np.array_equal(ref, rows[::997]), workers
    print("%d worker(s) on one GPU: %.2f M frames/s, %.1f GB/s over PCIe both ways" % (workers, frames / best / 1e6, frames * (8192 + 8196) / best / 1e9), flush=True)
