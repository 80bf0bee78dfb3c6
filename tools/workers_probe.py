"""glfer_hip_spectrogram_host_workers with 1, 2, 4, 8 workers sharing ONE GPU (the host side of the multi-GPU entry: a thread, a plan
and a chunk ring per worker; the GPU and its link are shared here, so the rate should hold, not scale), pinned ends; rows compared
with the one-worker run.   python tools/workers_probe.py
Round 4 measured 5.15 / 4.31 / 3.62 / 3.09 M frames/s with 1 / 2 / 4 / 8 workers: a device parked ONE chunk ring between calls, so all but one of the
workers that share the GPU allocated their pinned and device buffers inside every call.  Round 5: the entry keeps its workers (a plan and a ring EACH)
between calls, and a glfer_hip_workers handle does so explicitly -- both are timed here."""
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
    W = G.Workers(params, [0] * workers, hint_frames=frames)
    hb = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        nf, _ = W.run_host(pcm, rows, phases=False)
        hb = min(hb, time.perf_counter() - t0)
    _, ph = W.run_host(pcm, rows)
    W.close()
    assert nf == frames and np.array_equal(ref, rows[::997]), workers
    print("%d worker(s) on one GPU: stateless entry %.2f M frames/s (%.1f GB/s over PCIe both ways), through a handle %.2f M frames/s; phases of one more call (with timing events) %s"
          % (workers, frames / best / 1e6, frames * (8192 + 8196) / best / 1e9, frames / hb / 1e6, {k: round(v, 4) if isinstance(v, float) else v for k, v in ph.items()}), flush=True)
