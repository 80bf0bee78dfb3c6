"""cfg.sub_mean = 1 on C1 / C2: the separate means launch (default), round 4's fused launch, and round 5's LOCK-STEPPED fused launch
(GLFER_MEANS_PRODUCERS x GLFER_FUSED_BLOCK_FRAMES x GLFER_FUSED_LOOK), M frames/s on 2^30-sample streams with a DC level; rows compared
with the default path's.   python tools/lockstep_time.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import glfer_amd as G

quick = len(sys.argv) > 1
dev = torch.device("cuda", 0)


def rate(sp, x, out, reps=3):
    for _ in range(2):
        sp.run(x, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(3):
            sp.run(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 3)
    return out.shape[0] / best / 1e6


for wl in ("fft1k", "fft"):
    name, n, overlap, _, _, frames, _ = bench.WORKLOADS[wl]
    if quick:
        frames //= 4
    off = G.Spectrogram(bench.make_params(G, wl))
    x = bench.synth_on_device(torch, frames * off.hop, dev, seed=0, fs=8000.0 if wl == "fft1k" else 48000.0) + 0.1
    out = torch.empty((frames, off.bins), dtype=torch.float32, device=dev)
    r_off = rate(off, x, out)
    off.close()
    sp = G.Spectrogram(bench.make_params(G, wl, sub_mean=G.SUBMEAN_EXACT))
    for k in ("GLFER_MEANS_PRODUCERS", "GLFER_FUSED_BLOCK_FRAMES", "GLFER_FUSED_LOOK"):
        os.environ.pop(k, None)
    r_def = rate(sp, x, out)
    want = out[::1013].clone()
    last = out[-300:].clone()
    print("%s: mean removal off %.1f | separate means launch %.1f (%.2f)" % (wl, r_off, r_def, r_def / r_off), flush=True)
    os.environ["GLFER_MEANS_PRODUCERS"] = "512"
    r = rate(sp, x, out)
    print("   round 4's fused launch, 512 producers: %.1f (%.2f) rows equal: %s" % (r, r / r_off, torch.equal(out[::1013], want)), flush=True)
    grid = [(p_, b_, l_) for p_ in (64, 128, 256) for b_ in (16, 32, 64) for l_ in (1024, 4096)]
    if quick:
        grid = [(64, 64, 1 << 20), (128, 64, 1 << 20), (64, 32, 2048), (64, 64, 2048), (128, 32, 2048), (128, 64, 4096), (64, 16, 1024), (128, 16, 4096)]
    for prod, bf, look in grid:
        if True:
            if True:
                os.environ["GLFER_MEANS_PRODUCERS"] = str(prod)
                os.environ["GLFER_FUSED_BLOCK_FRAMES"] = str(bf)
                os.environ["GLFER_FUSED_LOOK"] = str(look)
                out.zero_()
                r = rate(sp, x, out, reps=2)
                ok = torch.equal(out[::1013], want) and torch.equal(out[-300:], last)
                print("   lock-stepped: producers %3d, %2d frames per consumer workgroup, look %4d hops: %.1f (%.2f) rows equal: %s" % (prod, bf, look, r, r / r_off, ok), flush=True)
    for k in ("GLFER_MEANS_PRODUCERS", "GLFER_FUSED_BLOCK_FRAMES", "GLFER_FUSED_LOOK"):
        os.environ.pop(k, None)
    sp.close()
    del x, out
    torch.cuda.empty_cache()
