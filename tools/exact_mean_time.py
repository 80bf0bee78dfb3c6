"""Throughput with per-hop mean removal off / the reference's own summation order (cfg.sub_mean = 1 = GLFER_SUBMEAN_EXACT:
hop means by submean_seq.hip, handed to the kernels as a table, piece by piece beside the estimator launches) / the kernels'
own sums (GLFER_SUBMEAN_FAST), 2^30-sample f32 streams with a DC offset.
    python tools/exact_mean_time.py            the table DESIGN quotes (product defaults)
    python tools/exact_mean_time.py sweep      + the knobs of glfer_hip.cpp launch_body_with_reference_means
                                                 (GLFER_EXACT_PIECE_MB x GLFER_EXACT_STREAMS x GLFER_MEANS_HPW x GLFER_MEANS_BLOCKS)"""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
CASES = (("C1 periodogram N=1024 50%", G.FftParams, dict(n=1024, window_type=0, overlap=0.5)),
         ("C2 periodogram N=4096 75%", G.FftParams, dict(n=4096, window_type=0, overlap=0.75)),
         ("C3 multitaper N=4096 5 tapers", G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4)),
         ("C3 at 75 %", G.MtmParams, dict(n=4096, overlap=0.75, w=2.5, kmax=4)),
         ("glfer default: N=1024 Kaiser ovl 0", G.FftParams, dict(n=1024, window_type=7, overlap=0.0)),
         ("glfer default MTM: N=1024 8 tapers ovl 0", G.MtmParams, dict(n=1024, overlap=0.0, w=4.0, kmax=7)),
         ("C4 multitaper N=16384 9 tapers", G.MtmParams, dict(n=16384, overlap=0.0, w=4.5, kmax=8)))
KNOBS = ("GLFER_EXACT_PIECE_MB", "GLFER_EXACT_STREAMS", "GLFER_MEANS_HPW", "GLFER_MEANS_BLOCKS")


def rate(cls, kw, mode, env=None):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    sp = G.Spectrogram(cls(sub_mean=mode, **kw))
    frames = min((1 << 30) // sp.hop, 1 << 21)
    x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2 + 0.1
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    best = 1e9
    for rep in range(3):
        sp.run(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            sp.run(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 4)
    n = out.shape[0]
    del x, out, sp
    torch.cuda.empty_cache()
    for k in KNOBS:
        os.environ.pop(k, None)
    return n / best / 1e6


sweep = len(sys.argv) > 1 and sys.argv[1] == "sweep"
big = len(sys.argv) > 1 and sys.argv[1] == "big"          # a few LARGE pieces on two / three streams: does the means pass hide beside the estimator?
for name, cls, kw in (CASES[:3] if (sweep or big) else CASES):
    off, exact, fast = (rate(cls, kw, m) for m in (G.SUBMEAN_OFF, G.SUBMEAN_EXACT, G.SUBMEAN_FAST))
    print("%-44s  off %8.1f M  reference order %8.1f M (%.3f of the in-kernel sums)  in-kernel sums %8.1f M"
          % (name, off, exact, exact / fast, fast), flush=True)
    if big:
        for streams in (1, 2, 3):
            for piece in (512, 1024, 2048):
                for hpw in (64, 16):
                    r = rate(cls, kw, G.SUBMEAN_EXACT, dict(GLFER_EXACT_PIECE_MB=piece, GLFER_EXACT_STREAMS=streams, GLFER_MEANS_HPW=hpw))
                    print("    streams %d  piece %4d MB  hops/wavefront %2d: %8.1f M (%.3f)" % (streams, piece, hpw, r, r / fast), flush=True)
        continue
    if not sweep:
        continue
    for streams in (1, 2, 3):
        for piece in ((0,) if streams == 1 else ()) + (16, 32, 48, 64, 96, 128):
            for hpw in (64, 16, 4):
                for blocks in ((0,) if streams == 1 else (0, 64, 128, 256)):
                    if hpw == 64 and blocks:
                        continue
                    r = rate(cls, kw, G.SUBMEAN_EXACT, dict(GLFER_EXACT_PIECE_MB=piece, GLFER_EXACT_STREAMS=streams, GLFER_MEANS_HPW=hpw,
                                                           GLFER_MEANS_BLOCKS=blocks))
                    print("    streams %d  piece %4d MB  hops/wavefront %2d  means blocks %4d: %8.1f M (%.3f)"
                          % (streams, piece, hpw, blocks, r, r / fast), flush=True)
