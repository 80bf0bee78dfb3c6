"""Throughput with per-hop mean removal off / in-kernel sums (GLFER_SUBMEAN_FAST) / the reference's own summation order
(GLFER_SUBMEAN_EXACT: hop_means_seq_kernel + the means table handed to the kernels, or the corrected copy where a form
takes no table), 2^30-sample f32 streams."""
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
for name, cls, kw in CASES:
    line = "%-44s" % name
    for mode in (0, 1, 2):
        sp = G.Spectrogram(cls(sub_mean=mode, **kw))
        frames = min((1 << 30) // sp.hop, 1 << 21)
        x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2 + 0.1
        out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
        best = 1e9
        for rep in range(3):
            sp.run(x, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4): sp.run(x, out=out)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4)
        line += "  %s %8.1f M" % (("off", "fast", "exact")[mode], out.shape[0] / best / 1e6)
        del x, out, sp
        torch.cuda.empty_cache()
    print(line, flush=True)
