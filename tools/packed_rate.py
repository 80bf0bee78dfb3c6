"""M frames/s of the packed kernel's (spectro16.hip) forms: even taper counts and the general path (limiter), by block size.
    python tools/packed_rate.py            (GLFER_LIB_PATH selects a variant library for same-box A/B runs)"""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

CASES = [("mtm 4 tapers N=4096 ovl 0", G.MtmParams, dict(n=4096, overlap=0.0, w=2.0, kmax=3)),
         ("mtm 8 tapers N=4096 ovl 50%", G.MtmParams, dict(n=4096, overlap=0.5, w=4.0, kmax=7)),
         ("mtm 4 tapers N=4096 ovl 0 +mean(fast)", G.MtmParams, dict(n=4096, overlap=0.0, w=2.0, kmax=3, sub_mean=2)),
         ("fft N=4096 limiter ovl 75%", G.FftParams, dict(n=4096, window_type=0, overlap=0.75, limiter=1)),
         ("fft N=4096 limiter ovl 0", G.FftParams, dict(n=4096, window_type=0, overlap=0.0, limiter=1)),
         ("mtm 4 tapers N=512 ovl 0", G.MtmParams, dict(n=512, overlap=0.0, w=2.0, kmax=3)),
         ("mtm 4 tapers N=256 ovl 0", G.MtmParams, dict(n=256, overlap=0.0, w=2.0, kmax=3)),
         ("fft N=512 limiter ovl 50%", G.FftParams, dict(n=512, window_type=0, overlap=0.5, limiter=1)),
         ("mtm 4 tapers N=2048 ovl 0", G.MtmParams, dict(n=2048, overlap=0.0, w=2.0, kmax=3))]
for name, P, kw in CASES:
    sp = G.Spectrogram(P(**kw))
    frames = min((1 << 28) // sp.hop, 1 << 20)
    x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2
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
    print("%-40s %8.1f M frames/s" % (name, out.shape[0] / best / 1e6))
