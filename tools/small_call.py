"""Cost of a SMALL device-resident call (launch-bound territory): microseconds per call of Spectrogram.run on 1 / 8 / 64 frames, asynchronous
(calls queued back to back, one synchronize at the end) and synchronous (a synchronize per call).   python tools/small_call.py"""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

CASES = [("C1 fft N=1024 ovl 50%", G.FftParams, dict(n=1024, window_type=0, overlap=0.5)),
         ("C1 + mean (reference order)", G.FftParams, dict(n=1024, window_type=0, overlap=0.5, sub_mean=1)),
         ("C3 mtm N=4096 5 tapers", G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4)),
         ("C3 + mean (reference order)", G.MtmParams, dict(n=4096, overlap=0.0, w=2.5, kmax=4, sub_mean=1)),
         ("C4 mtm N=16384 9 tapers", G.MtmParams, dict(n=16384, overlap=0.0, w=4.5, kmax=8)),
         ("C5 hparma t=128 p_e=32", G.HparmaParams, dict(n=4096, overlap=0.0, t=128, p_e=32)),
         ("LMP N=1024 lmp_av 4", G.LmpParams, dict(n=1024, overlap=0.0, avg=4))]
for name, P, kw in CASES:
    sp = G.Spectrogram(P(**kw))
    line = "%-30s" % name
    for frames in (1, 8, 64):
        x = torch.randn(frames * sp.hop + (sp.n - sp.hop), device='cuda') * 0.2
        out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
        for _ in range(20):
            sp.run(x, out=out)
        torch.cuda.synchronize()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            sp.run(x, out=out)
        torch.cuda.synchronize()
        a = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(reps):
            sp.run(x, out=out)
            torch.cuda.synchronize()
        s = (time.perf_counter() - t0) / reps * 1e6
        line += "   %2d frames: %6.1f us queued, %6.1f us with a sync" % (frames, a, s)
    print(line)
    sp.close()
