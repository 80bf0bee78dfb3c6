"""M frames/s of the per-bin statistics of SURVEY 8(f-4): LMP (lmp.c:101-181) and the harmonic F-test (mtm.c:165-233), device-resident.
    python tools/stats_rate.py"""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

def best_of(fn, reps=3):
    b = 1e9
    for _ in range(reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b

for n, ovl, avg in ((1024, 0.0, 4), (4096, 0.75, 4), (4096, 0.0, 16)):
    sp = G.Spectrogram(G.LmpParams(n=n, overlap=ovl, avg=avg))
    frames = min((1 << 28) // sp.hop, 1 << 20)
    x = torch.randn(frames * sp.hop + (n - sp.hop), device='cuda') * 0.2
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    dt = best_of(lambda: sp.run(x, out=out))
    nb = out.shape[0] * (4 * sp.hop + 4 * sp.bins)
    print("LMP N %5d overlap %.2f lmp_av %2d: %8.1f M frames/s   (samples in + statistic out: %.2f TB/s)" % (n, ovl, avg, out.shape[0] / dt / 1e6, nb / dt / 1e12))
    sp.close()
for n, kmax in ((4096, 4), (1024, 7)):
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=2.5 if kmax == 4 else 4.0, kmax=kmax))
    frames = min((1 << 28) // sp.hop, 1 << 18)
    x = torch.randn(frames * sp.hop, device='cuda') * 0.2
    dt = best_of(lambda: sp.ftest(x))
    print("F-test N %5d %d tapers: %8.1f M frames/s" % (n, kmax + 1, frames / dt / 1e6))
    sp.close()
