"""Throughput of the estimator's other modes and outputs on BASELINE-sized streams: LMP (lmp.c), the harmonic F-test
(mtm.c:165-233), halfcomplex spectra, prepare_audio frames, HP-ARMA; each beside the plain PSD run it is built on."""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G

def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def line(label, frames, dt, bytes_per_frame):
    print("%-58s %9.2f M frames/s  %6.0f GB/s algorithmic (%.3f ms)" % (label, frames / dt / 1e6, frames * bytes_per_frame / dt / 1e9, dt * 1e3), flush=True)

n = 4096
for ovl, frames in ((0.75, 1 << 19), (0.0, 1 << 18)):
    sp = G.Spectrogram(G.FftParams(n=n, window_type=0, overlap=ovl))
    x = torch.rand(frames * sp.hop + n, device='cuda') - 0.5
    out = torch.empty((frames, sp.bins), device='cuda')
    line("periodogram N=4096 overlap %.2f" % ovl, frames, timed(lambda: sp.run(x, nframes=frames, out=out)), 4 * sp.hop + 4 * sp.bins)
    lm = G.Spectrogram(G.LmpParams(n=n, overlap=ovl, avg=4))
    line("LMP avg=4 N=4096 overlap %.2f" % ovl, frames, timed(lambda: lm.run(x, nframes=frames, out=out)), 4 * sp.hop + 4 * sp.bins)
    line("periodogram + halfcomplex spectrum, overlap %.2f" % ovl, frames, timed(lambda: sp.run(x, nframes=frames, spectrum=True)), 4 * sp.hop + 4 * sp.bins + 4 * n)
    line("prepare_audio frames (inbuf_fft), overlap %.2f" % ovl, frames, timed(lambda: sp.prepare(x, nframes=frames)), 4 * sp.hop + 4 * n)
frames = 1 << 17
mt = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=2.5, kmax=4))
x = torch.rand(frames * mt.hop + n, device='cuda') - 0.5
out = torch.empty((frames, mt.bins), device='cuda')
line("multitaper N=4096 5 tapers", frames, timed(lambda: mt.run(x, nframes=frames, out=out)), 4 * mt.hop + 4 * mt.bins)
line("multitaper F statistic (mu live)", frames, timed(lambda: mt.ftest(x, nframes=frames)), 4 * mt.hop + 4 * mt.bins)
