"""Periodogram throughput by block size (f32, Hanning, 50 % overlap), for A/B runs of two builds:
GLFER_LIB_PATH=<other libglfer_hip.so> python3 tools/fft_sizes.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"):
    A.LIB_PATH = os.environ["GLFER_LIB_PATH"]
FMT = os.environ.get("GLFER_FMT", "f32")
for n, frames in ((512, 2097152), (1024, 1048576), (2048, 524288), (4096, 524288), (8192, 131072), (16384, 65536)):
    sp = G.Spectrogram(G.FftParams(n=n, overlap=0.5, window_type=0,
                                   sample_format={"f32": G.SAMPLES_F32, "s16": G.SAMPLES_S16, "u8": G.SAMPLES_U8}[FMT]))
    x = torch.randn(frames * sp.hop + (n - sp.hop), device='cuda') * 0.2
    if FMT == "s16": x = (x * 32767).clamp(-32768, 32767).to(torch.int16)
    if FMT == "u8": x = (x * 127 + 128).clamp(0, 255).to(torch.uint8)
    out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
    best = 1e9
    for rep in range(3):
        for _ in range(2): sp.run(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): sp.run(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    nf = out.shape[0]
    print("FFT " + FMT + " n=%d overlap=0.50: %.1f M frames/s, %.0f GB/s algorithmic" % (n, nf / best / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / best / 1e9), flush=True)
