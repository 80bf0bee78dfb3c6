"""Periodogram throughput by block size and overlap (f32, or GLFER_FMT=s16|u8; Hanning) for A/B runs:
GLFER_LIB_PATH=<other libglfer_hip.so> python3 tools/fft_policy_sizes.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
if os.environ.get("GLFER_LIB_PATH"):
    A.LIB_PATH = os.environ["GLFER_LIB_PATH"]
tag = os.path.basename(os.path.dirname(os.environ.get("GLFER_LIB_PATH", "x/product/lib")))
for n in [int(v) for v in os.environ.get("GLFER_SIZES", "4096,8192,16384").split(",")]:
    for overlap in (0.0, 0.5, 0.75):
        hop = int(n * (1 - overlap))
        frames = (1 << 29) // hop
        sp = G.Spectrogram(G.FftParams(n=n, overlap=overlap, window_type=0, sample_format={'s16': A.SAMPLES_S16, 'u8': A.SAMPLES_U8}.get(os.environ.get('GLFER_FMT'), A.SAMPLES_F32)))
        x = torch.randn(frames * sp.hop + (n - sp.hop), device='cuda') * 0.2
        if os.environ.get("GLFER_FMT") == "s16": x = (x * 32768.0).clamp(-32768, 32767).to(torch.int16)
        if os.environ.get("GLFER_FMT") == "u8": x = (x * 128.0 + 128.0).clamp(0, 255).to(torch.uint8)
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
        print("%-8s n=%5d overlap=%.2f: %7.1f M frames/s, %5.0f GB/s algorithmic" % (tag, n, overlap, nf / best / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / best / 1e9), flush=True)
        del x, out
