"""C3 / C2 / C4 by sample format (f32, s16, u8): the rate of the estimator kernels when the stream is the 16-bit PCM a WAV file
holds (half the sample bytes; the conversion sits in the gather).  python3 tools/formats_time.py"""
import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
import glfer_amd.api as A
CASES = (("C3 mtm N=4096 5 tapers overlap 0", lambda sf: G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=sf), 262144),
         ("C3 at 75 % overlap", lambda sf: G.MtmParams(n=4096, overlap=0.75, w=2.5, kmax=4, sample_format=sf), 262144),
         ("C2 fft N=4096 Hanning 75 %", lambda sf: G.FftParams(n=4096, overlap=0.75, window_type=0, sample_format=sf), 1048576),
         ("C1 fft N=1024 Hanning 50 %", lambda sf: G.FftParams(n=1024, overlap=0.5, window_type=0, sample_format=sf), 2097152),
         ("C4 mtm N=16384 9 tapers overlap 0", lambda sf: G.MtmParams(n=16384, overlap=0.0, w=4.5, kmax=8, sample_format=sf), 65536))
for name, mk, frames in CASES:
    row = []
    for fmt, sf in (("f32", A.SAMPLES_F32), ("s16", A.SAMPLES_S16), ("u8", A.SAMPLES_U8)):
        sp = G.Spectrogram(mk(sf))
        ns = frames * sp.hop + (sp.n - sp.hop)
        x = torch.randn(ns, device='cuda') * 0.2
        if fmt == "s16": x = (x * 32768.0).clamp(-32768, 32767).to(torch.int16)
        if fmt == "u8": x = (x * 128.0 + 128.0).clamp(0, 255).to(torch.uint8)
        out = torch.empty((sp.num_frames(x.numel()), sp.bins), device='cuda')
        best = 1e9
        for rep in range(3):
            for _ in range(3): sp.run(x, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): sp.run(x, out=out)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 8)
        row.append("%s %8.2f" % (fmt, out.shape[0] / best / 1e6))
        del x, out
        sp.close()
    print("%-36s M frames/s:  %s" % (name, "   ".join(row)), flush=True)
