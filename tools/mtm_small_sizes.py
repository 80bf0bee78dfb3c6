import sys, time
sys.path.insert(0, '.')
import torch
import glfer_amd as G
for n, nw, k, frames in ((256, 2.5, 4, 4194304), (256, 4.0, 7, 2097152), (512, 2.5, 4, 2097152), (1024, 2.5, 4, 1048576), (1024, 4.0, 7, 524288), (2048, 4.0, 7, 262144), (2048, 2.5, 3, 524288)):
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=0.0, w=nw, kmax=k))
    x = torch.randn(frames * sp.hop, device='cuda')
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
    tr = (k + 1) / 2.0
    print("MTM n=%d tapers=%d: %.1f M frames/s, %.0f GB/s algorithmic, %.0f M N-point transforms/s = %.0f M 4096-equivalents/s" % (n, k + 1, nf / best / 1e6, nf * (4 * sp.hop + 4 * sp.bins) / best / 1e9, nf * tr / best / 1e6, nf * tr / best / 1e6 * (n * (n.bit_length() - 1)) / (4096 * 12)), flush=True)
