import sys; sys.path.insert(0,'.')
import torch, bench, glfer_amd as G
frames = int(sys.argv[1]) if len(sys.argv)>1 else 262144
sp = G.Spectrogram(bench.make_params(G, "fft"))
x = bench.synth_on_device(torch, frames * sp.hop, torch.device("cuda", 0), seed=0)
rows = sp.run(x)
want_avg, want_ret = G.update_avg(G.AVG_PLAIN, rows, 4, 0, sp.bins)
avg, ret, psd = sp.run_avg(x, G.AVG_PLAIN, 4, 0, sp.bins, want_psd=True)
torch.cuda.synchronize()
bad_psd = (psd != rows).any(dim=1).nonzero().flatten()
bad_avg = (avg != want_avg).any(dim=1).nonzero().flatten()
bad_pk = (ret[:,1] != want_ret[:,1]).nonzero().flatten()
print("frames", frames, "bad psd rows", bad_psd.numel(), bad_psd[:10].tolist(), "bad avg rows", bad_avg.numel(), bad_avg[:10].tolist(), bad_avg[-5:].tolist(), "bad peak", bad_pk.numel(), bad_pk[:10].tolist())
if bad_avg.numel():
    f = int(bad_avg[0]); cols = (avg[f] != want_avg[f]).nonzero().flatten()
    print("row", f, "cols", cols.numel(), cols[:8].tolist(), avg[f, cols[:4]].tolist(), want_avg[f, cols[:4]].tolist())
