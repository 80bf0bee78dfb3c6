"""bench.py's end_to_end row after everything bench.py does before it, repeated: does the slow state pass?"""
import sys, time
sys.path.insert(0, '.')
import torch, bench
import glfer_amd as G

torch.cuda.set_device(0)
for wl, k in (("mtm", 20),) + bench.SECONDARY:
    bench.measure(torch, G, None, wl, 0, k, 3, 1, 0, 0, False)
for wl, k in (("fft1k", 5), ("fft", 5), ("mtm", 5)):
    bench.measure(torch, G, None, wl, 0, k, 3, 1, 0, 0, False, params_kw=dict(sub_mean=G.SUBMEAN_EXACT), dc=0.1)
bench.measure(torch, G, None, "fft", 0, 5, 3, 1, 0, 0, False, params_kw=dict(psd_pitch=2112))
bench.measure(torch, G, None, "fft", 262144, 5, 3, 1, 0, 0, False, avg_depth=4)
bench.parity_vs_oracle(torch, G, "mtm", 0)
print("reserved %.1f GiB, allocated %.1f GiB" % (torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30), flush=True)
for i in range(4):
    t0 = time.perf_counter()
    r = bench.end_to_end(torch, G, 0, reps=3)
    print("end_to_end call %d: %.2f M frames/s  %.1f GB/s   (%.1f s)" % (i, r["value"] / 1e6, r["pcie_gbs_both_directions"], time.perf_counter() - t0), flush=True)
