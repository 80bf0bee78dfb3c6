"""bench.py's end_to_end row alone and after each thing bench.py does before it (it read 3.8 M frames/s inside the full run, 5.06 M alone)."""
import sys
sys.path.insert(0, '.')
import torch, bench
import glfer_amd as G

def e2e(tag):
    r = bench.end_to_end(torch, G, 0, reps=2)
    print("%-44s %.2f M frames/s  %.1f GB/s" % (tag, r["value"] / 1e6, r["pcie_gbs_both_directions"]), flush=True)

torch.cuda.set_device(0)
e2e("fresh process")
res = bench.measure(torch, G, None, "mtm", 0, 3, 1, 1, 0, 0, False)
e2e("after measure(mtm)")
torch.cuda.empty_cache()
e2e("after empty_cache")
par = bench.parity_vs_oracle(torch, G, "mtm", 0)
e2e("after parity_vs_oracle")
for wl in ("fft1k", "fft", "mtm16k", "hparma"):
    bench.measure(torch, G, None, wl, 0, 2, 1, 1, 0, 0, False)
    e2e("after measure(%s)" % wl)
torch.cuda.empty_cache()
e2e("after empty_cache")
