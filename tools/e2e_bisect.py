"""Why bench.py's end_to_end row is slower after torch has cached device memory (tools/e2e_alone.py): which allocation's timing matters."""
import ctypes as C
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch, bench
import glfer_amd as G

frames = 131072
torch.cuda.set_device(0)

def buffers():
    pcm = G.pinned_empty((frames * 4096,), np.int16)
    pcm[:] = (np.random.default_rng(1).standard_normal(pcm.size) * 6000).clip(-32768, 32767).astype(np.int16)
    rows = G.pinned_empty((frames, 2049), np.float32)
    return pcm, rows

def rate(tag, pcm, rows, new_plan=True, sp=[None]):
    if new_plan or sp[0] is None:
        if sp[0] is not None:
            sp[0].close()
        sp[0] = G.Spectrogram(G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=G.SAMPLES_S16), device=0)
    best = 1e9
    for r in range(3):
        nf = C.c_size_t(0)
        t0 = time.perf_counter()
        rc = G.api.lib().glfer_hip_spectrogram_host(sp[0]._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
        dt = time.perf_counter() - t0
        assert rc == 0
        if r:
            best = min(best, dt)
    print("%-70s %.2f M frames/s" % (tag, frames / best / 1e6), flush=True)

early = buffers()
rate("fresh process, buffers A", *early)
bench.measure(torch, G, None, "mtm", 0, 3, 1, 1, 0, 0, False)
rate("after measure(mtm): buffers A (pinned before), new plan", *early)
late = buffers()
rate("after measure(mtm): buffers B (pinned now), new plan", *late)
rate("after measure(mtm): buffers A again", *early)
print("torch reserved %.1f GiB" % (torch.cuda.memory_reserved() / 2**30))
G.api.lib().glfer_hip_scratch_trim(0, C.c_size_t(0))
rate("after scratch_trim(0) (ring dropped, rebuilt): buffers B", *late)
torch.cuda.empty_cache()
rate("after empty_cache: buffers B", *late)
rate("after empty_cache: buffers A", *early)

print("---- live device memory and the host path")
for gib in (1, 2, 3, 8):
    t = torch.empty(gib << 30, dtype=torch.uint8, device="cuda")
    rate("a live %d GiB torch tensor (untouched)" % gib, *late)
    t.zero_()
    torch.cuda.synchronize()
    rate("the same, written once" , *late)
    del t
    torch.cuda.empty_cache()
    rate("freed again", *late)
