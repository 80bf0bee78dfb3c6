"""What the box's PCIe link does with pinned memory: H2D alone, D2H alone, both at once on two streams, by transfer size.
    python tools/pcie_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
for mb in (16, 64, 256):
    n = mb << 20
    h_up, h_dn = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    d_up, d_dn = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = max(4, 2048 // mb)

    def run(up, dn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if up:
                with torch.cuda.stream(s1):
                    d_up.copy_(h_up, non_blocking=True)
            if dn:
                with torch.cuda.stream(s2):
                    h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9

    run(True, True)
    u, d, b = run(True, False), run(False, True), run(True, True)
    print("%4d MiB transfers: H2D alone %5.1f GB/s   D2H alone %5.1f GB/s   both at once %5.1f + %5.1f GB/s" % (mb, u, d, b, b))

# the ingest ring's shape (ingest.cpp run_job): two streams, chunk c on stream c % 2 = [upload, kernel, download], the stream drained
# before its buffers are reused
mb = 128
n = mb << 20
chunks = 16
h_src = torch.empty(chunks * n, dtype=torch.uint8).pin_memory()
h_dst = torch.empty(chunks * n, dtype=torch.uint8).pin_memory()
d_in = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
d_out = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
st = [torch.cuda.Stream(), torch.cuda.Stream()]
for kernel in (False, True):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(chunks):
            b = c & 1
            st[b].synchronize()
            with torch.cuda.stream(st[b]):
                d_in[b].copy_(h_src[c * n:(c + 1) * n], non_blocking=True)
                if kernel:
                    d_out[b].copy_(d_in[b])
                h_dst[c * n:(c + 1) * n].copy_(d_out[b], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("ring of two streams, %d chunks of %d MiB up and down%s: %.1f + %.1f GB/s" % (chunks, mb, ", a device copy between" if kernel else "", chunks * n / dt / 1e9, chunks * n / dt / 1e9))
