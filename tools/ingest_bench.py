"""PCIe-inclusive rates of the host-side entries (never bench.py's `value`): a WAV file streamed from the
page cache through glfer_hip_spectrogram_wav, and a host buffer through glfer_hip_spectrogram_host,
both with the PSD rows copied back to host memory."""
import os, struct, sys, time
sys.path.insert(0, '.')
import numpy as np
import glfer_amd as G

def write_wav(path, raw, rate):
    bits = 16 if raw.dtype == np.int16 else 8
    data = raw.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * bits // 8, bits // 8, bits) + b"data" + struct.pack("<I", len(data)))
        f.write(data)

rng = np.random.default_rng(1)
frames = 131072
for name, params in (("C2 periodogram n=4096 overlap 0.75", dict(kind="fft", n=4096, overlap=0.75)),
                     ("C3 multitaper n=4096 5 tapers overlap 0", dict(kind="mtm", n=4096, overlap=0.0))):
    hop = int(4096 * (1 - params["overlap"]))
    ns = frames * hop
    raw = (rng.standard_normal(ns) * 6000).clip(-32768, 32767).astype(np.int16)
    path = "/dev/shm/glfer_ingest_%d.wav" % os.getpid()
    write_wav(path, raw, 48000)
    mk = (lambda f: G.FftParams(n=4096, overlap=params["overlap"], window_type=0, sample_format=f)) if params["kind"] == "fft" \
        else (lambda f: G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4, sample_format=f))
    sp = G.Spectrogram(mk(G.SAMPLES_S16))
    sp.run_wav(path, max_frames=4096)
    t0 = time.perf_counter(); out = sp.run_wav(path); dt = time.perf_counter() - t0
    print("%s, s16 WAV (%.0f MiB) -> host PSD rows (%.0f MiB): %.2f M frames/s, %.1f GB/s over PCIe both ways"
          % (name, ns * 2 / 2**20, out.nbytes / 2**20, out.shape[0] / dt / 1e6, (ns * 2 + out.nbytes) / dt / 1e9), flush=True)
    os.unlink(path)
    sp.run_host(raw[: 4096 * hop])
    t0 = time.perf_counter(); out = sp.run_host(raw); dt = time.perf_counter() - t0
    print("%s, s16 host buffer -> host PSD rows: %.2f M frames/s, %.1f GB/s over PCIe both ways"
          % (name, out.shape[0] / dt / 1e6, (ns * 2 + out.nbytes) / dt / 1e9), flush=True)
    # (that call also grew the plan's ring to the job's chunk size: pinned staging of 2 x 256 MiB; the same call again:)
    t0 = time.perf_counter(); out = sp.run_host(raw); dt = time.perf_counter() - t0
    print("%s, s16 host buffer -> host PSD rows, ring at size: %.2f M frames/s, %.1f GB/s over PCIe both ways"
          % (name, out.shape[0] / dt / 1e6, (ns * 2 + out.nbytes) / dt / 1e9), flush=True)
    del out
    # rows into pinned memory (glfer_hip_host_alloc): DMA straight into the caller's array, no staging copy
    import ctypes as C
    pin = G.PinnedArray((frames, sp.bins), np.float32)
    nf = C.c_size_t(0)
    lib = G.api.lib()
    lib.glfer_hip_spectrogram_host(sp._h, raw.ctypes.data, raw.size, pin.ptr, C.byref(nf))
    t0 = time.perf_counter()
    lib.glfer_hip_spectrogram_host(sp._h, raw.ctypes.data, raw.size, pin.ptr, C.byref(nf))
    dt = time.perf_counter() - t0
    print("%s, s16 host buffer -> PINNED host PSD rows: %.2f M frames/s, %.1f GB/s over PCIe both ways"
          % (name, nf.value / dt / 1e6, (ns * 2 + pin.nbytes) / dt / 1e9), flush=True)
    # ... and the samples in pinned memory too: uploaded from where they lie
    pin_in = G.PinnedArray((raw.size,), np.int16)
    pin_in.array[:] = raw
    lib.glfer_hip_spectrogram_host(sp._h, pin_in.ptr, raw.size, pin.ptr, C.byref(nf))
    t0 = time.perf_counter()
    lib.glfer_hip_spectrogram_host(sp._h, pin_in.ptr, raw.size, pin.ptr, C.byref(nf))
    dt = time.perf_counter() - t0
    print("%s, PINNED s16 host buffer -> PINNED host PSD rows: %.2f M frames/s, %.1f GB/s over PCIe both ways"
          % (name, nf.value / dt / 1e6, (ns * 2 + pin.nbytes) / dt / 1e9), flush=True)
    pin_in.free()
    pin.free()
    # waterfall: RGB + levbuf back instead of float PSD (5 bytes per bin instead of 4 -- or 3 without levbuf)
    for want_lev in (True, False):
        d = G.Display(scale_type=G.SCALE_LOG, autoscale=1, overlap=params["overlap"], palette=0)
        sp.waterfall_host(raw[: 4096 * hop], d, want_lev=want_lev)
        d = G.Display(scale_type=G.SCALE_LOG, autoscale=1, overlap=params["overlap"], palette=0)
        t0 = time.perf_counter(); rgb, lev = sp.waterfall_host(raw, d, want_lev=want_lev); dt = time.perf_counter() - t0
        back = rgb.nbytes + (lev.nbytes if want_lev else 0)
        print("%s, s16 host buffer -> host RGB%s: %.2f M frames/s, %.1f GB/s over PCIe both ways"
              % (name, " + levbuf" if want_lev else "", rgb.shape[0] / dt / 1e6, (ns * 2 + back) / dt / 1e9), flush=True)
        del rgb, lev
