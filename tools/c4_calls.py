"""Call by call: a 1-hour WAV through a workers handle, the phases of each of the first calls (timing events on), then calls without them.
python tools/c4_calls.py [workers]"""
import os, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, glfer_amd as G
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, nsamples = 16384, 3600 * 48000
path = "/dev/shm/glfer_c4_calls_%d.wav" % os.getpid()
x = bench.synth_on_device(torch, nsamples, torch.device("cuda", 0), seed=4)
pcm = (x * 32767.0).round().to(torch.int16).cpu().numpy()
del x
with open(path, "wb") as f:
    f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 48000, 96000, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes))
    pcm.tofile(f)
try:
    frames = nsamples // n
    params = G.MtmParams(n=n, overlap=0.0, w=4.5, kmax=8, sample_format=G.SAMPLES_S16)
    rows = G.pinned_empty((frames, n // 2 + 1), np.float32)
    t0 = time.perf_counter()
    W = G.Workers(params, [0] * w, hint_frames=frames)
    print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    for i in range(6):
        t0 = time.perf_counter()
        nf, ph = W.run_wav(path, rows, phases=(i < 4))
        dt = time.perf_counter() - t0
        print("call %d: %.2f ms  %s" % (i, dt * 1e3, {k: round(v * 1e3, 2) if isinstance(v, float) else v for k, v in (ph or {}).items()}), flush=True)
    time.sleep(0.5)
    for i in range(3):
        t0 = time.perf_counter()
        W.run_wav(path, rows, phases=False)
        print("after 0.5 s idle, call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    W.close()
    # is the slow first call the handle's, the process's, or the caller's row buffer's?  a SECOND handle in the same process, the same rows ...
    W2 = G.Workers(params, [0] * w, hint_frames=frames)
    for i in range(3):
        t0 = time.perf_counter()
        nf, ph = W2.run_wav(path, rows)
        print("second handle, call %d: %.2f ms  %s" % (i, (time.perf_counter() - t0) * 1e3, {k: round(v * 1e3, 2) if isinstance(v, float) else v for k, v in ph.items()}), flush=True)
    # ... and a fresh row buffer with the second handle (warm handle, cold rows)
    rows2 = G.pinned_empty((frames, n // 2 + 1), np.float32)
    for i in range(3):
        t0 = time.perf_counter()
        nf, ph = W2.run_wav(path, rows2)
        print("warm handle, fresh pinned rows, call %d: %.2f ms  %s" % (i, (time.perf_counter() - t0) * 1e3, {k: round(v * 1e3, 2) if isinstance(v, float) else v for k, v in ph.items()}), flush=True)
    # ... and a fresh FILE (same bytes, written now) with warm handle and warm rows
    path2 = path + ".2"
    with open(path, "rb") as fi, open(path2, "wb") as fo:
        fo.write(fi.read())
    for i in range(3):
        t0 = time.perf_counter()
        nf, ph = W2.run_wav(path2, rows2)
        print("warm handle, warm rows, fresh file, call %d: %.2f ms  %s" % (i, (time.perf_counter() - t0) * 1e3, {k: round(v * 1e3, 2) if isinstance(v, float) else v for k, v in ph.items()}), flush=True)
    os.unlink(path2)
    W2.close()
finally:
    os.unlink(path)
