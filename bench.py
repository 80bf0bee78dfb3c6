#!/usr/bin/env python3
"""bench.py -- headline benchmark: spectrogram frames/s + achieved HBM GB/s.

Workload (BASELINE.json metric): multitaper, N=4096, NW=2.5, mtm_k=4 (kmax+1 = 5 DPSS tapers,
mtm.c:189), overlap 0 (reference default, glfer.c:239), float32 mono 48 kHz synthetic stream
resident in HBM.  One "step" = one pass of the hot path (glfer_hip_spectrogram_device) over
this rank's batch of frames.  Frames are independent, so N GPUs = N disjoint frame ranges and no
data-path collective (weak scaling: frames per GPU fixed); torch.distributed (RCCL) is used only
for the barrier and the max-over-ranks of the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload mtm|fft|fft1k|mtm16k|mtm75|hparma] [--no-secondary]

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (one child process per GPU, started before anything in this process touches the GPU) and
prints rank 0's line; under `python -m torch.distributed.run` it is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TOPS = 78.6        # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-instructions/s
# what a stream of independent v_fma_f32 / v_add_f32 actually issues on this chip (tools/kbench,
# profiles/r01_kbench_ablation.txt): 62.6 T lane-results/s at 8 wavefronts per SIMD, 54.4 at 2, 44.4 at 3
VALU_STREAM_TOPS = 62.6
CLOCK_GHZ = 2.4               # MI355X_MICROARCH.md peak engine clock: the issue fractions below are against it

# workload -> (name, n, overlap, nw, kmax, default frames per GPU per step, frames of the CPU sample stream)
WORKLOADS = {
    # C1 is the reference's own CPU-runnable case (`glfer -f`, glfer.c:238-239,254: N=1024, Hanning, 50 %, 8 kHz): a parity case
    # first, timed here as a secondary row because BASELINE.md 3 lists it beside the others
    "fft1k": ("C1: periodogram Hanning N=1024, overlap 50%, 8 kHz mono f32", 1024, 0.5, 0.0, 0, 2097152, 262144),
    "mtm": ("C3: multitaper N=4096 NW=2.5 mtm_k=4 (5 tapers), overlap 0, 48 kHz mono f32", 4096, 0.0, 2.5, 4, 262144, 16384),
    "fft": ("C2: periodogram Hanning N=4096, overlap 75%, 48 kHz mono f32", 4096, 0.75, 0.0, 0, 1048576, 131072),
    # SURVEY 8(d)'s secondary row: BASELINE.json does not give C3's overlap; the reference's default (0) is the headline, C2's 75 % this one
    "mtm75": ("C3 at C2's overlap: multitaper N=4096 NW=2.5 mtm_k=4 (5 tapers), overlap 75%, 48 kHz mono f32", 4096, 0.75, 2.5, 4, 262144, 16384),
    "mtm16k": ("C4: multitaper N=16384 NW=4.5 mtm_k=8 (9 tapers), overlap 0, 48 kHz mono f32", 16384, 0.0, 4.5, 8, 65536, 1024),
    "hparma": ("C5: HP-ARMA t=128 p_e=32 N=4096, overlap 0, 48 kHz mono f32 (compute/latency bound, not HBM)", 4096, 0.0, 0.0, 0, 65536, 4096),
}
METRICS = {"fft1k": "spectrogram frames/sec + achieved HBM GB/s, N=1024 periodogram",
           "mtm": "spectrogram frames/sec + achieved HBM GB/s, N=4096 MTM K=4",
           "fft": "spectrogram frames/sec + achieved HBM GB/s, N=4096 periodogram",
           "mtm75": "spectrogram frames/sec + achieved HBM GB/s, N=4096 MTM K=4, overlap 75 %",
           "mtm16k": "spectrogram frames/sec + achieved HBM GB/s, N=16384 MTM K=8",
           "hparma": "spectrogram frames/sec, HP-ARMA t=128 p_e=32 N=4096"}
# the kernel that takes the body of the launch (glfer_hip.cpp launch_by_n) and its committed PMC summary
KERNELS = {"fft1k": ("spectro16h_kernel<10, ..., SHIFT 8> (register reuse across overlapped frames)", None),
           "mtm": ("spectro16y_kernel", "hbm_traffic.json"), "fft": ("spectro16h_kernel<12, ..., SHIFT 4> (register reuse across overlapped frames)", "hbm_traffic_fft.json"),
           "mtm16k": ("spectro16w_kernel<14> (multitaper form)", "hbm_traffic_mtm16k.json"),
           "mtm75": ("spectro16y_kernel", None),
           "hparma": ("hparma_kernel", None)}


# what binds each row (confirmed by the counter passes under profiles/): the roofline object's `bound`
BOUND = {"fft1k": "hbm", "fft": "hbm", "mtm": "valu", "mtm75": "valu", "mtm16k": "valu", "hparma": "fp64-valu"}

# Context for a row's fraction.  NOTHING in these strings is measured by this run: every figure is quoted from the committed
# file named beside it (earlier rounds' ablation and counter passes); this run's own measurements are the numeric fields.
ROOFLINE_NOTES = {
    "mtm": "from profiles/ (not this run): not HBM-bound (traffic 1.00x algorithmic): FP32 VALU at two wavefronts per SIMD (248 VGPRs, "
           "2 x 35 KB LDS); a wavefront executes VALU 51 % / LDS 10 % of its time and waits 34.5 %, VALU pipe 47-48 %, LDS array 33 % busy "
           "(profiles/r03_stall_picture.txt); ceiling of this instruction stream with the exchange removed 98 M frames/s "
           "(profiles/r01_xbench_exchange_ablation.txt), with the samples served from L2 instead of HBM 93-95 M "
           "(profiles/r03_y_sample_misses.txt)",
    "mtm75": "from profiles/ (not this run): FP32-VALU-bound at two wavefronts per SIMD, as the headline (profiles/r03_stall_picture.txt)",
    "mtm16k": "from profiles/ (not this run): bound by its barrier and exchanges at two wavefronts per SIMD, VALU 41 % / waiting 45 % of a "
              "wavefront's time (profiles/r03_stall_picture.txt)",
    "fft": "from profiles/ (not this run): HBM-bound; a kernel that only streams this frame's 4 KB in : 8 KB out reaches 0.65-0.66 of the "
           "8 TB/s spec with aligned 16-byte stores and 0.58-0.61 into this interface's dense rows of 2049 floats, read-only 0.79 "
           "(tools/mixbench, profiles/r03_streaming_ceilings.txt)",
    "fft1k": "from profiles/ (not this run): HBM-bound; a kernel that only streams this frame's 2 KB in : 2 KB out reaches 0.63-0.64 of the "
             "8 TB/s spec (read-only 0.79, write-only 0.71, copy 0.69; profiles/r03_streaming_ceilings.txt)",
    "hparma": "not HBM-bound (24.6 KB per frame of algorithmic traffic): bound by the double-precision vector instructions of the Jacobi "
              "rotations, see valu (FP64)",
}


def synth_on_device(torch, nsamples, device, seed, fs=48000.0):
    """x[n] = 0.5 sin(2 pi 1000 n/fs) + 0.25 sin(2 pi 7350.5 n/fs) + 0.05 N(0,1), clipped."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(nsamples, dtype=torch.float32, device=device)
    chunk = 1 << 24
    for s in range(0, nsamples, chunk):
        e = min(nsamples, s + chunk)
        n = torch.arange(s, e, device=device, dtype=torch.float64)
        x = 0.5 * torch.sin(2 * torch.pi * 1000.0 / fs * n) + 0.25 * torch.sin(2 * torch.pi * 7350.5 / fs * n)
        x = x.float() + 0.05 * torch.randn(e - s, device=device, generator=g)
        out[s:e] = x.clamp_(-1.0, 0.9999999)
    return out


def cpu_baseline(workload, n, overlap, nw, kmax, frames):
    """The oracle (CPU restatement of fft_do+fft_psd / mtm_do, pinned to the reference's
    fft_radix2.c) timed on the host on a bounded prefix of the same workload: (i) one thread, the
    shape of the reference (source.c:130-170 is single-threaded); (ii) one independent estimator
    instance per thread over the host cores this process may use."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _signals import synth
    hop = O.hop(n, overlap)
    x = synth(frames * hop, seed=0)

    def one_pass(stream):
        if workload in ("mtm", "mtm16k", "mtm75"):
            O.spectrogram_mtm(stream, n, overlap, nw, kmax)
        elif workload == "hparma":
            O.spectrogram_hparma(stream, n, overlap, 128, 32)
        else:
            O.spectrogram_fft(stream, n, overlap, O.WINDOWS["hanning"])

    # (i) the same `frames`-frame stream is processed repeatedly until >= 10 s of CPU work is timed
    # (memory stays bounded; every pass is the full per-frame work of the reference path)
    done, dt = 0, 0.0
    while dt < 10.0:
        t0 = time.perf_counter()
        one_pass(x)
        dt += time.perf_counter() - t0
        done += frames
    rate1 = done / dt
    # (ii) every thread runs its own instance over a prefix sized for ~6 s at the one-thread rate
    # (ctypes releases the GIL; the oracle keeps no global state).  A 1-GPU box gives one job
    # 16 of the host's cores, so the pool is capped there.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    per_thread = max(1, min(frames, int(rate1 * 6.0)))
    part = x[:per_thread * hop + (n - hop)]
    per_thread = int(O.num_frames(part.size, n, overlap))
    with ThreadPoolExecutor(max_workers=cores) as pool:
        t0 = time.perf_counter()
        list(pool.map(one_pass, [part] * cores))
        dtn = time.perf_counter() - t0
    return {"value": rate1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same workload (a %d-frame stream, repeated), single thread, "
                      "oracle/glfer_oracle.c (gcc -O2, radix-2 recurrence FFT as fft_radix2.c), %.1f s"
                      % (done, frames, dt),
            "all_cores": {"value": cores * per_thread / dtn, "unit": "frames/s", "cores": cores, "kind": "port",
                          "sample": "%d threads (of %d host cores; a 1-GPU box's share is 16) x %d frames, one estimator "
                                    "instance per thread, %.1f s" % (cores, os.cpu_count() or 0, per_thread, dtn)},
            "host_cores_available": os.cpu_count()}


def self_launch(args, argv):
    """Start one rank per GPU as child processes and relay rank 0's JSON line.  Nothing here
    imports torch or touches HIP: the children are started before any GPU call of this process
    (a process that has initialised the GPU must never be replaced or forked)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read on a thread while every child is polled: a rank that dies (a bad device,
    # an RCCL error) must not leave the others waiting at a barrier for ever
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()                           # these exact children, by PID
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    for ln in out.decode(errors="replace").splitlines():       # the JSON line on stdout, anything else on stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return rc


def dry_run(args, world, rank):
    """--dry-run: the launch path only (rendezvous, barrier, max-over-ranks, rank 0 prints), on
    gloo, no GPU.  tests/test_bench_contract.py uses it to exercise `--gpus 2` as typed."""
    import torch
    import torch.distributed as dist
    if os.environ.get("GLFER_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    per_rank = [float(rank + 1)]
    host_phase = None
    if world > 1:
        dist.barrier()
        per_rank = gather_per_rank(torch, dist, t, world)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # the host-side phase of the real run: rank 0 works (here: sleeps), the others wait on the CPU
        host_phase = host_side_phase(dist, rank, None, lambda: {"slept_s": time.sleep(0.2) or 0.2})
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": t.item(), "steps": args.steps,
                          "warmup": args.warmup, "workload": args.workload, "per_rank": per_rank, "host_phase": host_phase}))


def gather_per_rank(torch, dist, t, world):
    """Every rank's own value of the one-element tensor t, in rank order (the MAX over them is what the contract reports)."""
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    return [x.item() for x in lst]


def host_side_phase(dist, rank, cpu_group, work):
    """"scaling risk is host-side only" -- so after the device-resident timing ONE process (rank 0) drives every GPU of
    the job through the host entries (one worker thread, plan and pinned ring per GPU: glfer_hip_spectrogram_host_multi,
    glfer_hip_spectrogram_wav_multi) while the other ranks wait at a barrier on the CPU (a gloo group: an RCCL barrier would
    keep their GPUs spinning under rank 0's workers).  Returns work()'s result on rank 0, None elsewhere."""
    out = None
    if rank == 0:
        try:
            out = work()
        except Exception as exc:                     # the line and the other ranks must not be lost to a host-side row
            out = {"error": "%s: %s" % (type(exc).__name__, exc)}
    dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()
    return out


def cpu_model():
    """The host CPU's model string (BASELINE.md 3 asks for it beside the CPU baseline)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def make_params(G, workload, **kw):
    name, n, overlap, nw, kmax, _, _ = WORKLOADS[workload]
    if workload in ("mtm", "mtm16k", "mtm75"):
        return G.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax, **kw)
    if workload == "hparma":
        return G.HparmaParams(n=n, overlap=overlap, t=128, p_e=32, **kw)
    return G.FftParams(n=n, window_type=G.WINDOWS["hanning"], overlap=overlap, **kw)


def pmc_summary(pmc_name):
    """Committed rocprofv3 PMC summary of a workload (newest round first): (dict, path) or (None, None)."""
    if pmc_name is None:
        return None, None
    for rnd in ("r05_", "r04_", "r03_", "r02_", "r01_"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", rnd + pmc_name))), "profiles/" + rnd + pmc_name
        except Exception:
            pass
    return None, None


def measure(torch, G, dist, workload, frames, steps, warmup, world, rank, local, rehearse, params_kw=None, avg_depth=0, dc=0.0, avg_form="two"):
    """W untimed + K timed passes of one workload over this rank's frame range; the timed region is
    bracketed by barrier + synchronize on both sides and the MAX over ranks is taken.  Returns a dict
    (every rank; only rank 0 uses it)."""
    name, n, overlap, nw, kmax, default_frames, _ = WORKLOADS[workload]
    dev = torch.device("cuda", local)
    frames = frames or default_frames               # a 2^30-sample stream per GPU
    fs = 8000.0 if workload == "fft1k" else 48000.0
    sp = G.Spectrogram(make_params(G, workload, **(params_kw or {})), device=local)
    hop, bins = sp.hop, sp.bins
    # Weak scaling: the job is world*frames frames of one long stream; this rank owns the
    # contiguous frame range frame_range() gives it and holds only the samples of its window
    # (its hops + the N-H history halo on the left; the stream starts with zero history).
    from glfer_amd.shard import frame_range, run_shard, sample_window
    first, count = frame_range(frames * world, rank, world)
    begin, end = sample_window(first, count, hop, n)
    shard = synth_on_device(torch, end - begin, dev, seed=rank, fs=fs)
    if dc:
        shard += dc                                   # a DC level: what mean removal is for (and where the order of the hop's sum shows)
    psd = torch.empty((count, sp.pitch), dtype=torch.float32, device=dev)       # (sp.pitch = bins unless cfg.psd_pitch)
    avg_out = None
    if avg_depth:                                     # update_avg_plain (avg.c:108-159) over the rows, the fused kernel: 8 B out per bin
        avg_out = (torch.empty((count, bins), dtype=torch.float64, device=dev), torch.empty((count, 4), dtype=torch.float64, device=dev))
    import ctypes as C

    def step():
        if avg_depth and avg_form != "two":
            # fft_do + fft_psd + update_avg_plain in ONE launch (glfer_hip_spectrogram_avg_device: the average taken on the |X|^2
            # values in registers); "fused": the PSD rows are never stored, "fused+rows": they are
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = G.api.lib().glfer_hip_spectrogram_avg_device(
                sp._h, C.c_void_p(shard.data_ptr() - begin * shard.element_size()), begin + shard.numel(), first, count, G.AVG_PLAIN, avg_depth, 0, bins, 0,
                bins, C.c_void_p(psd.data_ptr() if avg_form == "fused+rows" else None), C.c_void_p(avg_out[0].data_ptr()), C.c_void_p(avg_out[1].data_ptr()), st)
            assert rc == 0, rc
            return
        run_shard(sp, shard, begin, first, count, out=psd)
        if avg_depth:
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = G.api.lib().glfer_hip_avg_device(G.AVG_PLAIN, psd.data_ptr(), count, bins, bins, avg_depth, 0, bins, 0,
                                                  avg_out[0].data_ptr(), avg_out[1].data_ptr(), st)
            assert rc == 0, rc

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    # events on the stream the kernels are launched on (run_shard hands torch's current stream
    # to the C-ABI, and torch.cuda.Event records on that same stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                 # this rank's own K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    each = [a.elapsed_time(b) for a, b in ev]
    kernel_ms = sum(each) / steps
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
    per_rank_ms = [dt_own / steps * 1e3]
    if world > 1:
        per_rank_ms = [v / steps * 1e3 for v in gather_per_rank(torch, dist, torch.tensor([dt_own], dtype=torch.float64, device="cpu" if rehearse else dev), world)]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    dt_local = dt
    res = {"workload": workload, "name": name, "n": n, "overlap": overlap, "hop": hop, "bins": bins, "frames": frames, "dt_local": dt_local, "per_rank_ms": per_rank_ms,
           "ntapers": sp.ntapers, "dt": dt, "kernel_ms": kernel_ms, "fps": frames * world * steps / dt,
           "ms_per_step": dt / steps * 1e3, "b_alg": 4 * hop + 4 * bins,
           "kernel_ms_first": each[0], "kernel_ms_min": min(each), "kernel_ms_max": max(each)}
    res["achieved_gbs"] = frames * res["b_alg"] / (kernel_ms * 1e-3) / 1e9
    del shard, psd, avg_out
    sp.close()
    torch.cuda.empty_cache()
    return res


def valu_view(res):
    """FP32 VALU view of a launch: butterflies 3*M*log2(M) per complex M-point transform + inter-pass
    twiddles, taper multiply and |Z|^2.  MTM at N <= 4096: one N-point transform per taper PAIR, the odd
    taper shared by two frames (ntap/2 transforms per frame); periodogram and MTM at N >= 8192: one
    N/2-point transform per taper + the real-input split (8 ops per bin)."""
    import math
    n, ntap, frames, kernel_ms, workload = res["n"], res["ntapers"], res["frames"], res["kernel_ms"], res["workload"]
    m = n // 2
    real_input = 3 * m * math.log2(m) + 4 * (m - m // 64) + 2 * m + 8 * m
    if workload in ("fft", "fft1k"):
        lane_ops = real_input
    elif workload == "mtm16k":
        lane_ops = ntap * real_input
    else:
        lane_ops = (ntap / 2.0) * (3 * n * math.log2(n) + 4 * (n - n // 64) + 2 * n + 2 * n)
    tops = frames * lane_ops / (kernel_ms * 1e-3) / 1e12
    return {"lane_ops_per_frame": lane_ops, "achieved_Tops": tops, "peak_Tops": VALU_PEAK_TOPS, "frac": tops / VALU_PEAK_TOPS,
            "measured_stream_Tops": VALU_STREAM_TOPS, "frac_of_measured_stream": tops / VALU_STREAM_TOPS}


def hparma_view(res):
    """HP-ARMA is neither HBM- nor FP32-bound: a counted FP64 model of hparma_do (hparma.c:74-157 + compute_svd,
    util.c:261-386) at t = 128, p_e = 32 ON THIS STREAM: t lags x N double multiply-adds; the one-sided Jacobi SVD -- the
    three length-t dot products of every TESTED column pair and, for every pair actually ROTATED, two columns of the
    t x 33 matrix and of the 33 x 33 accumulator; Horner over N/2+1 bins x 33 coefficients, complex.  The loop runs until a
    sweep rotates nothing: on the bench stream 8-9 sweeps of 528 tested pairs and 2 567-2 959 rotations per frame
    (tools/hparma_sweep_count.py); the model takes the smallest (8 x 528 tested, 2 567 rotated): a LOWER bound.  (Rounds 1-3
    counted 12 full sweeps -- sweepmax's floor, which the loop does not wait for -- and so overstated the fraction by 1.7x.)"""
    n, frames, kernel_ms = res["n"], res["frames"], res["kernel_ms"]
    t_, ncol = 128, 33
    tested, rotated = 8 * (ncol * (ncol - 1) // 2), 2567
    flops = 2.0 * t_ * n + tested * (3 * 2 * t_) + rotated * (6 * t_ + 6 * ncol) + (n / 2 + 1) * ncol * 8
    tf = frames * flops / (kernel_ms * 1e-3) / 1e12
    view = {"model": "counted FP64 flops of hparma_do on this stream, lower bound (8 sweeps of 528 tested pairs, 2 567 rotations: tools/hparma_sweep_count.py)",
            "flops_per_frame_f64": flops,
            "achieved_TFLOPs": tf, "peak_TFLOPs": 78.6, "frac": tf / 78.6,
            "note": "bound by the ISSUE of double-precision vector instructions (4 clocks each on a SIMD): multiplies and adds the reference rounds "
                    "one by one (no fma in a rotation), float <-> double conversions and an angle per step are instructions, not counted flops.  "
                    "Round 4: column-disjoint rotations side by side (a static schedule of eight per step over 8-lane octets, 80 steps instead of 528 "
                    "rotations), the shape as compile-time constants (straight-line steps), the autocorrelation two lags a lane and four terms "
                    "a round, frames from a queue: 0.526 -> 2.13 M frames/s (profiles/r04_hparma_schedule.txt)"}
    pmc, src = pmc_summary("hparma_issue.json")
    if pmc:
        per_frame = pmc["SQ_INSTS_VALU_per_frame"]
        fps = frames / (kernel_ms * 1e-3)
        view["issue"] = {"valu_wave_instr_per_frame_measured": per_frame, "clocks_per_instr": 4,
                         "simd_clocks_per_s": 1024 * 2.4e9, "frac": per_frame * 4 * fps / (1024 * 2.4e9),
                         "is": "share of the chip's SIMD clocks (1024 SIMDs x 2.4 GHz nominal) this launch spent issuing vector instructions at 4 clocks each",
                         "source": "SQ_INSTS_VALU, rocprofv3 PMC (" + src + "), not this run"}
    return view


def stats_rows(torch, G, local, steps=5, warmup=3):
    """SURVEY 8(f-4)'s two per-bin statistics as rows of the line (device-resident, timed like the secondary rows, with events on the
    launch stream): LMP (lmp.c:101-181) at the reference's defaults (lmp_av = 4, glfer.c:252; N = 1024, overlap 0) and the harmonic
    F-test (mtm.c:165-233) on the headline's estimator (N = 4096, 5 tapers)."""
    dev = torch.device("cuda", local)
    rows = []

    def timed(fn, frames):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return frames * steps / dt, dt / steps * 1e3, sum(a.elapsed_time(b) for a, b in ev) / steps

    sp = G.Spectrogram(G.LmpParams(n=1024, overlap=0.0, avg=4), device=local)
    frames = 1 << 20
    x = synth_on_device(torch, frames * sp.hop, dev, seed=3, fs=8000.0)
    out = torch.empty((frames, sp.bins), dtype=torch.float32, device=dev)
    fps, ms, kms = timed(lambda: sp.run(x, out=out), frames)
    b_alg = 4 * sp.hop + 4 * sp.bins
    rows.append({"workload": "f-4: LMP detection statistic N=1024 lmp_av=4, overlap 0 (rectangular periodogram + lmp_ring_kernel)", "key": "lmp",
                 "value": fps, "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": ms, "kernel_ms": kms, "frames_per_step": frames,
                 "roofline": {"bound": "hbm", "achieved": frames * b_alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": frames * b_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": b_alg,
                              "note": "algorithmic bytes: the hop in, the statistic's row out; the launches also write the periodogram rows and read them back once (three row passes)"}})
    sp.close()
    del x, out
    sp = G.Spectrogram(G.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4), device=local)
    frames = 1 << 17
    x = synth_on_device(torch, frames * sp.hop, dev, seed=4)
    fout = torch.empty((frames, sp.bins), dtype=torch.float32, device=dev)
    import ctypes as C
    stq = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def ftest_once():
        rc = G.api.lib().glfer_hip_mtm_ftest_device(sp._h, x.data_ptr(), x.numel(), 0, frames, fout.data_ptr(), 1, stq)
        assert rc == 0, rc

    fps, ms, kms = timed(ftest_once, frames)
    rows.append({"workload": "f-4: harmonic F-test N=4096 NW=2.5 mtm_k=4, overlap 0 (six one-taper transforms per frame, fused)", "key": "ftest",
                 "value": fps, "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": ms, "kernel_ms": kms, "frames_per_step": frames,
                 "roofline": {"bound": "valu", "achieved": frames * (4 * sp.hop + 4 * sp.bins) / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": frames * (4 * sp.hop + 4 * sp.bins) / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "not HBM-bound: hn and the five tapers are six full N-point transforms of one real frame each (the statistic needs every taper's own spectrum), 2.4 x the headline's transform work"}})
    sp.close()
    del x, fout
    torch.cuda.empty_cache()
    return rows


def parity_vs_oracle(torch, G, workload, local, frames=64):
    """max|d|/max and L2 error of the HIP rows against the CPU oracle on the first
    `frames` frames of the workload's synthetic stream (seed 0) -- the checker beside the number, not
    in the timed region."""
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _signals import synth
    import numpy as np
    name, n, overlap, nw, kmax, _, _ = WORKLOADS[workload]
    sp = G.Spectrogram(make_params(G, workload), device=local)
    x = synth(frames * sp.hop, fs=8000.0 if workload == "fft1k" else 48000.0, seed=0)
    got = sp.run(torch.from_numpy(x).to("cuda:%d" % local)).cpu().numpy().astype(np.float64)
    sp.close()
    if workload in ("mtm", "mtm16k", "mtm75"):
        want = O.spectrogram_mtm(x, n, overlap, nw, kmax)
    elif workload == "hparma":
        want = O.spectrogram_hparma(x, n, overlap, 128, 32)
    else:
        want = O.spectrogram_fft(x, n, overlap, O.WINDOWS["hanning"])
    want = want.astype(np.float64)
    d = got - want
    per_frame_max = np.abs(d).max(axis=1) / np.abs(want).max(axis=1)
    per_frame_l2 = np.sqrt((d * d).sum(axis=1) / (want * want).sum(axis=1))
    return {"max_rel": float(per_frame_max.max()), "l2_rel": float(per_frame_l2.max()), "frames": int(got.shape[0]),
            "against": "oracle/glfer_oracle.c (CPU restatement pinned bit-exact to the reference's fft_radix2.c / g-l_dpss.c objects)",
            "norms": "per frame max|d|/max(ref) and ||d||2/||ref||2, worst frame", "tolerance": 1e-5}


def roofline_of(res, workload, traffic=None, traffic_src=None, kernel_ms_triplet=None):
    """The contract's roofline object.  achieved / peak / frac are ALGORITHMIC HBM bytes over the kernel's time against
    the 8 TB/s spec -- the fraction north_star asks for on every row -- and `bound` says which resource actually binds
    the row; where that is not HBM the binding resource's own fraction rides along as `binding`."""
    v = hparma_view(res) if workload == "hparma" else valu_view(res)
    r = {"bound": BOUND[workload], "achieved": res["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": res["achieved_gbs"] / HBM_PEAK_GBS, "traffic": traffic,
         "algorithmic_bytes_per_frame": res["b_alg"], "algorithmic_bytes_per_launch": res["frames"] * res["b_alg"],
         "kernel": KERNELS[workload][0], "kernel_ms": res["kernel_ms"], "note": ROOFLINE_NOTES.get(workload, "")}
    if traffic_src:
        r["traffic_unit"] = "bytes per launch, rocprofv3 PMC (%s)" % traffic_src
    if kernel_ms_triplet:
        r["kernel_ms_first_min_max"] = kernel_ms_triplet
    if BOUND[workload] == "valu":
        r["binding"] = {"resource": "fp32 vector ALU", "achieved": v["achieved_Tops"], "peak": v["peak_Tops"], "unit": "T lane-op/s",
                        "frac": v["frac"], "peak_is": "nominal: 256 CU x 4 SIMD x 32 lanes x %.1f GHz; the counter passes ran this kernel at "
                        "2.03-2.28 GHz (profiles/r03_stall_picture.txt)" % CLOCK_GHZ}
    elif BOUND[workload] == "fp64-valu":
        r["binding"] = {"resource": "fp64 vector ALU", "achieved": v["achieved_TFLOPs"], "peak": v["peak_TFLOPs"], "unit": "TFLOP/s (counted lower bound)",
                        "frac": v["frac"]}
    return r


def end_to_end(torch, G, local, frames=131072, reps=3):
    """SURVEY 8(d)'s separate row: the headline workload host memory to host memory -- a PINNED 16-bit PCM buffer (what a
    WAV file holds, wav_fmt.c:111-114) in, pinned float rows out, through glfer_hip_spectrogram_host's chunk ring (uploads on one stream, kernels and downloads on two)
    (source.c:112-171 as a batch).  PCIe-inclusive: never the headline value."""
    import numpy as np
    name, n, overlap, nw, kmax, _, _ = WORKLOADS["mtm"]
    sp = G.Spectrogram(G.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax, sample_format=G.SAMPLES_S16), device=local)
    x = synth_on_device(torch, frames * sp.hop, torch.device("cuda", local), seed=7)
    pcm = G.pinned_empty((frames * sp.hop,), np.int16)
    pcm[:] = (x * 32767.0).round().to(torch.int16).cpu().numpy()
    del x
    import ctypes as C
    rows = G.pinned_empty((frames, sp.bins), np.float32)       # (allocated once: pinning 1 GB takes longer than the whole job)
    # The engine in its steady state, not the driver's housekeeping: for a while after gigabytes of device memory have gone back
    # to the driver (the earlier rows' tensors, `x` above) copies over the link run at three quarters of their rate (tools/e2e_bisect2.py:
    # 3.8 against 5.0 M frames/s for ~0.5 s after a 12 GiB hipFree; tools/e2e_alone.py: up to ~2 s at this point of the full run).  The
    # call is therefore repeated for three seconds, a quarter of a second apart, and the best one is the row; every call's rate is kept.
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    best, first, rates = 1e9, None, []
    t_begin = time.perf_counter()
    while True:
        nf = C.c_size_t(0)
        t0 = time.perf_counter()
        rc = G.api.lib().glfer_hip_spectrogram_host(sp._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
        dt = time.perf_counter() - t0
        assert rc == 0 and nf.value == frames, (rc, nf.value)
        if first is None:
            first = dt                                  # (the first call makes the plan's ring: pinned and device buffers)
        else:
            best = min(best, dt)
            rates.append(round(frames / dt / 1e6, 2))
        if len(rates) >= reps and (time.perf_counter() - t_begin >= 3.0 or len(rates) >= 40):
            break
        time.sleep(0.25)
    del rows
    nbytes = frames * (2 * sp.hop + 4 * sp.bins)
    sp.close()
    return {"workload": name + ", 16-bit PCM", "path": "pinned host samples -> glfer_hip_spectrogram_host -> pinned host rows",
            "frames": frames, "value": frames / best, "unit": "frames/s", "seconds": best, "first_call_seconds": first,
            "pcie_gbs_both_directions": nbytes / best / 1e9, "bytes_in_per_frame": 2 * sp.hop, "bytes_out_per_frame": 4 * sp.bins,
            "calls_M_frames_per_s": rates,
            "note": "best of %d calls after the first, spread over three seconds; pinned copies on this pool's boxes run at 57 GB/s one way and 48 + 48 GB/s both ways (tools/pcie_probe.py, profiles/r04_ingest_pcie_inclusive.txt)" % len(rates)}


def end_to_end_multi(torch, G, local, devices, frames_per_gpu=32768, reps=3):
    """The N > 1 line's host-side row: the headline workload from ONE process over every GPU of the job -- pinned 16-bit PCM in
    host memory in, pinned rows out, through glfer_hip_spectrogram_host_workers (one worker thread + plan + pinned ring per
    entry of `devices`; frame ranges, no collective).  Where scaling can be lost (PCIe, NUMA placement, thread wake-ups: SURVEY
    8(e)) and the device-resident value cannot show it."""
    import ctypes as C
    import numpy as np
    name, n, overlap, nw, kmax, _, _ = WORKLOADS["mtm"]
    params = G.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax, sample_format=G.SAMPLES_S16)
    cfg = G.make_config(params, devices[0])
    frames = frames_per_gpu * len(devices)
    hop, bins = n, n // 2 + 1
    pcm = G.pinned_empty((frames * hop,), np.int16)
    piece = 1 << 26
    for s0 in range(0, frames * hop, piece):                        # (the stream is made on this rank's GPU, a piece at a time)
        e0 = min(frames * hop, s0 + piece)
        x = synth_on_device(torch, e0 - s0, torch.device("cuda", local), seed=7 + s0 // piece)
        pcm[s0:e0] = (x * 32767.0).round().to(torch.int16).cpu().numpy()
        del x
    rows = G.pinned_empty((frames, bins), np.float32)
    devs = (C.c_int * len(devices))(*devices)
    times = []
    for _ in range(reps + 1):
        nf = C.c_size_t(0)
        t0 = time.perf_counter()
        rc = G.api.lib().glfer_hip_spectrogram_host_workers(C.byref(cfg), devs, len(devices), pcm.ctypes.data, pcm.size, rows.ctypes.data, C.byref(nf))
        times.append(time.perf_counter() - t0)
        assert rc == 0 and nf.value == frames, (rc, nf.value)
    del rows, pcm
    best = min(times[1:])
    nbytes = frames * (2 * hop + 4 * bins)
    return {"workload": name + ", 16-bit PCM, one process driving %d GPUs" % len(devices), "devices": list(devices),
            "path": "pinned host samples -> glfer_hip_spectrogram_host_workers (one worker per GPU) -> pinned host rows",
            "frames": frames, "frames_per_gpu": frames_per_gpu, "value": frames / best, "unit": "frames/s", "seconds": best,
            "first_call_seconds": times[0], "calls_seconds": times, "pcie_gbs_both_directions_aggregate": nbytes / best / 1e9}


def c4_as_worded(torch, G, local, devices=None):
    """BASELINE config 4 as worded: a 1-hour 48 kHz mono 16-bit WAV (346 MB, written to /dev/shm here) through
    glfer_hip_spectrogram_wav_multi on every visible GPU -- multitaper N = 16384, 9 tapers -- wall seconds, file to rows
    in host memory (source.c:193, wav_fmt.c:45-121 as a batch)."""
    import struct
    import tempfile
    name, n, overlap, nw, kmax, _, _ = WORKLOADS["mtm16k"]
    nsamples = 3600 * 48000
    d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(d, "glfer_c4_%d.wav" % os.getpid())
    try:
        x = synth_on_device(torch, nsamples, torch.device("cuda", local), seed=4)
        pcm = (x * 32767.0).round().to(torch.int16).cpu().numpy()
        del x
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 48000, 96000, 2, 16))
            f.write(b"data" + struct.pack("<I", pcm.nbytes))
            pcm.tofile(f)
        del pcm
        import ctypes as C
        import numpy as np
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        params = G.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax, sample_format=G.SAMPLES_S16)
        cfg = G.make_config(params, devices[0])
        frames = nsamples // n
        rows = G.pinned_empty((frames, n // 2 + 1), np.float32)     # the application's row buffer, made once (pinned: rows arrive by DMA)
        devs = (C.c_int * len(devices))(*devices)             # (one worker per entry; an ordinal may repeat: the one-GPU rehearsal)
        # (a) through a kept set of workers (glfer_hip_workers: plans, tables and one chunk ring PER WORKER made once, outside the calls)
        t0 = time.perf_counter()
        W = G.Workers(params, devices, hint_frames=frames)
        create_s = time.perf_counter() - t0
        times = []
        for _ in range(6):
            t0 = time.perf_counter()
            nf, _ = W.run_wav(path, rows, phases=False)
            times.append(time.perf_counter() - t0)
            assert nf == frames, nf
        # the phase split, from two more calls that record timing events around every copy and kernel (and run a little slower for it)
        phases = [W.run_wav(path, rows)[1] for _ in range(2)]
        W.close()
        best = min(range(1, len(times)), key=lambda i: times[i])
        # (b) the stateless entry (it keeps the two most recently used sets of workers itself: its first call pays for making them)
        stateless = []
        for _ in range(4):
            nf = C.c_size_t(0)
            t0 = time.perf_counter()
            rc = G.api.lib().glfer_hip_spectrogram_wav_workers(C.byref(cfg), devs, len(devices), os.fsencode(path), rows.ctypes.data, frames, C.byref(nf), 0)
            stateless.append(time.perf_counter() - t0)
            assert rc == 0 and nf.value == frames, (rc, nf.value)
        del rows
        return {"workload": "C4 as worded: multitaper N=16384 NW=4.5 mtm_k=8 over a 1-hour 48 kHz 16-bit mono WAV", "file_bytes": 44 + 2 * nsamples,
                "gpus": len(set(devices)), "workers": len(devices), "frames": frames, "wall_seconds": times[best], "first_call_seconds": times[0], "calls_seconds": times,
                "workers_create_seconds": create_s,
                "value": frames / times[best], "unit": "frames/s",
                "phases": phases[1],
                "phases_are": "seconds, per field the largest over the workers: set-up inside the call, reading the file into pinned memory, uploads, kernels, "
                              "downloads (sums over a worker's chunks, which overlap one another) and the call's wall time",
                "link_bound_seconds": 2 * nsamples / 48e9,
                "stateless_entry_seconds": stateless,
                "path": "glfer_hip_workers_spectrogram_wav: every worker reads its own part of the file into its pinned ring, rows by DMA into a pinned host buffer",
                "note": "file in %s (page cache); glfer_hip_workers_create (plans, DPSS tables, a chunk ring per worker) is outside the calls and reported beside them; "
                        "best of the calls after the first; the kernel alone runs this file's 10 546 frames in ~1.2 ms; 346 MB each way over a 48 + 48 GB/s link is 7.2 ms" % d}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


# the default run's secondary rows (BASELINE.json configs other than the headline): workload -> timed steps
SECONDARY = (("fft1k", 5), ("fft", 5), ("mtm75", 5), ("mtm16k", 5), ("hparma", 5))
SECONDARY_WARMUP = 3    # untimed passes per secondary row: the first pass over a freshly allocated 8 GB of rows runs 10 % slow on a fresh box


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0 = the workload's default)")
    ap.add_argument("--workload", default="mtm", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary rows (C1, C2, C3 at 75 %% overlap, C4, C5) and the parity object")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import glfer_amd as G

    # GLFER_BENCH_REHEARSE=1: every rank on cuda:0 with gloo for the barrier and the max -- lets a
    # one-GPU box run the N > 1 path end to end (same sharding, same launches); never a measurement
    rehearse = os.environ.get("GLFER_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    res = measure(torch, G, dist, args.workload, args.frames, args.steps, args.warmup, world, rank, local, rehearse)

    if rank == 0:
        frames, n, hop, b_alg, kernel_ms = res["frames"], res["n"], res["hop"], res["b_alg"], res["kernel_ms"]
        kernel, pmc_name = KERNELS[args.workload]
        # measured HBM bytes per frame of this kernel, from the committed rocprofv3 PMC passes
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is) -- see profiles/
        prof, traffic_src = pmc_summary(pmc_name)
        traffic = prof["hbm_traffic_bytes_per_frame_corrected"] * frames if prof else None
        # VALU wavefront-instructions per frame, counted by the hardware (SQ_INSTS_VALU pass of the same command)
        valu_instr = prof.get("sq_counters_per_launch", {}).get("SQ_INSTS_VALU") if prof else None
        if valu_instr is not None:
            valu_instr /= prof["frames_per_launch"]
        line = {
            "metric": METRICS[args.workload],
            "value": res["fps"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU)" if rehearse else ""),
            "config": {"workload": res["name"], "frames_per_gpu_per_step": frames, "n": n, "hop": hop,
                       "tapers": res["ntapers"], "sharding": "frame ranges, no collective"},
            "roofline": roofline_of(res, args.workload, traffic, traffic_src, [res["kernel_ms_first"], res["kernel_ms_min"], res["kernel_ms_max"]]),
            "hbm_gbs_aggregate": res["fps"] * b_alg / 1e9,
        }
        line["valu"] = valu_view(res)
        if valu_instr is not None:
            # The same in the hardware's own count: VALU wavefront-instructions per frame (SQ_INSTS_VALU / frames,
            # committed PMC pass) x frames / kernel time, per CU and clock.  Nominal peak 2 (four SIMDs, a wave64 f32
            # instruction in two clocks: what peak_Tops assumes); a stream of independent v_fma/v_add reaches 1.59
            # (tools/kbench, 8 wavefronts per SIMD).
            per_clock = valu_instr * frames / (kernel_ms * 1e-3) / (256 * CLOCK_GHZ * 1e9)
            line["valu"]["issue"] = {"wave_instr_per_frame_measured": valu_instr, "lane_instr_per_frame_measured": 64 * valu_instr,
                                     "per_cu_clock": per_clock, "peak_per_cu_clock": 2.0,
                                     "measured_stream_per_cu_clock": VALU_STREAM_TOPS * 1e12 / 64 / 256 / (CLOCK_GHZ * 1e9),
                                     "clock_ghz": CLOCK_GHZ, "source": "SQ_INSTS_VALU, rocprofv3 PMC (%s)" % traffic_src}
        if args.workload == "hparma":
            line["valu"] = hparma_view(res)

    # ---- the other BASELINE.json configs, each timed the same way for a few steps (N = 1 only: they are
    # single-GPU rows; every rank would have to take part in their barriers otherwise)
    if world == 1 and not args.no_secondary and args.workload == "mtm":
        rows = []
        for wl, k in SECONDARY:
            r = measure(torch, G, dist, wl, 0, k, SECONDARY_WARMUP, world, rank, local, rehearse)
            row = {"workload": r["name"], "key": wl, "value": r["fps"], "unit": "frames/s", "steps": k, "warmup": SECONDARY_WARMUP,
                   "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"], "kernel": KERNELS[wl][0],
                   "frames_per_step": r["frames"], "roofline": roofline_of(r, wl)}
            v = hparma_view(r) if wl == "hparma" else valu_view(r)
            row["valu_frac"] = v["frac"]
            rows.append(row)
        # the reference's DEFAULT setting -- per-hop mean removal on (glfer.c:275), the hop summed in the reference's own order
        # (cfg.sub_mean = 1) -- on the two periodogram rows and the headline, on a stream with a DC level
        for wl, k in (("fft1k", 5), ("fft", 5), ("mtm", 5)):
            r = measure(torch, G, dist, wl, 0, k, SECONDARY_WARMUP, world, rank, local, rehearse, params_kw=dict(sub_mean=G.SUBMEAN_EXACT), dc=0.1)
            rows.append({"workload": r["name"] + ", per-hop mean removal on (the reference's summation order)", "key": wl + "+mean", "value": r["fps"],
                         "unit": "frames/s", "steps": k, "warmup": SECONDARY_WARMUP, "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"],
                         "frames_per_step": r["frames"], "roofline": roofline_of(r, wl)})
        # the same C2 into rows at a 64-byte-multiple pitch (cfg.psd_pitch = 2112 floats: an interface addition); the
        # algorithmic bytes stay the P = 2049 bins of a row, not the pitch
        r = measure(torch, G, dist, "fft", 0, 5, SECONDARY_WARMUP, world, rank, local, rehearse, params_kw=dict(psd_pitch=2112))
        rows.append({"workload": r["name"] + ", rows at a pitch of 2112 floats (cfg.psd_pitch)", "key": "fft+pitch2112", "value": r["fps"],
                     "unit": "frames/s", "steps": 5, "warmup": SECONDARY_WARMUP, "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"],
                     "frames_per_step": r["frames"], "roofline": roofline_of(r, "fft")})
        # SURVEY 8(d)'s avg-on row: C2 with update_avg_plain, depth 4, over the whole band (avg.c:108-159).  "fft+avg": the average
        # taken inside the periodogram launch, PSD rows never stored (north_star: |X|^2 + block-average fused in-register);
        # "fft+avg+rows": the same launch storing the rows too; "fft+avg two launches": rows, then avg_fused_kernel over them (round 4's row)
        for form, key, what, extra in (("fused", "fft+avg", "one launch, the average taken in registers, PSD rows not stored", 0),
                                       ("fused+rows", "fft+avg+rows", "one launch, the average taken in registers, PSD rows stored as well", 4),
                                       ("two", "fft+avg two launches", "rows, then avg_fused_kernel over them: the row in again, 8 B per bin out", 8)):
            r = measure(torch, G, dist, "fft", 262144, 5, SECONDARY_WARMUP, world, rank, local, rehearse, avg_depth=4, avg_form=form)
            b = 4 * r["hop"] + 8 * r["bins"] + 32 + extra * r["bins"]
            rows.append({"workload": r["name"] + " + update_avg_plain depth 4 (%s)" % what, "key": key, "value": r["fps"],
                         "unit": "frames/s", "steps": 5, "warmup": SECONDARY_WARMUP, "ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"],
                         "frames_per_step": r["frames"],
                         "roofline": {"bound": "hbm", "achieved": r["frames"] * b / (r["kernel_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": r["frames"] * b / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": b,
                                      "note": "bytes this form moves by construction: the hop in, 8 B per bin of averages + 32 B of return values out"
                                              + (", 4 B per bin of PSD row out" if extra == 4 else (", the PSD row out and in again" if extra == 8 else ""))}})
        # ... and the same one launch under the reference's default, per-hop mean removal in its own order (the means pass in front, given to the kernel)
        r = measure(torch, G, dist, "fft", 262144, 5, SECONDARY_WARMUP, world, rank, local, rehearse, params_kw=dict(sub_mean=G.SUBMEAN_EXACT), avg_depth=4,
                    avg_form="fused", dc=0.1)
        b = 4 * r["hop"] + 8 * r["bins"] + 32
        rows.append({"workload": r["name"] + " + update_avg_plain depth 4, per-hop mean removal on (the reference's order; one estimator launch behind the means pass)",
                     "key": "fft+mean+avg", "value": r["fps"], "unit": "frames/s", "steps": 5, "warmup": SECONDARY_WARMUP, "ms_per_step": r["ms_per_step"],
                     "kernel_ms": r["kernel_ms"], "frames_per_step": r["frames"],
                     "roofline": {"bound": "hbm", "achieved": r["frames"] * b / (r["kernel_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": r["frames"] * b / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": b}})
        rows.extend(stats_rows(torch, G, local))
        line["secondary"] = rows
        line["parity"] = parity_vs_oracle(torch, G, args.workload, local)
        line["end_to_end"] = end_to_end(torch, G, local)
        line["c4_as_worded"] = c4_as_worded(torch, G, local)
    # ---- N > 1: the host-side rows (rank 0 drives every GPU of the job from one process; the other ranks wait on the CPU)
    if world > 1 and not args.no_secondary and args.workload == "mtm":
        try:
            cpu_group = None if rehearse else dist.new_group(backend="gloo")
        except Exception:                            # no gloo beside RCCL on this stack: the default group's barrier (the waiting ranks' GPUs spin)
            cpu_group = None
        devs = [0] * world if rehearse else list(range(world))
        host = host_side_phase(dist, rank, cpu_group, lambda: {"end_to_end": end_to_end_multi(torch, G, local, devs),
                                                               "c4_as_worded": c4_as_worded(torch, G, local, devs)})
        if rank == 0:
            line["host_side"] = host
            line["host_side"]["note"] = ("one process (rank 0) over all %d GPUs while the other ranks wait at a CPU barrier: the path on which "
                                         "multi-GPU scaling can be lost (SURVEY 8(e)); never part of `value`" % world)
    if rank == 0:
        line["per_rank_ms_per_step"] = res["per_rank_ms"]
        if world == 1 and not args.no_cpu_baseline:
            _, n, overlap, nw, kmax, _, cpu_frames = WORKLOADS[args.workload]
            line["cpu_baseline"] = cpu_baseline(args.workload, n, overlap, nw, kmax, cpu_frames)
            line["cpu_baseline"]["cpu_model"] = cpu_model()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
