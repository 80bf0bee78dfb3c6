#!/usr/bin/env python3
"""bench.py -- headline benchmark: spectrogram frames/s + achieved HBM GB/s.

Workload (BASELINE.json metric): multitaper, N=4096, NW=2.5, mtm_k=4 (kmax+1 = 5 DPSS tapers,
mtm.c:189), overlap 0 (reference default, glfer.c:239), float32 mono 48 kHz synthetic stream
resident in HBM.  One "step" = one pass of the hot path (glfer_hip_spectrogram_device) over
this rank's batch of frames.  Frames are independent, so N GPUs = N disjoint frame ranges and no
data-path collective (weak scaling: frames per GPU fixed); torch.distributed (RCCL) is used only
for the barrier and the max-over-ranks of the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload mtm|fft]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TOPS = 78.6        # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-instructions/s


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
    fft_radix2.c) timed on one host core on a bounded prefix of the same workload."""
    import numpy as np
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _signals import synth
    hop = O.hop(n, overlap)
    x = synth(frames * hop, seed=0)
    # the same `frames`-frame stream is processed repeatedly until >= 12 s of CPU work is timed
    # (memory stays bounded; every pass is the full per-frame work of the reference path)
    done, dt = 0, 0.0
    while dt < 12.0:
        t0 = time.perf_counter()
        if workload == "mtm":
            O.spectrogram_mtm(x, n, overlap, nw, kmax)
        elif workload == "hparma":
            O.spectrogram_hparma(x, n, overlap, 128, 32)
        else:
            O.spectrogram_fft(x, n, overlap, O.WINDOWS["hanning"])
        dt += time.perf_counter() - t0
        done += frames
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same workload (a %d-frame stream, repeated), single thread, "
                      "oracle/glfer_oracle.c (gcc -O2, radix-2 recurrence FFT as fft_radix2.c), %.1f s"
                      % (done, frames, dt),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=262144, help="frames per GPU per step")
    ap.add_argument("--workload", default="mtm", choices=["mtm", "fft", "hparma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import glfer_amd as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.workload == "mtm":
        n, overlap, nw, kmax = 4096, 0.0, 2.5, 4
        params = G.MtmParams(n=n, overlap=overlap, w=nw, kmax=kmax)
        name = "C3: multitaper N=4096 NW=2.5 mtm_k=4 (5 tapers), overlap 0, 48 kHz mono f32"
    elif args.workload == "hparma":
        n, overlap, nw, kmax = 4096, 0.0, 0.0, 0
        params = G.HparmaParams(n=n, overlap=overlap, t=128, p_e=32)
        name = "C5: HP-ARMA t=128 p_e=32 N=4096, overlap 0, 48 kHz mono f32 (compute/latency bound, not HBM)"
        if args.frames == 262144:
            args.frames = 16384
    else:
        n, overlap, nw, kmax = 4096, 0.75, 0.0, 0
        params = G.FftParams(n=n, window_type=G.WINDOWS["hanning"], overlap=overlap)
        name = "C2: periodogram Hanning N=4096, overlap 75%, 48 kHz mono f32"
        if args.frames == 262144:
            args.frames = 1048576                     # a 2^30-sample stream per GPU
    sp = G.Spectrogram(params, device=local)
    hop, bins = sp.hop, sp.bins
    frames = args.frames
    # Weak scaling: the job is world*frames frames of one long stream; this rank owns the
    # contiguous frame range frame_range() gives it and holds only the samples of its window
    # (its hops + the N-H history halo on the left; the stream starts with zero history).
    from glfer_amd.shard import frame_range, run_shard, sample_window
    first, count = frame_range(frames * world, rank, world)
    begin, end = sample_window(first, count, hop, n)
    shard = synth_on_device(torch, end - begin, dev, seed=rank)
    psd = torch.empty((count, bins), dtype=torch.float32, device=dev)

    def step():
        run_shard(sp, shard, begin, first, count, out=psd)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    if rank == 0:
        total_frames = frames * world * args.steps
        fps = total_frames / dt
        b_alg = 4 * hop + 4 * bins                      # compulsory read of H new samples + P bins out
        achieved = frames * b_alg / (kernel_ms * 1e-3) / 1e9
        ntap = sp.ntapers
        # measured HBM bytes per frame of this kernel, from the committed rocprofv3 PMC passes
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is) -- see profiles/
        traffic = None
        try:
            pname = {"mtm": "r01_hbm_traffic.json", "fft": "r01_hbm_traffic_fft.json"}[args.workload]
            prof = json.load(open(os.path.join(ROOT, "profiles", pname)))
            traffic = prof["hbm_traffic_bytes_per_frame_corrected"] * frames
        except Exception:
            pass
        line = {
            "metric": {"mtm": "spectrogram frames/sec + achieved HBM GB/s, N=4096 MTM K=4",
                       "fft": "spectrogram frames/sec + achieved HBM GB/s, N=4096 periodogram",
                       "hparma": "spectrogram frames/sec, HP-ARMA t=128 p_e=32 N=4096"}[args.workload],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "frames_per_gpu_per_step": frames, "n": n, "hop": hop,
                       "tapers": ntap, "sharding": "frame ranges, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch, rocprofv3 PMC (profiles/r01_hbm_traffic*.json)",
                         "algorithmic_bytes_per_launch": frames * b_alg,
                         "kernel": {"hparma": "hparma_kernel", "fft": "spectro16h_kernel<12>", "mtm": "spectro16y_kernel"}[args.workload],
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_frame": b_alg,
                         "note": "FP32-VALU-bound on this chip (SURVEY 7): see valu.frac"},
            "hbm_gbs_aggregate": fps * b_alg / 1e9,
        }
        # FP32 VALU view of the same launch: butterflies 3*M*log2(M) per complex M-point transform
        # + inter-pass twiddles, taper multiply and |Z|^2.  MTM: one N-point transform per taper
        # PAIR, the odd taper shared by two frames (ntap/2 transforms per frame); periodogram: one
        # N/2-point transform per frame + the real-input split (8 ops per bin)
        import math
        if args.workload == "fft":
            m = n // 2
            lane_ops = 3 * m * math.log2(m) + 4 * (m - m // 64) + 2 * m + 8 * m
        else:
            lane_ops = (ntap / 2.0) * (3 * n * math.log2(n) + 4 * (n - n // 64) + 2 * n + 2 * n)
        line["valu"] = {"lane_ops_per_frame": lane_ops, "achieved_Tops": frames * lane_ops / (kernel_ms * 1e-3) / 1e12,
                        "peak_Tops": VALU_PEAK_TOPS,
                        "frac": frames * lane_ops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS}
        if world == 1 and not args.no_cpu_baseline:
            cpu_frames = {"mtm": 16384, "fft": 131072, "hparma": 4096}[args.workload]
            line["cpu_baseline"] = cpu_baseline(args.workload, n, overlap, nw, kmax, cpu_frames)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
