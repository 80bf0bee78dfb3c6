#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- golden input/output vectors for the hot path.

The reference ships no fixtures.  These vectors come from the CPU oracle
(oracle/glfer_oracle.c), run in the build container right after tests/test_oracle_pinning.py
has shown it bit-identical to the reference's own fft_radix2.c / g-l_dpss.c / avg.c / util.c
objects (oracle/_ref/).  Each file holds data only: parameters, the input samples and the
expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from _signals import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
W = O.WINDOWS


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print(name, {k: getattr(v, "shape", v) for k, v in kw.items()})


def fft_case(name, n, overlap, window, frames, fs, seed, **opt):
    h = O.hop(n, overlap)
    x = synth(frames * h, fs=fs, seed=seed)
    psd = O.spectrogram_fft(x, n, overlap, W[window], opt.get("a", 0.0), opt.get("limiter", 0),
                            opt.get("sub_mean", 0), opt.get("history_mode", 0))
    save(name, mode="fft", n=n, overlap=np.float32(overlap), window=W[window], fs=fs, seed=seed,
         a=np.float32(opt.get("a", 0.0)), limiter=opt.get("limiter", 0),
         sub_mean=opt.get("sub_mean", 0), history_mode=opt.get("history_mode", 0),
         x=x, psd=psd, win=O.window(W[window], n))


def mtm_case(name, n, overlap, nw, kmax, frames, fs, seed, **opt):
    h = O.hop(n, overlap)
    x = synth(frames * h, fs=fs, seed=seed)
    psd = O.spectrogram_mtm(x, n, overlap, nw, kmax, opt.get("sub_mean", 0), opt.get("history_mode", 0))
    v, sig = O.dpss(n, kmax, nw)
    save(name, mode="mtm", n=n, overlap=np.float32(overlap), nw=nw, kmax=kmax, fs=fs, seed=seed,
         sub_mean=opt.get("sub_mean", 0), history_mode=opt.get("history_mode", 0),
         x=x, psd=psd, sig=sig, tapers_head=v[:, :64].copy(), tapers_sum=v.sum(axis=1))


def main():
    os.makedirs(OUT, exist_ok=True)
    # BASELINE.json configs 1-3 (+ C3 at 75 % overlap), seeds 0/1
    fft_case("c1_fft1024_hann_ovl50_s0", 1024, 0.5, "hanning", 16, 8000.0, 0)
    fft_case("c2_fft4096_hann_ovl75_s0", 4096, 0.75, "hanning", 12, 48000.0, 0)
    mtm_case("c3_mtm4096_nw25_k4_ovl0_s0", 4096, 0.0, 2.5, 4, 6, 48000.0, 0)
    mtm_case("c3_mtm4096_nw25_k4_ovl75_s1", 4096, 0.75, 2.5, 4, 10, 48000.0, 1)
    # BASELINE config 4: N=16384, NW=4.5, 9 tapers (three frames keep the file small)
    mtm_case("c4_mtm16384_nw45_k8_ovl0_s0", 16384, 0.0, 4.5, 8, 3, 48000.0, 0)
    # BASELINE config 5: HP-ARMA t=128, p_e=32, N=4096 (psd + AR vector + rank per frame)
    x = synth(4 * 4096, seed=5)
    fr = O.hparma_frames(x, 4096, 0.0, 128, 32)
    save("c5_hparma4096_t128_p32_s5", mode="hparma", n=4096, overlap=np.float32(0.0), t=128, p_e=32, seed=5, x=x,
         psd=np.array([f[0] for f in fr]), ar=np.array([f[1] for f in fr]), rank=np.array([f[2] for f in fr]))
    # edge fixtures
    fft_case("e_fft1024_kaiser_submean", 1024, 0.5, "kaiser", 10, 8000.0, 2, sub_mean=1)
    fft_case("e_fft1024_hann_zero_always", 1024, 0.5, "hanning", 6, 8000.0, 2, history_mode=1)
    fft_case("e_fft1024_blackman_ra9mb", 1024, 0.25, "blackman", 6, 8000.0, 1, a=0.001)
    fft_case("e_fft1024_hamming_limiter", 1024, 0.0, "hamming", 4, 8000.0, 1, limiter=1)
    fft_case("e_fft1024_rect_ovl90", 1024, 0.9, "rectangular", 40, 8000.0, 0)
    fft_case("e_fft256_welch", 256, 0.5, "welch", 8, 8000.0, 0)
    fft_case("e_fft2048_gauss", 2048, 0.5, "gaussian", 6, 48000.0, 0)
    fft_case("e_fft512_bartlett", 512, 0.0, "bartlett", 5, 8000.0, 0)
    mtm_case("e_mtm1024_nw4_k7_submean", 1024, 0.5, 4.0, 7, 8, 8000.0, 1, sub_mean=1)
    # halfcomplex spectrum of one Hanning frame (what fft_do leaves in outbuf)
    x = synth(4096, seed=3)
    w = O.window(W["hanning"], 4096)
    save("spec_fft4096_hann", x=x, win=w, halfcomplex=O.rfft_halfcomplex(w * x))
    # averaging (avg.c) and floor (fft.c:240-294) on a C1-like PSD sequence
    x = synth(20 * 512, fs=8000.0, seed=4)
    psd = O.spectrogram_fft(x, 1024, 0.5, W["hanning"])
    avg_out = {}
    for mode in ("plain", "sumextreme", "sumavg"):
        for max0 in (0, 1):
            a = O.Averager(1024, 4)
            rows, rets = [], []
            for f in range(psd.shape[0]):
                r, avg, peak, var = a.update(mode, psd[f], 25, 450, max0=max0, n=513)
                rows.append(avg)
                rets.append([r, peak, var])
            avg_out["%s_max%d_avg" % (mode, max0)] = np.array(rows)
            avg_out["%s_max%d_ret" % (mode, max0)] = np.array(rets)
    fl = np.array([O.floor_stats(p) for p in psd], np.float64)
    save("avg_floor_fft1024", psd=psd, depth=4, minbin=25, maxbin=450, floor=fl, **avg_out)
    # display mapping (g_main.c:1099-1236) of that PSD sequence: the default look (log scale,
    # autoscale, HSV) and a fixed-level linear THRESH-palette one with a 20 % threshold; plus
    # the averaged (double) source in log scale and all eight palettes
    disp = {}
    cases = {"log_auto_hsv": dict(palette_id=0, scale_log=True, autoscale=True, overlap=0.5),
             "lin_fixed_thresh": dict(palette_id=1, scale_log=False, autoscale=False, max_level_db=-25.0,
                                      min_level_db=-70.0, thr_level=20.0),
             "log_fixed_bone": dict(palette_id=5, scale_log=True, autoscale=False, max_level_db=-20.0,
                                    min_level_db=-80.0, thr_level=5.0)}
    for name, kw in cases.items():
        rgb, lev, levels, _ = O.display(psd[:, :513], fl, **kw)
        disp[name + "_rgb"], disp[name + "_lev"], disp[name + "_levels"] = rgb, lev, levels
    rgb, lev, levels, _ = O.display(avg_out["plain_max0_avg"], fl, palette_id=7, scale_log=True, autoscale=True,
                                    overlap=0.5)
    disp["avg_log_auto_otd_rgb"], disp["avg_log_auto_otd_lev"], disp["avg_log_auto_otd_levels"] = rgb, lev, levels
    save("display_fft1024", psd=psd[:, :513], stats=fl.astype(np.float32), avg=avg_out["plain_max0_avg"],
         palettes=np.array([O.palette(i) for i in range(8)]), **disp)
    round2()


def round2():
    """Round-2 rows: LMP (l_*), MTM harmonic F-test (f_*), WAV files with a trailing partial block (w_*)."""
    x = synth(14 * 512, fs=8000.0, seed=7)
    save("l_lmp1024_av4_ovl50", n=1024, overlap=np.float32(0.5), nl=4, sub_mean=0, x=x,
         out=O.spectrogram_lmp(x, 1024, 0.5, 4))
    x = synth(9 * 2048, fs=48000.0, seed=8)
    save("l_lmp2048_av3_submean", n=2048, overlap=np.float32(0.0), nl=3, sub_mean=1, x=x,
         out=O.spectrogram_lmp(x, 2048, 0.0, 3, sub_mean=1))
    for live in (1, 0):
        x = synth(5 * 1024, fs=8000.0, seed=9)
        psd, ft = O.spectrogram_mtm_ftest(x, 1024, 0.0, 2.5, 4, mu_live=live)
        save("f_mtm1024_nw25_k4_mu%d" % live, n=1024, overlap=np.float32(0.0), nw=2.5, kmax=4, mu_live=live, x=x,
             psd=psd, ftest=ft)
    x = synth(4 * 1024, fs=48000.0, seed=10)
    psd, ft = O.spectrogram_mtm_ftest(x, 4096, 0.75, 4.0, 7, mu_live=1)
    save("f_mtm4096_nw4_k7_ovl75_mu1", n=4096, overlap=np.float32(0.75), nw=4.0, kmax=7, mu_live=1, x=x, psd=psd, ftest=ft)
    x = synth(9 * 512 + 200, fs=8000.0, seed=11)
    pcm = np.round(x * 30000).astype(np.int16)
    save("w_s16_fft1024_kaiser_ovl50_submean_tail200", bits=16, mode="fft", n=1024, overlap=np.float32(0.5), window=7,
         sub_mean=1, nw=0.0, kmax=0, pcm=pcm, psd=O.wav_spectrogram(pcm, 16, "fft", 1024, 0.5, 7, sub_mean=1))
    pcm8 = np.clip(np.round(x[:5 * 1024 + 333] * 120 + 128), 0, 255).astype(np.uint8)
    save("w_u8_mtm1024_nw25_k4_tail333", bits=8, mode="mtm", n=1024, overlap=np.float32(0.0), window=5, sub_mean=0,
         nw=2.5, kmax=4, pcm=pcm8, psd=O.wav_spectrogram(pcm8, 8, "mtm", 1024, 0.0, nw=2.5, kmax=4))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        round2()
    else:
        main()
