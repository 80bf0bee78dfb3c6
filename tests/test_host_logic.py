"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol
include/glfer_hip.h declares, builds the same tables as the reference, and refuses to compute
without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "glfer_hip.h")).read()
    declared = set(re.findall(r"\b(glfer_hip_[a-z_]+)\s*\(", hdr))
    assert declared == set(lib.api.EXPORTS), declared ^ set(lib.api.EXPORTS)
    L = ctypes.CDLL(lib.api.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert "gfx950" in lib.version()


def test_compat_library_exports_reference_entry_points():
    path = os.path.join(ROOT, "glfer_amd", "lib", "libglfer_compat.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    import torch  # noqa: F401  (one HIP runtime per process: torch's, loaded first)
    L = ctypes.CDLL(path)
    for sym in ("fft_init", "fft_do", "fft_psd", "fft_close", "mtm_init", "mtm_do", "mtm_close",
                "hparma_init", "hparma_do", "hparma_close", "compute_floor", "init_avg", "alloc_avg", "delete_avg", "update_avg_plain",
                "update_avg_sumextreme", "update_avg_sumavg", "lmp_init", "lmp_do", "lmp_close", "prepare_audio",
                "open_wav_file", "wav_read", "close_wav_file", "glfer_compat_readahead", "glfer_compat_readahead_served"):   # wav_fmt.h:24-26 (round 3)
        assert hasattr(L, sym), sym


def test_compat_reads_the_programs_own_globals(tmp_path):
    """Boundary: glfer defines `opt_t opt; glfer_t glfer;` (glfer.c:56-57) and the estimators read
    opt.autoscale / glfer.first_buffer directly (fft.c:186, fft.c:99).  A C program that defines
    the two globals and links libglfer_compat.so -- no glue file, no hook -- must be read through."""
    import subprocess
    libdir = os.path.join(ROOT, "glfer_amd", "lib")
    exe = tmp_path / "c_compat_globals"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_compat_globals.c"), "-o", str(exe), "-L", libdir, "-lglfer_compat",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + libdir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr)
    # LP64 layout of glfer.h:62-139: three pointers ahead of init_done / first_buffer in glfer_t
    size_opt, off_auto, size_glfer, off_first = map(int, r.stdout.split())
    assert (size_opt, off_auto, size_glfer, off_first) == (184, 148, 88, 28)


def test_kept_objects_link_against_the_library(tmp_path):
    """glfer relinks with libglfer_compat.so in place of fft.o fft_radix2.o mtm.o g-l_dpss.o avg.o hparma.o
    lmp.o wav_fmt.o only if the library has EVERY symbol the kept objects (glfer.c source.c g_main.c g_options.c) take from
    those -- the data symbols fft_windows[] / num_fft_windows of fft.c:47-59 included (g_options.c:47-48, 579-583).  A C file
    that references all of them is linked with --no-undefined; run without arguments it only walks the window table."""
    import subprocess
    libdir = os.path.join(ROOT, "glfer_amd", "lib")
    exe = tmp_path / "c_compat_link_all"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_compat_link_all.c"), "-o", str(exe), "-Wl,--no-undefined", "-L", libdir,
                    "-lglfer_compat", "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + libdir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "1", (r.returncode, r.stdout, r.stderr)
    # and the list itself: where the reference tree is mounted, every non-static function or object the dropped files define
    # that a kept file names must be in the library
    ref = "/root/reference"
    if os.path.isdir(ref):
        dropped = ["fft.c", "fft_radix2.c", "mtm.c", "g-l_dpss.c", "avg.c", "hparma.c", "lmp.c", "wav_fmt.c"]
        kept = [f for f in os.listdir(ref) if f.endswith(".c") and f not in dropped and f != "bell-p-w.c"]
        defined = set()
        for f in dropped:
            for line in open(os.path.join(ref, f), errors="replace"):
                m = re.match(r"^(?!static)[A-Za-z_][\w \*]*?\**\b(\w+)\s*(\(|\[\]\s*=|=)", line)
                if m and not line.rstrip().endswith(";") or (m and "=" in line):
                    defined.add(m.group(1))
        kept_text = " ".join(open(os.path.join(ref, f), errors="replace").read() for f in kept)
        needed = sorted(s for s in defined if re.search(r"\b" + s + r"\b", kept_text) and s not in ("main", "opt", "glfer"))
        assert "fft_windows" in needed and "num_fft_windows" in needed and "fft_do" in needed
        import torch  # noqa: F401
        L = ctypes.CDLL(os.path.join(libdir, "libglfer_compat.so"))
        missing = [s for s in needed if not hasattr(L, s)]
        assert not missing, missing


def test_compat_struct_fields_follow_the_reference_header():
    """opt_t / glfer_t in include/glfer_compat.h list the fields of glfer.h:62-139 in the same
    order with the same scalar types (GTK pointers as void *).  Reads the reference header as text
    where the reference tree is mounted (build container); skipped elsewhere."""
    ref = "/root/reference/glfer.h"
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted")

    def fields(text, name):
        body = re.search(r"typedef struct\s*\{([^}]*)\}\s*" + name + r"\s*;", re.sub(r"/\*.*?\*/", "", text, flags=re.S), re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(.*?)(\**)\s*(\w+)$", decl)
            typ, ptr, ident = m.group(1).strip(), m.group(2), m.group(3)
            if ptr and typ in ("GtkWidget", "GtkTooltips", "void"):
                typ = "void"
            out.append((typ + ptr, ident))
        return out

    ours = open(os.path.join(ROOT, "include", "glfer_compat.h")).read()
    theirs = open(ref).read()
    for name in ("opt_t", "glfer_t"):
        assert fields(ours, name) == fields(theirs, name), name


@pytest.mark.parametrize("n", [256, 1024, 4096, 16384])
def test_windows_match_reference_formulas(lib, oracle, n):
    for name, t in lib.WINDOWS.items():
        assert np.array_equal(lib.make_window(t, n), oracle.window(t, n)), name


@pytest.mark.parametrize("n,kmax,nw", [(1024, 4, 2.5), (4096, 4, 2.5), (4096, 7, 4.0), (16384, 8, 4.5)])
def test_dpss_matches_reference_algorithm(lib, oracle, n, kmax, nw):
    v, s = lib.make_dpss(n, kmax, nw)
    v2, s2 = oracle.dpss(n, kmax, nw)
    # same classical Jacobi scheme, same operation order: identical doubles
    assert np.array_equal(v, v2) and np.array_equal(s, s2)


def test_argument_errors(lib):
    api = lib.api
    for bad in (dict(n=1000), dict(n=4), dict(n=1 << 21), dict(n=1 << 24), dict(overlap=1.0), dict(overlap=-0.1), dict(window_type=9)):
        kw = dict(n=1024, overlap=0.0, window_type=0)
        kw.update(bad)
        with pytest.raises(api.GlferHipError, match="bad argument"):
            lib.Spectrogram(lib.FftParams(**kw))
    with pytest.raises(api.GlferHipError, match="bad argument"):
        lib.Spectrogram(lib.MtmParams(n=1024, w=0.0, kmax=3))
    # cfg.psd_pitch (round 4): at least N/2+1 floats, not negative, not with the LMP statistic
    for pitch in (512, 1, -16):
        with pytest.raises(api.GlferHipError, match="bad argument"):
            lib.Spectrogram(lib.FftParams(n=1024, overlap=0.0, window_type=0, psd_pitch=pitch))
    lp = lib.LmpParams(n=1024, overlap=0.5, avg=4)
    lp.psd_pitch = 528
    with pytest.raises(api.GlferHipError, match="bad argument"):
        lib.Spectrogram(lp)
    with pytest.raises(api.GlferHipError):
        lib.make_window(12, 64)


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.GlferHipError, match="HIP runtime error"):
        lib.Spectrogram(lib.FftParams(n=1024))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "glfer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").replace("not the oracle", ""), os.path.join(dirpath, f)


def test_wav_header_parse_fixed_width(lib, tmp_path):
    """open_wav_file (wav_fmt.c:45-80) restated with fixed-width fields: host code, no GPU."""
    import struct
    import wave
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(np.arange(1000, dtype=np.int16).tobytes())
    info = lib.wav_probe(str(p))
    assert (info.format, info.channels, info.sample_rate, info.bits_per_sample) == (1, 2, 48000, 16)
    assert info.data_offset == 44 and info.nsamples == 1000     # channels are not interpreted (wav_fmt.c ignores modus)
    # the same 44 bytes, field by field as wav_fmt.h:34-52 lays them out
    hd = open(p, "rb").read(44)
    riff, _, wavetag, fmt_, sc_len, fmt, modus, fq, bps, bpspl, bits, data, dlen = struct.unpack("<4sI4s4sIHHIIHH4sI", hd)
    assert (riff, wavetag, fmt_, data) == (b"RIFF", b"WAVE", b"fmt ", b"data") and sc_len == 16
    assert (fmt, modus, fq, bits, dlen) == (1, 2, 48000, 16, 2000)
    # rejected: float WAV (format 3), 24-bit PCM
    bad = bytearray(hd)
    bad[20:22] = struct.pack("<H", 3)
    q = tmp_path / "float.wav"
    q.write_bytes(bytes(bad) + b"\0" * 64)
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        lib.wav_probe(str(q))
    bad = bytearray(hd)
    bad[34:36] = struct.pack("<H", 24)
    q.write_bytes(bytes(bad) + b"\0" * 64)
    with pytest.raises(lib.GlferHipError, match="bad argument"):
        lib.wav_probe(str(q))


def _riff(chunks):
    body = b"WAVE" + b"".join(cid + len(data).to_bytes(4, "little") + data + (b"\0" if len(data) & 1 else b"") for cid, data in chunks)
    return b"RIFF" + len(body).to_bytes(4, "little") + body


def test_wav_chunk_walk(lib, tmp_path):
    """glfer_hip_wav_probe walks the RIFF chunks (round 3): LIST / fact chunks before and after `data`, an
    18-byte fmt chunk, a data size field of 0 (a recording that was cut off), trailing bytes that are not a chunk
    (the reference reads them as samples, wav_fmt.c:102) -- host code, no GPU."""
    import struct
    pcm = (np.arange(5000) % 2000 - 1000).astype(np.int16)
    fmt = struct.pack("<HHIIHH", 1, 1, 8000, 16000, 2, 16)
    plain = _riff([(b"fmt ", fmt), (b"data", pcm.tobytes())])
    (tmp_path / "plain.wav").write_bytes(plain)
    i = lib.wav_probe(str(tmp_path / "plain.wav"))
    assert (i.data_offset, i.nsamples, i.sample_rate, i.bits_per_sample) == (44, 5000, 8000, 16)
    listed = _riff([(b"fmt ", fmt + b"\0\0"), (b"LIST", b"INFOISFT" + (13).to_bytes(4, "little") + b"some encoder\0"),
                    (b"fact", (5000).to_bytes(4, "little")), (b"data", pcm.tobytes()),
                    (b"LIST", b"INFOICMT" + (6).to_bytes(4, "little") + b"after\0")])
    (tmp_path / "listed.wav").write_bytes(listed)
    i = lib.wav_probe(str(tmp_path / "listed.wav"))
    assert i.nsamples == 5000 and i.data_offset == listed.index(b"data") + 8 and i.data_offset > 44
    assert listed[i.data_offset:i.data_offset + 10000] == pcm.tobytes()
    cut = bytearray(plain)
    cut[40:44] = (0).to_bytes(4, "little")
    (tmp_path / "cut.wav").write_bytes(bytes(cut))
    assert lib.wav_probe(str(tmp_path / "cut.wav")).nsamples == 5000
    (tmp_path / "stray.wav").write_bytes(plain + b"\x7f\x01\x02")          # not a chunk: samples, as the reference reads them
    assert lib.wav_probe(str(tmp_path / "stray.wav")).data_bytes == 10003
    (tmp_path / "nowave.wav").write_bytes(b"RIFF" + plain[4:8] + b"XXXX" + plain[12:])   # no WAVE tag: the fixed 44-byte layout
    j = lib.wav_probe(str(tmp_path / "nowave.wav"))
    assert (j.data_offset, j.nsamples) == (44, 5000)


@pytest.mark.parametrize("bits", [16, 8])
def test_exported_reader_follows_the_reference_reader(oracle, tmp_path, bits):
    """open_wav_file / wav_read / close_wav_file as libglfer_compat.so exports them (round 3) against the oracle's
    restatement of wav_fmt.c:45-141 -- itself bit-identical to the reference's own object (test_oracle_extras.py):
    every block handed out, the short last block over the stale buffer, the estimator's mean removal in the
    reader's buffer between reads.  Host code: no GPU is touched by the reader."""
    import ctypes as C
    import wave
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    compat = C.CDLL(os.path.join(root, "glfer_amd", "lib", "libglfer_compat.so"))
    compat.open_wav_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    compat.wav_read.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
    rng = np.random.default_rng(3)
    hop = 700
    n = hop * 9 + 333
    pcm = (rng.integers(-20000, 20000, n).astype(np.int16) if bits == 16 else rng.integers(0, 256, n).astype(np.uint8))
    path = tmp_path / "r.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(bits // 8)
        w.setframerate(11025)
        w.writeframes(pcm.tobytes())

    def mutate(block):                       # what prepare_audio does to the reader's buffer (fft.c:93-95)
        block -= np.float32(block.mean())

    want = oracle.wav_blocks(pcm, bits, hop, mutate=mutate)
    speed = C.c_int(0)
    compat.open_wav_file(os.fsencode(str(path)), hop, C.byref(speed))
    assert speed.value == 11025
    got = []
    buf, nblk = C.POINTER(C.c_float)(), C.c_int(0)
    while True:
        compat.wav_read(C.byref(buf), C.byref(nblk))
        if nblk.value == 0:
            break
        blk = np.ctypeslib.as_array(buf, shape=(hop,))
        got.append(blk.copy())
        mutate(blk)
    compat.close_wav_file()
    assert len(got) == len(want) == 10
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_palettes_match_reference_tables(lib, oracle):
    # glfer_hip_palette is a host table builder (set_palette, g_main.c:651-762): no GPU needed
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "display_fft1024.npz"))
    for name, p_n in lib.PALETTES.items():
        assert np.array_equal(lib.palette(p_n), g["palettes"][p_n]), name
        assert np.array_equal(lib.palette(p_n), oracle.palette(p_n)), name
    assert np.array_equal(lib.palette(-3), lib.palette(lib.PALETTES["bw"]))


def test_headers_compile_as_c99(tmp_path):
    """include/*.h are the boundary a C caller (glfer itself) binds: a C99 translation unit that
    uses both headers must compile cleanly, and its undefined symbols must all be exported."""
    import subprocess
    obj = tmp_path / "probe.o"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                    os.path.join(ROOT, "tests", "c_abi_probe.c"), "-o", str(obj)], check=True)
    und = subprocess.run(["nm", "-u", str(obj)], check=True, capture_output=True, text=True).stdout.split()
    wanted = {s for s in und if s.startswith(("glfer_", "fft_", "mtm_"))}
    have = set()
    for so in ("libglfer_hip.so", "libglfer_compat.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "glfer_amd", "lib", so)], check=True,
                             capture_output=True, text=True).stdout
        have |= {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert wanted and wanted <= have, wanted - have


def test_numa_lookups_for_the_worker_placement(lib, tmp_path):
    """a multi-GPU worker thread is bound to the CPUs of its GPU's NUMA node before it allocates its
    pinned ring (ingest.cpp on_workers).  The two look-ups it uses, against a fake sysfs tree: PCI bus id -> node
    (upper-case ids as hipDeviceGetPCIBusId prints them, a missing domain, the kernel's -1, a missing device, a
    malformed file) and node -> CPU mask (ranges, singles, a trailing newline, a malformed list)."""
    import ctypes as C
    L = lib.api.lib()
    root = tmp_path / "sys"
    for bus, node in (("0000:c1:00.0", "3\n"), ("0000:05:00.0", "0\n"), ("0001:e5:00.0", "-1\n"), ("0000:75:00.0", "garbage\n")):
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node)
    for node, cpus in ((0, "0-15,128-143\n"), (3, "48-63,176-191"), (5, "7\n"), (6, "3-1\n"), (7, "0-3,x\n")):
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)
    r = str(root).encode()
    f = L.glfer_hip_numa_node_of_bus_id
    assert f(b"0000:C1:00.0", r) == 3 and f(b"0000:c1:00.0", r) == 3 and f(b"c1:00.0", r) == 3
    assert f(b"0000:05:00.0", r) == 0
    assert f(b"0001:E5:00.0", r) == -1                       # the kernel says -1: unknown
    assert f(b"0000:75:00.0", r) == -1                       # not a number
    assert f(b"0000:99:00.0", r) == -1                       # no such device
    assert f(b"../../etc", r) == -1 and f(b"", r) == -1 and f(None, r) == -1
    mask = (C.c_ubyte * 128)()

    def cpus(node):
        n = L.glfer_hip_numa_node_cpus(node, r, mask, len(mask))
        return n, [c for c in range(1024) if mask[c >> 3] >> (c & 7) & 1]
    n, got = cpus(0)
    assert n == 32 and got == list(range(0, 16)) + list(range(128, 144))
    n, got = cpus(3)
    assert n == 32 and got == list(range(48, 64)) + list(range(176, 192))
    assert cpus(5) == (1, [7])
    assert cpus(6)[0] == -1 and cpus(7)[0] == -1 and cpus(9)[0] == -1 and cpus(-1)[0] == -1
    # the real tree of this host, if it has one: a node's list parses or the call says so -- never a crash
    assert L.glfer_hip_numa_node_cpus(0, None, mask, len(mask)) >= -1


def test_round5_entries_fail_cleanly_without_a_gpu(lib):
    """The entries added in round 5 on a host with no usable HIP device (this container): an error code, never a crash and never a
    CPU computation -- glfer_hip_workers_create, glfer_hip_spectrogram_avg_device (bad arguments are refused before any device is touched)."""
    import ctypes as C
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the GPU suite covers these entries")
    except ImportError:
        pass
    L = lib.api.lib()
    assert L.glfer_hip_abi_version() == 5
    cfg = lib.make_config(lib.MtmParams(n=4096, overlap=0.0, w=2.5, kmax=4), 0)
    devs = (C.c_int * 2)(0, 0)
    h = C.c_void_p()
    rc = L.glfer_hip_workers_create(C.byref(cfg), devs, 2, 1000, C.byref(h))
    assert rc in (-1, -2) and not h.value                      # GLFER_E_ARG / GLFER_E_HIP
    L.glfer_hip_workers_destroy(None)
    nf = C.c_size_t(7)
    assert L.glfer_hip_workers_spectrogram_wav(None, b"/nonexistent.wav", None, 0, C.byref(nf), 0, None) == -1
    assert L.glfer_hip_spectrogram_avg_device(None, None, 0, 0, 0, 2, 4, 0, 10, 0, 100, None, None, None, None) == -1
