"""Frame-range sharding of a spectrogram over the GPUs of one node.

Frame f depends only on samples [f*H - (N-H), f*H + H) (fft.c:98-113), so the frame index
range is cut into contiguous blocks, one per rank, each rank reading its hops plus a left halo
of the N-H history samples, rounded up to whole hops (zeros for rank 0, as fft.c:103-108).  There is no exchange step: no collective
on the data path, only the outputs' row ranges are disjoint.
"""


FRAME_ALIGN = 32      # GLFER_FRAME_ALIGN (include/glfer_hip.h): cuts on multiples of it keep every
                      # frame bit-identical to the single-launch run


def frame_range(total_frames, rank, world, align=FRAME_ALIGN):
    """(first, count) of this rank's contiguous block.  The stream is dealt out in units of
    `align` frames (the remainder units go to the low ranks, the last partial unit to whoever
    holds the end), so every boundary is a multiple of `align`."""
    units = -(-total_frames // align)
    base, rem = divmod(units, world)
    u_first = rank * base + min(rank, rem)
    u_count = base + (1 if rank < rem else 0)
    first = min(total_frames, u_first * align)
    last = min(total_frames, (u_first + u_count) * align)
    return first, last - first


def halo_samples(hop, n, history_mode=0):
    """Samples a rank needs to the left of its first hop: the N-H history rounded UP to whole
    hops.  Whole hops because per-hop mean removal (cfg.sub_mean, fft.c:86-96) corrects every
    history sample by the mean of the hop it arrived in, so the engine reads ceil((N-H)/H) complete
    hops back (glfer_hip.cpp run_device) -- the same rule the chunked ingest uses between chunks.
    history_mode 1 (history zeroed in every frame) needs no halo."""
    if history_mode:
        return 0
    return -(-(n - hop) // hop) * hop


def sample_window(first, count, hop, n, history_mode=0, extra_frames=0):
    """[begin, end) of the stream samples the frames [first, first+count) read.  extra_frames:
    whole frames recomputed in front of the block instead of carried over a boundary -- lmp_av - 1
    for the LMP estimator's ring (lmp.c:85), D - 1 for a moving average of depth D."""
    if count == 0:
        return 0, 0
    begin = max(0, (first - extra_frames) * hop - halo_samples(hop, n, history_mode))
    return begin, (first + count) * hop


def run_shard(sp, local, begin, first, count, out=None):
    """Frames [first, first+count) from `local`, a device tensor holding stream samples
    [begin, begin+len(local)) as sample_window() prescribes.  The C-ABI addresses samples
    relative to sample 0 of the stream, so it is handed the virtual base local - begin."""
    import ctypes as C
    import torch
    from . import api
    esz = local.element_size()
    if out is None:
        out = torch.empty((count, sp.pitch), dtype=torch.float32, device=local.device)      # (sp.pitch = bins unless cfg.psd_pitch)
    if count == 0:
        return out
    st = C.c_void_p(torch.cuda.current_stream(local.device).cuda_stream)
    api._check(api.lib().glfer_hip_spectrogram_device(
        sp._h, C.c_void_p(local.data_ptr() - begin * esz), begin + local.numel(), first, count,
        out.data_ptr(), st), "glfer_hip_spectrogram_device")
    return out
