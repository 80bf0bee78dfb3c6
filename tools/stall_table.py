#!/usr/bin/env python3
"""Condense gpurun_out/stall_<tag>/summary.txt (tools/stall_pass.sh: SQ counter groups, one rocprofv3 --pmc pass
each, bench.py --no-secondary) into the per-wavefront time budget and the pipe occupancies of the body kernel.
usage: stall_table.py <tag> <kernel-substring> <frames-per-launch> <kernel-ms-of-the-stats-pass>"""
import re, sys
tag, kname, frames, ms = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
cur, c = None, {}
for ln in open("gpurun_out/stall_%s/summary.txt" % tag):
    if not ln.startswith(" "):
        cur = ln.strip()
        continue
    if kname in cur:
        m = re.match(r"\s+(\S+)\s+(\d+)", ln)
        c[m.group(1)] = float(m.group(2))
cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # per XCD: the kernel's duration in shader clocks
wave = c["SQ_WAVE_CYCLES"]                            # wavefront residency, in units of 4 clocks
print("kernel           : %s" % kname)
print("duration         : %.0f clocks per XCD (GRBM_GUI_ACTIVE / 8) under the counter pass = %.2f GHz x the %.3f ms of the --stats pass"
      % (cyc, cyc / (ms * 1e6), ms))
print("wavefronts       : %d per launch, resident %.2f per SIMD on average (SQ_WAVE_CYCLES x 4 / (1024 SIMDs x duration))"
      % (c["SQ_WAVES"], wave * 4 / (1024 * cyc)))
print("a wavefront's time (shares of SQ_WAVE_CYCLES; an instruction of a wave64 occupies the wave for >= 4 clocks = 1 unit):")
for n, label in (("SQ_ACTIVE_INST_VALU", "executing VALU"), ("SQ_ACTIVE_INST_LDS", "executing LDS instructions"),
                 ("SQ_ACTIVE_INST_VMEM", "executing vector memory instructions"), ("SQ_ACTIVE_INST_SCA", "executing scalar instructions"),
                 ("SQ_ACTIVE_INST_MISC", "branches, s_barrier issue, messages"), ("SQ_ACTIVE_INST_ANY", "= executing anything"),
                 ("SQ_WAIT_INST_ANY", "waiting at an s_waitcnt (any counter)"), ("SQ_WAIT_INST_LDS", "  of which: for LDS (lgkmcnt)"),
                 ("SQ_WAIT_ANY", "waiting for anything else (barrier partners, issue arbitration, instruction fetch)")):
    print("   %-22s %5.1f %%   %s" % (n, 100.0 * c[n] / wave, label))
print("pipes (per SIMD / per CU, against the kernel's duration):")
valu = c["SQ_INSTS_VALU"]
print("   VALU   %7.0f wave-instructions per frame; one per %.2f clocks per SIMD (floor 2: two wavefronts taking turns; a lone wavefront issues one per 4)"
      % (valu / frames, 1024 * cyc / valu))
print("          => vector ALU busy %.0f %% of the clocks at 2 clocks per instruction" % (100 * 2 * valu / (1024 * cyc)))
print("   LDS    %7.0f wave-instructions per frame; LDS array active %.0f %% of the clocks (SQ_LDS_IDX_ACTIVE / (256 CUs x duration)); bank conflicts %.0f cycles"
      % (c["SQ_INSTS_LDS"] / frames, 100 * c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), c["SQ_LDS_BANK_CONFLICT"]))
print("   VMEM   %7.1f loads + %.1f stores (wave-instructions) per frame; on average %.1f outstanding per CU (SQ_INST_LEVEL_VMEM / duration / 256)"
      % (c["SQ_INSTS_VMEM_RD"] / frames, c["SQ_INSTS_VMEM_WR"] / frames, c["SQ_INST_LEVEL_VMEM"] * 4 / cyc / 256))
print("   SALU   %7.0f per frame, branches %.0f;  instruction fetches %.0f per frame" % (c["SQ_INSTS_SALU"] / frames, c["SQ_INSTS_BRANCH"] / frames, c["SQ_IFETCH"] / frames))
