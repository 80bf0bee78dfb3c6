"""Condense tools/stall_pass.sh's counters of the HP-ARMA kernel into profiles/<tag>_hparma_issue.json (per frame, and the shares of a
wavefront's time).  usage: python tools/hparma_issue.py gpurun_out/stall_<tag>/summary.txt <frames per launch> <round tag>"""
import json
import os
import sys

path, frames, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
c, kernel = {}, None
for line in open(path):
    w = line.split()
    if "hparma_kernel" in line:
        kernel = line.strip()
    elif kernel and len(w) >= 2 and w[0].isupper():
        c[w[0]] = float(w[1])
dur = c["GRBM_GUI_ACTIVE"] / 8.0                                   # clocks per XCD
wave = c["SQ_WAVE_CYCLES"]                                         # units of 4 clocks
out = {"kernel": kernel, "frames_per_launch": frames, "clocks_per_xcd": dur,
       "wavefronts_per_simd_resident": wave * 4 / (1024 * dur),
       "SQ_INSTS_VALU_per_frame": c["SQ_INSTS_VALU"] / frames, "SQ_INSTS_LDS_per_frame": c["SQ_INSTS_LDS"] / frames,
       "SQ_INSTS_SALU_per_frame": c["SQ_INSTS_SALU"] / frames,
       "valu_clocks_per_instr_per_simd": 1024 * dur / c["SQ_INSTS_VALU"],
       "valu_issue_share_at_4_clocks": 4 * c["SQ_INSTS_VALU"] / (1024 * dur),
       "lds_array_busy": c["SQ_LDS_IDX_ACTIVE"] / (256 * dur), "lds_bank_conflict_share_of_active": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
       "wave_time_shares": {k: c[k] / wave for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
                                                     "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY") if k in c},
       "counters_per_launch": c, "made_by": "tools/stall_pass.sh hparma <tag>; tools/hparma_issue.py"}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", tag + "_hparma_issue.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters_per_launch"}, indent=1))
