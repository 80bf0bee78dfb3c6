#!/usr/bin/env python3
"""Instruction-mix summary of a hipcc -save-temps .s file: per kernel, counts by class."""
import collections
import re
import sys

def main(path, pat=""):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for idx, (i, name) in enumerate(starts):
        if pat not in name:
            continue
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        c = collections.Counter()
        inloop = 0
        for l in lines[i:end]:
            t = l.strip()
            if not t or t[0] in ".;/" or t.endswith(":"):
                continue
            op = t.split()[0]
            if op == "s_endpgm":
                break
            c[op] += 1
        g = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_accvgpr"): g["v_accvgpr"] += v
            elif k.startswith(("v_fma", "v_fmac")): g["v_fma*"] += v
            elif k.startswith(("v_mul", "v_add_f32", "v_sub")): g["v_mul/add/sub_f32"] += v
            elif k.startswith("v_pk"): g["v_pk*"] += v
            elif k.startswith("v_mov"): g["v_mov"] += v
            elif k.startswith("v_"): g["v_other"] += v
            elif k.startswith("ds_"): g[k] += v
            elif k.startswith(("global_", "buffer_", "scratch_", "flat_")): g[k] += v
            elif k.startswith("s_waitcnt"): g["s_waitcnt"] += v
            else: g["s_other"] += v
        tot = sum(c.values())
        print(name)
        print("  total %d  valu %d" % (tot, sum(v for k, v in c.items() if k.startswith("v_"))))
        print("  " + ", ".join("%s=%d" % kv for kv in sorted(g.items(), key=lambda x: -x[1])))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
