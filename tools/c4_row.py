"""bench.py's c4_as_worded row alone (a 1-hour WAV through the workers handle), with WORKERS workers sharing GPU 0.  python tools/c4_row.py [workers]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, glfer_amd as G
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print(json.dumps(bench.c4_as_worded(torch, G, 0, [0] * w), indent=1))
