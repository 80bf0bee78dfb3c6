#!/bin/bash
# Host-to-host rate with the calling process on the socket FAR from the GPU, pinned memory placed where the thread runs (GLFER_NUMA_BIND=0)
# against beside the GPU (the default: glfer_hip_host_alloc and the ring's staging bind the allocating thread to the GPU's node for the call).
cd "${GRAFT_REPO_ROOT:-.}"
NODE=$(python3 - <<'PY'
import glob, os
# the visible GPU: the first display-class PCI function ROCm exposes is not knowable from here; take the node the library reports
import sys
sys.path.insert(0, '.')
import ctypes as C
import torch
import glfer_amd as G
L = G.api.lib()
bus = C.create_string_buffer(64)
import subprocess
hip = C.CDLL("libamdhip64.so")
hip.hipInit(0)
hip.hipDeviceGetPCIBusId(bus, 64, 0)
print(L.glfer_hip_numa_node_of_bus_id(bus.value.lower(), None))
PY
)
echo "GPU 0 sits on NUMA node $NODE"
FAR=$((1 - NODE))
CPUS=$(cat /sys/devices/system/node/node$FAR/cpulist)
for B in 0 1; do
  echo "== process on node $FAR (cpus $CPUS), GLFER_NUMA_BIND=$B"
  GLFER_NUMA_BIND=$B timeout -k 10 200 taskset -c $CPUS python3 tools/chunk_probe.py 2>/dev/null | grep "dflt"
done
NEAR=$(cat /sys/devices/system/node/node$NODE/cpulist)
echo "== process on node $NODE (cpus $NEAR), GLFER_NUMA_BIND=0"
GLFER_NUMA_BIND=0 timeout -k 10 200 taskset -c $NEAR python3 tools/chunk_probe.py 2>/dev/null | grep "dflt"
