"""Where the GPU and this process's CPUs sit: the GPU's NUMA node (the library's own sysfs look-up), the node of every CPU the
process may use, and what pinned allocations therefore get.  python tools/numa_info.py"""
import ctypes as C
import glob
import os
import sys
sys.path.insert(0, '.')
import torch
import glfer_amd as G

L = G.api.lib()
bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
print("allowed CPUs:", sorted(os.sched_getaffinity(0)))
nodes = {}
for d in glob.glob("/sys/devices/system/node/node*"):
    n = int(d.rsplit("node", 1)[1])
    nodes[n] = open(d + "/cpulist").read().strip()
print("nodes:", nodes)
for f in glob.glob("/sys/bus/pci/devices/*/numa_node"):
    cls = open(os.path.dirname(f) + "/class").read().strip()
    if cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200"):
        print(os.path.dirname(f).rsplit("/", 1)[1], "class", cls, "numa_node", open(f).read().strip())
