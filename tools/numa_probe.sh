#!/bin/bash
# Where the GPU sits relative to the CPUs and memory this job may use, then the host <-> device copy rates (tools/pcie_bw.py).
echo "cpus allowed: $(grep Cpus_allowed_list /proc/self/status)"; echo "mems allowed: $(grep Mems_allowed_list /proc/self/status)"
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) $(grep MemFree $n/meminfo | tr -s ' ')"; done
python3 - <<'P'
import torch, os, glob
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
print("gpu", bdf)
for f in ("numa_node", "local_cpulist"):
    try:
        print(f, open("/sys/bus/pci/devices/%s/%s" % (bdf, f)).read().strip())
    except Exception as e:
        print(f, "?", e)
print("this process runs on cpu", os.sched_getcpu())
P
timeout -k 10 120 python3 tools/pcie_bw.py
