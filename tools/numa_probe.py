#!/usr/bin/env python3
"""Developer probe: where the GPU sits (PCI bus id -> NUMA node, local CPUs), where this process may run, and on which NUMA
node pinned host memory from mi_blur_host_alloc lands (move_pages query), first-touched from a local and from a remote CPU."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def node_of(addr, libc):
    pages = (C.c_void_p * 1)(addr)
    status = (C.c_int * 1)(-1)
    rc = libc.syscall(279, 0, 1, pages, None, status, 0)        # move_pages(pid 0, query)
    return status[0] if rc == 0 else f"errno {C.get_errno()}"


def main():
    import torch
    libc = C.CDLL("libc.so.6", use_errno=True)
    print("affinity mask:", len(os.sched_getaffinity(0)), "cpus; cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
    try:
        print("cpuset.cpus.effective:", open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip())
        print("cpuset.mems.effective:", open("/sys/fs/cgroup/cpuset.mems.effective").read().strip())
    except OSError as e:
        print("cpuset:", e)
    p = torch.cuda.get_device_properties(0)
    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    print("GPU 0:", p.name, bdf)
    base = f"/sys/bus/pci/devices/{bdf}"
    for f in ("numa_node", "local_cpulist", "current_link_speed", "current_link_width", "max_link_speed", "max_link_width"):
        try:
            print(" ", f, "=", open(os.path.join(base, f)).read().strip())
        except OSError as e:
            print(" ", f, ":", e)
    for n in sorted(os.listdir("/sys/devices/system/node")):
        if n.startswith("node"):
            print(n, "cpus", open(f"/sys/devices/system/node/{n}/cpulist").read().strip())
    print("this thread runs on cpu", libc.sched_getcpu())
    pkg = entry.load_package()
    L = pkg.lib()
    a = L.mi_blur_host_alloc(64 << 20)
    print("mi_blur_host_alloc(64 MiB): page 0 on node", node_of(a, libc), " page mid on node", node_of(a + (32 << 20), libc))
    L.mi_blur_host_free(a)


if __name__ == "__main__":
    main()
