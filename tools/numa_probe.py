"""Which NUMA node does each GPU hang off, as sysfs tells it, and does torch's PCI identity of device 0 find its sysfs entry?"""
import glob, os, torch
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        real = os.path.basename(os.path.realpath(d))
        rd = lambda f: open(os.path.join(d, f)).read().strip()
        print(real, "numa_node", rd("numa_node"), "local_cpulist", rd("local_cpulist"), "vendor", rd("vendor"), "device", rd("device"))
    except OSError as e:
        print(d, e)
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(os.path.basename(n), open(os.path.join(n, "cpulist")).read().strip())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i)
    addr = "%04x:%02x:%02x.0" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
    path = "/sys/bus/pci/devices/" + addr
    print("torch device", i, addr, "exists" if os.path.isdir(path) else "MISSING", open(path + "/numa_node").read().strip() if os.path.isdir(path) else None)
print("affinity", len(os.sched_getaffinity(0)))
