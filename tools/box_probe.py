"""What kind of box is this?  The pool has boxes on which the same build runs every kernel 25-70 % longer.  Prints
the device's clocks / power state / partition modes as the driver reports them, a device-to-device copy rate, and the times of
the step's kernels on the bench scene, so that the two kinds can be told apart from their records."""
import glob, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout.strip()
    except Exception as e:  # noqa
        return f"<{e}>"


print("== host:", sh("nproc"), "cpus;", sh("lscpu | grep -E 'Model name|Socket|Thread|NUMA node\\(s\\)' | tr -s ' ' | tr '\\n' ';'"))
print("== rocm-smi")
print(sh("rocm-smi --showclocks --showperflevel --showpower --showmaxpower --showcomputepartition --showmemorypartition --showtemp 2>&1 | grep -v '^=*$' | head -60"))
for f in sorted(glob.glob("/sys/class/drm/card*/device/power_dpm_force_performance_level") + glob.glob("/sys/class/drm/card*/device/pp_dpm_*clk")
                + glob.glob("/sys/class/drm/card*/device/current_compute_partition") + glob.glob("/sys/class/drm/card*/device/current_memory_partition")):
    try:
        print(f, "->", open(f).read().strip().replace("\n", " | "))
    except OSError as e:
        print(f, "->", e)
p = torch.cuda.get_device_properties(0)
print("== torch:", p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2 ** 30, 1), "clock MHz", getattr(p, "clock_rate", 0) / 1e3,
      "mem clock MHz", getattr(p, "memory_clock_rate", 0) / 1e3, "bus", getattr(p, "memory_bus_width", None), "L2 MiB", getattr(p, "L2_cache_size", 0) / 2 ** 20)
dev = torch.device("cuda:0")
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("== device copy 1 GiB: %.1f us -> %.2f TB/s (read + write)" % (dt * 1e6, 2 * a.numel() * 4 / dt / 1e12))
del a, b
print("== VALU rate / shader clock under load")
print(sh(os.path.join(ROOT, "tools/ubench/valu_rate") + " 2>&1 | grep -E 'v_fma_f32|v_exp_f32|v_pk_fma' | head -6"))
print("== clocks right after load")
print(sh("rocm-smi --showclocks 2>&1 | grep -E 'sclk|mclk|fclk|socclk' | head -8"))
