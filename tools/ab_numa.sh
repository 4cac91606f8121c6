#!/bin/bash
# does the host side of the loops care which socket it runs on?  bench under taskset: the GPU's own NUMA node, the other one, unpinned
cd "$GRAFT_REPO_ROOT"
read ADDR LOCAL OTHER <<< $(python - <<'PY'
import torch, os
p = torch.cuda.get_device_properties(0)
addr = "%04x:%02x:%02x.0" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
node = open("/sys/bus/pci/devices/%s/numa_node" % addr).read().strip()
loc = open("/sys/bus/pci/devices/%s/local_cpulist" % addr).read().strip()
oth = open("/sys/devices/system/node/node%d/cpulist" % (1 - int(node))).read().strip()
print(addr, loc, oth)
PY
)
echo "GPU $ADDR local cpus $LOCAL other node $OTHER"
run() { timeout 300 "$@" python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-10s bwd %.2f us  fwd %.2f us  | dropin %.0f  torch-l1 %.0f  train.py-loss %.0f  synced %.0f  run-ahead %.0f it/s' % ('$TAG', r['avg_kernel_ms']*1e3, r['composite_fwd']['avg_kernel_ms']*1e3, d['value'], d['iters_per_sec_dropin_reference_loop_torch_l1'], d['iters_per_sec_dropin_reference_loop_train_py_loss'], d['iters_per_sec_one_call_synced'], d['iters_per_sec_run_ahead']))"; }
for rep in 1 2; do
  TAG=unpinned; run env
  TAG=local; run taskset -c $LOCAL
  TAG=remote; run taskset -c $OTHER
done
