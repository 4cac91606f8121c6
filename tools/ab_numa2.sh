#!/bin/bash
# finer than the node: one CCD (8 cores sharing an L3) of the GPU's node, with and without the SMT siblings
cd "$GRAFT_REPO_ROOT"
read ADDR LOCAL <<< $(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
addr = "%04x:%02x:%02x.0" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
print(addr, open("/sys/bus/pci/devices/%s/local_cpulist" % addr).read().strip())
PY
)
F=${LOCAL%%-*}; S=$((F+128))
echo "GPU $ADDR local cpus $LOCAL; first CCD $F-$((F+7)) siblings $S-$((S+7)); L3 of cpu $F: $(cat /sys/devices/system/cpu/cpu$F/cache/index3/shared_cpu_list)"
run() { timeout 300 "$@" python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-12s dropin %.0f  torch-l1 %.0f  train.py-loss %.0f  synced %.0f  run-ahead %.0f it/s' % ('$TAG', d['value'], d['iters_per_sec_dropin_reference_loop_torch_l1'], d['iters_per_sec_dropin_reference_loop_train_py_loss'], d['iters_per_sec_one_call_synced'], d['iters_per_sec_run_ahead']))"; }
for rep in 1 2; do
  TAG=node; run env
  TAG=ccd+smt; run taskset -c $F-$((F+7)),$S-$((S+7))
  TAG=ccd; run taskset -c $F-$((F+7))
  TAG=2ccd; run taskset -c $F-$((F+15))
done
