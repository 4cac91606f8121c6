#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  echo -n "default:                      "; timeout 200 python tools/dropin_rate.py 1500 2>/dev/null
  echo -n "ROC_ACTIVE_WAIT_TIMEOUT=2000: "; ROC_ACTIVE_WAIT_TIMEOUT=2000 timeout 200 python tools/dropin_rate.py 1500 2>/dev/null
  echo -n "HSA_ENABLE_INTERRUPT=0:       "; HSA_ENABLE_INTERRUPT=0 timeout 200 python tools/dropin_rate.py 1500 2>/dev/null
done
