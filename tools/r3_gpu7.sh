#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -25; done
