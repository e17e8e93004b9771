#!/bin/bash
# re-run one failing batch-fuzz worker under knob settings (NAME:ENV=V,ENV=V ...)
seed=${SEED:-3203}; iters=${ITERS:-5}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    echo "== $name: $(python3 tests/fuzz_gpu.py --batch --iters $iters --seed $seed 2>&1 | grep -c MISMATCH) mismatches" )
done
