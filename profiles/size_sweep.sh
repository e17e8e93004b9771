# throughput across keypoint counts / estimators (looks for cliffs in the split heuristics); one line per config
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), 'pairs/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms'].items()})"; }
for K in 250 500 1000 2000 4000 8000; do
  F=500; if [ $K -ge 4000 ]; then F=100; fi
  for E in "fixed 4096" "ransac 487"; do set -- $E
    python bench.py --frames $F --kpts $K --estimator $1 --hyp $2 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | show "kpts=$K frames=$F $1 H=$2"
  done
done
