for q in 0 2 3 4; do
  PUTSLAM_HIP_QSPLIT=$q python3 bench.py --streams 1 --steps 10 --warmup 5 --repeats 3 --error-version 0 --estimator ransac --hyp 487 --no-cpu-baseline --no-other-modes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qsplit $q', round(d['value']), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
