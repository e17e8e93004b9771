# repeatability of the multi-stream bench configurations (bench.py --streams S --join step|end)
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"; }
for rep in 1 2 3; do
for S in 2 3; do for J in step end; do $B --streams $S --join $J 2>&1 | tail -1 | show "S=$S $J"; done; done
done
