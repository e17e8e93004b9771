# multi-stream bench configurations (bench.py --streams S --join step|end), three repeats of the two candidates
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})"; }
for S in 1 2 3 4; do for J in step end; do $B --streams $S --join $J 2>&1 | tail -1 | show "S=$S $J"; done; done
for rep in 1 2; do $B --streams 2 --join end 2>&1 | tail -1 | show "repeat S=2 end"; $B --streams 3 --join end 2>&1 | tail -1 | show "repeat S=3 end"; done
