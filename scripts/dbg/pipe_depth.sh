# the fresh-batch cadence of bench.py --mode step with builds started 1 / 2 steps ahead, for two lengths of the timed loop;
# usage: bash scripts/dbg/pipe_depth.sh [idgcn|idgin] [centres]
set -o pipefail
MODEL=${1:-idgcn}
CEN=${2:-4096}
for cfg in "1 20" "2 20" "1 30" "2 30"; do
  set -- $cfg
  MP_PIPE_DEPTH=$1 timeout -k 10 240 python bench.py --mode step --step-model $MODEL --centres $CEN --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d.get('step', d)
print('$MODEL c$CEN depth $1 steps $2:', {k: (round(v, 2) if isinstance(v, float) else v) for k, v in s.items() if k in ('ms_per_step', 'ms_per_step_fresh_batch', 'ms_per_step_fresh_batch_serial', 'batch_build_ms', 'host_enqueue_ms_per_fresh_step', 'driver_allocs_in_fresh_steps', 'driver_frees_in_fresh_steps', 'reserved_gb_after_fresh_steps', 'allocator_settings')})
" || echo "cfg $cfg failed"
done
