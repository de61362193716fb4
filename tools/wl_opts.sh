mkdir -p gpurun_out/s4g
for o in "" "--opt neq_persist=0" "--opt nodes_per_block=8" "--opt nodes_per_block=4" "--opt nodes_per_block=4 --opt neq_persist=0" "--opt nodes_per_block=8 --opt neq_persist=0"; do
  for c in "" "--cells"; do
    timeout 120 python bench.py --mode search --engine worklist $c $o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$c $o', round(d['ms_per_step'], 1), 'ms', '%.3g' % d['config']['nodes_per_s'], 'nodes/s', d['config']['nodes'])
"
  done
done > gpurun_out/s4g/wl.txt 2>&1
cat gpurun_out/s4g/wl.txt
