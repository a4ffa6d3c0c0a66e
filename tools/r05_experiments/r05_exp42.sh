#!/bin/bash
# round 5, experiment 42: the reduce kernel alone (--no-pipeline) against next to the previous batch's second stage: cfg 5, cfg 3, headline
O=gpurun_out/r05_exp42.log
: > $O
Q="--steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest"
for r in 1 2; do
  for c in "--config 5" "--config 5 --no-pipeline" "--config 3" "--config 3 --no-pipeline" "" "--no-pipeline"; do
    python3 bench.py $Q $c 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-28s %8.0f fps  step %.4f  kernel %.4f  whole %.3f  stages %s' % ('$c', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['whole_path_frac'], j.get('stage_ms_per_step')))" >> $O
  done
done
echo done >> $O
