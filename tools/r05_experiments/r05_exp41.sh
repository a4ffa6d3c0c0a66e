#!/bin/bash
# round 5, experiment 41: cfg 5 with 16 / 32 / 48 / 64 frames per step (the turnaround between two reduce kernels is per step), one box, interleaved
O=gpurun_out/r05_exp41.log
: > $O
Q="--config 5 --steps 20 --warmup 5 --min-seconds 0.7 --no-cpu-baseline --no-ingest"
for r in 1 2 3; do
  for b in 32 64 16 48; do
    python3 bench.py $Q --batch $b --stack $((2*b)) 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('B=%s  %8.0f fps  step %.4f  kernel %.4f  whole %.3f  verified %s' % ('$b', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['whole_path_frac'], j['verified']))" >> $O
  done
done
echo done >> $O
