#!/bin/bash
# The HOST side of an N-rank run, measured on ONE GPU (the only part of the 8-GPU scaling run a one-GPU box can show): N ranks share
# cuda:0 (at most 6 processes may use the card on the pool's boxes, and the launcher's agent counts: 5 ranks), each with its own step loop, stack and ctx.  --no-collective: the
# ranks meet only at the fences, so what is compared with the single rank is N Python loops + N x the launches on the box's granted
# cores; the gloo forms add the rehearsal backend's host copies per gather.  usage: tools/host_side_rehearsal.sh <outdir> [N...]
OUT=$1; shift
RANKS=${@:-"2 4 5"}
mkdir -p $OUT
COMMON="--stack 64 --steps 100 --warmup 10 --min-seconds 1.0"
python3 bench.py $COMMON --no-cpu-baseline --no-ingest > $OUT/ranks1.json 2> $OUT/ranks1.err
for n in $RANKS; do
  python3 bench.py --gpus $n --shared-gpu --dist-backend gloo --no-collective $COMMON > $OUT/ranks${n}_no_collective.json 2> $OUT/ranks${n}_no_collective.err
done
n=$(echo $RANKS | awk '{print $NF}')
python3 bench.py --gpus $n --shared-gpu --dist-backend gloo --gather-every 0 $COMMON > $OUT/ranks${n}_gloo_gather_once.json 2> $OUT/ranks${n}_gloo_gather_once.err
python3 bench.py --gpus $n --shared-gpu --dist-backend gloo $COMMON > $OUT/ranks${n}_gloo_gather_every_step.json 2> $OUT/ranks${n}_gloo_gather_every_step.err
python3 bench.py --gpus $n --shared-gpu --dist-backend gloo --no-collective --no-pin $COMMON > $OUT/ranks${n}_no_collective_unpinned.json 2> $OUT/ranks${n}_no_collective_unpinned.err
python3 - $OUT <<'PY'
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append((os.path.basename(f), "FAILED: %r" % e)); continue
    rows.append((os.path.basename(f), "%d ranks  %9.0f frames/s  %.4f ms/step  enqueue %.1f us/step (%.3f of a step)  issue-in-region %.1f us  pinned %s  verified %s gather %s" % (
        j["n_gpus"], j["value"], j["ms_per_step"], j["host_enqueue_us_per_step"], j["host_enqueue_frac_of_step"], j["issue_us_per_step_in_timed_region"],
        j["config"].get("cpus_pinned"), j["verified"], j["gather_verified"])))
for r in rows: print("%-44s %s" % r)
PY
