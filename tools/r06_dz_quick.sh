#!/bin/bash
# round 6: the device-zlib tests + a short bench line of --scheme 0 --device-zlib (and whatever extra bench arguments follow)
set -o pipefail
tag=${1:-dz}; shift
python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -4 || exit 1
python bench.py --scheme 0 --device-zlib --no-ingest --no-cpu-baseline "$@" > gpurun_out/r06_${tag}.json 2> gpurun_out/r06_${tag}.err || { tail -5 gpurun_out/r06_${tag}.err; exit 1; }
python - <<PY
import json
r = json.loads(open("gpurun_out/r06_${tag}.json").read().strip().splitlines()[-1])
print("${tag}: %.0f frames/s  whole %.4f  kernel %.4f ms (frac %.4f)  step %.4f ms  rec %.0f B  verified %s" % (r["value"], r["roofline"]["whole_path_frac"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["ms_per_step"], r["config"]["record_bytes_per_frame"], r["verified"]))
PY
