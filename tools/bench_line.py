"""Development helper: run bench.py with the given arguments and print one short line (value, ms/step, kernel ms, fractions)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-ingest"] + sys.argv[1:], capture_output=True, text=True).stdout
j = json.loads(out.strip().splitlines()[-1])
r = j["roofline"]
print("%s: %.0f %s  %.4f ms/step  kernel %.4f ms  whole_path_frac %.3f  frac %.3f  verified %s" %
      (" ".join(sys.argv[1:]), j["value"], j["unit"], j["ms_per_step"], r.get("kernel_ms", 0), r.get("whole_path_frac", 0), r["frac"], j.get("verified")))
