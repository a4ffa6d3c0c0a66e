"""Copies the artefacts of tools/final_pass.sh <tag> from gpurun_out/ into profiles/ (what the judge reads):
   profiles/<round>_final_lines/*, <round>_bench_profiles.md, <round>_bench_kernel_stats_<cfg>.csv, <round>_bench_under_rocprof_<cfg>.json
usage: python tools/collect_final.py <tag> <round>      e.g.  r02c r02"""
import glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
ARGS = {"lz4": "", "zstd": "--scheme 1", "zstd_fast": "--scheme 1 --clevel 0", "read_s1": "--read --scheme 1", "read_s2": "--read --scheme 2",
        "cfg5": "--ny 8184 --nx 11520 --batch 32 --stack 64 --sparsity-ppm 50000 --scheme 1 --depth 12",
        "cfg5_b16": "--ny 8184 --nx 11520 --batch 16 --stack 32 --sparsity-ppm 50000 --scheme 1 --depth 12",
        "cfg4": "--scheme 8 --level 2 --sparsity-ppm 1000", "d12": "--depth 12",
        "det_lz4": "--clustered --sparsity-ppm 11000 --depth 12", "det_zstd": "--clustered --sparsity-ppm 11000 --depth 12 --scheme 1",
        "u32": "--source-bytes 4", "u8": "--source-bytes 1", "k2": "--ny 3710 --nx 3838 --batch 64 --stack 128",
        "l2_1pct": "--level 2 --sparsity-ppm 10000", "l2_clustered": "--level 2 --clustered --sparsity-ppm 2000 --depth 12"}
os.makedirs("profiles/%s_final_lines" % rnd, exist_ok=True)
for f in glob.glob("gpurun_out/final_%s/*.json" % tag):
    shutil.copy(f, "profiles/%s_final_lines/%s" % (rnd, os.path.basename(f)))
hp = "profiles/%s_bench_profiles.md" % rnd
head = (open(hp).read().split("\n## ")[0].rstrip() if os.path.exists(hp) else
        "# %s: bench.py under rocprofv3 (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes), tools/final_pass.sh %s" % (rnd, tag)) + "\n"
out = [head]
for cfg, args in ARGS.items():
    d = "gpurun_out/prof_%s_%s" % (tag, cfg)
    if not os.path.isdir(d):
        continue
    line = [l for l in open(d + "/bench_under_rocprof.json").read().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    r = j["roofline"]
    open("profiles/%s_bench_under_rocprof_%s.json" % (rnd, cfg), "w").write(line + "\n")
    ks = sorted(glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True), key=os.path.getsize)[-1]
    shutil.copy(ks, "profiles/%s_bench_kernel_stats_%s.csv" % (rnd, cfg))
    out.append("\n## %s  (`bench.py %s`)\n\nbench line under rocprofv3: %.0f frames/s, %.4f ms/step, `roofline.kernel_ms` %.4f (HIP events), frac %.4f, whole_path_frac %.4f, "
               "%.0f B/record, verified=%s\n\n%s" % (cfg, args, j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["whole_path_frac"],
                                                      j["config"].get("record_bytes_per_frame", 0), j.get("verified"), open(d + "/summary.md").read()))
open("profiles/%s_bench_profiles.md" % rnd, "w").write("".join(out))
# profiles/traffic.json: PMC-derived HBM bytes per launch of the dominant kernel, per configuration key (bench.py reads it)
import re
KEYS = {"lz4": "4096x4096_b64_ppm10000_d16_s2", "zstd": "4096x4096_b64_ppm10000_d16_s1", "cfg5": "8184x11520_b32_ppm50000_d12_s1",
        "cfg5_b16": "8184x11520_b16_ppm50000_d12_s1", "cfg4": "4096x4096_b64_ppm1000_d16_s8_l2", "d12": "4096x4096_b64_ppm10000_d12_s2",
        "zstd_fast": None, "det_lz4": "4096x4096_b64_ppm11000_d12_s2_clustered", "det_zstd": "4096x4096_b64_ppm11000_d12_s1_clustered",
        "u32": "4096x4096_b64_ppm10000_d20_s2_u32", "u8": "4096x4096_b64_ppm10000_d8_s2_u8", "k2": "3710x3838_b64_ppm10000_d16_s2",
        "l2_1pct": "4096x4096_b64_ppm10000_d16_s2_l2", "l2_clustered": "4096x4096_b64_ppm2000_d12_s2_clustered_l2"}
tj = json.load(open("profiles/traffic.json"))
for cfg, key in KEYS.items():
    d = "gpurun_out/prof_%s_%s" % (tag, cfg)
    if not key or not os.path.isdir(d):
        continue
    rows = [l for l in open(d + "/summary.md") if l.startswith("| k_reduce_tiles")]
    rows.sort(key=lambda l: -int(l.split("|")[2]))          # the instantiation with the most calls = the steady-state kernel
    c = [x.strip() for x in rows[0].split("|")]
    tj[key] = {"reduce_kernel_hbm_bytes_per_launch": int(round(float(c[7]) / 1e6) * 1e6),
               "source": "profiles/%s_bench_profiles.md (%s): (2*FETCH_SIZE + WRITE_SIZE) KiB of %s, mean over the dispatches of the two PMC runs" % (rnd, cfg, c[1])}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
print("collected", tag, "->", rnd)
