#!/bin/bash
# End-of-round measurement pass on the MI355X box: tools/final_pass.sh <round-tag> [lines]   (lines: step 1 only)
#   1. unprofiled bench lines of every configuration (+ reader lines + ingest) -> gpurun_out/final_<tag>/
#   2. tools/prof_round.sh passes (kernel-trace stats + FETCH_SIZE / WRITE_SIZE) for the profiled configurations
T=$1
OUT=gpurun_out/final_$T
mkdir -p $OUT
line() { name=$1; shift; python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; tail -c 700 $OUT/$name.json | head -c 300; echo; }
Q="--no-cpu-baseline --no-ingest --min-seconds 1"
line headline
line cfg1 --config 1 $Q
line cfg3_zstd --config 3 --no-cpu-baseline
line zstd_fast --scheme 1 --clevel 0 $Q
line lz4_level0 --clevel 0 $Q
line cfg4 --config 4 $Q
line cfg5_b16 --config 5 --batch 16 --stack 32 $Q
line cfg5_b32 --config 5 $Q
line l3_lz4 --level 3 $Q
line mode0 --scheme 0 $Q
line lz4_d12 --depth 12 $Q
line zstd_d12 --scheme 1 --depth 12 $Q
line detector_like_lz4 --clustered --sparsity-ppm 11000 --depth 12 $Q
line detector_like_zstd --clustered --sparsity-ppm 11000 --depth 12 --scheme 1 $Q
line rehearsal_2ranks_gloo_shared_gpu --gpus 2 --shared-gpu --dist-backend gloo --stack 128 --min-seconds 1
line uint8_sources_lz4 --source-bytes 1 $Q
line uint8_sources_zstd --source-bytes 1 --scheme 1 $Q
line uint32_sources_lz4 --source-bytes 4 $Q
line uint32_sources_zstd --source-bytes 4 --scheme 1 $Q
line k2_3838x3710_lz4 --ny 3710 --nx 3838 --batch 64 --stack 128 $Q
line k2_3838x3710_zstd_d12 --ny 3710 --nx 3838 --batch 64 --stack 128 --scheme 1 --depth 12 $Q
line odd_1023x1023_lz4 --ny 1023 --nx 1023 --batch 1024 --stack 2048 $Q
line lz4_2pct_d12 --sparsity-ppm 20000 --depth 12 $Q
line zstd_2pct_d12 --sparsity-ppm 20000 --depth 12 --scheme 1 $Q
line l2_1pct --level 2 --sparsity-ppm 10000 $Q
line l2_clustered --level 2 --clustered --sparsity-ppm 2000 --depth 12 $Q
line dense_10pct --sparsity-ppm 100000 --stack 64 --batch 32 $Q
line dense_30pct --sparsity-ppm 300000 --stack 64 --batch 32 $Q
line dense_60pct --sparsity-ppm 600000 --stack 32 --batch 16 $Q
line headline_events_every_launch --kernel-events-every 1 $Q
line read_zstd --read --scheme 1 --steps 30 --warmup 5 --min-seconds 1
line read_lz4 --read --scheme 2 --steps 30 --warmup 5 --min-seconds 1
line read_zstd_fast --read --scheme 1 --clevel 0 --steps 30 --warmup 5 --min-seconds 1
line read_zstd_blob_on_device --read --scheme 1 --blob-on-device --steps 30 --warmup 5 --min-seconds 1
line read_cfg5 --read --scheme 1 --ny 8184 --nx 11520 --batch 16 --sparsity-ppm 50000 --depth 12 --steps 10 --warmup 2 --min-seconds 1
python3 tools/foreign_read_rate.py 1 64 32 > $OUT/foreign_zstd.log 2>&1; tail -2 $OUT/foreign_zstd.log
python3 tools/foreign_read_rate.py 2 64 32 > $OUT/foreign_lz4.log 2>&1; tail -2 $OUT/foreign_lz4.log
if [ "$2" = "lines" ]; then echo "done (lines only)"; exit 0; fi
echo "== profiles"
tools/prof_round.sh ${T}_lz4 > $OUT/prof_lz4.log 2>&1
tools/prof_round.sh ${T}_zstd --scheme 1 > $OUT/prof_zstd.log 2>&1
tools/prof_round.sh ${T}_cfg5 --config 5 > $OUT/prof_cfg5.log 2>&1
tools/prof_round.sh ${T}_cfg5_b16 --config 5 --batch 16 --stack 32 > $OUT/prof_cfg5_b16.log 2>&1
tools/prof_round.sh ${T}_cfg4 --config 4 > $OUT/prof_cfg4.log 2>&1
tools/prof_round.sh ${T}_d12 --depth 12 > $OUT/prof_d12.log 2>&1
tools/prof_round.sh ${T}_det_lz4 --clustered --sparsity-ppm 11000 --depth 12 > $OUT/prof_det_lz4.log 2>&1
tools/prof_round.sh ${T}_det_zstd --clustered --sparsity-ppm 11000 --depth 12 --scheme 1 > $OUT/prof_det_zstd.log 2>&1
tools/prof_round.sh ${T}_u32 --source-bytes 4 > $OUT/prof_u32.log 2>&1
tools/prof_round.sh ${T}_u8 --source-bytes 1 > $OUT/prof_u8.log 2>&1
tools/prof_round.sh ${T}_k2 --ny 3710 --nx 3838 --batch 64 --stack 128 > $OUT/prof_k2.log 2>&1
for s in 1 2; do tools/prof_bench.sh ${T}_read_s$s --read --scheme $s --steps 30 --warmup 5 --min-seconds 0.5 > $OUT/prof_read_s$s.log 2>&1; done
echo done
