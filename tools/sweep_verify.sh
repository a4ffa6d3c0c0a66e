run() { timeout -k 10 120 python3 bench.py "$@" --steps 10 --warmup 3 --min-seconds 0.2 --no-cpu-baseline --no-ingest 2>/dev/null | python3 -c "
import sys, json
try:
    j = json.loads(sys.stdin.readlines()[-1]); print('%-66s %9.0f fps  whole %.3f  rec %9.0f %s' % (sys.argv[1], j['value'], j['roofline']['whole_path_frac'], j['config']['record_bytes_per_frame'], 'ok' if j['verified'] else 'NOT VERIFIED'))
except Exception as e: print('%-66s ERROR %r' % (sys.argv[1], e))" "$*"; }
while read -r cfg; do [ -z "$cfg" ] && continue; run $cfg; done <<CFGS
--source-bytes 1 --depth 1
--source-bytes 1 --depth 5 --scheme 1
--source-bytes 1 --depth 8 --scheme 8
--source-bytes 1 --depth 8 --level 3
--source-bytes 1 --depth 8 --level 2
--source-bytes 1 --depth 7 --scheme 0
--source-bytes 1 --depth 8 --clustered --sparsity-ppm 11000
--source-bytes 1 --depth 8 --ny 3710 --nx 3838 --stack 128
--source-bytes 4 --depth 17
--source-bytes 4 --depth 24 --scheme 1
--source-bytes 4 --depth 31 --scheme 8
--source-bytes 4 --depth 32 --scheme 0
--source-bytes 4 --depth 20 --clustered --sparsity-ppm 11000
--source-bytes 4 --depth 20 --sparsity-ppm 50000 --batch 32 --stack 64
--source-bytes 4 --depth 20 --ny 1023 --nx 1023 --batch 256 --stack 512
--level 2 --depth 12 --sparsity-ppm 1000
--level 2 --scheme 0 --sparsity-ppm 1000
--level 2 --clustered --sparsity-ppm 2000 --depth 12
--level 2 --ny 3710 --nx 3838 --stack 128 --sparsity-ppm 1000
--level 3 --ny 1023 --nx 1023 --batch 512 --stack 1024
--scheme 1 --ny 1023 --nx 1023 --batch 512 --stack 1024
--scheme 8 --ny 3710 --nx 3838 --stack 128
--clevel 0 --sparsity-ppm 20000 --depth 12
--scheme 1 --clevel 0 --clustered --sparsity-ppm 11000 --depth 12
--config 5 --batch 8 --stack 16
--ny 8184 --nx 11520 --batch 8 --stack 16 --scheme 2 --sparsity-ppm 50000 --depth 12
--ny 8184 --nx 11520 --batch 8 --stack 16 --level 2 --scheme 8 --sparsity-ppm 1000
CFGS
