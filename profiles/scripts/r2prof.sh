#!/bin/bash
# usage: profiles/scripts/r2prof.sh <tag>   (run on the GPU box from the repo root)
set -o pipefail
T=$1; OUT=gpurun_out/$T; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_pipe_256.json 2> $OUT/bench.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 bench.py --no-cpu-baseline > $OUT/bench_pipe_256_under_rocprofv3.json 2> $OUT/stats.err && echo "stats done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o b -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err && echo "fetch done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o b -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 > $OUT/pmc_write.json 2> $OUT/pmc_write.err && echo "write done" &&
python3 profiles/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/traffic.json 10 0 10 > $OUT/traffic.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats
python3 - <<PY
import json
for f in ("bench_pipe_256","bench_pipe_256_under_rocprofv3"):
    j=json.load(open("$OUT/%s.json"%f)); print(f, j["ms_per_step"], j["value"], j["roofline"]["avg_launch_ms"], {k:(v["ms_total"]/max(v["launches"],1)) for k,v in j["kernel_ms"].items()})
PY
head -12 $OUT/kernel_stats.csv
