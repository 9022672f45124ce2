#!/bin/bash
# usage: profiles/scripts/r3prof.sh <tag>   (run on the GPU box from the repo root)
# 1. the driver's command (bench.py with the 512^3 target leg and the CPU baseline)   2. rocprofv3 --kernel-trace --stats of the
# headline part alone (--no-target-512: the 512^3 leg launches the same kernel on other sizes and would mix into the average)
# 3. the two --pmc passes for the HBM traffic of the collide kernel (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only)
set -o pipefail
T=$1; OUT=gpurun_out/$T; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench.err && echo "driver command done" &&
python3 bench.py --no-cpu-baseline --no-target-512 > $OUT/bench_pipe_256.json 2>> $OUT/bench.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 bench.py --no-cpu-baseline --no-target-512 > $OUT/bench_pipe_256_under_rocprofv3.json 2> $OUT/stats.err && echo "stats done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o b -- python3 bench.py --no-cpu-baseline --no-target-512 --steps 40 --warmup 10 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err && echo "fetch done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o b -- python3 bench.py --no-cpu-baseline --no-target-512 --steps 40 --warmup 10 > $OUT/pmc_write.json 2> $OUT/pmc_write.err && echo "write done" &&
python3 profiles/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/traffic.json 10 0 10 > $OUT/traffic.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats
python3 - <<PY
import json
for f in ("bench_driver_command", "bench_pipe_256","bench_pipe_256_under_rocprofv3"):
    j=json.load(open("$OUT/%s.json"%f)); print(f, j["ms_per_step"], j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], {k:(v["ms_total"]/max(v["launches"],1)) for k,v in j["kernel_ms"].items()})
PY
head -12 $OUT/kernel_stats.csv; cat $OUT/traffic.json | head -12
