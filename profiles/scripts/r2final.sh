#!/bin/bash
# final measurements of a build: tests, smoke, headline bench (+ rocprofv3 stats, PMC traffic), other workloads, facade, harness, self-loop
T=$1; OUT=gpurun_out/$T; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash profiles/scripts/r2prof.sh $T
cp $OUT/traffic.json profiles/${2:-r02_x}_traffic.json   # bench.py reads the PMC figure of this build from profiles/
python3 bench.py > $OUT/bench_pipe_256_with_traffic.json 2> $OUT/bench2.err
python3 bench.py --no-cpu-baseline --nx 512 --ny 512 --nz 512 --steps 60 --warmup 20 > $OUT/bench_pipe_512.json 2> $OUT/b512.err
python3 bench.py --no-cpu-baseline --nx 512 --ny 256 --nz 256 --plt-ratio 0.07 --steps 100 --warmup 20 > $OUT/bench_config3_512x256x256_rbc_plt.json 2> $OUT/bc3.err
python3 bench.py --no-cpu-baseline --nx 512 --ny 512 --nz 512 --periodic-box --steps 60 --warmup 20 > $OUT/bench_c5_periodic_512.json 2> $OUT/bc5.err
python3 examples/pipe/run_headline.py /tmp/headline > $OUT/facade_headline.txt 2>&1
python3 examples/run_reference_performance_testing.py /tmp/pt > $OUT/reference_harness.txt 2>&1
python3 examples/rccl_selfloop.py 256 100 > $OUT/rccl_selfloop_256.txt 2>&1
python3 - <<PY
import json
for f in ("bench_pipe_256_with_traffic","bench_pipe_512","bench_config3_512x256x256_rbc_plt","bench_c5_periodic_512"):
    j=json.load(open("$OUT/%s.json"%f)); r=j["roofline"]
    print(f, round(j["ms_per_step"],4), round(j["value"]), "fluid", round(j["mlups_fluid_nodes"]), "frac", round(r["frac"],3), "real", r["frac_real_traffic"], "alone", j["roofline_alone"] and round(j["roofline_alone"]["frac"],3), "whole", round(j["whole_step_hbm_frac"],3), "copy", round(r["copy_GBps_this_gpu"]))
PY
tail -2 $OUT/facade_headline.txt; head -1 $OUT/reference_harness.txt; tail -3 $OUT/rccl_selfloop_256.txt
