#!/bin/bash
# record of a dropped experiment: the HEMOCELL_PIPELINE_CHUNKS switch this script drives was removed again (profiles/r03_f_pipelined_velocity_update_ab.txt)
mkdir -p gpurun_out/r3k
for r in 1 2 3; do
  for mode in 1 4 8; do
    HEMOCELL_PIPELINE_CHUNKS=$mode python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 > gpurun_out/r3k/ab_${mode}_$r.json 2>> gpurun_out/r3k/ab.err || exit 1
    python - <<PY
import json
j=json.load(open("gpurun_out/r3k/ab_${mode}_$r.json"))
k=j["kernel_ms"]
print("chunks $mode round $r: %.4f ms/step  collide avg %.4f  interp %.4f x%d  spread total %.2f ms" % (j["ms_per_step"], j["roofline"]["avg_launch_ms"],
  k["ibm_interpolate"]["ms_total"]/max(k["ibm_interpolate"]["launches"],1), k["ibm_interpolate"]["launches"], k["ibm_spread"]["ms_total"]))
PY
  done
done
HEMOCELL_PIPELINE_CHUNKS=1 python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 --nx 512 --ny 512 --nz 512 --steps 25 --warmup 5 > gpurun_out/r3k/big_1.json 2>> gpurun_out/r3k/ab.err
HEMOCELL_PIPELINE_CHUNKS=4 python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 --nx 512 --ny 512 --nz 512 --steps 25 --warmup 5 > gpurun_out/r3k/big_4.json 2>> gpurun_out/r3k/ab.err
HEMOCELL_PIPELINE_CHUNKS=8 python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 --nx 512 --ny 512 --nz 512 --steps 25 --warmup 5 > gpurun_out/r3k/big_8.json 2>> gpurun_out/r3k/ab.err
python - <<PY
import json
for m in (1,4,8):
    j=json.load(open("gpurun_out/r3k/big_%d.json"%m)); print("512^3 chunks", m, "%.4f ms/step"%j["ms_per_step"])
PY
