#!/bin/bash
# A/B in one job on one box: default (spread beside the collide), spread after the collide, no overlap -- alternately, three rounds
mkdir -p gpurun_out/r3e
for r in 1 2 3; do
  for mode in default after none; do
    case $mode in default) flag="";; after) flag="--spread-after-collide";; none) flag="--no-overlap";; esac
    python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 $flag > gpurun_out/r3e/ab_${mode}_$r.json 2>> gpurun_out/r3e/ab.err || exit 1
    python - <<PY
import json
j=json.load(open("gpurun_out/r3e/ab_${mode}_$r.json"))
k=j["kernel_ms"]
print("$mode $r: %.4f ms/step  collide avg %.4f (alone %.4f x%d, beside %.4f x%d)  spread %.4f  frac %.3f" % (j["ms_per_step"], j["roofline"]["avg_launch_ms"],
  k["collide_stream_alone"]["ms_total"]/max(k["collide_stream_alone"]["launches"],1), k["collide_stream_alone"]["launches"],
  k["collide_stream_beside"]["ms_total"]/max(k["collide_stream_beside"]["launches"],1), k["collide_stream_beside"]["launches"],
  k["ibm_spread"]["ms_total"]/max(k["ibm_spread"]["launches"],1), j["roofline"]["frac"]))
PY
  done
done
