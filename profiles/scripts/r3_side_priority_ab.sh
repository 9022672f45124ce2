#!/bin/bash
mkdir -p gpurun_out/r3j
for r in 1 2 3; do
  for mode in high low; do
    HEMOCELL_SIDE_PRIORITY=$mode python bench.py --no-cpu-baseline --no-target-512 --copy-reps 4 > gpurun_out/r3j/ab_${mode}_$r.json 2>> gpurun_out/r3j/ab.err || exit 1
    python - <<PY
import json
j=json.load(open("gpurun_out/r3j/ab_${mode}_$r.json"))
k=j["kernel_ms"]
print("$mode $r: %.4f ms/step  collide avg %.4f (alone %.4f, beside %.4f)  spread %.4f" % (j["ms_per_step"], j["roofline"]["avg_launch_ms"],
  k["collide_stream_alone"]["ms_total"]/max(k["collide_stream_alone"]["launches"],1),
  k["collide_stream_beside"]["ms_total"]/max(k["collide_stream_beside"]["launches"],1),
  k["ibm_spread"]["ms_total"]/max(k["ibm_spread"]["launches"],1)))
PY
  done
done
