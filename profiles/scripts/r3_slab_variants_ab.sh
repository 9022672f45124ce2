#!/bin/bash
mkdir -p gpurun_out/r3p
for r in 1 2 3; do
  for nx in 64 128; do
    for v in 0 1 2 3; do
      HEMOCELL_SLAB_VARIANT=$v python examples/rccl_selfloop.py 256 100 rccl $nx > gpurun_out/r3p/s_${nx}_${v}_$r.txt 2>&1
      echo "nx $nx variant $v round $r: $(grep 'hc_iterate' gpurun_out/r3p/s_${nx}_${v}_$r.txt | awk '{print $5}') -> $(grep 'slab schedule' gpurun_out/r3p/s_${nx}_${v}_$r.txt | awk '{print $5}')"
    done
  done
done
