mkdir -p gpurun_out/r2y; : > gpurun_out/r2y/overlap_ab.txt
for round in 1 2 3; do
for which in overlap no-overlap; do
  if [ $which = overlap ]; then flag=""; else flag="--no-overlap"; fi
  python bench.py --no-cpu-baseline $flag > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 1; }
  python - $which >> gpurun_out/r2y/overlap_ab.txt <<'PY'
import json,sys
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],4), {k:round(v["ms_total"]/max(v["launches"],1),4) for k,v in d["kernel_ms"].items()})
PY
done
done
cat gpurun_out/r2y/overlap_ab.txt
