mkdir -p gpurun_out/r2y; : > gpurun_out/r2y/bench_ab.txt
cp hemocell_amd/lib/libhemocell_amd.so /tmp/lib_after.so
for round in 1 2 3; do
for which in before after; do
  if [ $which = before ]; then cp scratch/ab/lib_before.so hemocell_amd/lib/libhemocell_amd.so; else cp /tmp/lib_after.so hemocell_amd/lib/libhemocell_amd.so; fi
  python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 1; }
  python - $which >> gpurun_out/r2y/bench_ab.txt <<'PY'
import json,sys
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],4), {k:round(v["ms_total"]/max(v["launches"],1),4) for k,v in d["kernel_ms"].items()})
PY
done
done
cp /tmp/lib_after.so hemocell_amd/lib/libhemocell_amd.so
cat gpurun_out/r2y/bench_ab.txt
