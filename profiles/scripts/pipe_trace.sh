out=$PWD/gpurun_out/r2z; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/trace -o t -- python3 $root/scratch/pipe_trace.py > $out/trace.txt 2>&1 || { tail -5 $out/trace.txt; exit 1; }
cd $root; python3 examples/trace_timeline.py $(find $out/trace -name "*.db" | head -1) 0 400 > $out/timeline.txt
grep -n "collide" $out/timeline.txt | tail -40 | head -5
