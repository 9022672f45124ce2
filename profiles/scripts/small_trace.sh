out=$PWD/gpurun_out/r2z; mkdir -p $out; root=$PWD
w=/tmp/shear; rm -rf $w; cp -r tests/golden/shear_case $w; chmod -R u+w $w
sed -i "s#<tmax>[^<]*</tmax>#<tmax> 3000 </tmax>#; s#<tmeas>[^<]*</tmeas>#<tmeas> 100000 </tmeas>#" $w/config.xml
cd /tmp && export TMPDIR=/tmp
cd $w && rocprofv3 --kernel-trace --stats -d $out/small -o s -- $root/build/ref_drivers/oneCellShear config.xml > $out/small.txt 2>&1 || { tail -5 $out/small.txt; exit 1; }
cd $root; python3 examples/trace_timeline.py $(find $out/small -name "*.db" | head -1) 6000 24
t0=$(date +%s%N); (cd $w && rm -rf tmp && $root/build/ref_drivers/oneCellShear config.xml > /dev/null 2>&1); t1=$(date +%s%N); echo "3000 iterations wall: $(( (t1-t0)/1000000 )) ms (with start-up)"
