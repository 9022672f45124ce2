set -u
out=$PWD/gpurun_out/r2x; mkdir -p $out
root=$PWD
: > $out/unbounded.summary
for tmax in 200 1200; do
  w=/tmp/unb_$tmax; rm -rf $w; cp -r scratch/unbounded_case $w
  sed -i "s#<tmax>[^<]*</tmax>#<tmax> $tmax </tmax>#; s#<tmeas>[^<]*</tmeas>#<tmeas> 1000000 </tmeas>#" $w/config.xml
  t0=$(date +%s%N)
  (cd $w && timeout -k 10 900 $root/build/ref_drivers/unbounded_nohdf5 config.xml > $out/unbounded_$tmax.stdout 2>&1) || exit 1
  t1=$(date +%s%N)
  echo "tmax $tmax: $(( (t1 - t0) / 1000000 )) ms wall" | tee -a $out/unbounded.summary
done
cd /tmp && export TMPDIR=/tmp
w=/tmp/unb_prof; rm -rf $w; cp -r $root/scratch/unbounded_case $w
sed -i "s#<tmax>[^<]*</tmax>#<tmax> 200 </tmax>#; s#<tmeas>[^<]*</tmeas>#<tmeas> 1000000 </tmeas>#" $w/config.xml
cd $w && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o unb -- $root/build/ref_drivers/unbounded_nohdf5 config.xml > $out/unbounded_prof.stdout 2>&1
ls -R $out/prof | head
