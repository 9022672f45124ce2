set -u
root=$PWD; out=$root/gpurun_out/r2xmf; rm -rf $out; mkdir -p $out
g++ -std=c++14 -O2 -Wno-deprecated-declarations -DHEMOCELL_WITH_HDF5 -I/opt/conda/include -Iinclude -Ihemocell_amd/compat examples/pipe/pipe_synthetic.cpp -o /tmp/pipe_drv -Lhemocell_amd/lib -lhemocell_amd -Wl,-rpath,$root/hemocell_amd/lib build/ref_drivers/hdf5lib/libhdf5_hl.so.100 build/ref_drivers/hdf5lib/libhdf5.so.103 -Wl,-rpath,$root/build/ref_drivers/hdf5lib || exit 1
for world in 1 2; do
  w=/tmp/xmf_$world; rm -rf $w; mkdir -p $w; cp examples/pipe/{config.xml,RBC.xml,PLT.xml,RBC.pos,PLT.pos} $w/
  if [ $world = 1 ]; then (cd $w && /tmp/pipe_drv config.xml > $out/run_$world.txt 2>&1) || exit 1
  else
    (cd $w && OMPI_COMM_WORLD_RANK=0 OMPI_COMM_WORLD_SIZE=2 OMPI_COMM_WORLD_LOCAL_RANK=0 HEMOCELL_PORT=35111 HEMOCELL_TRANSPORT=tcp /tmp/pipe_drv config.xml > $out/run_2_r0.txt 2>&1) &
    p0=$!
    (cd $w && OMPI_COMM_WORLD_RANK=1 OMPI_COMM_WORLD_SIZE=2 OMPI_COMM_WORLD_LOCAL_RANK=1 HEMOCELL_PORT=35111 HEMOCELL_TRANSPORT=tcp /tmp/pipe_drv config.xml > $out/run_2_r1.txt 2>&1) &
    p1=$!
    wait $p0 || exit 1; wait $p1 || exit 1
  fi
  mkdir -p $out/ranks_$world/hdf5; cp -r $w/tmp_pipe/hdf5/000000000400 $out/ranks_$world/hdf5/; cp -r $w/tmp_pipe/csv $out/ranks_$world/
done
du -sh $out/*
