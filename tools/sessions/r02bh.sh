# GPU session r02bh: walk micro-benchmark -- what each ingredient of the step costs on this box
O=gpurun_out/r02bh; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/walk_steps.hip -o $O/walk_steps 2> $O/build.log || { tail $O/build.log; exit 1; }
timeout -k 10 300 $O/walk_steps > $O/walk_steps.txt 2>&1; echo "rc=$?"; cat $O/walk_steps.txt; rm -f $O/walk_steps
exit 0
