# GPU session r02z: hardware probe -- 64-bit shift with the amount in the wave's last allocated VGPR
O=gpurun_out/r02z; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/shift64_probe tools/probes/shift64_probe.hip 2> $O/build.log || { tail $O/build.log; exit 1; }
timeout -k 10 120 $O/shift64_probe > $O/probe.txt 2>&1; echo "rc=$?"; cat $O/probe.txt
rm -f $O/shift64_probe
exit 0
