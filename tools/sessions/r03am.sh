# GPU session r03am: one LSD pass with 8-, 10- and 11-bit digits in the product's kernel structure (tools/micro/radix_digits.hip)
O=gpurun_out/r03am; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/radix_digits tools/micro/radix_digits.hip 2> $O/build.log || { tail $O/build.log; exit 1; }
timeout -k 10 300 $O/radix_digits 28 > $O/radix_digits_2p28.txt 2>&1; echo "rc=$?"; cat $O/radix_digits_2p28.txt
rm -f $O/radix_digits
