# GPU session r02l: text at 2^32 (dense rounds at the main path's largest n), real text untimed
set -o pipefail
O=gpurun_out/r02l; mkdir -p $O
timeout -k 10 300 python tools/check_text_2p32.py 32 > $O/text_2p32.log 2>&1; echo "text32 rc=$?"; cat $O/text_2p32.log
timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.log 2>&1; echo "text31 rc=$?"; cat $O/text_2p31.log
for t in check_realtext check_versions_text; do sed 's/ctx.set_timing(2)/ctx.set_timing(0)/' tools/$t.py > tools/_untimed_$t.py; timeout -k 10 300 python tools/_untimed_$t.py 2>&1 | grep -E "real text|versions|roundtrip" | cut -c1-160; rm -f tools/_untimed_$t.py; done
