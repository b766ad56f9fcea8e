# GPU session r03bj: the larger manual checks on the final build: versions of real text (215 MiB), word-level text, deep repeats, smoke()
O=gpurun_out/r03bj; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.txt)"
timeout -k 10 400 python tools/check_versions_text.py > $O/versions_text.txt 2>&1; echo "versions rc=$?"; tail -3 $O/versions_text.txt
timeout -k 10 400 python tools/check_wordtext.py > $O/wordtext.txt 2>&1; echo "wordtext rc=$?"; tail -3 $O/wordtext.txt
timeout -k 10 400 python tools/check_deep_repeats.py > $O/deep_repeats.txt 2>&1; echo "deep rc=$?"; tail -3 $O/deep_repeats.txt
