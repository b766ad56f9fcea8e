# Top-level convenience targets (the reference's Makefile had `all`, `test`, `test-enwik8`: Makefile:5-7,30-38).
PY ?= python

all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

# golden / oracle / C-ABI / CLI-contract checks that need no GPU
test: all
	$(PY) -m pytest tests -q -m "not gpu"

# HIP path vs oracle, through the C-ABI (needs an MI355X)
test-gpu: all
	$(PY) -m pytest tests -q -m gpu

bench: all
	$(PY) bench.py

clean:
	$(MAKE) -C bijective-bwt_amd clean
	$(MAKE) -C oracle clean

.PHONY: all test test-gpu bench clean
