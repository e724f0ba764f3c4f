#!/bin/bash
# First GPU call of a round (about 3-4 minutes of box time): the native columnar check, the whole
# -m gpu suite, smoke(), and one default bench line.  Everything lands under gpurun_out/first/.
out=gpurun_out/first
mkdir -p $out
export TMPDIR=/tmp
if [ -x tools/native/col_check ]; then
  timeout 60 ./tools/native/col_check > $out/col_check.txt 2>&1; echo "col_check exit $?"; tail -4 $out/col_check.txt
fi
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench exit $?"; cut -c1-1500 $out/bench_default.json
