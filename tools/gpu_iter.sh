#!/bin/bash
# one GPU iteration: family parity tests, then short cfg3f benches (1 B rows)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "${1:-baseline_family or large_property}" > gpurun_out/pytest_iter.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/pytest_iter.log
for rows in ${2:-1e9}; do
  timeout 300 python bench.py --config cfg3f --rows $rows --steps 3 --warmup 1 --no-cpu-baseline --verify > gpurun_out/bench_iter_$rows.log 2>&1
  echo "== bench $rows exit $?"; tail -c 1800 gpurun_out/bench_iter_$rows.log
done
