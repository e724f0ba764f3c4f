#!/bin/bash
# Soak run of the GPU fuzz tests with fresh seeds: gpu_soak.sh <first_seed> <n_seeds> [iterations per seed]
first=${1:-5000}; n=${2:-6}; iters=${3:-300}
out=gpurun_out/soak
mkdir -p $out
export PYTHONUNBUFFERED=1
for s in $(seq $first $((first + n - 1))); do
  MI355Q_FUZZ_SEED=$s MI355Q_FUZZ_ITERS=$iters timeout 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x \
      -k "fuzzed_row_plans or joins_match_oracle_on_random_plans" > $out/seed_$s.log 2>&1
  echo "seed $s exit $? $(tail -1 $out/seed_$s.log | cut -c1-120)"
done
