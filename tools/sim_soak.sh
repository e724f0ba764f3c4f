#!/bin/bash
# Soak run of the GPU fuzz tests WITHOUT a GPU: fresh seeds through the real kernels on the host simulation (tests/hostsim,
# MI355Q_HOSTSIM=real).  sim_soak.sh <first_seed> <n_seeds> [iterations per seed]
# (round 3: seeds 7101-7103 x 150 and 7200-7203, 7210-7213 x 300 iterations of the fuzzed row plans and the random joins: all pass)
first=${1:-7300}; n=${2:-4}; iters=${3:-300}
out=/tmp/mi355q_sim_soak
mkdir -p $out
cd "$(dirname "$0")/.."
for s in $(seq $first $((first + n - 1))); do
  MI355Q_HOSTSIM=real HOSTSIM_WATCHDOG=300 MI355Q_FUZZ_SEED=$s MI355Q_FUZZ_ITERS=$iters timeout 2400 python -u -m pytest tests/test_gpu_parity.py \
      -m gpu -q -p no:cacheprovider -x -k "fuzzed_row_plans or joins_match_oracle_on_random_plans" > $out/seed_$s.log 2>&1
  echo "seed $s exit $? $(tail -1 $out/seed_$s.log | cut -c1-120)"
done
