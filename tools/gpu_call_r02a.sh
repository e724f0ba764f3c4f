#!/bin/bash
# Round 2, first GPU call: host facts, the -m gpu suite (incl. the BASELINE-size oracle parity tests),
# smoke, the headline experiments (pair pacing window x chunk size), one default bench line, the SQ
# instruction-mix counters of phase 2.  Everything lands under gpurun_out/r02a/.
out=gpurun_out/r02a
mkdir -p $out
export TMPDIR=/tmp
(nproc; free -g; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > $out/host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --ignore=tests/test_zz_gpu_baseline_sizes.py > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 $out/pytest_gpu.log
timeout 1200 python -m pytest tests/test_zz_gpu_baseline_sizes.py -m gpu -q -s -p no:cacheprovider --durations=12 > $out/pytest_sizes.log 2>&1
echo "pytest sizes exit $?"; tail -40 $out/pytest_sizes.log | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 900 python tools/headline_experiments.py > $out/experiments.jsonl 2> $out/experiments.err
echo "experiments exit $?"; cat $out/experiments.jsonl; grep -a "Mcycles\|gave up" $out/experiments.err | tail -8
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench exit $?"; cut -c1-3000 $out/bench_default.json
bash tools/gpu_pmc2.sh r02a > $out/pmc_sq.txt 2>&1; cat $out/pmc_sq.txt | cut -c1-200
