#!/bin/bash
# Round 3, call 5 (second session): the whole -m gpu suite on the tree as it stands, the default bench line (+ the
# rocprofv3 kernel summary of the same command), a kernel trace of the reference-benchmark queries that are still
# far below the roofline, smoke().
out=gpurun_out/r03e
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 520 python -u -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 --durations=12 -rf > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -25 $out/pytest_gpu.log | cut -c1-220
timeout 200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench exit $?"; cut -c1-600 $out/bench_default.json
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg3f -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_rocprof.json 2> $out/bench_rocprof.err; echo "rocprof bench exit $?"
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg3f_kernel_stats.csv \;
rm -rf $out/trace
head -8 $out/cfg3f_kernel_stats.csv | cut -c1-180
timeout 150 rocprofv3 --kernel-trace --stats -f csv -d $out/trace2 -o ref -- python tools/refbench.py --rows 128e6 --steps 2 --budget-ms 4000 --only PHS004,PHS005,BH002,BH003,BH004,BH007,S001,MSBS001 > $out/refbench_cliffs.jsonl 2> $out/refbench_cliffs.err; echo "rocprof refbench exit $?"
cut -c1-260 $out/refbench_cliffs.jsonl
find $out/trace2 -name "*kernel_stats.csv" -exec cp {} $out/refbench_cliffs_kernel_stats.csv \;
rm -rf $out/trace2
head -30 $out/refbench_cliffs_kernel_stats.csv | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
