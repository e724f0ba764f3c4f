#!/bin/bash
# Everything profiles/r03_* holds, in one GPU session (round 3 collected it in ten short calls, 14 GPU-minutes in all):
#   1. the -m gpu suite                       2. plain bench.py + its rocprofv3 kernel summary
#   3. FETCH_SIZE / WRITE_SIZE passes -> profiles/traffic.json (tied to this build of kernels_part.hip)
#   4. the reference's synthetic benchmark at 1 B and 128 M rows (57 steps)
#   5. per-launch trace of the baseline LDS retry chain     6. the typed LDS group-by microbenchmark
# usage: gpu_round_r03.sh [out-dir]          (copy what is meant to be read from there into profiles/)
out=${1:-gpurun_out/round_r03}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 1200 python -u -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 -rf > $out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench exit $?"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o cfg3f -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_rocprof.json 2> $out/bench_rocprof.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/cfg3f_kernel_stats.csv \; ; rm -rf $out/trace
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$grp -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$grp.log 2>&1
  python tools/rocpd_stats.py $out/pmc_$grp/pmc_results.db > $out/pmc_${grp}_stats.txt 2>&1; rm -rf $out/pmc_$grp
done
python tools/make_traffic_json.py $out/pmc_FETCH_SIZE_stats.txt $out/pmc_WRITE_SIZE_stats.txt 1e10 r03
timeout 600 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1; echo "refbench 1B exit $?"
timeout 300 python tools/refbench.py --rows 128e6 --steps 3 --budget-ms 1500 --out $out/refbench_128m.jsonl > $out/refbench_128m.log 2>&1; echo "refbench 128M exit $?"
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $out/trace2 -o ref -- python tools/refbench.py --rows 128e6 --steps 3 --budget-ms 4000 \
  --only PHS003,PHS004,BH001,BH002,BH003,BH004,BH007,MSBS001,MSBS002,MSPHS002 > $out/refbench_lds_shapes.jsonl 2> $out/refbench_lds_shapes.err
find $out/trace2 -name "*kernel_trace.csv" -exec cp {} $out/lds_retry_chain_kernel_trace.csv \; ; rm -rf $out/trace2
[ -x tools/microbench/groupby_lds_typed ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/groupby_lds_typed tools/microbench/groupby_lds_typed.hip
timeout 60 ./tools/microbench/groupby_lds_typed 1e9 > $out/microbench_groupby_lds_typed.txt 2>&1; cat $out/microbench_groupby_lds_typed.txt
timeout 60 ./integration/glue_check > $out/glue_check.log 2>&1; echo "glue_check exit $?"
python - <<'PY' > $out/explain_routes.txt 2>&1
import sys
sys.path.insert(0, "tools")
import refbench
from heavydb_amd import capi
from heavydb_amd.executor import Executor
capi.load_library()
names, descs, _ = refbench.schema()
ex = Executor(0)
for q in refbench.queries():
    ra, _ = refbench.build_unit(refbench.queries()[q], names, descs, 10**9)
    print(q, "->", ex.explain(ra, [32_000_000] * 31 + [8_000_000]))
PY
head -8 $out/explain_routes.txt
