#!/bin/bash
# round 4, call 6: what bounds the keyed payload probe (SQ counters; two workgroups per CU), the gpu tests this round touched
out=${1:-gpurun_out/r04_call6}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_refbench.py tests/test_zz_gpu_columnar.py -m gpu -q -p no:cacheprovider -x --durations=3 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -6 $out/pytest.log
for args in "--sparse" "--sparse --blocks-per-cu 2"; do
  tag=$(echo "$args" | tr -d ' -' )
  timeout 300 python bench.py --config cfg4 $args --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
  echo "cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['roofline'].get('whole_step_frac'))" 2>&1)"
done
timeout 120 python bench.py --config cfg1 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_cfg1.json 2> $out/bench_cfg1.err
python -c "import json; d=json.load(open('$out/bench_cfg1.json')); print('cfg1', d['ms_per_step'], d['roofline'].get('whole_step_frac'))"
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/$out/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --sparse --rows 3.2e9 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/$out/pmc_$i/pmc_results.db > $GRAFT_REPO_ROOT/$out/pmc_${i}_stats.txt 2>&1; rm -rf $GRAFT_REPO_ROOT/$out/pmc_$i
  grep -E "k_part_probe|k_part_scatter" $GRAFT_REPO_ROOT/$out/pmc_${i}_stats.txt | cut -c1-160
done
