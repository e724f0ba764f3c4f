#!/bin/bash
# round 4, call 11: what bounds the keyed probe now (four lanes per sequence): pacing off, SQ + L2 counters on one launch
out=${1:-gpurun_out/r04_call11}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
for args in "--sparse --sum-dim" "--sparse --sum-dim --opt-flags 4" "--sparse --sum-dim --probe-passes 2" "--sparse --opt-flags 4"; do
  tag=$(echo "$args" | tr -d ' -' )
  timeout 300 python bench.py --config cfg4 $args --steps 2 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
  echo "cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['verify'])" 2>&1)"
done
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/$out/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --sparse --sum-dim --rows 3.2e9 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/$out/pmc_$i/pmc_results.db > $GRAFT_REPO_ROOT/$out/pmc_${i}_stats.txt 2>&1; rm -rf $GRAFT_REPO_ROOT/$out/pmc_$i
  grep -E "k_part_probe" $GRAFT_REPO_ROOT/$out/pmc_${i}_stats.txt | cut -c1-170
done
