#!/bin/bash
# SQ instruction-mix counters for one bench command: gpu_pmc2.sh <tag> <bench args...>
tag=$1; shift
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc/${tag}_$i -o pmc -- python bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/${tag}_$i.log 2>&1
  python tools/rocpd_stats.py gpurun_out/pmc/${tag}_$i/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part" 
done
