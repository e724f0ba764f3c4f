#!/bin/bash
# Round 3, call 7 (the last seconds): per-launch kernel trace of the baseline LDS retry chain (BH003: 2 attempts, BH004: 3,
# MSBS001: FLOAT key, 2) at 128 M rows — where does the ~11 ms per lost attempt go?
out=gpurun_out/r03g
mkdir -p $out
export TMPDIR=/tmp
timeout 75 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o ref -- python tools/refbench.py --rows 128e6 --steps 2 --budget-ms 4000 --only BH003,BH004,MSBS001 > $out/refbench.jsonl 2> $out/refbench.err
echo "exit $?"; cat $out/refbench.jsonl | cut -c1-200
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/trace -name "*kernel_trace.csv" -exec cp {} $out/kernel_trace.csv \;
rm -rf $out/trace
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r03g/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    if "k_generate" in n: continue
    short = n.split("(")[0].replace("void mq::(anonymous namespace)::", "").replace("mq::(anonymous namespace)::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e3:9.1f} us  {short}  grid={r.get('Grid_Size_X', r.get('Grid_Size'))} lds={r.get('LDS_Block_Size', '')}")
PY
