#!/bin/bash
# Round 3, call 8 (the last ~70 s of GPU time): the LDS group-by after the shared "replica full" word and the 32-bit key
# mix — its gpu tests (case matrix planned, the reference benchmark small + windowed, random plans), then the per-launch
# trace of the retry chain again and the timing of the shapes it serves at 128 M rows.
out=gpurun_out/r03i
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
t0=$SECONDS
timeout 22 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 40 -x \
  -k "(windowed or small or (test_hip_matches_oracle and planned) or random_plans or retry_ladder) and not 16m" > $out/pytest.log 2>&1
echo "pytest exit $? at $((SECONDS - t0)) s"; tail -4 $out/pytest.log | cut -c1-200
timeout $(( 32 - (SECONDS - t0) )) rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o ref -- python tools/refbench.py --rows 128e6 --steps 3 --budget-ms 4000 --only PHS003,PHS004,BH001,BH002,BH003,BH004,BH007,MSBS001,MSBS002,MSPHS002 > $out/refbench.jsonl 2> $out/refbench.err
echo "refbench exit $? at $((SECONDS - t0)) s"; cut -c1-220 $out/refbench.jsonl
find $out/trace -name "*kernel_trace.csv" -exec cp {} $out/kernel_trace.csv \;
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r03i/kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_groupby_lds" in n:
        d[n[n.index("k_groupby_lds"):][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, "launches", len(v), "us:", " ".join(f"{x:.0f}" for x in v[:60]))
PY
echo "done at $((SECONDS - t0)) s"
