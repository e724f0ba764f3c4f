#!/bin/bash
# Round 3, call 10 (the last seconds of GPU time): mi355q_explain on the real device (it had only met the host simulation) and the
# native HeavyDB-binding check with its new explain call.
out=gpurun_out/r03j
mkdir -p $out
timeout 12 ./integration/glue_check > $out/glue_check.log 2>&1; echo "glue_check exit $?"; grep -E "route|agree|FAIL|fail" $out/glue_check.log | head -8
timeout 12 python - <<'PY' 2>&1 | tail -12
import sys
sys.path.insert(0, "tools")
import refbench
from heavydb_amd import capi
from heavydb_amd.executor import Executor
capi.load_library()
names, descs, _ = refbench.schema()
ex = Executor(0)
for q in ["NGA03", "PHS004", "BH005", "BH007", "MSBS003", "MSPHS009", "S002"]:
    ra, _ = refbench.build_unit(refbench.queries()[q], names, descs, 10**9)
    print(q, "->", ex.explain(ra, [32_000_000] * 31 + [8_000_000]))
PY
