#!/usr/bin/env python
"""Writes tests/golden/ref_data_types_basic5_numeric.json: the numeric columns of the reference's own fixture table
`data_types_basic5` (Tests/Import/datafiles/data_types_basic5.csv.gz, DDL at Tests/ExecuteTest.cpp:10472-10500), which
its Select.CountIf / Select.SumIf tests (ExecuteTest.cpp:4020-4198) run on.  Empty CSV fields are SQL NULLs.
Run in the build container only (reads /root/reference); the JSON is what travels."""
import csv
import gzip
import json
import os

SRC = "/root/reference/Tests/Import/datafiles/data_types_basic5.csv.gz"
COLS = {"Tiny_int": "int8_t", "Small_int": "int16_t", "Int_": "int32_t", "Big_int": "int64_t", "Float_": "float", "Double_": "double"}


def main():
    csv.field_size_limit(1 << 30)  # the geometry columns are long
    with gzip.open(SRC, "rt") as f:
        rows = list(csv.DictReader(f))
    out = {"source": "Tests/Import/datafiles/data_types_basic5.csv.gz of the reference (numeric columns only)", "types": COLS, "columns": {}}
    for c, t in COLS.items():
        vals = []
        for r in rows:
            s = r[c].strip()
            vals.append(None if s == "" else (float(s) if t in ("float", "double") else int(s)))
        out["columns"][c] = vals
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_data_types_basic5_numeric.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, len(rows), "rows")


if __name__ == "__main__":
    main()
