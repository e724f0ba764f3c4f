#!/bin/bash
out=gpurun_out/r02s
mkdir -p $out
timeout 300 ./tools/microbench/groupby_small > $out/groupby_small.txt 2>&1; cat $out/groupby_small.txt
echo finished
