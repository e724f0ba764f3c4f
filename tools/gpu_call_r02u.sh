#!/bin/bash
out=gpurun_out/r02u
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/headline_experiments.py --steps 3 --settings "0:0,1:0,2:0,3:0,4:0,8:0,16:0,0:0" --trace "2:0" > $out/experiments.jsonl 2> $out/experiments.err
echo "experiments exit $?"; cat $out/experiments.jsonl; grep "phase 2" $out/experiments.err | tail -4
echo finished
