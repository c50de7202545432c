#!/bin/bash
# Same-box A/B of library builds: scratch/libs/<name>/libshennong_hip.so, each run through the A/B harness
# (tools/ab_fbank512.cpp, which takes the library from LD_LIBRARY_PATH before its RUNPATH), alternated ROUNDS
# times.   tools/ab_libs.sh <kind> <reps> <rounds> name1 name2 ...
kind=$1; reps=$2; rounds=$3; shift 3
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for name in "$@"; do
    LD_LIBRARY_PATH=$PWD/scratch/libs/$name scratch/ab512 10000 $kind $reps -- $name= 2>&1 | grep "kernel ms" | cut -c1-75
  done
done
