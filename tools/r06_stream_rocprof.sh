#!/bin/bash
# kernel trace of the streamed pipeline (25 h, pinned, njobs 1): which kernels the 0.3 s of GPU time per 125 h are
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_stream_prof -o stream -- env ONLY=pinned python tools/profile_streamed.py 25 none > gpurun_out/r06_stream_prof.log 2>&1
find gpurun_out/r06_stream_prof -name "*kernel_stats*" | head
f=$(find gpurun_out/r06_stream_prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
# keep only the summaries (the traces are tens of MB)
find gpurun_out/r06_stream_prof -name "*kernel_trace*" -delete
