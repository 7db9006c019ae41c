#!/bin/bash
# kernel trace of the bench command -> per-queue launch list of one step (gpurun_out/<tag>_streams.txt)
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_${tag}_s
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -o kt -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $out/kt.log 2>&1
python tools/rocprof_summary.py streams $out/kt 0 > gpurun_out/${tag}_streams.txt
find $out -name "*.db" -size +30M -delete
