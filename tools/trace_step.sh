#!/bin/bash
# kernel-trace only (no counters): per-kernel table of one step, idle gaps, multi-stream overlap.  usage: tools/trace_step.sh <tag>
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
cmd="python bench.py --steps 3 --warmup 6 --no-cpu-baseline --no-profile"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $cmd > $out/kt.log 2>&1
sha=$(python tools/source_hash.py)
python tools/rocprof_summary.py stats $out/kt "rocprofv3 --kernel-trace --stats -- $cmd   ($tag, source_sha256 $sha, git ${GIT_SHA:-unknown})" > gpurun_out/${tag}_kernel_stats.txt
python tools/rocprof_summary.py gaps $out/kt > gpurun_out/${tag}_gaps.txt 2>&1
python tools/rocprof_summary.py overlap $out/kt > gpurun_out/${tag}_overlap.txt 2>&1
python tools/rocprof_summary.py phases $out/kt > gpurun_out/${tag}_phases.txt 2>&1
head -40 gpurun_out/${tag}_kernel_stats.txt | cut -c1-160; cat gpurun_out/${tag}_overlap.txt; head -12 gpurun_out/${tag}_gaps.txt; cat gpurun_out/${tag}_phases.txt | head -30
find $out -name "*.db" -size +30M -delete; find $out -name "*.csv" -size +30M -delete
