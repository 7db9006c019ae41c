#!/bin/bash
# kernel-trace stats of a short bench run only (no PMC passes): tools/quick_stats.sh <tag> [pattern] [launch-regex] [phases]  -> gpurun_out/<tag>_quick_stats.txt
tag=${1:-q}
pat=${2:-.}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
cmd="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $cmd > $out/kt.log 2>&1
python tools/rocprof_summary.py stats $out/kt "quick $tag" > gpurun_out/${tag}_quick_stats.txt
grep -E "$pat" gpurun_out/${tag}_quick_stats.txt | head -40 | cut -c1-170
if [ -n "$3" ]; then python tools/rocprof_summary.py launches $out/kt "$3" | cut -c1-150; fi
if [ -n "$4" ]; then python tools/rocprof_summary.py phases $out/kt; fi
rm -rf $out
