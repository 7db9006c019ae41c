#!/bin/bash
# Round profile of the bench command on the GPU box: kernel-trace stats, then the two PMC passes for HBM traffic
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains are combined with --pmc).
# usage: [GIT_SHA=<head>] tools/profile_step.sh <tag>     -> gpurun_out/<tag>_kernel_stats.txt, <tag>_gaps.txt, <tag>_pmc_traffic.json
# (both carry tools/source_hash.py's hash of the tree they were measured on; bench.py only quotes a matching counter file)
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
cmd="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $cmd > $out/kt.log 2>&1
sha=$(python tools/source_hash.py)
python tools/rocprof_summary.py stats $out/kt "rocprofv3 --kernel-trace --stats -- $cmd   ($tag, source_sha256 $sha, git ${GIT_SHA:-unknown})" > gpurun_out/${tag}_kernel_stats.txt
python tools/rocprof_summary.py gaps $out/kt > gpurun_out/${tag}_gaps.txt 2>&1
# the same step on ONE stream (every kernel alone on the chip): what bench.py's in-situ HIP-event profile of its roofline line measures
ALDI_WGRAD_STREAM=0 ALDI_TEACHER_STREAM=0 ALDI_AUX_STREAM=0 ALDI_SGD_STREAM=0 rocprofv3 --kernel-trace --stats -d $out/kt1 -o kt -- $cmd > $out/kt1.log 2>&1
python tools/rocprof_summary.py stats $out/kt1 "ALDI_WGRAD_STREAM=0 ALDI_TEACHER_STREAM=0 ALDI_AUX_STREAM=0 ALDI_SGD_STREAM=0 rocprofv3 --kernel-trace --stats -- $cmd   ($tag, single stream, source_sha256 $sha, git ${GIT_SHA:-unknown})" > gpurun_out/${tag}_kernel_stats_single_stream.txt
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- $cmd > $out/write.log 2>&1
python tools/rocprof_summary.py pmc $out/fetch $out/write $sha "${GIT_SHA:-unknown}" > gpurun_out/${tag}_pmc_traffic.json
head -12 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150; cat gpurun_out/${tag}_pmc_traffic.json
# keep the merge-back small
find $out -name "*.db" -size +30M -delete; find $out -name "*.csv" -size +30M -delete
