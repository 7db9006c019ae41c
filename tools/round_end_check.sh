# Round-end check on the GPU box (run through gpurun from the repo root): full -m gpu suite, the three bench workloads, smoke(), rocprofv3 kernel stats.
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 500 python bench.py > gpurun_out/bench_r50.log 2>&1; tail -1 gpurun_out/bench_r50.log | cut -c1-250
for wl in vitdet_b convnext_l; do timeout 400 python bench.py --workload $wl > gpurun_out/bench_$wl.log 2>&1; tail -1 gpurun_out/bench_$wl.log | cut -c1-200; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
for wl in vitdet_b convnext_l; do rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o v -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 3 > /tmp/prof.log 2>&1; ps=2; [ $wl = convnext_l ] && ps=1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats /tmp/prof "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 3 --warmup 3   (r01, ALDI step, N=1)" adamw_kernel $ps > $GRAFT_REPO_ROOT/gpurun_out/r01_${wl}_kernel_stats.txt 2>&1; done
rm -rf /tmp/prof; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats /tmp/prof "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile   (r01_v13)" > $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats_v13.txt 2>&1
echo done
