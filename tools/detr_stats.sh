#!/bin/bash
# kernel-trace stats of the Deformable-DETR workload (whole run: per-kernel totals over warm-up + timed steps)  -> gpurun_out/<tag>_detr_kernel_stats.txt
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_${tag}_detr
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --workload detr --steps 3 --warmup 1 > $out/kt.log 2>&1
python - <<PY > gpurun_out/${tag}_detr_kernel_stats.txt
import sqlite3, glob, collections
db = glob.glob("$out/kt/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("# rocprofv3 --kernel-trace --stats -- python bench.py --workload detr --steps 3 --warmup 1   ($tag; all %d launches of the run: 2 set-up + 1 warm-up + 3 timed steps; %.1f ms of kernels)" % (len(rows), tot / 1e3))
print("%-110s %7s %11s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-110s %7d %11.1f %9.1f %6.1f" % (n[:110], c, t, t / c, 100 * t / tot))
PY
find $out -name "*.db" -size +30M -delete
head -30 gpurun_out/${tag}_detr_kernel_stats.txt | cut -c1-175
