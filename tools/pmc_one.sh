#!/bin/bash
# usage: pmc_one.sh "<counters>" <kernel-substring> -- <command...>   (one rocprofv3 --pmc pass, bounded by timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ctr="$1"; pat="$2"; shift 3
out=gpurun_out/pmc_one; rm -rf $out; mkdir -p $out
timeout 120 rocprofv3 --pmc $ctr -d $out/a -o p --output-format csv -- "$@" > $out/a.log 2>&1
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(res.items()):
    print("%-28s %14.1f   (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $out
