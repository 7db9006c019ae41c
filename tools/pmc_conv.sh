#!/bin/bash
# PMC profile of one conv shape (igemm or wgrad).  usage: pmc_conv.sh <tag> N H W Cin Cout k [wgrad]
# Counters are collected in separate passes (SQ 8 slots, TCC 4 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
args="$@"
shape="${@:1:6}"; mode="${7:-}"
run() { rocprofv3 --pmc $1 -d $out/$2 -o p --output-format csv -- python tools/conv_micro.py $shape 6 $mode > $out/$2.log 2>&1; }
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" sq1
run "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" sq2
run "FETCH_SIZE GRBM_GUI_ACTIVE" fetch
run "WRITE_SIZE TCC_HIT TCC_MISS" write
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for d in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"]:
                res[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== $tag $args  (per-dispatch means over the conv launches)")
for k, v in sorted(res.items()):
    print("%-28s %14.1f   (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
