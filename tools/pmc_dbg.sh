#!/bin/bash
# usage: pmc_dbg.sh <dbg> N H W Cin Cout k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
dbg=$1; shift
out=gpurun_out/pmc_dbg$dbg; rm -rf $out; mkdir -p $out
ALDI_IGEMM_DBG=$dbg rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $out/a -o p --output-format csv -- python tools/conv_micro.py $@ 4 > $out/a.log 2>&1
ALDI_IGEMM_DBG=$dbg rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/b -o p --output-format csv -- python tools/conv_micro.py $@ 4 > $out/b.log 2>&1
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== dbg=$dbg $@")
for k, v in sorted(res.items()):
    print("%-28s %14.1f   (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
