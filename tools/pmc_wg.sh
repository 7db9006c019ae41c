#!/bin/bash
# usage: pmc_wg.sh <lean> N H W Cin Cout k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lean=$1; shift
out=gpurun_out/pmc_wg$lean; rm -rf $out; mkdir -p $out
run() { ALDI_WGRAD_LEAN=$lean rocprofv3 --pmc $1 -d $out/$2 -o p --output-format csv -- python tools/conv_micro.py $3 $4 $5 $6 $7 $8 4 wgrad > $out/$2.log 2>&1; }
run "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" a $@
run "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" b $@
run "FETCH_SIZE TCC_HIT TCC_MISS" c $@
run "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" d $@
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== lean=$lean $@")
for k, v in sorted(res.items()):
    print("%-28s %14.1f   (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
