# PMC pass over the ViTDet backbone (tools/bench_vit.py): LDS and MFMA activity of the attention kernels.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_vit.py > /tmp/pmc.log 2>&1
tail -2 /tmp/pmc.log | cut -c1-200
python - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "attn" not in k: continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r01_pmc_attn.txt", "w")
hdr = "# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/bench_vit.py\n# per kernel, summed over launches; ratios: lds_active / busy_cu, bank_conflict / lds_active, mfma_busy / (4 SIMD * busy_cu)\n"
out.write(hdr); print(hdr)
for k, c in sorted(agg.items()):
    busy = c.get("SQ_BUSY_CU_CYCLES", 0) or 1
    line = "%-34s launches %3d  gui %.3e  busy_cu %.3e  lds_active %.3e (%.2f)  bank_conflict %.3e (%.2f of lds)  mfma_busy %.3e (%.2f)" % (
        k, n[k], c.get("GRBM_GUI_ACTIVE", 0), busy, c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_IDX_ACTIVE", 0) / busy,
        c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1),
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * busy))
    out.write(line + "\n"); print(line)
PY
