for TW in 4 2; do
echo "== TILE_WAVES=$TW"
MONORTM_TILE_WAVES=$TW MONORTM_FAR_WAVES=1 PMC_MATCH="far_kernel" PMC_PROFILES=1 PMC_SETS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" tools/pmc_libs.sh c3 - > /dev/null
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcl_-/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "far_kernel" in r["Kernel_Name"]:
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, cs in sorted(acc.items()):
    print("grid", g, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()})
PY
done
