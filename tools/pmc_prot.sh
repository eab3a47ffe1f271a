set -eo pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_prot
mkdir -p $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python tools/bench_surface_b.py 20000000 10000000 > "$out/$name.log" 2>&1
done
python - <<'PY'
import csv,glob
c={}
for path in glob.glob("gpurun_out/pmc_prot/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_classify_prot9" in row["Kernel_Name"]:
            c.setdefault(row["Counter_Name"],[]).append(float(row["Counter_Value"]))
for k,v in sorted(c.items()): print(k, ["%.3g"%x for x in v[:3]], "...", ["%.3g"%x for x in v[-3:]])
PY
