set -eo pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_gen
mkdir -p $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/$name" -o pmc -- python bench.py --steps 2 --warmup 1 --batches 2 --no-cpu-baseline --no-gather > "$out/$name.log" 2>&1
done
python - <<'PY'
import csv,glob,json
c={}
for path in glob.glob("gpurun_out/pmc_gen/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        n=row["Kernel_Name"]
        if "k_classify9" in n:
            key=("general<64" if "general<64" in n or "generalILi64" in n else "general<128" if "general" in n else "lca*")
            c.setdefault(key,{}).setdefault(row["Counter_Name"],[]).append(float(row["Counter_Value"]))
print(json.dumps({k:{cc:[round(x/1e6,1) for x in v] for cc,v in d.items()} for k,d in c.items()}, indent=0))
PY
