#!/bin/bash
# SQ counters of the one-kernel MLP alone (tools/probes/mlp_fused_probe.py launches): three passes -> gpurun_out/r04_mlp_pmc/
R=$PWD; O=$R/gpurun_out/r04_mlp_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
C="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
for S in A B C; do
  eval "set_=\$$S"
  rm -rf $O/$S
  PYTHONPATH=$R timeout 300 rocprofv3 --pmc $set_ --output-format csv -d $O/$S -o pmc -- python $R/tools/probes/mlp_fused_probe.py > $O/$S.log 2>&1
  tail -2 $O/$S.log
done
cd $R
python - <<PY
import csv, glob, collections
for S in "ABC":
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % S, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            if "mlp_fused" not in k and "glds" not in k: continue
            key = (k, row.get("Grid_Size", ""))
            per[key][row["Counter_Name"]] += float(row["Counter_Value"]); n[key].add(row["Dispatch_Id"])
    for key in sorted(per):
        print(S, key, len(n[key]), {c: round(v / len(n[key])) for c, v in per[key].items()})
PY
rm -rf $O/A $O/B $O/C
