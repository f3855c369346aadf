#!/bin/bash
# SQ counters of the "mixed" step's kernels (the split-bf16 GEMMs / weight gradients): where their waves' time goes.  Three own passes,
# counters only.   tools/probes/split_pmc.sh -> gpurun_out/split_pmc/{pmc_mfma_busy.txt, pmc_c.txt}
R=$PWD; O=$R/gpurun_out/split_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
rm -rf $O/a $O/b $O/c
for s in a b c; do
  case $s in a) set -- $A;; b) set -- $B;; c) set -- $C;; esac
  PRECISION=mixed OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=2 timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/$s -o pmc -- python $R/tools/step_profile.py > $O/$s.log 2>&1
done
cd $R
python tools/pmc_mfma_summary.py $O/a,$O/b $O/pmc_mfma_busy "PRECISION=mixed OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=2 rocprofv3 --pmc <set> -- python tools/step_profile.py" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$O/c/**/*counter_collection.csv", recursive=True):
    rd = csv.DictReader(open(f, newline="")); cols = {c.lower(): c for c in rd.fieldnames}
    for row in rd:
        k = row[cols["kernel_name"]]; per[k][row[cols["counter_name"]]] += float(row[cols["counter_value"]])
        n[k].add(row.get(cols.get("dispatch_id", ""), ""))
with open("$O/pmc_c.txt", "w") as fh:
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        if "split" in k or "f32" in k or "glds" in k:
            fh.write(k[:90] + " launches %d " % len(n[k]) + " ".join("%s=%.3g" % (c, x / max(len(n[k]), 1)) for c, x in sorted(v.items())) + "\n")
PY
grep -E "split|f32" $O/pmc_mfma_busy.txt | cut -c1-260
cat $O/pmc_c.txt | cut -c1-400
rm -rf $O/a $O/b $O/c
