#!/bin/bash
# EXPERIMENT: PMC counters of the row-shared kernel variants on two shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0 0" "0 1 1"; do
set -- $cfg
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  rm -rf gpurun_out/pmc_tmp
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o p -- env AID_W4R_MODE=$1 AID_W4R_RAW=$2 AID_W4R_IL=$3 "W4R_SHAPES=8,256,384,64,4;8,64,64,2048,1" python tools/w4r_ab.py 3 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "wino4r" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in agg:
    print("mode$1 raw$2 il$3", k, {c: "%.4g"%(v/n[k][c]) for c,v in agg[k].items()})
PY
done
done
rm -rf gpurun_out/pmc_tmp
