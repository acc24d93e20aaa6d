#!/bin/bash
# PMC counters of the weight-gradient kernels over tools/wgrad_probe.py (two passes; averages per kernel).  usage: tools/pmc_wgrad.sh > gpurun_out/r02_wgrad_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  rm -rf gpurun_out/pmc_tmp
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o p -- python tools/wgrad_probe.py 4 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "wgrad" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in agg:
    d={c: v/n[k][c] for c,v in agg[k].items()}
    extra=""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d: extra=" | matrix pipe busy %.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"]/d["GRBM_GUI_ACTIVE"]/128)
    if "SQ_LDS_BANK_CONFLICT" in d: extra=" | LDS conflict share %.3f, VALU per MFMA %.2f" % (d["SQ_LDS_BANK_CONFLICT"]/max(1.0,d["SQ_LDS_IDX_ACTIVE"]), d["SQ_INSTS_VALU"]/max(1.0,d["SQ_INSTS_MFMA"]))
    print(k, {c: "%.4g"%v for c,v in d.items()}, extra)
PY
done
rm -rf gpurun_out/pmc_tmp
