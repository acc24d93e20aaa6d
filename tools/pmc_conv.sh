#!/bin/bash
# PMC counters of the Winograd kernels on one shape (two passes).  usage: tools/pmc_conv.sh "<shape>" <V 0|1|2>   (V: 1 = F(4,3) Winograd-domain input, 2 = F(8,3))
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape="$1"; V=$2
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o p -- env PROBE_V=$V PROBE_WINO=30 python tools/conv_probe.py $shape 5 -1 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "wino4" not in k and "wino8" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in agg:
    print("$shape V=$V", k, {c: "%.4g"%(v/n[k][c]) for c,v in agg[k].items()})
PY
done
rm -rf gpurun_out/pmc_tmp
