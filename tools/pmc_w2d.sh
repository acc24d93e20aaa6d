#!/bin/bash
# PMC counters of the three kernels of the 2-D Winograd form on one level (tools/w2d_probe.py layer), separate passes as MI355X_MICROARCH.md prescribes:
# busy / wait cycles, instruction mix + LDS conflicts, FETCH_SIZE, WRITE_SIZE.   usage: tools/pmc_w2d.sh "<label prefix, e.g. L5 C256>" <batch>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
only="$1"; B=${2:-4}
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o p -- env PROBE_ONLY="$only" python tools/w2d_probe.py layer $B > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "w2d_" not in k and "wino8r" not in k and "wino4r" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in sorted(agg):
    print("$only B$B", k, {c: "%.4g"%(v/n[k][c]) for c,v in agg[k].items()}, "launches", max(n[k].values()))
PY
done
rm -rf gpurun_out/pmc_tmp
