#!/bin/bash
# HBM traffic of the F(8,3) (V=2) and F(4,3) (V=1) row-shared kernels per layer shape: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md), FETCH doubled for gfx950, against the algorithmic bytes of the launch (Winograd-domain input once, residual once, output once, weights once).
# usage (GPU box): tools/pmc_traffic_shapes.sh > gpurun_out/r04_wino8_traffic_by_shape.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "8 64 64 64 2048 5 3 1" "8 96 96 192 512 5 3 4" "8 128 128 256 256 5 3 4" "8 128 128 320 128 5 3 4" "8 256 256 384 64 5 3 4" "8 256 256 448 32 5 3 4" "4 256 256 384 64 5 3 4" "4 256 256 448 32 5 3 4"; do
  for V in 2 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/pmc_t_$c
      timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_t_$c -o p -- env PROBE_V=$V PROBE_WINO=30 python tools/conv_probe.py $shape 5 -1 > /dev/null 2>&1
    done
    python - "$shape" $V <<'PY'
import csv, glob, sys
B, Cin, Cout, F, T = [int(v) for v in sys.argv[1].split()[:5]]
V = int(sys.argv[2])
def avg(c):
    f = glob.glob("gpurun_out/pmc_t_%s/*counter_collection.csv" % c)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and ("wino8r" in r["Kernel_Name"] or "wino4r" in r["Kernel_Name"])]
    return sum(v) / max(1, len(v)), len(v)
fe, n = avg("FETCH_SIZE"); wr, _ = avg("WRITE_SIZE")
n_el = B * F * T
xin = (1.25 if V == 2 else 1.5) * 4 * n_el * Cin
alg = xin + 4 * n_el * Cout * 2 + 4 * Cin * Cout * (50 if V == 2 else 30)      # input (Winograd domain) + residual + output + transformed weights
hbm = (2 * fe + wr) * 1024
print("B%d C%d F%d T%d %s: launches %d  FETCH x2 %.1f MB  WRITE %.1f MB  total %.1f MB | algorithmic %.1f MB (input %.1f, residual + output %.1f, weights %.1f) | x%.2f; read / (input + residual + weights) = x%.2f"
      % (B, Cin, F, T, "F(8,3)" if V == 2 else "F(4,3)", n, 2 * fe * 1024 / 1e6, wr * 1024 / 1e6, hbm / 1e6, alg / 1e6, xin / 1e6, 8 * n_el * Cout / 1e6,
         4 * Cin * Cout * (50 if V == 2 else 30) / 1e6, hbm / alg, 2 * fe * 1024 / (xin + 4 * n_el * Cout + 4 * Cin * Cout * (50 if V == 2 else 30))))
PY
  done
done
rm -rf gpurun_out/pmc_t_FETCH_SIZE gpurun_out/pmc_t_WRITE_SIZE
