out=gpurun_out/r04_streams_after_ks.txt; : > $out
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for s in "" "--streams 3" "--streams 4" "--split 3,3,2" "" "--streams 3"; do
  echo "batch 8 $s: $(timeout 900 python bench.py $s --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> $out
done
cat $out
