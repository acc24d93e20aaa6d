cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 1 > /dev/null 2>&1
done
python tools/conv_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r02_conv_traffic.json > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python tools/wgrad_probe.py 4 > $O/wgrad_probe_base.txt 2>&1
