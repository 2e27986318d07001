cd $GRAFT_REPO_ROOT
for n in 64 256 512; do
  for k in wave wg; do
    echo "inspectors $n kernel $k: $(SUAMD_ST_KERNEL=$k timeout 600 python tools/analyzer_bench.py $n 2>&1 | tail -1)"
  done
done
