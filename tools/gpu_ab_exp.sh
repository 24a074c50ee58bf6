cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in var_old.so libextrack_hip.so; do
    EXTRACK_HIP_LIB=$PWD/extrack_amd/$lib python bench.py --no-cpu-baseline --no-extra --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4))"
  done
done
