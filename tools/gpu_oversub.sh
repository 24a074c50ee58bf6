#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for ov in 4 8 12 16 32; do
  EXTRACK_OVERSUB=$ov python bench.py --no-cpu-baseline --no-extra --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('oversub $ov', round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4), d['config']['launch']['blocks'])"
done; done
