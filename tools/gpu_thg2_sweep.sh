#!/bin/bash
# tile geometry sweep of the (sequence, track)-lane gradient kernel: tools/gpu_thg2_sweep.sh c2|c3 "NT:TT NT:TT ..."
cd $GRAFT_REPO_ROOT
for cfg in $2; do
  NT=${cfg%%:*}; TT=${cfg##*:}
  echo "== threads $NT TT $TT"
  EXTRACK_THG2_THREADS=$NT EXTRACK_THG2_TT=$TT timeout -k 10 200 python tools/gpu_thgrad_time.py $1 2>&1 | tail -1 | sed 's/| objective + gradient/\n   | objective + gradient/' | cut -c1-260
done
