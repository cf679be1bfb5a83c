#!/bin/bash
# Run ON THE GPU BOX: the "B" plans (one persistent 8-wave workgroup per CU) at several channel counts, product against variant builds, separate processes.
#   usage: tools/bplan_ab.sh <variant> ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/variants/libjsg_$v.so"; fi
  for cfg in "2048 8" "2048 2" "4096 8" "4096 2" "4096 1"; do set -- $cfg
    SP_LIB=$L SP_N=$1 SP_CHANNELS=$2 SP_BPC_1=0 SP_BPC_2=0 SP_ROUNDS=6 python tools/strided_probe_c3.py 2>/dev/null | grep "B, default" | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$v', '$1', '$2', j['kernel'][:9], j['us_per_batch'], j['ffts_per_s'])"
  done
done; done
