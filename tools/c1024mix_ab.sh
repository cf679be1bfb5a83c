#!/bin/bash
# Run ON THE GPU BOX: 1024-point strided dispatches with 2 and 8 channels mixed per column, product against variant builds, separate processes.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/variants/libjsg_$v.so"; fi
  for ch in 2 8; do
    SP_LIB=$L TP_CHANNELS=$ch TP_BATCHES=$((64/ch)) TP_ROUNDS=7 python tools/tail_probe.py 2>/dev/null | grep "tail plane (pitch" | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', $ch, j['us_per_dispatch_median'], j['best'])"
  done
done; done
