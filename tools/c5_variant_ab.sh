#!/bin/bash
# Run ON THE GPU BOX: the C5 image dispatch (Cfg4096B, ARGB out) with variant builds of the library, each in its own process, two interleaved rounds.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/variants/libjsg_$v.so"; fi
  SP_LIB=$L C5_BPC=1 C5_ROUNDS=7 python tools/c5_grid_probe.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['us_per_dispatch_median'], j['best'], j['columns_per_s'])"
done; done
