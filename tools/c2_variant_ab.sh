#!/bin/bash
# Run ON THE GPU BOX: the C2 strided dispatch (Cfg1024, tail-plane layout) with variant builds of the library, each in its own process, two interleaved rounds.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/variants/libjsg_$v.so"; fi
  SP_LIB=$L TP_ROUNDS=7 python tools/tail_probe.py 2>/dev/null | grep "tail plane (pitch" | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['us_per_dispatch_median'], j['best'], j['frac_of_8_median'])"
done; done
