#!/bin/bash
# Run ON THE GPU BOX: the C3 dispatch (Cfg2048B) with variant builds of the library, each in its own process, two interleaved rounds.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then L=""; else L="tools/variants/libjsg_$v.so"; fi
  SP_LIB=$L PP_PLANS=2 PP_ROUNDS=7 python tools/pair_probe.py 2>/dev/null | grep '"plan_select"' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['us_per_dispatch_median'], j['best'], j['fft_per_s_median'])"
done; done
