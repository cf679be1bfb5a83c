#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product xcd3 xcd4 xcd6 xcd7 xcd8; do
  if [ $v = product ]; then L=""; else L="tools/variants/libjsg_$v.so"; fi
  SP_LIB=$L TP_ROUNDS=7 python tools/tail_probe.py 2>/dev/null | grep "tail plane (pitch" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['us_per_dispatch_median'], j['frac_of_8_median'], j['frac_of_8_best'])"
done; done
