#!/bin/bash
# Run ON THE GPU BOX: the small-workgroup 2048- and 4096-point plans with and without the early first layer (variant e1all), separate processes.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in product e1all; do
  if [ $v = product ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/variants/libjsg_$v.so"; fi
  for cfg in "2048 8" "2048 1" "4096 2" "4096 1"; do set -- $cfg
    SP_LIB=$L SP_N=$1 SP_CHANNELS=$2 SP_BPC_1=0 SP_BPC_2=0 SP_ROUNDS=6 python tools/strided_probe_c3.py 2>/dev/null | grep -v '"batches"' | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print('$v', '$1', '$2', {k:j[k] for k in j if k in ('kernel','us_per_batch','ffts_per_s')})"
  done
done; done
