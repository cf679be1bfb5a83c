#!/bin/bash
# DEV TOOL: A/B two builds of libjsg.so on the same box (interleaved, so clock / box drift hits both alike).
#   usage: tools/ab_lib.sh <other-lib> [reps]      (other-lib e.g. jadespectrogram_amd/libjsg_old.so.keep)
set -e
cd "$(dirname "$0")/.."
other=$1; reps=${2:-3}
cp jadespectrogram_amd/libjsg.so /tmp/libjsg_new.so
for r in $(seq $reps); do
  for which in new other; do
    if [ $which = new ]; then cp /tmp/libjsg_new.so jadespectrogram_amd/libjsg.so; else cp "$other" jadespectrogram_amd/libjsg.so; fi
    for b in 1 8; do
      timeout -k 10 300 python bench.py --no-cpu-baseline --blocks-per-cu $b 2>&1 | tail -1 > /tmp/b.json
      python3 -c "
import json; j=json.load(open('/tmp/b.json')); print('$which', 'bpc', $b, '%.4g' % j['value'], 'in-order us', round(j['roofline']['avg_launch_us'],3), 'conc', round(j['roofline']['concurrent_frac'],3))"
    done
  done
done
cp /tmp/libjsg_new.so jadespectrogram_amd/libjsg.so
