#!/bin/bash
# DEV TOOL: A/B two builds of libjsg.so with tools/kbench.py (interleaved).  usage: tools/ab_kbench.sh <other-lib> <set> [reps]
set -e
cd "$(dirname "$0")/.."
other=$1; kset=${2:-frames}; reps=${3:-2}
cp jadespectrogram_amd/libjsg.so /tmp/libjsg_new.so
for r in $(seq $reps); do
  for which in new other; do
    if [ $which = new ]; then cp /tmp/libjsg_new.so jadespectrogram_amd/libjsg.so; else cp "$other" jadespectrogram_amd/libjsg.so; fi
    echo "== $which (rep $r)"
    timeout -k 10 300 python tools/kbench.py --set $kset 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('  n', j['n'], 'hop', j['hop'], 'frames', j['frames'], 'ch', j['ch'], 'us', j['us'], 'frac', j['frac'])"
  done
done
cp /tmp/libjsg_new.so jadespectrogram_amd/libjsg.so
