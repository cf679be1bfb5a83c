#!/bin/bash
# DEV TOOL (GPU box): what core clock, memory clock and power does the card run at while the bench workloads are on it?
#   tools/clock_watch.sh <out-dir>      -> <out-dir>/clock_<cfg>.txt: one rocm-smi sample every ~0.3 s while `bench.py --config <cfg>` runs
# rocm-smi only reads sysfs here (no settings are changed; an ordinary user cannot change them anyway).
out=${1:-gpurun_out/clock}
mkdir -p "$out"
B="--no-extra --no-boundary --no-calibration --no-parity --no-cpu-baseline --no-single"
rocm-smi --showclocks --showpower > "$out/idle.txt" 2>&1
for cfg in c2 c3 c5; do
  python bench.py --config $cfg --steps 400 --warmup 20 $B > "$out/bench_$cfg.json" 2> "$out/bench_$cfg.err" &
  pid=$!
  : > "$out/clock_$cfg.txt"
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> "$out/clock_$cfg.txt"
    echo "--" >> "$out/clock_$cfg.txt"
    sleep 0.3
  done
  wait $pid
done
