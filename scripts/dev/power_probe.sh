#!/bin/bash
# Developer aid: board power / clocks while the edge stream runs back to back (is the kernel held at a power-limited clock?)
# usage (on the GPU box): bash scripts/dev/power_probe.sh [ab_stream.py args]
python scripts/ab_stream.py --variants tile32w:0 --rounds 400 "$@" > /tmp/power_probe_run.log 2>&1 &
PID=$!
sleep 45
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|edge" | tr '\n' ';'
  echo
  sleep 1.5
done
wait $PID
tail -2 /tmp/power_probe_run.log
