#!/bin/bash
# Samples socket power and shader clock (rocm-smi) while the 7x7 head conv runs back to back (tools/halo7_probe.py with many repetitions):
#   bash tools/power_probe.sh            (GPU box)   -> "power / sclk" samples during the run, then the probe's own output
export KG_PROBE_REPS=${KG_PROBE_REPS:-1200}
python tools/halo7_probe.py > /tmp/probe.out 2>&1 &
PID=$!
for i in $(seq 1 60); do
  sleep 0.7
  echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level" | sed 's/GPU\[0\]//; s/[\t ]\+/ /g' | tr '\n' ' ')"
  kill -0 $PID 2>/dev/null || break
done
wait $PID
cat /tmp/probe.out
