#!/bin/bash
# Energy per launch of the first-layer 7x7 head conv: the 8-wave kernel alone, the 4-wave kernel with the blocked accumulation for every launch, and the
# default routing (blocked 4-wave kernel for the wide 3-product launches, 128 x 128 wave tiles where no blocked accumulation is needed):
# tools/halo7_probe.py runs its six bench launches KG_PROBE_REPS times back to back per shape while rocm-smi samples the socket power; the
# accumulated-energy counter is read before and after when the box exposes it.     bash tools/energy_probe.sh > profiles/r05_energy_probe.txt
export KG_PROBE_REPS=${KG_PROBE_REPS:-400}
for V in 0 2 1; do
  case $V in 0) ENVV="KG_HALO7_W4=0 KG_HALO7_NB2=0"; LBL='8-wave kernel only: 64 couts x 64 px per wave, 8 ds_read_b128 per 16 MFMAs';;
             2) ENVV="KG_HALO7_W4=2 KG_HALO7_NB2=0"; LBL='4-wave kernel for every launch: 64 couts x 128 px per wave, 12 ds_read_b128 per 32 MFMAs, blocked accumulation';;
             1) ENVV="KG_HALO7_W4=1 KG_HALO7_NB2=1"; LBL='default routing: 8-wave | 4-wave blocked (wide 3-product) | 4-wave 128 couts x 128 px, 16 reads per 64 MFMAs (single product, 64 -> 192 forward)';; esac
  E0=$(rocm-smi --showenergycounter 2>/dev/null | grep -i "Accumulated Energy" | grep -oE "[0-9.]+" | tail -1)
  T0=$(date +%s.%N)
  env $ENVV python tools/halo7_probe.py > /tmp/probe_$V.out 2>&1 &
  PID=$!
  : > /tmp/pw_$V.txt
  while kill -0 $PID 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level" | sed 's/GPU\[0\]//; s/[\t ]\+/ /g' | tr '\n' ' ' >> /tmp/pw_$V.txt
    echo >> /tmp/pw_$V.txt
    sleep 0.5
  done
  wait $PID
  T1=$(date +%s.%N)
  E1=$(rocm-smi --showenergycounter 2>/dev/null | grep -i "Accumulated Energy" | grep -oE "[0-9.]+" | tail -1)
  echo "== $ENVV ($LBL)"
  grep -v amdgpu.ids /tmp/probe_$V.out
  python3 - "$V" "$E0" "$E1" "$T0" "$T1" <<'PY'
import re, sys
v, e0, e1, t0, t1 = sys.argv[1:6]
pw, ck = [], []
for line in open(f"/tmp/pw_{v}.txt"):
    m = re.search(r"Power \(W\): ([\d.]+)", line)
    c = re.search(r"\((\d+)Mhz\)", line)
    if m and float(m.group(1)) > 900:      # samples taken while the kernels run
        pw.append(float(m.group(1)))
        if c: ck.append(float(c.group(1)))
tot = None
for line in open(f"/tmp/probe_{v}.out"):
    m = re.match(r"sum ([\d.]+) ms", line)
    if m: tot = float(m.group(1))
if pw and tot:
    mp = sum(pw) / len(pw)
    print(f"   socket power while running: mean {mp:.0f} W over {len(pw)} samples (min {min(pw):.0f}, max {max(pw):.0f}); shader clock mean {sum(ck) / max(len(ck), 1):.0f} MHz")
    print(f"   energy of the six launches (3 shapes x {{3 products, 1 product}}): {tot:.3f} ms x {mp:.0f} W = {tot * mp / 1e3:.2f} J")
try:
    print(f"   accumulated-energy counter over the whole process (incl. set-up): {float(e1) - float(e0):.0f} counter units in {float(t1) - float(t0):.1f} s")
except Exception:
    print("   (no accumulated-energy counter on this box)")
PY
done
