#!/usr/bin/env python
"""Recomputes the bench line's roofline of the dominant kernel from the two committed profiles of the same command:
  profiles/<tag>_bench_launches.txt      per launch shape: ms/step, achieved algorithmic TFLOP/s, launches/step, products (HIP events)
  profiles/<tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats: per kernel calls / total ns / ms per step
usage: python tools/roofline_check.py [tag]   ->  algorithmic FLOPs per step of conv_halo<7,1> (k1skip excluded: another instantiation),
MFMA-issued FLOPs (x products), and both divided by the profiler's duration of conv_halo_kernel<7, 1, 8, 0> and by 2.5 PFLOP/s."""
import csv, re, sys, os
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
alg = issued = ms_ev = 0.0
n = 0
for line in open(os.path.join(root, f"{tag}_bench_launches.txt")):
    m = re.match(r"\s*([\d.]+) ms/step\s+([\d.]+) TF\s+x\s*([\d.]+)\s+(\S.*?)\s{2,}(.*)", line)
    nm = m.group(4).replace(" ", "") if m else ""
    if not m or not (nm in ("conv_halo<7,1>", "conv_halo_kernel<7,1,8,0>") or nm.startswith("conv_halo7_w4_kernel<")):      # (incl. "w4<*,2>+conv_halo_kernel<7,1,8,0>": one call, two launches)
        continue
    ms, tf, cnt, desc = float(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(5)
    prod = int(re.search(r"products=(\d+)", desc).group(1)) if "products=" in desc else 1
    fl = ms * 1e-3 * tf * 1e12
    alg += fl; issued += fl * prod; ms_ev += ms; n += cnt
prof_ms = calls = 0.0      # (the family: the 8-wave kernel + its 4-wave sibling with the blocked accumulation, round 5)
for r in csv.DictReader(open(os.path.join(root, f"{tag}_bench_kernel_stats.csv"))):
    nm = r["Name"].replace(" ", "")
    if nm in ("voidconv_halo_kernel<7,1,8,0>(HaloArgs)", "voidconv_halo_kernel<7,1,8,0,false>(HaloArgs)") or nm.startswith("voidconv_halo7_w4_kernel<"):
        prof_ms += float(r["MsPerStep"]); calls += float(r["CallsPerStep"])
print(f"conv_halo<7,1,8,0> + conv_halo7_w4: {n:.0f} launches/step (profiler: {calls:.0f}), algorithmic {alg / 1e12:.2f} TFLOP/step, MFMA-issued {issued / 1e12:.2f} TFLOP/step "
      f"(x{issued / alg:.3f} products per multiply)")
print(f"  HIP events  : {ms_ev:.2f} ms/step -> {alg / ms_ev / 1e9:.1f} TFLOP/s algorithmic, {issued / ms_ev / 1e9:.1f} issued = {issued / ms_ev / 1e9 / 2500:.3f} of 2.5 PFLOP/s")
print(f"  rocprofv3   : {prof_ms:.2f} ms/step -> {alg / prof_ms / 1e9:.1f} TFLOP/s algorithmic, {issued / prof_ms / 1e9:.1f} issued = {issued / prof_ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s")
