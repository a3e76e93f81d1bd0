#!/usr/bin/env python
"""Which roof is each conv_gather launch of the bench step under?  Reads the per-launch dump of bench.py (KG_BENCH_DUMP ->
profiles/<tag>_bench_launches.txt) and, per launch shape of conv_gather_kernel / conv1x1_stream_kernel, prints
  * algorithmic HBM bytes (input rows read once: M_in x cin x xP planes x 2 B; output rows written once: M x cout x yP x 2 B; packed weights once)
    / time  ->  GB/s against the ~6.3 TB/s a copy kernel reaches on MI355X,
  * 16-bit MFMA FLOPs issued (algorithmic x plane products) / time against 2.5 PFLOP/s,
  * workgroups of the launch (256 x 128 tiles, or 128 x 64 for the <*, 1> variant) against the 256 CUs,
and the verdict: HBM (>= 50 % of the copy rate), MFMA (>= 40 % of peak), else LATENCY (too few / too short workgroups: the launch is a chain of
load latencies -- the K split and the conv_tiny route exist for those).      python tools/gather_bound_table.py r05"""
import os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = []
for line in open(os.path.join(root, f"{tag}_bench_launches.txt")):
    m = re.match(r"\s*([\d.]+) ms/step\s+([\d.]+) TF\s+x\s*([\d.]+)\s+(\S.*?)\s{2,}(.*)", line)
    if not m or not ("conv_gather" in m.group(4) or "conv1x1_stream" in m.group(4)):
        continue
    ms, tf, cnt, name, desc = float(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4), m.group(5)
    kv = dict(re.findall(r"(\w+)=([\d.]+)", desc))
    M, cout = int(kv["M"]), int(kv["cout"])
    if "K" in kv and "cinp" not in kv:       # conv1x1 wrapper: K = cin_pad of ONE plane set, planes unknown -> both operands two-plane in the forward, one in the backward
        continue
    k, cinp, mode = int(kv["k"]), int(kv["cinp"]), int(kv["mode"])
    xP, yP, prod, stride = int(kv.get("xP", 1)), int(kv.get("yP", 1)), int(kv.get("products", 1)), int(kv.get("stride", 1))
    cin = cinp                                                   # (cin_pad of one plane; the packed weights hold `products` copies)
    per = ms / cnt
    m_in = M * stride * stride if mode in (0, 2) else M // (stride * stride) if stride > 1 else M
    bytes_ = m_in * cin * xP * 2 + M * cout * max(yP, 1) * 2 + cout * k * k * cinp * prod * 2
    gbs = bytes_ / (per * 1e-3) / 1e9
    issued = tf * prod
    wgs = -(-M // (128 if name.endswith(", 1>") else 256)) * -(-cout // (64 if name.endswith(", 1>") else 128))
    verdict = "HBM" if gbs >= 0.5 * 6300 else "MFMA" if issued >= 0.4 * 2500 else "LATENCY"
    rows.append((ms, per, cnt, name, M, cin, cout, k, mode, prod, gbs, issued, wgs, verdict))
rows.sort(reverse=True)
print(f"{'ms/step':>8} {'ms':>7} {'x':>3}  {'kernel':28s} {'M':>8} {'cin':>5} {'cout':>5} k mode prod {'GB/s':>6} {'%copy':>5} {'TF issued':>9} {'%peak':>5} {'WGs':>6}  bound")
tot = {}
for ms, per, cnt, name, M, cin, cout, k, mode, prod, gbs, issued, wgs, v in rows:
    print(f"{ms:8.3f} {per:7.3f} {cnt:3.0f}  {name:28s} {M:8d} {cin:5d} {cout:5d} {k} {mode:4d} {prod:4d} {gbs:6.0f} {100 * gbs / 6300:5.0f} {issued:9.0f} {100 * issued / 2500:5.0f} {wgs:6d}  {v}")
    tot[v] = tot.get(v, 0.0) + ms
print("ms/step by verdict:", {k: round(v, 2) for k, v in tot.items()})
