#!/usr/bin/env python3
"""Read a rocprofv3 kernel trace (csv) of one bench run and print, for the last chunk in it, how the P-frame chain's time divides:
busy time per kernel family, gaps between consecutive chain kernels, and how much of the entropy kernels' time lies inside the chain's span.
usage: tools/trace_chain.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    fam = None
    for k, f in (("recon_inter_pre", "pre"), ("recon_sb_kernel", "walk"), ("cdef_sb", "cdef"), ("deblock", "deblock"), ("lr_unit", "lr"), ("motion_search", "me"),
                 ("subpel_refine", "refine"), ("me64", "me"), ("symbolize", "symbolize"), ("rangecode", "rangecode"), ("pack_tiles", "pack"), ("sse_kernel", "sse"), ("tile_order", "order")):
        if k in n:
            fam = f
            break
    if fam:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
ev.sort()
# last chunk: from the last 'pack' backwards to the previous 'pack'
packs = [i for i, e in enumerate(ev) if e[2] == "pack"]
lo = packs[-2] + 1 if len(packs) > 1 else 0
seg = ev[lo:packs[-1] + 1]
t0, t1 = seg[0][0], seg[-1][1]
print("chunk span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
busy = collections.Counter(); cnt = collections.Counter()
for s, e, f in seg:
    busy[f] += e - s; cnt[f] += 1
for f, v in busy.most_common():
    print("  %-10s %4d launches  %8.2f ms total  %7.1f us avg" % (f, cnt[f], v / 1e6, v / 1e3 / cnt[f]))
chain = [x for x in seg if x[2] in ("pre", "walk", "cdef", "deblock", "lr")]
gaps = [chain[i + 1][0] - chain[i][1] for i in range(len(chain) - 1)]
print("chain: %d kernels, busy %.2f ms, gaps %.2f ms (mean %.1f us, max %.1f us), first start +%.2f ms, last end +%.2f ms" % (
    len(chain), sum(e - s for s, e, _ in chain) / 1e6, sum(g for g in gaps if g > 0) / 1e6, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3, (chain[0][0] - t0) / 1e6, (chain[-1][1] - t0) / 1e6))
big = sorted(((g, chain[i][2], chain[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:8]
print("largest gaps (us, after, before):", [(round(g / 1e3, 1), a, b) for g, a, b in big])
by = collections.Counter(); n_by = collections.Counter()
for i, g in enumerate(gaps):
    by[(chain[i][2], chain[i + 1][2])] += g; n_by[(chain[i][2], chain[i + 1][2])] += 1
print("gap by transition:", {k: (round(v / n_by[k] / 1e3, 1), n_by[k]) for k, v in by.items()})
if len(sys.argv) > 2:
    print("head of the chunk:")
    for s, e, f in seg[:int(sys.argv[2])]:
        print("   +%8.1f us  %-10s %7.1f us" % ((s - t0) / 1e3, f, (e - s) / 1e3))
    print("tail of the chunk:")
    for s, e, f in seg[-10:]:
        print("   +%8.1f us  %-10s %7.1f us" % ((s - t0) / 1e3, f, (e - s) / 1e3))
