#!/usr/bin/env python3
"""What sits between two decode steps on the device: from a rocprofv3 kernel-trace database (rocpd), the kernels and idle
gaps between the last kernel of a decode graph (`pick_final`) and the first attention launch of the next step, as medians
over the last N step seams.   usage: tools/step_gaps.py <db> [N]"""
import sqlite3
import statistics
import sys


def main(path, n=40):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    short = lambda k: k.split("(")[0].replace("void ", "").replace("mi::", "")[:60]  # noqa: E731
    ends = [i for i, r in enumerate(rows) if "pick_final" in r[0] and "pairs" not in r[0]]
    seams = []
    for i in ends:
        j = i + 1
        while j < len(rows) and "paged_attn_decode" not in rows[j][0] and j - i < 12:
            j += 1
        if j < len(rows) and "paged_attn_decode" in rows[j][0]:
            seams.append((i, j))
    # group the seams by what runs in between (the engine's steps have the token / metadata copies there, the bench's
    # bare graph replays have nothing)
    groups = {}
    for i, j in seams:
        groups.setdefault(tuple(short(rows[k][0]) for k in range(i + 1, j)), []).append((i, j))
    for names, ss in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        ss = ss[-n:]
        total = [(rows[j][1] - rows[i][2]) / 1000.0 for i, j in ss]
        busy = [sum(rows[k][2] - rows[k][1] for k in range(i + 1, j)) / 1000.0 for i, j in ss]
        print(f"{len(ss)} seams, end of pick_final -> start of the next step's first attention launch: median "
              f"{statistics.median(total):.2f} us (kernels {statistics.median(busy):.2f}, idle "
              f"{statistics.median([t - b for t, b in zip(total, busy)]):.2f})  min {min(total):.2f}  max {max(total):.2f}")
        for pos, name in enumerate(names):
            gaps = [(rows[a + 1 + pos][1] - rows[a + pos][2]) / 1000.0 for a, b in ss]
            durs = [(rows[a + 1 + pos][2] - rows[a + 1 + pos][1]) / 1000.0 for a, b in ss]
            print(f"  {name:60s} idle before it {statistics.median(gaps):7.2f} us   runs {statistics.median(durs):6.2f} us")
        last = [(rows[b][1] - rows[b - 1][2]) / 1000.0 for a, b in ss]
        print(f"  {'first attention launch of the next step':60s} idle before it {statistics.median(last):7.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
