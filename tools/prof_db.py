#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel count / avg / min / max / total."""
import sqlite3
import sys


def main(path, top=30):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"""select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0,
            max(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0
            from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 6 desc limit {top}"""
    rows = list(cur.execute(q))
    total = sum(r[5] for r in cur.execute(f"select 0,0,0,0,0,sum(end-start)/1000.0 from {kd}"))
    print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_us':>11s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:7d} {r[2]:9.2f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:11.1f} {100 * r[5] / total:6.2f}")


def last_n(path, kernel_substr, n):
    """Average duration of the LAST n dispatches of the kernels whose name contains kernel_substr
    (bench.py's roofline loop is the tail of the run)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"""select (d.end - d.start) / 1000.0 from {kd} d join {ks} s on d.kernel_id = s.id
            where s.kernel_name like ? order by d.start"""
    vals = [r[0] for r in cur.execute(q, (f"%{kernel_substr}%",))]
    tail = vals[-n:]
    print(f"last {len(tail)} of {len(vals)} dispatches of *{kernel_substr}*: avg {sum(tail) / len(tail):.2f} us "
          f"min {min(tail):.2f} max {max(tail):.2f}")


def window(path, anchor_substr, n_anchor, label):
    """Device timeline between the first and the last of the LAST n_anchor dispatches of the anchor kernel
    (e.g. the 28 prefill-attention launches of the bench's final prefill step): busy time per kernel, idle gaps."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"""select s.kernel_name, d.start, d.end from {kd} d join {ks} s
                                on d.kernel_id = s.id order by d.start"""))
    anchors = [i for i, r in enumerate(rows) if anchor_substr in r[0]][-n_anchor:]
    lo, hi = anchors[0], anchors[-1]
    span = (rows[hi][2] - rows[lo][1]) / 1000.0
    per, busy, prev_end = {}, 0.0, rows[lo][1]
    gaps = 0.0
    for name, st, en in rows[lo:hi + 1]:
        c = per.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += (en - st) / 1000.0
        busy += (en - st) / 1000.0
        gaps += max(0, st - prev_end) / 1000.0
        prev_end = max(prev_end, en)
    print(f"{label}: {n_anchor} x *{anchor_substr}* first start -> last end: {span:.1f} us "
          f"({span / max(1, n_anchor - 1) :.1f} us per anchor interval), kernels busy {busy:.1f} us, idle gaps {gaps:.1f} us")
    for name, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {name[:96]:96s} {cnt:5d} {tot / cnt:9.2f} us avg {tot:10.1f} us {100 * tot / span:5.1f} %")


def edges(path, anchor_substr, n_anchor, around=14):
    """The kernels just before the first and just behind the last of the LAST n_anchor dispatches of the anchor kernel
    (a prefill step's head: embedding .. first attention, and tail: last attention .. sampler), with start offsets,
    durations and the idle gap in front of each - what a step spends outside its layers."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"""select s.kernel_name, d.start, d.end from {kd} d join {ks} s
                                on d.kernel_id = s.id order by d.start"""))
    anchors = [i for i, r in enumerate(rows) if anchor_substr in r[0]][-n_anchor:]
    lo, hi = anchors[0], anchors[-1]
    t0 = rows[lo][1]
    for title, a, b in (("head", max(0, lo - around), lo + 1), ("tail", hi, min(len(rows), hi + around))):
        print(f"{title}: (us relative to the first anchor's start; gap = idle time in front of the kernel)")
        for i in range(a, b):
            name, st, en = rows[i]
            gap = (st - rows[i - 1][2]) / 1000.0 if i else 0.0
            print(f"   {(st - t0) / 1000.0:10.1f}  {(en - st) / 1000.0:8.1f} us  gap {gap:8.1f}  {name[:90]}")


def layers(path, anchor_substr, n_anchor, groups=2):
    """The last `groups` runs of n_anchor dispatches of the anchor kernel (the bench's prefill steps, newest last) side by
    side: start-to-start interval of consecutive anchors, i.e. the time of each layer, and the anchor's own duration -
    where a step that is slower than its successor loses its time (round 5: the first step behind an idle device)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"""select d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id
                                where s.kernel_name like ? order by d.start""", (f"%{anchor_substr}%",)))
    runs = [rows[len(rows) - (g + 1) * n_anchor: len(rows) - g * n_anchor] for g in range(groups)][::-1]
    print(f"layer by layer, the last {groups} runs of {n_anchor} x *{anchor_substr}* (oldest first): "
          "anchor-to-anchor interval us / anchor duration us")
    for i in range(n_anchor):
        cells = []
        for r in runs:
            iv = (r[i + 1][0] - r[i][0]) / 1000.0 if i + 1 < n_anchor else float("nan")
            cells.append(f"{iv:8.1f} / {(r[i][1] - r[i][0]) / 1000.0:6.1f}")
        print(f"   layer {i:2d}   " + "      ".join(cells))
    for r in runs:
        print(f"   run: first anchor start -> last anchor end {(r[-1][1] - r[0][0]) / 1000.0:9.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[2] == "--layers":
        layers(sys.argv[1], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 2)
        sys.exit(0)
    if len(sys.argv) > 4 and sys.argv[2] == "--edges":
        edges(sys.argv[1], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    if len(sys.argv) > 5 and sys.argv[2] == "--window":
        window(sys.argv[1], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    elif len(sys.argv) > 4 and sys.argv[2] == "--last":
        last_n(sys.argv[1], sys.argv[3], int(sys.argv[4]))
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
