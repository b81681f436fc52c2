#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc counter from the rocpd database."""
import sqlite3
import sys


def main(path, top=12):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = {t.split("_0000")[0]: t for (t,) in cur.execute("select name from sqlite_master where type='table'")}
    pe, ip, kd, ks = (tabs[k] for k in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch",
                                          "rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
    q = f"""select s.kernel_name, p.name, count(*), avg(e.value), min(e.value), max(e.value)
            from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id
            join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name order by sum(e.value) desc limit {top}"""
    try:
        rows = list(cur.execute(q))
    except sqlite3.OperationalError as err:
        print("schema:", cols, err)
        return
    print(f"{'kernel':90s} {'counter':12s} {'n':>6s} {'avg':>14s} {'min':>14s} {'max':>14s}")
    for r in rows:
        print(f"{r[0][:90]:90s} {r[1]:12s} {r[2]:6d} {r[3]:14.1f} {r[4]:14.1f} {r[5]:14.1f}")


def last_n(path, kernel_substr, n):
    """Average of each counter over the LAST n dispatches of the kernels whose name contains
    kernel_substr (bench.py's roofline loop is the tail of the run)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = {t.split("_0000")[0]: t for (t,) in cur.execute("select name from sqlite_master where type='table'")}
    pe, ip, kd, ks = (tabs[k] for k in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch",
                                          "rocpd_info_kernel_symbol"))
    q = f"""select p.name, d.start, sum(e.value) from {pe} e join {ip} p on e.pmc_id = p.id
            join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
            where s.kernel_name like ? group by p.name, d.event_id order by d.start"""
    per = {}
    for name, _, v in cur.execute(q, (f"%{kernel_substr}%",)):
        per.setdefault(name, []).append(v)
    for name, vals in per.items():
        tail = vals[-n:]
        print(f"{name}: last {len(tail)} of {len(vals)} dispatches of *{kernel_substr}*: avg {sum(tail) / len(tail):.4f} "
              f"min {min(tail):.4f} max {max(tail):.4f}")


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[2] == "--last":
        last_n(sys.argv[1], sys.argv[3], int(sys.argv[4]))
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
