"""Do the forked branch (warm_l2_kernel) and the tensor-parallel seam (allreduce_add_rmsnorm_kernel) overlap on the
device?  Reads the rocpd databases of `rocprofv3 --kernel-trace -- python tools/tp_on_one_gpu.py` (one per rank process)
and counts, per database, the warm launches whose [start, end] intersects a seam launch.
usage: python tools/seam_overlap_trace.py <dir with *.db>"""
import glob
import os
import sqlite3
import sys


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
        if not kd or not ks:
            continue
        rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd[0]} d join {ks[0]} s on d.kernel_id = s.id "
                                "order by d.start"))
        warm = [(a, b) for n, a, b in rows if "warm_l2_kernel" in n]
        seam = [(a, b) for n, a, b in rows if "allreduce_add_rmsnorm_kernel" in n]
        if not warm or not seam:
            print(f"{os.path.basename(path)}: {len(warm)} warm launches, {len(seam)} seam launches")
            continue
        j, hit, shared = 0, 0, 0.0
        for a, b in warm:
            while j < len(seam) and seam[j][1] < a:
                j += 1
            k = j
            while k < len(seam) and seam[k][0] <= b:
                lo, hi = max(a, seam[k][0]), min(b, seam[k][1])
                if hi > lo:
                    hit += 1
                    shared += (hi - lo) / 1000.0
                    break
                k += 1
        wavg = sum(b - a for a, b in warm) / len(warm) / 1000.0
        savg = sum(b - a for a, b in seam) / len(seam) / 1000.0
        print(f"{os.path.basename(path)}: {len(warm)} warm launches (avg {wavg:.2f} us), {len(seam)} seam launches "
              f"(avg {savg:.2f} us): {hit} warm launches run concurrently with a seam launch, {shared / max(hit, 1):.2f} us "
              "shared on average")


if __name__ == "__main__":
    main(sys.argv[1])
