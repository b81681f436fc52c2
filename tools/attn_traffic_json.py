#!/usr/bin/env python3
"""profiles/rNN_attn_traffic.json from the two PMC passes of tools/final_validation.sh: measured HBM bytes per launch of
the fused decode attention kernel (FETCH_SIZE x 2 - the gfx950 correction of MI355X_MICROARCH.md - + WRITE_SIZE, both in
KB, separate rocprofv3 --pmc passes over `python bench.py`) against the algorithmic K/V bytes of the same launches
(bench.py's roofline loop: the last 560 launches of the run, bs 32, every sequence at the run's last context).

usage: python tools/attn_traffic_json.py <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> <bench_under_trace.json> <out.json>"""
import json
import re
import sys


def last_avg(path):
    for line in reversed(open(path).read().splitlines()):
        m = re.search(r"paged_attn_decode_kernel\*: avg ([0-9.]+)", line)
        if m:
            return float(m.group(1))
    raise SystemExit(f"no paged_attn_decode_kernel summary in {path}")


def main(fetch_path, write_path, bench_path, out_path):
    fetch_kb, write_kb = last_avg(fetch_path), last_avg(write_path)
    line = json.loads(open(bench_path).read().strip().splitlines()[-1])
    algo = int(line["roofline"]["bytes_per_launch"])
    batch, ctx_last = line["config"]["batch"], line["config"]["ctx_last"]
    traffic = (fetch_kb * 2 + write_kb) * 1024
    out = {
        "kernel": "paged_attn_decode_kernel<2,8,FUSE,PIPE,false,4> (q/k-norm + RoPE + KV store + paged attention in one launch)",
        "config": f"bs={batch}, per-sequence context {algo // (batch * 4096)}, block 16 (bench.py roofline loop: the last 560 launches of the run; "
                  f"the run's decode steps ended at context {ctx_last})",
        "command": "tools/final_validation.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --no-cpu-baseline ; "
                   "same with --pmc WRITE_SIZE (separate passes)",
        "FETCH_SIZE_kb_avg": fetch_kb, "WRITE_SIZE_kb_avg": write_kb,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM)",
        "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": algo,
        "traffic_over_algorithmic": traffic / algo, "batch": batch, "ctx_sum": algo // 4096,
        "note": "the fused launch also reads the step's qkv rows (262 KB), the RoPE table rows and the norm weights, and writes "
                "the new K/V rows; r04: 1.0099, r03: 1.0100, r02: 1.0098",
    }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
