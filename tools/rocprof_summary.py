#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite DB, the default output of `rocprofv3 --kernel-trace --stats`)
into a small CSV + markdown table that can be committed under profiles/.

usage: tools/rocprof_summary.py gpurun_out/prof1/r01_results.db profiles/r01_kernel_stats [--pmc]
"""
import csv
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:80]


def main():
    db_path, out = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for n, c, t, a, p in rows:
            w.writerow([short(n), c, round(t, 3), round(a, 3), round(p, 3)])
    with open(out + ".md", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats summary of `{db_path}` (durations in microseconds)\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for n, c, t, a, p in rows:
            f.write(f"| {short(n)} | {c} | {t:.1f} | {a:.2f} | {p:.2f} |\n")
    if "--pmc" in sys.argv:
        try:
            q = ("select k.name, p.name, avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                 "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol k on d.kernel_id = k.id "
                 "group by k.name, p.name")
            with open(out + "_pmc.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "counter", "avg_per_dispatch"])
                for k, c, v in db.execute(q):
                    w.writerow([short(k), c, v])
        except Exception as e:  # schema differs between rocprofv3 builds
            print("pmc query failed:", e)
    print("wrote", out + ".csv", out + ".md")


if __name__ == "__main__":
    main()
