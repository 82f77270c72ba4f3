#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/<name>.md
"""
import sqlite3
import sys


def main(path, title=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6, min(d.end-d.start)/1000.0,"
        f" max(d.end-d.start)/1000.0, d.grid_size_x/d.workgroup_size_x, d.workgroup_size_x, max(s.arch_vgpr_count),"
        f" max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id = s.id"
        f" group by s.kernel_name, d.grid_size_x order by 4 desc").fetchall()
    total = sum(r[3] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary {title}\n")
    print(f"source: `{path}` — total kernel time {total:.1f} ms\n")
    print("| kernel | calls | avg us | total ms | % | min us | max us | workgroups | wg size | vgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if r[3] / total < 0.0005:
            continue
        name = r[0].replace(".kd", "")
        print(f"| `{name[:80]}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {100 * r[3] / total:.1f} | {r[4]:.1f} | {r[5]:.1f} |"
              f" {r[6]} | {r[7]} | {r[8]} | {r[9]} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
