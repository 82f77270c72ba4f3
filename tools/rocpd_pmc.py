#!/usr/bin/env python
"""Per-kernel averages of one PMC counter from a rocprofv3 (rocpd sqlite) counter-collection run.

    python tools/rocpd_pmc.py gpurun_out/pmc_FETCH_SIZE/r_results.db [min_calls]
"""
import sqlite3
import sys


def table(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    pe, ip, kd, ks = T("rocpd_pmc_event_"), T("rocpd_info_pmc_"), T("rocpd_kernel_dispatch_"), T("rocpd_info_kernel_symbol_")
    # a counter has one row per hardware instance (XCD x SE ...) and dispatch: sum the instances of a dispatch first,
    # then average over the dispatches of a (kernel, grid)
    rows = cur.execute(
        f"select kname, wgs, cname, count(*), avg(v), min(v), max(v) from ("
        f" select s.kernel_name as kname, d.grid_size_x / d.workgroup_size_x as wgs, i.name as cname, sum(p.value) as v"
        f" from {pe} p join {ip} i on p.pmc_id = i.id join {kd} d on d.event_id = p.event_id join {ks} s on d.kernel_id = s.id"
        f" group by p.event_id, i.name) group by kname, wgs, cname order by 1, 2, 3").fetchall()
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1])
    mc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("| kernel | workgroups | counter | calls | avg | min | max |\n|---|---|---|---|---|---|---|")
    for r in rows:
        if r[3] >= mc:
            print(f"| `{r[0].replace('.kd', '')[:70]}` | {r[1]} | {r[2]} | {r[3]} | {r[4]:.1f} | {r[5]:.1f} | {r[6]:.1f} |")
