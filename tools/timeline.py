#!/usr/bin/env python
"""Print the kernel timeline of one sampling step from a rocprofv3 (rocpd sqlite) kernel trace (start, duration, gap to
the previous kernel's end, queue, workgroups).   usage: python tools/timeline.py r_results.db [step_index] [max_rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks = T("rocpd_kernel_dispatch_"), T("rocpd_info_kernel_symbol_")
rows = cur.execute(f"select d.start,d.end,s.kernel_name,d.queue_id,d.grid_size_x/d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_step_all' in r[2]] or [i for i, r in enumerate(rows) if 'k_advance' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prev_end = None; t0 = rows[a][0]
for r in rows[a:min(b + 1, a + mx)]:
    name = r[2].replace('_ZN2dd', '').replace('.kd', '')[:40]
    gap = (r[0] - prev_end) / 1000 if prev_end else 0
    print(f"{(r[0]-t0)/1000:8.1f} +{(r[1]-r[0])/1000:6.1f}us gap {gap:6.1f} q{r[3]} wg{r[4]:5d} {name}")
    prev_end = max(prev_end or 0, r[1])
