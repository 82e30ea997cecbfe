"""Kernel timeline (start offset, duration, stream/queue, name) from a rocprofv3 rocpd .db — for looking at launch gaps and overlap.
usage: python tools/rocpd_timeline.py x_results.db [name-substring] [max-rows]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 80
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
t0 = rows[0][1]
prev_end = None
n = 0
for r in rows:
    name = re.sub(r"\s+", " ", r[0])
    if sub and sub not in name:
        continue
    gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(r[1] - t0) / 1e3:11.1f} us  +{(r[2] - r[1]) / 1e3:8.1f} us  gap {gap:8.1f}  q={r[3] if qcol else '-'}  {name[:90]}")
    prev_end = r[2]
    n += 1
    if n >= mx:
        break
