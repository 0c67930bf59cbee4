#!/usr/bin/env python3
"""Kernel timeline of the last bench step of a rocprofv3 --kernel-trace run: python tools/timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "fz_merge_lists" in r[0]]
last, prev = idx[-1], idx[-2]
t0 = None
for r in rows[prev + 1:last + 1]:
    if "fz_" not in r[0]:
        continue
    if t0 is None:
        t0 = r[1]
    print("%-44s start %7.1f  dur %7.1f  end %7.1f us" % (r[0][:44], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3))
