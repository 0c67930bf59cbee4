#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite result: per-kernel time stats and, when present, PMC counter
sums per kernel.  Usage: python tools/rocpd_summary.py results.db [more.db ...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        print(f"== {path}")
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc").fetchall()
        print("kernel,calls,total_ns,avg_ns,min_ns,max_ns")
        for r in rows:
            print(",".join(str(x) for x in r))
        try:
            ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
            rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                              "group by kernel_name, counter_name order by 1, 2").fetchall()
            if rows:
                print("kernel,counter,dispatches,sum")
                for r in rows:
                    print(",".join(str(x) for x in r))
        except sqlite3.Error as e:
            print("no counters:", e, ccols if 'ccols' in dir() else "")


if __name__ == "__main__":
    main()
