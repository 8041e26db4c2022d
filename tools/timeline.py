#!/usr/bin/env python3
"""A text timeline out of a rocprofv3 result database (--kernel-trace --memory-copy-trace): kernels and copies above a minimum
duration with their start and end relative to the first event of the last `window_ms` of the trace, by queue / stream.
    python tools/timeline.py <run_results.db> [window_ms] [min_us]"""
import sqlite3
import sys


def cols(con, view):
    try:
        return [d[0] for d in con.execute(f"select * from {view} limit 1").description]
    except Exception:
        return []


def main():
    db = sys.argv[1]
    window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    for view, kind in (("kernels", "K"), ("memory_copies", "C")):
        c = cols(con, view)
        if not c:
            print(f"(no view {view}; have: {[n for n in names if 'rocpd' not in n][:40]})")
            continue
        pick = lambda *alts: next((a for a in alts if a in c), None)      # noqa: E731
        st, en, nm = pick("start"), pick("end"), pick("name", "kernel_name")
        q = pick("queue_id", "queue", "stream_id", "stream", "dst_agent_abs_index")
        extra = pick("size", "grid_x", "grid_size")
        if not (st and en and nm):
            print(f"({view}: columns {c})")
            continue
        sel = f"select {st},{en},{nm},{q or 'null'},{extra or 'null'} from {view}"
        for a, b, n, qq, x in con.execute(sel):
            ev.append((a, b, kind, str(n), qq, x))
    if not ev:
        return
    ev.sort()
    t_end = max(e[1] for e in ev)
    lo = t_end - window_ms * 1e6
    ev = [e for e in ev if e[0] >= lo and (e[1] - e[0]) >= min_us * 1e3]
    t0 = ev[0][0]
    for a, b, kind, n, qq, x in ev:
        n = n.replace("void ", "")
        if len(n) > 60:
            n = n[:57] + "..."
        print(f"{(a - t0) / 1e6:9.2f} .. {(b - t0) / 1e6:9.2f} ms  {kind} q={qq!s:>4}  {(b - a) / 1e6:8.2f} ms  {n}  [{x}]")


if __name__ == "__main__":
    main()
