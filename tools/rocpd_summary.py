"""Summarise a rocprofv3 rocpd (.db) result: per-kernel duration stats and, if present, PMC counter sums.

    python tools/rocpd_summary.py results.db out_prefix      -> out_prefix_kernel_stats.csv, out_prefix_pmc.csv
"""
import csv
import sqlite3
import sys


def main(db, prefix):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(
        """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])
    print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms total kernel time")
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
        pcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
        n = cur.execute("select count(*) from rocpd_pmc_event").fetchone()[0]
    except sqlite3.Error:
        n = 0
    if not n:
        return
    # pmc_event(event_id -> dispatch event), info_pmc(name); dispatch has event_id
    q = """select s.kernel_name, p.name, sum(e.value), count(*)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on d.event_id = e.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, p.name"""
    try:
        pm = cur.execute(q).fetchall()
    except sqlite3.Error as ex:
        print("pmc join failed:", ex, cols, pcols)
        return
    names = sorted({r[1] for r in pm})
    by = {}
    for k, c, v, cnt in pm:
        by.setdefault(k, {})[c] = v
    dur = {r[0]: (r[1], r[2]) for r in rows}
    with open(prefix + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs"] + names)
        for k in sorted(by, key=lambda k: -dur.get(k, (0, 0))[1]):
            w.writerow([k, dur.get(k, (0, 0))[0], dur.get(k, (0, 0))[1]] + [by[k].get(c, "") for c in names])
    print("pmc counters:", names)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])


def gaps(db, lo=0.4, hi=0.8):
    """Idle time between kernels in the middle of the trace (steady-state graph replays): prints the busy
    fraction, the number of gaps and the gap-length histogram."""
    con = sqlite3.connect(db)
    rows = con.execute("select start, end from rocpd_kernel_dispatch order by start").fetchall()
    n = len(rows)
    rows = rows[int(n * lo):int(n * hi)]
    t0, busy_end, idle, hist = rows[0][0], rows[0][1], 0, {}
    for s, e in rows[1:]:
        if s > busy_end:
            g = s - busy_end
            idle += g
            b = min(int(g / 1000), 20)
            hist[b] = hist.get(b, 0) + 1
        busy_end = max(busy_end, e)
    span = busy_end - t0
    print(f"window: {len(rows)} kernels, span {span / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms ({100 * idle / span:.1f}%), "
          f"mean gap {idle / max(1, sum(hist.values())) / 1e3:.2f} us")
    print("gap histogram (us bucket: count):", dict(sorted(hist.items())))


def exclusive(db, prefix, lo=0.4, hi=0.8):
    """Wall-clock attribution: a kernel is charged only for the time by which it pushes the busy front forward
    (end_i - max(start_i, latest end so far)), so the overlap between a draining kernel and its successor is not
    counted twice.  Written as <prefix>_exclusive.csv over the middle (steady-state) part of the trace."""
    con = sqlite3.connect(db)
    rows = con.execute("""select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    n = len(rows)
    rows = rows[int(n * lo):int(n * hi)]
    front = rows[0][0]
    acc = {}
    for s_, e_, name in rows:
        ex = max(0, e_ - max(s_, front))
        front = max(front, e_)
        a = acc.setdefault(name, [0, 0, 0])
        a[0] += 1
        a[1] += ex
        a[2] += e_ - s_
    span = front - rows[0][0]
    with open(prefix + "_exclusive.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "ExclusiveNs", "ExclusivePct", "InclusiveNs"])
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], a[1], round(100.0 * a[1] / span, 3), a[2]])
    print(f"exclusive attribution over {len(rows)} kernels, span {span / 1e6:.3f} ms -> {prefix}_exclusive.csv")


def sequence(db, prefix, n_last):
    """The last `n_last` dispatches in start order -> <prefix>_sequence.csv (one steady-state step when n_last = launches per
    step): start offset, duration, queue / stream when the schema has them, grid size, kernel name."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)")]
    extra = [c for c in ("queue_id", "stream_id", "grid_size_x", "workgroup_size_x") if c in cols]
    sel = "".join(f", d.{c}" for c in extra)
    rows = con.execute(f"""select d.start, d.end, s.kernel_name{sel} from rocpd_kernel_dispatch d
                           join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    rows = rows[-n_last:]
    t0 = rows[0][0]
    with open(prefix + "_sequence.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["start_us", "dur_us"] + extra + ["name"])
        for r in rows:
            w.writerow([round((r[0] - t0) / 1e3, 2), round((r[1] - r[0]) / 1e3, 2)] + list(r[3:]) + [r[2]])
    print(f"{len(rows)} dispatches -> {prefix}_sequence.csv")


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[3] == "seq":
    sequence(sys.argv[1], sys.argv[2], int(sys.argv[4]))
if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "gaps":
    gaps(sys.argv[1])
    exclusive(sys.argv[1], sys.argv[2])
