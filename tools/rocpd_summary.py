"""Summarise a rocprofv3 rocpd SQLite database: per-kernel time stats and (if present) PMC counter values.
usage: python tools/rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def short(name: str) -> str:
    for key in ("msm_accumulate", "msm_reduce", "msm_window_sum", "msm_digits", "msm_offsets", "msm_count", "msm_ntask",
                "msm_fold", "synth_bases", "poseidon_kernel", "ntt_", "g16_", "pow_table"):
        if key in name:
            i = name.find(key)
            return name[i:i + 40].split("(")[0]
    return name[:70]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# {path}")
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for name, n, total, avg, mn, mx in rows[:25]:
        print(f"{short(name):58s} {n:6d} {total / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * total / tot:6.1f}")
    try:
        pm = cur.execute("select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from counters_collection p "
                         "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by 4 desc").fetchall()
    except sqlite3.Error as e:
        pm = []
        try:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            print("# counters_collection columns:", ccols)
            pm = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                             "group by kernel_name, counter_name order by 4 desc").fetchall()
        except sqlite3.Error as e2:
            print("# no PMC data:", e2)
    if pm:
        print(f"{'kernel':58s} {'counter':>14s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s}")
        for name, cname, n, s, a in pm[:30]:
            print(f"{short(name):58s} {cname:>14s} {n:10d} {s:16.1f} {a:16.1f}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
