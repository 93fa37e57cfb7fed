"""How busy was the GPU?  From a rocprofv3 --kernel-trace database: the union of all kernel intervals against the wall time of the
busiest window, the mean number of kernels in flight, and the per-kernel time inside that window.  Used on the pipelined-proofs leg of
bench.py to tell a GPU-bound pipeline (union ~ wall) from a host-bound one (gaps).
usage: python tools/trace_busy.py <results.db> [window_seconds=1.0]"""
import sqlite3
import sys


def main(path, window_s=1.0):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        raise SystemExit("no kernels")
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    win = int(window_s * 1e9)
    # busiest window by kernel count: slide in steps of win / 4
    best, best_lo = -1, t0
    starts = [r[1] for r in rows]
    import bisect
    lo = t0
    while lo + win <= t1:
        n = bisect.bisect_left(starts, lo + win) - bisect.bisect_left(starts, lo)
        if n > best:
            best, best_lo = n, lo
        lo += win // 4
    lo, hi = best_lo, best_lo + win
    sel = [(max(s, lo), min(e, hi), n) for n, s, e in rows if e > lo and s < hi]
    sel.sort()
    union, cur_s, cur_e, total = 0, None, None, 0
    for s, e, _ in sel:
        total += e - s
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    print(f"# {path}: {len(rows)} kernels over {(t1 - t0) / 1e9:.2f} s; busiest {window_s:.2f} s window holds {len(sel)} kernels")
    print(f"GPU busy (union of kernel intervals) {100.0 * union / win:.1f} % of the window; mean kernels in flight while busy {total / max(1, union):.2f}")
    per = {}
    for s, e, n in sel:
        k = n.split("(")[0][-60:]
        per[k] = per.get(k, 0) + (e - s)
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:18]:
        print(f"  {v / 1e6:9.2f} ms  {100.0 * v / total:5.1f} %  {k}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
