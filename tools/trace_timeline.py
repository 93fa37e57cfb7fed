"""Timeline of a rocprofv3 --kernel-trace database: which kernels ran beside which.  Two views:
  (1) per kernel name: calls, mean duration, and the share of its run time during which a kernel of ANOTHER stream / queue was also running -
      a grid that saturates the device shows neighbours only at its ends;
  (2) a literal listing of one window: start offset, duration, queue / stream, name.
usage: python tools/trace_timeline.py <results.db> [from_ms=auto] [len_ms=12] [max_rows=120]"""
import sqlite3
import sys


def short(name):
    for key in ("msm_accumulate", "msm_reduce", "msm_window", "msm_digits", "msm_offsets", "msm_count", "msm_ntask", "msm_fold", "dedup_", "ntt_", "g16_",
                "wf_", "poseidon", "onesweep", "radix", "scan", "histogram", "lookback"):
        i = name.find(key)
        if i >= 0:
            return name[i:i + 44].split("(")[0]
    return name[:44]


def main(path, from_ms=None, len_ms=12.0, max_rows=120):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    lane = next((c for c in ("stream_id", "queue_id", "tid") if c in cols), None)
    q = f"select name, start, end, {lane or '0'} from kernels order by start"
    rows = cur.execute(q).fetchall()
    if not rows:
        raise SystemExit("no kernels")
    t0 = rows[0][1]
    print(f"# {path}: {len(rows)} kernels over {(rows[-1][2] - t0) / 1e6:.1f} ms; lane column = {lane}; columns: {cols}")
    # (1) overlap shares: sweep over the sorted interval boundaries
    ev = []
    for i, (n, s, e, l) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active, last = set(), None
    alone, shared = [0] * len(rows), [0] * len(rows)
    for t, kind, i in ev:
        if last is not None and active:
            lanes = {rows[j][3] for j in active}
            for j in active:
                if len(lanes) > 1 or len(active) > 1 and lane is None:
                    shared[j] += t - last
                else:
                    alone[j] += t - last
        if kind:
            active.add(i)
        else:
            active.discard(i)
        last = t
    per = {}
    for i, (n, s, e, l) in enumerate(rows):
        k = short(n)
        p = per.setdefault(k, [0, 0, 0, 0])
        p[0] += 1
        p[1] += e - s
        p[2] += shared[i]
        p[3] = max(p[3], e - s)
    print(f"{'kernel':46s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'max_us':>9s} {'beside_other_lane_%':>20s}")
    for k, (c, tot, sh, mx) in sorted(per.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{k:46s} {c:6d} {tot / 1e6:9.2f} {tot / c / 1e3:9.1f} {mx / 1e3:9.1f} {100.0 * sh / max(1, tot):20.1f}")
    # (2) one window, literally
    if from_ms is None:
        from_ms = (rows[len(rows) // 2][1] - t0) / 1e6
    lo, hi = t0 + int(from_ms * 1e6), t0 + int((from_ms + len_ms) * 1e6)
    print(f"# window [{from_ms:.2f}, {from_ms + len_ms:.2f}) ms: start_ms  dur_us  lane  kernel")
    shown = 0
    for n, s, e, l in rows:
        if e > lo and s < hi and shown < max_rows:
            print(f"{(s - t0) / 1e6:10.3f} {(e - s) / 1e3:9.1f}  {str(l):>6s}  {short(n)}")
            shown += 1


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], float(a[1]) if len(a) > 1 and a[1] != "auto" else None, float(a[2]) if len(a) > 2 else 12.0, int(a[3]) if len(a) > 3 else 120)
