"""Per-launch HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass: MI355X_MICROARCH.md, counter-slot table).  Both counters are reported in KiB.  gfx950 correction (same
guide, HBM section): FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is used as reported.
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <kernel substring> <out.json> [command string]"""
import json
import sqlite3
import sys


def per_launch(path, counter, kernel):
    cur = sqlite3.connect(path).cursor()
    try:
        rows = cur.execute("select k.name, count(*), sum(p.value) from counters_collection p join kernels k on "
                           "k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall()
    except sqlite3.Error:
        rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                           "group by kernel_name", (counter,)).fetchall()
    n = tot = 0
    for name, cnt, s in rows:
        if kernel in name:
            n += cnt
            tot += s
    if not n:
        raise SystemExit(f"no dispatch of '{kernel}' with {counter} in {path}")
    return tot / n * 1024.0, n


if __name__ == "__main__":
    fetch_db, write_db, kernel, out = sys.argv[1:5]
    f, nf = per_launch(fetch_db, "FETCH_SIZE", kernel)
    w, nw = per_launch(write_db, "WRITE_SIZE", kernel)
    doc = {"kernel": kernel, "fetch_bytes_per_launch_raw": round(f), "fetch_bytes_per_launch": round(2 * f),
           "write_bytes_per_launch": round(w), "traffic_bytes_per_launch": round(2 * f + w), "dispatches": [nf, nw],
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE as reported; separate --pmc passes",
           "command": sys.argv[5] if len(sys.argv) > 5 else None}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))
