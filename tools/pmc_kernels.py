"""Per-launch HBM-side traffic of SEVERAL kernels from two rocprofv3 PMC passes of the same command (FETCH_SIZE and WRITE_SIZE cannot
share a pass: MI355X_MICROARCH.md, counter-slot table).  Both counters are reported in KiB.  For wide coalesced streaming reads the
guide's gfx950 figure applies (FETCH_SIZE tallies 128-B requests at 64 B: x2); both the raw and the doubled value are written, and
each entry says which one its kernel's access pattern calls for.

A kernel that is launched several times per operation (the tree's one launch per level) is given the number of OPERATIONS the traced
command ran: its traffic is then the sum over all dispatches divided by that (bytes per operation, not per launch).

usage: python tools/pmc_kernels.py <fetch.db> <write.db> <out.json> --stamp S --command "..." name=substring[:stream|:gather[:runs]] ..."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    try:
        rows = cur.execute("select k.name, p.value from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
                           "where p.counter_name = ? order by p.dispatch_id", (counter,)).fetchall()
    except sqlite3.Error:
        rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    fetch_db, write_db, out = args[:3]
    rest, stamp, command, specs = args[3:], None, None, []
    while rest:
        if rest[0] == "--stamp":
            stamp, rest = rest[1], rest[2:]
        elif rest[0] == "--command":
            command, rest = rest[1], rest[2:]
        else:
            specs.append(rest[0])
            rest = rest[1:]
    fr, wr = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    doc = {"source_stamp": stamp, "command": command, "kernels": {}}
    for spec in specs:
        name, sub = spec.split("=", 1)
        pattern, runs = "stream", 0
        # trailing ":stream" / ":gather" and ":<operations>" are options; the substring itself may hold "::" (C++ names)
        while True:
            head, sep, tail = sub.rpartition(":")
            if sep and tail.isdigit():
                runs, sub = int(tail), head
            elif sep and tail in ("stream", "gather"):
                pattern, sub = tail, head
            else:
                break
        f = [v for k, v in fr if sub in k]
        w = [v for k, v in wr if sub in k]
        if not f or not w:
            doc["kernels"][name] = {"error": f"no dispatch matching '{sub}'"}
            continue
        fa, wa = sum(f) / (runs or len(f)) * 1024.0, sum(w) / (runs or len(w)) * 1024.0
        factor = 2.0 if pattern == "stream" else 1.0
        doc["kernels"][name] = {"match": sub, "dispatches": [len(f), len(w)], "per": ("operation (%d in the traced command)" % runs) if runs else "launch",
                                "fetch_bytes_per_launch_raw": round(fa), "fetch_factor": factor,
                                "write_bytes_per_launch": round(wa), "traffic_bytes_per_launch": round(factor * fa + wa),
                                "correction": ("FETCH_SIZE x2: wide coalesced streaming reads (the guide's gfx950 figure)" if factor == 2.0 else
                                               "FETCH_SIZE as reported: 16-byte gathers of 112 / 224-byte records (x1.008 calibrated on this pattern, profiles/r03_pmc_traffic.json)")}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))
