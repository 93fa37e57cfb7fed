"""Per-kernel sums of arbitrary PMC counters from one rocprofv3 --pmc pass (rocpd sqlite database): for each kernel name the number of
dispatches and, per counter, the sum over its dispatches; plus any ratio asked for as NUM/DEN.
usage: python tools/pmc_generic.py <results.db> COUNTER [COUNTER ...] [--ratio NUM/DEN ...] [--top N]"""
import re
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    db, counters, ratios, top = args[0], [], [], 25
    rest = args[1:]
    while rest:
        if rest[0] == "--ratio":
            ratios.append(rest[1].split("/")); rest = rest[2:]
        elif rest[0] == "--top":
            top = int(rest[1]); rest = rest[2:]
        else:
            counters.append(rest[0]); rest = rest[1:]
    cur = sqlite3.connect(db).cursor()
    acc = {}
    for c in counters:
        try:
            rows = cur.execute("select k.name, p.value from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
                               "where p.counter_name = ?", (c,)).fetchall()
        except sqlite3.Error:
            rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (c,)).fetchall()
        for name, v in rows:
            short = re.sub(r"\(.*", "", name).replace("void ", "").replace("bzk::", "")[:70]
            e = acc.setdefault(short, {"n": {}, "s": {}})
            e["n"][c] = e["n"].get(c, 0) + 1
            e["s"][c] = e["s"].get(c, 0.0) + float(v)
    key = counters[0]
    rows = sorted(acc.items(), key=lambda kv: -kv[1]["s"].get(key, 0))[:top]
    print(f"{'kernel':70s} {'calls':>6s} " + " ".join(f"{c[-18:]:>18s}" for c in counters) + " " + " ".join(f"{a[-9:] + '/' + b[-9:]:>20s}" for a, b in ratios))
    for name, e in rows:
        line = f"{name:70s} {max(e['n'].values()):6d} " + " ".join(f"{e['s'].get(c, 0):18.4g}" for c in counters)
        for a, b in ratios:
            d = e["s"].get(b, 0)
            line += f" {(e['s'].get(a, 0) / d if d else float('nan')):20.4f}"
        print(line)


if __name__ == "__main__":
    main()
