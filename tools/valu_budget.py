"""Where does a proof's vector-issue time go?  From ONE rocprofv3 --pmc pass (SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE, with
--kernel-trace; counter collection serialises the kernels) over tools/prove_serial.py: per kernel the time it took ALONE and the vector-ALU-active cycles it
issued, the latter also expressed in "accumulation-equivalent milliseconds" - the time the same issue work takes at the G1 accumulation's own issue
density (the kernel DESIGN 3.2 shows to be issue-bound).  The sum of that column over a proof is what a perfectly packed pipeline would need per proof;
the distance between it and the sum of the stand-alone times is what overlapping streams can win, and the distance to the measured pipelined time per
proof is what is still lost to scheduling ("first MEASURE it").
usage: python tools/valu_budget.py <results.db> <n_proofs> [pipelined_ms_per_proof]"""
import re
import sqlite3
import sys


def main(db, n_proofs, pipelined_ms=None):
    cur = sqlite3.connect(db).cursor()
    names = {}
    for did, name, s, e in cur.execute("select dispatch_id, name, start, end from kernels"):
        names[did] = (re.sub(r"\(.*", "", name).replace("void ", "").replace("bzk::", "")[:64], e - s)
    ctr = {}
    try:
        rows = cur.execute("select dispatch_id, counter_name, value from counters_collection").fetchall()
    except sqlite3.Error as ex:
        raise SystemExit(f"no counters_collection: {ex}")
    for did, c, v in rows:
        ctr.setdefault(did, {})[c] = ctr.setdefault(did, {}).get(c, 0.0) + float(v)
    per = {}
    for did, (nm, dur) in names.items():
        p = per.setdefault(nm, {"n": 0, "ns": 0, "valu": 0.0, "busy": 0.0, "wavecyc": 0.0, "waves": 0.0, "gui": 0.0})
        c = ctr.get(did, {})
        p["n"] += 1
        p["ns"] += dur
        p["valu"] += c.get("SQ_ACTIVE_INST_VALU", 0.0)
        p["busy"] += c.get("SQ_BUSY_CYCLES", 0.0)
        p["wavecyc"] += c.get("SQ_WAVE_CYCLES", 0.0)
        p["waves"] += c.get("SQ_WAVES", 0.0)
        p["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
    ref = next((v for k, v in per.items() if k.startswith("msm_accumulate_kernel<G1Fast")), None)
    if not ref or not ref["valu"]:
        raise SystemExit("no G1 accumulation with counters in this trace")
    unit = ref["ns"] / ref["valu"]  # ns of stand-alone accumulation per VALU-active count
    print(f"# {db}: {len(names)} dispatches, {n_proofs} proofs; calibration: msm_accumulate<G1> {ref['ns'] / 1e6 / n_proofs:.3f} ms per proof stand-alone (profiled clocks)")
    print(f"{'kernel':64s} {'calls/pf':>8s} {'alone ms/pf':>11s} {'issue-eq ms/pf':>14s} {'issue density':>13s} {'valu/wavecyc':>12s}")
    tot_alone = tot_eq = 0.0
    rows = sorted(per.items(), key=lambda kv: -kv[1]["valu"])
    for nm, p in rows:
        if "synth" in nm or "setup" in nm or "fixed_base" in nm or "table_build" in nm or "at::native" in nm or "pow_table" in nm:
            continue  # one-off work (CRS, tables), not part of a proof
        alone, eq = p["ns"] / 1e6 / n_proofs, p["valu"] * unit / 1e6 / n_proofs
        tot_alone += alone
        tot_eq += eq
        if alone < 0.01 and eq < 0.01:
            continue
        print(f"{nm:64s} {p['n'] / n_proofs:8.1f} {alone:11.3f} {eq:14.3f} {eq / alone if alone else 0:13.2f} {p['valu'] / p['wavecyc'] if p['wavecyc'] else 0:12.2f}")
    print(f"{'TOTAL per proof':64s} {'':8s} {tot_alone:11.3f} {tot_eq:14.3f}")
    print("# issue density = (issue-equivalent ms) / (stand-alone ms): 1.0 = as dense as the G1 accumulation, 0.1 = nine tenths of the device's issue slots idle while it runs alone")
    if pipelined_ms:
        print(f"# measured pipelined time per proof {pipelined_ms:.2f} ms: {100 * tot_eq / pipelined_ms:.0f} % of it is issue work at the accumulation's density; "
              f"{pipelined_ms - tot_eq:.2f} ms per proof are lost to scheduling / low-density phases that nothing overlaps")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else None)
