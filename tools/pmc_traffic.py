"""Per-launch HBM-side traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, counter-slot table).  Both counters are reported in KiB.

FETCH_SIZE correction.  The guide states the gfx950 reading for WIDE COALESCED streaming reads (x2: 128-B requests tallied at
64 B) and says other access widths are uncalibrated: "calibrate on a known byte count in your own access pattern".  The MSM
kernel gathers 112-byte records with 16-byte loads, so the factor is MEASURED on exactly that pattern: tools/ubench_batched_affine
calib runs k_gather_only (known requested bytes per launch) under the same --pmc FETCH_SIZE pass; factor = requested / reported,
taken from the run whose array is Infinity-Cache resident like the MSM's bases (117 MB) - the other (1.9 GB) is printed beside it.
Without a calibration database the guide's x2 is applied and the output says so.

usage: python tools/pmc_traffic.py <fetch.db> <write.db> <kernel substring> <out.json> [--calib calib_fetch.db requested_bytes]
                                   [--stamp SOURCE_STAMP] [--command "..."] [--calib-from earlier_pmc_traffic.json]
--calib-from: re-use the calibration of an earlier stamped file (the factor is a property of the counter on this access pattern,
not of the library build) when GPU minutes do not allow a fresh calibration pass; the output says where the factor came from."""
import json
import sqlite3
import sys


def per_launch(path, counter, kernel, which=None):
    cur = sqlite3.connect(path).cursor()
    try:
        rows = cur.execute("select k.name, p.value from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
                           "where p.counter_name = ? order by p.dispatch_id", (counter,)).fetchall()
    except sqlite3.Error:
        rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    vals = [v for name, v in rows if kernel in name]
    if not vals:
        raise SystemExit(f"no dispatch of '{kernel}' with {counter} in {path}")
    if which is not None:
        return vals, len(vals)
    return sum(vals) / len(vals) * 1024.0, len(vals)


if __name__ == "__main__":
    args = sys.argv[1:]
    fetch_db, write_db, kernel, out = args[:4]
    opt = args[4:]
    calib = stamp = command = calib_from = None
    while opt:
        if opt[0] == "--calib":
            calib, opt = (opt[1], int(opt[2])), opt[3:]
        elif opt[0] == "--calib-from":
            calib_from, opt = opt[1], opt[2:]
        elif opt[0] == "--stamp":
            stamp, opt = opt[1], opt[2:]
        elif opt[0] == "--command":
            command, opt = opt[1], opt[2:]
        else:
            command, opt = opt[0], opt[1:]
    f, nf = per_launch(fetch_db, "FETCH_SIZE", kernel)
    w, nw = per_launch(write_db, "WRITE_SIZE", kernel)
    factor, how = 2.0, "FETCH_SIZE x2 (the guide's gfx950 figure for wide coalesced reads; NOT calibrated for this kernel's 16-byte gathers)"
    cal_doc = None
    if calib:
        vals, _ = per_launch(calib[0], "FETCH_SIZE", "k_gather_only", which=True)
        # the tool launches the cache-resident array first (warm-up + 2 timed launches), then the 1.9 GB array: first / last thirds
        third = max(1, len(vals) // 2)
        small = sum(vals[:third]) / third * 1024.0
        big = sum(vals[-third:]) / third * 1024.0
        factor = calib[1] / small
        cal_doc = {"requested_bytes_per_launch": calib[1], "fetch_reported_cache_resident_117MB": round(small),
                   "fetch_reported_1p9GB_array": round(big), "factor_cache_resident": round(calib[1] / small, 4),
                   "factor_1p9GB_array": round(calib[1] / big, 4)}
        how = (f"FETCH_SIZE x{factor:.3f}: calibrated on k_gather_only (tools/ubench_batched_affine calib): 112-byte records gathered with "
               "16-byte loads from an Infinity-Cache-resident array, requested bytes / reported bytes; WRITE_SIZE as reported")
    elif calib_from:
        prev = json.load(open(calib_from))
        cal_doc = dict(prev["calibration"], taken_from=calib_from, taken_from_stamp=prev.get("source_stamp"))
        factor = float(prev["calibration"]["factor_cache_resident"])
        how = (f"FETCH_SIZE x{factor:.3f}: calibration re-used from {calib_from} (k_gather_only, tools/ubench_batched_affine calib: 112-byte records "
               "gathered with 16-byte loads from an Infinity-Cache-resident array); WRITE_SIZE as reported")
    doc = {"kernel": kernel, "fetch_bytes_per_launch_raw": round(f), "fetch_factor": round(factor, 4),
           "fetch_bytes_per_launch": round(factor * f), "write_bytes_per_launch": round(w),
           "traffic_bytes_per_launch": round(factor * f + w), "dispatches": [nf, nw], "correction": how, "calibration": cal_doc,
           "source_stamp": stamp, "command": command}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))
