"""Register / LDS / scratch footprint of libbzk's own kernels, read from the gfx950 code objects inside the built
bazuka_amd/csrc/_obj/*.o (llvm-objdump --offloading + llvm-readelf --notes; no GPU needed).
usage: python tools/kernel_resources.py > profiles/<name>.txt"""
import os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "bazuka_amd", "csrc", "_obj")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def extract_device_objects(tmp):
    """{object name: path of its gfx950 code object} for every built bazuka_amd/csrc/_obj/*.o"""
    out = {}
    for o in sorted(f for f in os.listdir(OBJ) if f.endswith(".o")):
        src = os.path.join(tmp, o)
        shutil.copy(os.path.join(OBJ, o), src)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", src], capture_output=True)
        hs = [f for f in os.listdir(tmp) if f.startswith(o + ".") and "gfx950" in f]
        if hs:
            out[o[:-2]] = os.path.join(tmp, hs[0])
    return out


def short_name(demangled):
    return re.sub(r"\(.*", "", demangled.replace("(anonymous namespace)::", "")).replace("bzk::", "").replace("void ", "")


def resources():
    """one dict per kernel of libbzk's own code objects: object, kernel (short demangled name), symbol, vgpr, agpr, sgpr, spill, scratch, lds, wg, waves"""
    tmp = tempfile.mkdtemp()
    rows = []
    try:
        for o, path in extract_device_objects(tmp).items():
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
                name = g("name")
                if "rocprim" in name or "at::native" in name:
                    continue
                rows.append({"object": o, "symbol": name, "agpr": int(blk.split()[0]), "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")),
                             "spill": int(g("vgpr_spill_count")), "scratch": int(g("private_segment_fixed_size")),
                             "lds": int(g("group_segment_fixed_size")), "wg": int(g("max_flat_workgroup_size"))})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    dm = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["kernel"] = short_name(dm.get(r["symbol"], r["symbol"]))
        r["waves"] = min(8, 512 // max(1, ((r["vgpr"] + 7) // 8) * 8))  # on gfx90a+ the unified count already includes the accumulation registers
    return rows


def instruction_counts(obj, symbol_substring):
    """{mnemonic: count} over the body of the first function of code object `obj` (e.g. "msm_g1") whose mangled name contains the substring"""
    tmp = tempfile.mkdtemp()
    try:
        path = extract_device_objects(tmp)[obj]
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    counts, inside = {}, False
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if inside:
                break
            inside = symbol_substring in m.group(1)
            continue
        if inside:
            t = line.split()
            if t and not t[0].startswith("//"):
                counts[t[0]] = counts.get(t[0], 0) + 1
    return counts


def loop_instruction_counts(obj, symbol_substring):
    """{mnemonic: count} over the LARGEST LOOP (the widest backward branch) of the first function of code object `obj` whose mangled name contains the
    substring - for the accumulation kernels: one pass = one point addition.  What must not be in there: scratch stores (a value parked per addition)."""
    tmp = tempfile.mkdtemp()
    try:
        path = extract_device_objects(tmp)[obj]
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    body, inside, name = [], False, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if inside:
                break
            inside = symbol_substring in m.group(1)
            name = m.group(1)
            continue
        if inside:
            a = re.search(r"//\s*([0-9A-Fa-f]+):", line)
            t = line.split()
            if a and t and not t[0].startswith("//"):
                body.append((int(a.group(1), 16), t[0], line))
    if not body:
        return {}
    base = body[0][0]
    best = None
    for addr, mn, line in body:
        if mn.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"<" + re.escape(name) + r"\+0x([0-9a-f]+)>", line)
            if m:
                tgt = base + int(m.group(1), 16)
                if tgt < addr and (best is None or addr - tgt > best[1] - best[0]):
                    best = (tgt, addr)
    if best is None:
        return {}
    counts = {}
    for addr, mn, _ in body:
        if best[0] <= addr <= best[1]:
            counts[mn] = counts.get(mn, 0) + 1
    return counts


def functions_clobbering_return_address():
    """[(object, mangled name)] of every NON-kernel device function that takes s[30:31] - its own return address - as the scratch pair of a
    long-branch expansion (`s_getpc_b64 s[30:31]`).  Round 5, run 4: the compiler did that in a 137 KB no-inline function whose early exits jump
    further than a conditional branch reaches; a wave whose lanes all left early then `returned` to the function's epilogue for ever."""
    tmp = tempfile.mkdtemp()
    bad = []
    try:
        for o, path in extract_device_objects(tmp).items():
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
            kernels = set(re.findall(r"\.name:\s+(\S+)", notes))
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur and cur not in kernels and "s_getpc_b64 s[30:31]" in line and (o, cur) not in bad:
                    bad.append((o, cur))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return bad


def main():
    print("# kernel resources of the gfx950 code objects in bazuka_amd/csrc/_obj (tools/kernel_resources.py)")
    print("# vgpr = unified register count (arch VGPRs + AGPRs, `agpr` of them accumulation registers); waves/SIMD = floor(512 / vgpr) capped at 8; scratch = private segment bytes per lane; LDS = static bytes per workgroup")
    print(f"{'object':9s} {'kernel':78s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s} {'wg max':>7s} {'waves/SIMD':>10s}")
    for r in resources():
        print(f"{r['object']:9s} {r['kernel'][:78]:78s} {r['vgpr']:>5d} {r['agpr']:>5d} {r['sgpr']:>5d} {r['spill']:>6d} {r['scratch']:>8d} {r['lds']:>7d} {r['wg']:>7d} {r['waves']:>10d}")


if __name__ == "__main__":
    main()
