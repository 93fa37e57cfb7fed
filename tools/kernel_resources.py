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


def main():
    tmp = tempfile.mkdtemp()
    rows = []
    try:
        for o in sorted(f for f in os.listdir(OBJ) if f.endswith(".o")):
            src = os.path.join(tmp, o)
            shutil.copy(os.path.join(OBJ, o), src)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", src], capture_output=True)
            hs = [f for f in os.listdir(tmp) if f.startswith(o + ".") and "gfx950" in f]
            if not hs:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, hs[0])], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
                name = g("name")
                if "rocprim" in name or "at::native" in name:
                    continue
                rows.append((o[:-2], name, blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                             g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    dm = demangle([r[1] for r in rows])
    print("# kernel resources of the gfx950 code objects in bazuka_amd/csrc/_obj (tools/kernel_resources.py)")
    print("# vgpr = unified register count (arch VGPRs + AGPRs, `agpr` of them accumulation registers); waves/SIMD = floor(512 / vgpr) capped at 8; scratch = private segment bytes per lane; LDS = static bytes per workgroup")
    print(f"{'object':9s} {'kernel':78s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s} {'wg max':>7s} {'waves/SIMD':>10s}")
    for o, name, agpr, vgpr, sgpr, spill, scratch, lds, wg in rows:
        short = re.sub(r"\(.*", "", dm.get(name, name)).replace("bzk::", "").replace("void ", "")[:78]
        tot = int(vgpr)  # on gfx90a+ the unified count already includes the accumulation registers
        occ = min(8, 512 // max(1, ((tot + 7) // 8) * 8))
        print(f"{o:9s} {short:78s} {vgpr:>5s} {agpr:>5s} {sgpr:>5s} {spill:>6s} {scratch:>8s} {lds:>7s} {wg:>7s} {occ:>10d}")


if __name__ == "__main__":
    main()
