"""Most of bench.py, the worker and the measurement tools only ever RUN on the GPU box: a misspelt name in there would pass every CPU test and
fail the round's bench.  This walks the symbol tables of those files (no execution) and reports every name that is read as a global but bound
nowhere - not at module level, not by an import, not a builtin."""
import builtins
import glob
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(set(
    [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    + glob.glob(os.path.join(ROOT, "bazuka_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py"))
    + glob.glob(os.path.join(ROOT, "tests", "tools", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py"))
    + [os.path.join(ROOT, "tests", f) for f in ("util.py", "mock_node.py", "pystate.py", "bincode_ref.py", "r1cs_scenarios.py")]))


def undefined_globals(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    if any(s.get_name() == "*" for s in top.get_symbols()):  # `from x import *`: cannot be decided statically
        return []
    known = module_names | set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    missing = []

    def walk(tab):
        # names a scope declares `global` and assigns are bound at module level by that assignment
        for s in tab.get_symbols():
            if tab.get_type() != "module" and s.is_declared_global() and s.is_assigned():
                known.add(s.get_name())
        for child in tab.get_children():
            walk(child)

    def check(tab):
        for s in tab.get_symbols():
            if s.is_referenced() and s.is_global() and s.get_name() not in known:
                missing.append((tab.get_name(), tab.get_lineno(), s.get_name()))
        for child in tab.get_children():
            check(child)

    walk(top)
    check(top)
    return missing


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, ROOT) for f in FILES])
def test_no_name_is_read_that_nothing_binds(path):
    assert undefined_globals(path) == []


def test_the_checker_sees_a_misspelt_name(tmp_path):
    f = tmp_path / "m.py"
    f.write_text("import os\nX = 1\ndef f(a):\n    def g():\n        return a + X + os.sep + len('x') + mispelt\n    return g\n")
    assert [n for _, _, n in undefined_globals(str(f))] == ["mispelt"]


def _instance_attributes(cls_name, path):
    import ast
    out = set()
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.ClassDef) and node.name == cls_name:
            for n in ast.walk(node):
                if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "self" and isinstance(n.ctx, ast.Store):
                    out.add(n.attr)
    return out


GPU_ONLY = sorted(set([os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py"), os.path.join(ROOT, "bazuka_amd", "worker.py")]
                      + glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "tools", "*.py"))
                      + glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py")) + [os.path.join(ROOT, "tests", "test_golden_gpu.py")]))


@pytest.mark.parametrize("path", GPU_ONLY, ids=[os.path.relpath(f, ROOT) for f in GPU_ONLY])
def test_calls_into_the_driver_name_things_that_exist(path):
    """`ctx.msm_g1_dev(..)`, `L.MpnWork.decode(..)`: every attribute read on the conventional names of a library context (`ctx`, `bzk`) or of the
    ctypes driver module (however the file imported it) exists on `Bzk` / in `bazuka_amd.lib` - a misspelt method would only fail on the GPU box"""
    import ast
    from bazuka_amd import lib as L
    inst = _instance_attributes("Bzk", os.path.join(ROOT, "bazuka_amd", "lib.py"))
    tree = ast.parse(open(path).read())
    aliases = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom) and n.module in ("bazuka_amd", None) and n.level <= 1:
            aliases |= {a.asname or a.name for a in n.names if a.name == "lib"}
    bad = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name):
            if n.value.id in aliases and not hasattr(L, n.attr):
                bad.append((n.lineno, f"{n.value.id}.{n.attr}"))
            if n.value.id in ("ctx", "bzk") and not n.attr.startswith("_") and not hasattr(L.Bzk, n.attr) and n.attr not in inst:
                bad.append((n.lineno, f"{n.value.id}.{n.attr}"))
    assert bad == []
