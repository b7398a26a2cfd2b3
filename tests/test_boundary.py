"""CPU: the C-ABI library loads and exports every symbol include/wavebreaking_hip.h declares (no compute
calls without a GPU), and nothing in the shipped package imports the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "wavebreaking_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    lib_path = g.build()
    lib = ctypes.CDLL(lib_path)
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libwb_hip.so does not export " + s
    lib.wb_backend.restype = ctypes.c_char_p
    assert lib.wb_backend() == b"hip"


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "wavebreaking_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/" in src and f.endswith(".py") and "import" in src.split("oracle/")[0][-40:]:
                    bad.append(f)
    assert not bad, bad


def test_product_workload_definition_matches_the_oracle_generator():
    """wavebreaking_amd.synthetic (what bench.py feeds the device generator) and oracle.synth (what the tests' numpy fields
    are made of) describe the same record."""
    import numpy as np
    from oracle import synth as osynth
    from wavebreaking_amd import synthetic as psynth
    assert psynth.SEED == osynth.SEED
    for seed in (psynth.SEED, 20241, 7):
        np.testing.assert_array_equal(psynth.wave_params(seed), osynth.wave_params(seed))
    for d in (0.25, 1.0, 2.5):
        for a, b in zip(psynth.grid(d), osynth.grid(d)):
            np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(psynth.times(5), osynth.times(5))


def test_bench_imports_the_oracle_in_its_cpu_baseline_leg_only():
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel, allowed in (("bench.py", {"cpu_baseline", "_cpu_step"}), (os.path.join("benchmarks", "bench_spatial.py"), set())):
        tree = ast.parse(open(os.path.join(root, rel)).read())
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            for node in (fn.body if isinstance(fn, ast.Module) else ast.walk(fn)):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    where = "<module>" if isinstance(fn, ast.Module) else fn.name
                    assert where in allowed, "%s imports the oracle in %s" % (rel, where)
