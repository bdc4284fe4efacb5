"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/bpgpu.h declares,
and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    return sorted(set(re.findall(r"\b(bpgpu_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import mpc_bulletproof_amd as m
    lib = ctypes.CDLL(m.lib.SO_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(m.lib.SYMBOLS) == names


def test_no_cpu_fallback_without_gpu():
    import mpc_bulletproof_amd as m
    if m.lib.device_count() > 0:
        return
    try:
        m.BpGpu(0)
    except m.BpGpuError as e:
        assert e.code == m.lib.E_DEVICE
    else:
        raise AssertionError("BpGpu() must fail without a HIP device")


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "mpc_bulletproof_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt.replace("oracle/ (tests/)", ""), os.path.join(dp, f)


def test_bench_uses_the_oracle_only_for_the_cpu_baseline():
    """bench.py may touch oracle/ (through tests/oracle_lib.py) only inside its cpu_baseline worker; the workload,
    the timed GPU legs and the secondary measurements must come from the product path."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    offenders = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            body = ast.get_source_segment(src, node)
            if ("oracle_lib" in body or "pymodel" in body or "liboracle" in body) and node.name != "_cpu_verify_chunk":
                offenders.append(node.name)
    assert offenders == [], offenders
    assert "import oracle_lib" not in src.split("def _cpu_verify_chunk")[0]
