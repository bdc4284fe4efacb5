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


CPU_WORKERS = ("_cpu_verify_chunk", "_cpu_prove_chunk", "_cpu_check_proofs")   # bench.py: cpu_baseline legs + the oracle's verdict on GPU-made proofs


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
            if ("oracle_lib" in body or "pymodel" in body or "liboracle" in body) and node.name not in CPU_WORKERS:
                offenders.append(node.name)
    assert offenders == [], offenders
    assert "import oracle_lib" not in src.split("def _cpu_verify_chunk")[0]
    # the workers run in a fork pool created before the GPU is initialised, and only through pool.map
    for w in CPU_WORKERS:
        for m in re.finditer(r"\b%s\b" % w, src):
            line = src[src.rfind("\n", 0, m.start()) + 1:src.find("\n", m.end())]
            assert line.startswith("def " + w) or "pool.map(" + w in line, line


def test_rust_sys_block_is_complete_and_in_step_with_the_header():
    """shim/src/sys.rs (the `extern "C"` block a Rust host links against; generated, uncompiled here) declares every
    function of include/bpgpu.h with the same parameter count, and is what tools/gen_rust_sys.py produces today."""
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]) == 0, \
        "shim/src/sys.rs is stale: run tools/gen_rust_sys.py"
    rs = open(os.path.join(ROOT, "shim", "src", "sys.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "bpgpu.h")).read(), flags=re.S)
    for name in _declared():
        m = re.search(r"pub fn %s\((.*?)\)( -> [^;]+)?;" % name, rs)
        assert m, name
        c = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, hdr, flags=re.S)
        nc = 0 if c.group(1).strip() in ("", "void") else c.group(1).count(",") + 1
        nr = 0 if not m.group(1).strip() else m.group(1).count(",") + 1
        assert nc == nr, (name, nc, nr)


def test_host_harness_lives_under_tests():
    """the reference's gadgets / tests restated and the flat C harness are test infrastructure (tests/host/), not product"""
    host = os.path.join(ROOT, "mpc_bulletproof_amd", "host")
    assert sorted(f for f in os.listdir(host) if f.endswith((".cpp", ".hpp"))) == ["mpc_bulletproof.cpp", "mpc_bulletproof.hpp"]
    assert os.path.exists(os.path.join(ROOT, "tests", "host", "capi.cpp"))


def test_thread_pool_nested_loops_do_not_deadlock():
    """The host mirror's persistent pool: a parallel_for started from inside a running loop -- on a worker or on the calling
    thread while it runs its own slice -- must run serially instead of re-locking the pool (Prover::prove_batch with nb >= 2
    provers of >= 4096 multipliers draws its blinding vectors that way).  CPU only: no device call."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "pool_test")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for threads in ("4", "2"):
        r = subprocess.run([exe], env=dict(os.environ, BPH_THREADS=threads), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "all passed" in r.stdout, r.stdout + r.stderr
    # the same binary also holds keccak256 known answers: once more with the scalar permutation forced (the default run takes the
    # AVX-512 one where the CPU has it), so both implementations are pinned wherever the suite runs
    r = subprocess.run([exe], env=dict(os.environ, BPH_KECCAK_SCALAR="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all passed" in r.stdout, r.stdout + r.stderr


def test_options_are_per_context_and_validated():
    """bpgpu_set_option / bpgpu_get_option need a context, i.e. a device: without one only the argument checks can run."""
    import ctypes as C
    import mpc_bulletproof_amd as m
    lib = C.CDLL(m.lib.SO_PATH)
    v = C.c_int64()
    assert lib.bpgpu_set_option(None, 1, C.c_int64(0)) == m.lib.E_ARG
    assert lib.bpgpu_get_option(None, 1, C.byref(v)) == m.lib.E_ARG
    # every option of the header has a name in the binding
    hdr = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    ids = {int(x) for x in re.findall(r"#define BPGPU_OPT_(?!COUNT)[A-Z_0-9]+ (\d+)", hdr)}
    assert ids == set(m.lib.OPT.values()) and max(ids) + 1 == int(re.search(r"#define BPGPU_OPT_COUNT (\d+)", hdr).group(1))


def test_round3_entry_points_reject_null_arguments_without_a_device():
    """The entry points added in round 3 return BPGPU_E_ARG for a missing context / operands before anything touches a device
    (the header's contract: functions never throw and never abort)."""
    import ctypes as C
    import mpc_bulletproof_amd as m
    lib = C.CDLL(m.lib.SO_PATH)
    E = m.lib.E_ARG
    z, n = C.c_size_t(0), C.c_size_t(4)
    nf = C.c_size_t(7)
    assert lib.bpgpu_r1cs_verify_stream(None, None, None, n, z, z, None, None, None, None) == E
    assert lib.bpgpu_r1cs_verify_stream_dev(None, None, None, n, z, z, None, None, None, None) == E
    assert lib.bpgpu_r1cs_verify_screened(None, None, None, n, z, z, None, None, None, None, None, C.byref(nf)) == E
    assert lib.bpgpu_r1cs_verify_screened_dev(None, None, None, n, z, z, None, None, None, None, None, C.byref(nf)) == E
    assert lib.bpgpu_r1cs_verify_screened_fs_dev(None, None, None, n, z, z, None, None, None, None, None, C.byref(nf)) == E
    assert lib.bpgpu_r1cs_verify_shard(None, None, None, z, z, None, None, None, None, z, C.c_size_t(1), None) == E
    assert lib.bpgpu_set_shard(None, z, C.c_size_t(1)) == E
    assert lib.bpgpu_r1cs_prover_commit(None, None, None, n, n, None, None, None, None, None, None, None, None) == E
    assert lib.bpgpu_r1cs_prover_session_polys(None, None, None, None, None, None, None) == E
    lib.bpgpu_profile_epoch.restype = C.c_void_p
    assert lib.bpgpu_profile_epoch(None) is None
    assert lib.bpgpu_profile_intervals(None, None, z, None, None, None, None) == E
