"""The multi-rank paths of mpc_bulletproof_amd.sharding on the HIP path: two fresh rank processes on ONE GPU (gloo for the
rendezvous and the all-gathers -- RCCL refuses two ranks per device; the 8-GPU RCCL run is the driver's), every sum
computed by libbpgpu.so, checked against the single-process GPU result and the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

import bp_helpers as bh
import oracle_lib as o

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_msm_and_combined_check_two_ranks_one_gpu(tmp_path):
    world, n_terms, n_bits, nb = 2, 2999, 8, 9
    port = _free_port()
    prefix = str(tmp_path / "rank")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rank_worker.py"), prefix, str(n_terms), str(n_bits), str(nb)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [json.load(open(f"{prefix}.{r}")) for r in range(world)]
    # term-range-sharded MSM: host-buffer and resident-operand forms, both ranks, == the oracle's single MSM
    sc = o.random_scalars(4100, n_terms)
    pts = ((o.gens("G", 512) + o.gens("H", 512)) * 3)[:64 * n_terms]
    want = o.msm(sc, pts).hex()
    assert all(r["big_host"] == want and r["big_dev"] == want for r in res)
    # proof-level sharding: the gathered accept bits and the sum of the per-rank combined-check partials
    recs, cap = bh.make_range_batch(n_bits, nb, tamper={1})
    sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    assert all(r["ok"] == [1 if s.rc == 0 else 0 for s in sessions] == [0 if i == 1 else 1 for i in range(nb)] for r in res)
    rho = o.random_scalars(777, nb)
    comb = o.msm(rho, b"".join(s.mega_check() for s in sessions), 1).hex()
    assert all(r["comb"] == comb for r in res) and comb != bytes(64).hex()
    assert [(r["lo"], r["hi"]) for r in res] == [(0, 5), (5, 9)] and all(r["tmax"] == 2.0 for r in res)


def test_points_sum(gpu_ctx):
    G = o.generator()
    pts = [o.point_mul(o.s2b(k), G) for k in (3, 5, 11)] + [bytes(64), o.point_mul(o.s2b(o.N - 3), G)]
    assert gpu_ctx.points_sum(b"".join(pts)) == o.point_mul(o.s2b(16), G)
    assert gpu_ctx.points_sum(b"") == bytes(64)
    assert gpu_ctx.points_sum(pts[0] + pts[4]) == bytes(64)                       # P + (-P)
    assert gpu_ctx.points_sum(pts[1] * 200) == o.point_mul(o.s2b(1000), G)        # more points than lanes' first pass
    import mpc_bulletproof_amd as m
    bad = bytearray(pts[0])
    bad[1] ^= 1
    with pytest.raises(m.BpGpuError):
        gpu_ctx.points_sum(bytes(bad) + pts[1])


@pytest.fixture(scope="module")
def gpu_ctx():
    import mpc_bulletproof_amd as m
    g = m.BpGpu(0)
    yield g
    g.close()
