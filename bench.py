#!/usr/bin/env python3
"""bench.py -- range-proof verifications/s on MI355X (BASELINE.json configs[1]).

One step = one pass of the verification hot path (reference src/r1cs/verifier.rs:457-553: constraint
flattening, inversions, verifier scalar assembly, the 154-term mega_check MSM, identity test) over a
batch of 1024 proofs of the 64-bit range gadget (tests/r1cs.rs:620-652, m = 1), inputs resident in
HBM (proof points/scalars + host-transcript challenges, boundary byte encodings of include/bpgpu.h).
Multi-GPU: one process per GPU, each rank verifies its own 1024 proofs (independent units, no
data-path collective) -> weak scaling; value = all ranks' proofs / max-over-ranks time.
`python bench.py --gpus N` without a launcher starts the N rank processes itself (fresh children,
before this process touches the GPU); under torch.distributed.run it reads RANK / WORLD_SIZE.

Prints ONE JSON line on rank 0 (contract in the task prompt) with `roofline` and `cpu_baseline`.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_BITS = 64
LABEL = b"RangeProofTest"
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# Mix-weighted VALU issue peak: 74 % of the instructions of the elliptic-curve kernels are v_mad_i64_i32, which holds a
# SIMD for 4.6 cycles per wave64 instruction (33.9 T lane-MAD/s measured chip-wide, profiles/r01_microbench_primitives.log);
# the doubling / mixed-addition micro-benchmarks, which ARE this mix, issue one wave64 instruction per 4.25 cycles per
# SIMD at 8 waves/SIMD.  1024 SIMDs x 2.4 GHz / 4.25.  (A kernel of plain 2-cycle VALU instructions would exceed it.)
VALU_ISSUE_PEAK_MIX = 256 * 4 * 2.4e9 / 4.25
MAD_PEAK_TOPS = 33.9            # measured v_mad_u64_u32 rate on MI355X (profiles/r01_microbench_primitives.log)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")


def _gen_workload(path, nb, seed0):
    """child process (forked before the parent touches the GPU): prove nb single-value 64-bit range proofs with the
    product path itself (C++ host mirror over the C ABI, provers in lock-step on the GPU), replay the verifier
    transcripts on the host and write the operands of bpgpu_r1cs_verify_batch to `path`."""
    import ctypes as C
    import pickle
    host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
    vals = [(0x9E3779B97F4A7C15 * (i + 1) + seed0) & ((1 << N_BITS) - 1) for i in range(nb)]
    arr = (C.c_uint64 * nb)(*vals)
    lab = (C.c_uint8 * len(LABEL)).from_buffer_copy(LABEL)
    proofs, coms, plen = (C.c_uint8 * (nb * 4096))(), (C.c_uint8 * (nb * 64))(), C.c_size_t(0)
    prove_s = None
    for rep in range(2):      # the second call has its workspaces and generator tables in place: that one is timed
        t0 = time.perf_counter()
        rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(1), C.c_size_t(N_BITS), lab, C.c_size_t(len(LABEL)), arr,
                                        C.c_uint64(seed0), C.c_size_t(N_BITS), proofs, C.byref(plen), coms)
        prove_s = time.perf_counter() - t0
        assert rc == 0, f"bph_range_prove_batch rc={rc}"
    pl = plen.value
    k = N_BITS.bit_length() - 1
    nvar = 11 + 1 + 2 * k
    pts, sc, ch = (C.c_uint8 * (nb * nvar * 64))(), (C.c_uint8 * (nb * 5 * 32))(), (C.c_uint8 * (nb * (6 + k) * 32))()
    init, dims = (C.c_uint8 * 32)(), (C.c_size_t * 6)()
    cap = 8 * N_BITS + 8
    rp, kind, idx, coeff = (C.c_uint32 * (2 * N_BITS + 2))(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint8 * (32 * cap))()
    rc = host.bph_range_verify_inputs(C.c_size_t(nb), C.c_size_t(N_BITS), lab, C.c_size_t(len(LABEL)), coms, proofs,
                                      C.c_size_t(pl), C.c_size_t(N_BITS), pts, sc, ch, init, dims, rp, kind, idx, coeff)
    assert rc == 0, f"bph_range_verify_inputs rc={rc}"
    n1, n, kk, m, q, nnz = list(dims)
    assert kk == k and m == 1
    G, H, B = (C.c_uint8 * (64 * N_BITS))(), (C.c_uint8 * (64 * N_BITS))(), (C.c_uint8 * 64)()
    assert host.bph_gens(ord("G"), 0, C.c_size_t(N_BITS), G) == 0 and host.bph_gens(ord("H"), 0, C.c_size_t(N_BITS), H) == 0
    host.bph_generator(B)
    # the same proofs in the reference's wire format (R1CSProof::to_bytes) + compressed commitments
    wire, wl_len = (C.c_uint8 * 4096)(), C.c_size_t(0)
    wires = []
    for i in range(nb):
        one = (C.c_uint8 * pl).from_buffer_copy(bytes(proofs)[i * pl:(i + 1) * pl])
        assert host.bph_proof_flat_to_wire(one, C.c_size_t(pl), wire, C.byref(wl_len)) == 0
        wires.append(bytes(wire)[:wl_len.value])
    ccom = (C.c_uint8 * (nb * 32))()
    assert host.bph_compress_points(coms, C.c_size_t(nb), ccom) == 0
    wl = {"wire_proofs": b"".join(wires), "wire_len": len(wires[0]), "wire_commitments": bytes(ccom),
          "proofs": bytes(proofs)[:nb * pl], "proof_len": pl, "commitments": bytes(coms), "points": bytes(pts),
          "scalars": bytes(sc), "challenges": bytes(ch), "init_state": bytes(init), "dims": (n1, n - n1, k, m),
          "csr": (list(rp)[:q + 1], list(kind)[:nnz], list(idx)[:nnz], bytes(coeff)[:32 * nnz]),
          "G": bytes(G), "H": bytes(H), "B": bytes(B), "prove_seconds": prove_s}
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        pickle.dump(wl, f)
    os.replace(tmp, path)


def _cpu_verify_chunk(args):
    import oracle_lib as o
    proofs, coms, plen = args
    nb = len(proofs) // plen
    t0 = time.perf_counter()
    ok = o.r1cs_verify_many(o.K_RANGE, N_BITS, LABEL, [], coms, 1, proofs, plen, nb, N_BITS)
    return sum(ok), time.perf_counter() - t0


def _cpu_baseline(wl, nb, ncpu):
    """The CPU oracle's Verifier::verify restatement (transcript replay + scalar assembly + 154-term Pippenger MSM) on a
    bounded sample of the same workload: all host cores, then ONE thread.  Fork pools, before any GPU initialisation."""
    plen = wl["proof_len"]
    per = (nb + ncpu - 1) // ncpu
    chunks = []
    for c in range(ncpu):
        lo, hi = c * per, min(nb, (c + 1) * per)
        if hi > lo:
            chunks.append((wl["proofs"][lo * plen:hi * plen], wl["commitments"][64 * lo:64 * hi], plen))
    with mp.get_context("fork").Pool(len(chunks)) as pool:
        pool.map(_cpu_verify_chunk, [(b"", b"", plen)] * len(chunks))     # start the workers, load the library
        t0 = time.perf_counter()
        res = pool.map(_cpu_verify_chunk, chunks, chunksize=1)
        wall = time.perf_counter() - t0
    assert sum(r[0] for r in res) == nb, "the CPU oracle rejects proofs made by the GPU prover"
    allc = {"value": nb / wall, "unit": "verifications/s", "cores": len(chunks), "kind": "port",
            "sample": f"{nb} proofs of the same workload, oracle cs_verify = the WHOLE of Verifier::verify (transcript replay "
                      f"+ scalar assembly + 154-term Pippenger MSM), {len(chunks)} processes, {sum(r[1] for r in res):.1f} s CPU; "
                      "the like-for-like GPU figure is `with_device_transcript`, not `value` (whose challenges are inputs)"}
    n1 = min(nb, 128)
    with mp.get_context("fork").Pool(1) as pool:
        pool.map(_cpu_verify_chunk, [(b"", b"", plen)])
        t0 = time.perf_counter()
        r1 = pool.map(_cpu_verify_chunk, [(wl["proofs"][:n1 * plen], wl["commitments"][:64 * n1], plen)])
        wall1 = time.perf_counter() - t0
    assert r1[0][0] == n1
    one = {"value": n1 / wall1, "unit": "verifications/s", "cores": 1, "kind": "port",
           "sample": f"the first {n1} proofs of the same workload, one process (BASELINE.md's single-thread column)"}
    return allc, one


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters; this parent never
    initialises the GPU) with the rendezvous environment torch.distributed.run would give them, relay rank 0's stdout."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if failed:                      # the other ranks would wait for the dead one in the next collective
                rc = abs(failed[0])
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    sys.exit(max([rc] + [abs(p.returncode or 0) for p in procs]) if rc == 0 else rc)


def _lib_hash():
    h = hashlib.sha256()
    with open(os.path.join(ROOT, "mpc_bulletproof_amd", "libbpgpu.so"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def main():
    # deep step pipelining needs hardware queues (ROCm default: 4) and one stream per context.  24 queues / 20 steps in
    # flight: the best short-burst AND steady-state setting of tools/burst_sweep.sh (20 steps after a device sync
    # 5.9 ms, 1024 steps 4.2 M/s); more than ~22 ACTIVE queues makes bursts erratic (24 / 24: 6 - 50 ms for 20 steps).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    os.environ.setdefault("BPGPU_SINGLE_STREAM", "1")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--window-bits", type=int, default=int(os.environ.get("BPGPU_WINDOW_BITS", "20")),
                    help="window of the resident generator tables: 20 bits = 13 table additions per generator term, a 57 GB "
                         "table for the 130 generators of the 64-bit gadget (16 bits: 16 additions, 4.5 GB)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("BPGPU_INFLIGHT", "20")),
                    help="steps in flight: consecutive steps alternate between this many independent contexts "
                         "(streams + workspaces), so the kernels of several batches overlap on the GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-combined", action="store_true", help="skip the secondary measurements (device transcript, wire format, "
                                                               "combined batch check, H2D-inclusive, single batch)")
    ap.add_argument("--no-prover", action="store_true", help="skip the secondary R1CS prover measurement (N = 1 only)")
    ap.add_argument("--workload-cache", default=None,
                    help="pickle of the generated workload (written if absent); lets a profiled run skip the fork pool")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus} "
              f"or let bench.py start the ranks itself (no WORLD_SIZE in the environment)", file=sys.stderr)
        sys.exit(2)
    nb = a.batch

    # ---- synthetic workload (setup, untimed), generated by the product path itself in a child process forked before
    # this process touches the GPU.  Every rank verifies the same batch (weak scaling): local rank 0 generates it
    # once and publishes it through a file, the other ranks wait for the file.  The CPU baseline (the oracle's
    # restatement) runs in fork pools on rank 0, also before any GPU initialisation here.
    import pickle
    import tempfile
    ncpu = max(1, min(os.cpu_count() or 1, 32))
    seed0 = 0xB0117E7
    cache = f"{a.workload_cache}.{nb}" if a.workload_cache else os.path.join(
        tempfile.gettempdir(), f"bpgpu_workload_{os.environ.get('MASTER_PORT', 'solo')}_{os.getppid()}_{nb}.pkl")
    if not os.path.exists(cache) and local_rank == 0:
        child = mp.get_context("fork").Process(target=_gen_workload, args=(cache, nb, seed0))
        child.start()
        child.join()
        if child.exitcode != 0:
            raise RuntimeError(f"workload generation failed (exit code {child.exitcode}): the HIP path is required, "
                               "there is no CPU fallback")
    deadline = time.time() + 1200
    while not os.path.exists(cache):
        if time.time() > deadline:
            raise RuntimeError("workload file never appeared")
        time.sleep(0.5)
    with open(cache, "rb") as f:
        wl = pickle.load(f)
    cpu = cpu1 = None
    if rank == 0 and not a.no_cpu_baseline:
        cpu, cpu1 = _cpu_baseline(wl, nb, ncpu)
    n1, n2, k, m = wl["dims"]
    rp, kind, idx, coeff = wl["csr"]
    pts, sc, ch = wl["points"], wl["scalars"], wl["challenges"]

    # ---- GPU
    import torch
    import torch.distributed as dist
    import mpc_bulletproof_amd as mb
    if world > 1:
        if os.environ.get("BPGPU_BENCH_REHEARSAL"):
            # control-flow rehearsal of the multi-rank path on a ONE-GPU box: all ranks share device 0, gloo instead
            # of RCCL (NCCL refuses two ranks on one device).  Not a measurement.
            dev = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            dev = local_rank
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dev = local_rank
    ctxs = [mb.BpGpu(dev) for _ in range(max(1, a.inflight))]
    gpu = ctxs[0]
    circ = gpu.circuit_create(rp, kind, idx, coeff, n1 + n2, m)
    gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], a.window_bits)
    d_pts, d_sc, d_ch = gpu.to_device(pts), gpu.to_device(sc), gpu.to_device(ch)
    d_oks = [gpu.malloc(4 * nb) for _ in ctxs]
    counter = [0]
    all_ok = (1).to_bytes(4, "little") * nb

    sample_period = [0]            # > 0: step number s carries HIP-event pairs when s % sample_period == 0 (see below)
    prof_state = [False] * len(ctxs)

    def step():
        n = counter[0]
        i = n % len(ctxs)
        counter[0] += 1
        if sample_period[0]:
            on = n % sample_period[0] == 0
            if on != prof_state[i]:
                ctxs[i].profile_enable(on)
                prof_state[i] = on
        ctxs[i].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[i])

    def sync_all():
        # one device-wide synchronisation (the contexts' streams are ordinary blocking HIP streams of this device);
        # a hipStreamSynchronize per context costs ~0.15 ms each on an idle stream
        torch.cuda.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, before=None):
        """EXACTLY `steps` calls of fn(i) between two fences; max over ranks.  The GPU is handed over BUSY: a burst of one
        un-timed call per context runs right up to the opening fence (result checks, profile reads and buffer setup before a
        leg leave the device idle for milliseconds, and an idle device starts the first kernels of a short run slowly);
        `before()` runs between that burst's completion and the fence (switches event timing on: a flag, no GPU work)."""
        for i in range(len(ctxs)):
            fn(i)
        sync_all()
        if before:
            before()
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        sync_all()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            from mpc_bulletproof_amd import sharding
            dt = sharding.max_over_ranks(dt)
        return dt

    # Per-kernel HIP-event timing (event pairs on the stream each kernel is launched on, recycled from a per-context pool).
    # In the timed region ONE STEP in `inflight + 1` carries events, so the sampled steps rotate over the contexts (an event
    # pair per launch on every context costs a short run a quarter of its throughput: ~12 barrier packets per step in each
    # queue; always the same context made that context lag, and the closing fence waited for it: 3-5 % of a 2 048-step run);
    # right after it the same K steps are REPLAYED with event pairs around every launch of every context: `roofline` is
    # computed from the replay (>= K launches per kernel) and quotes the timed region's own sample beside it.
    # BPGPU_PROF_EVERY overrides the sampling period.
    prof_every = max(1, int(os.environ.get("BPGPU_PROF_EVERY", str(len(ctxs) + 1))))
    noprof = bool(os.environ.get("BPGPU_BENCH_NOPROF"))
    def sampling_on():
        # in the timed region a sampled step times its dominant kernel only (bpgpu_profile_select): one event pair = two
        # barrier packets in that step's queue instead of twelve (a fully timed step in a 20-step run cost the run 4-5 %)
        for c in ctxs:
            c.profile_select(["verify_back"])
        sample_period[0] = 0 if noprof else prof_every

    def sampling_off():
        sample_period[0] = 0
        for i, c in enumerate(ctxs):
            c.profile_enable(False)
            c.profile_select(None)
            prof_state[i] = False

    for _ in range(max(a.warmup, len(ctxs))):
        step()
    sync_all()
    for c, d in zip(ctxs, d_oks):
        assert c.download(d, 4 * nb) == all_ok and c.input_flag() == 0, "GPU verification disagrees"
    # the checks above leave the GPU idle for milliseconds: the warm-up keeps submitting steps (untimed) for
    # BPGPU_WARM_SECONDS of wall time; timed() then hands the device over busy
    warm_s = float(os.environ.get("BPGPU_WARM_SECONDS", "0.3"))
    sampling_on()      # the warm-up also fills the sampled contexts' event pools: no hipEventCreate inside the timed region
    tw = time.perf_counter()
    while time.perf_counter() - tw < warm_s:
        for _ in range(2 * len(ctxs)):
            step()
    sync_all()
    warm_sample = {}
    for c in ctxs:           # the warm-up's sample (steady state, one step in 21) is kept; the events go back to the pools
        for name, (ms, cnt) in c.profile_read().items():
            pm, pc = warm_sample.get(name, (0.0, 0))
            warm_sample[name] = (pm + ms, pc + cnt)
    sampling_off()
    counter[0] = 0
    dt = timed(lambda i: step(), a.steps, before=sampling_on)
    sample_period[0] = 0      # (the per-context flags are cleared below, after the sample is read)

    def collect():
        acc = {}
        for c in ctxs:
            for name, (ms, cnt) in c.profile_read().items():
                pm, pc = acc.get(name, (0.0, 0))
                acc[name] = (pm + ms, pc + cnt)
        return acc

    prof_sample = collect()
    for c, d in zip(ctxs, d_oks):
        assert c.download(d, 4 * nb) == all_ok
    # instrumented replay of the same K steps (capped at 256): every launch of every context timed
    prof, replay_dt, replay_steps = prof_sample, None, 0
    if not noprof:
        def all_on():
            for c in ctxs:
                c.profile_enable(True)

        sampling_off()
        counter[0] = 0
        replay_steps = min(a.steps, 256)
        replay_dt = timed(lambda i: step(), replay_steps, before=all_on)
        prof = collect()
    for c in ctxs:
        c.profile_enable(False)

    fs = wire = comb = h2d = single = None
    if not a.no_combined:
        # ---- secondary: ONE batch at a time (no pipelining): the latency of a batch's kernel chain
        lat = []
        ctxs[0].set_latency_mode(True)      # the context-level hint for un-pipelined callers (include/bpgpu.h)
        for _ in range(9):
            sync_all()
            t0 = time.perf_counter()
            ctxs[0].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[0])
            ctxs[0].sync()
            lat.append(time.perf_counter() - t0)
        ctxs[0].set_latency_mode(False)
        assert ctxs[0].download(d_oks[0], 4 * nb) == all_ok
        lat = sorted(lat[2:])
        single = {"value": nb / lat[len(lat) // 2], "unit": "verifications/s", "ms_per_batch": lat[len(lat) // 2] * 1e3,
                  "min_ms": lat[0] * 1e3, "note": "one un-pipelined 1024-proof batch on an idle GPU, submit to host-visible completion "
                                                  "(median of 7, context in latency mode: bpgpu_set_latency_mode): front | scalars | windows | groups | back | verdict"}

        # ---- secondary: SURVEY 8d's metric as written -- the proof points, proof scalars and challenges of every step are
        # uploaded from page-locked host memory inside the timed region (asynchronous copies on the step's stream)
        packed = pts + sc + ch                      # one page-locked staging buffer, one asynchronous copy per step
        h_in = mb.lib.host_alloc(len(packed), packed)
        d_in = [c.malloc(len(packed)) for c in ctxs]
        import ctypes as C_

        def hstep(i):
            j = i % len(ctxs)
            c, dp = ctxs[j], d_in[j]
            c.upload_async(dp, h_in, len(packed))
            c.r1cs_verify_batch_dev(gens, circ, nb, n1, k, dp, C_.c_void_p(dp.value + len(pts)), C_.c_void_p(dp.value + len(pts) + len(sc)),
                                    d_oks[j])

        for i in range(len(ctxs)):
            hstep(i)
        sync_all()
        for c, d in zip(ctxs, d_oks):
            assert c.download(d, 4 * nb) == all_ok
        hdt = timed(hstep, a.steps)
        h2d = {"value": world * nb * a.steps / hdt, "unit": "verifications/s", "ms_per_step": hdt / a.steps * 1e3,
               "bytes_per_step": len(packed),
               "note": "as `value`, plus the upload of every step's proof points, proof scalars and challenges from page-locked "
                       "host memory inside the timed region (SURVEY 8d: 'incl. H2D of proof scalars + points'); never the headline"}

        # ---- secondary: the same per-proof verification with the Fiat-Shamir transcript replayed on the device
        # (SURVEY 8f N1): inputs are the proofs + one 32-byte initial chain state per proof, no host challenges
        d_init = gpu.to_device(wl["init_state"] * nb)

        def fstep(i):
            j = i % len(ctxs)
            ctxs[j].r1cs_verify_batch_fs_dev(gens, circ, nb, n1, k, d_init, d_pts, d_sc, d_oks[j])

        for i in range(len(ctxs)):
            fstep(i)
        sync_all()
        for c, d in zip(ctxs, d_oks):
            assert c.download(d, 4 * nb) == all_ok
        fdt = timed(fstep, a.steps)
        fs = {"value": world * nb * a.steps / fdt, "unit": "verifications/s", "ms_per_step": fdt / a.steps * 1e3,
              "note": "whole Verifier::verify incl. the transcript replay (keccak256 chain, hash_to_scalar) on the GPU; "
                      "per-proof accept bits.  This is the figure to hold against cpu_baseline (same work)"}

        # ---- secondary: the same proofs taken in the reference's WIRE format (SURVEY 8f N3 + N1): unpack, decompress the
        # 25 points of every proof (a square root in F_p each), transcript, verification -- all on the device
        d_wp, d_wc = gpu.to_device(wl["wire_proofs"]), gpu.to_device(wl["wire_commitments"])

        def wstep(i):
            j = i % len(ctxs)
            ctxs[j].r1cs_verify_batch_wire_dev(gens, circ, nb, n1, wl["wire_len"], d_wp, d_wc, d_init, d_oks[j])

        for i in range(len(ctxs)):
            wstep(i)
        sync_all()
        for c, d in zip(ctxs, d_oks):
            assert c.download(d, 4 * nb) == all_ok
        wdt = timed(wstep, a.steps)
        wire = {"value": world * nb * a.steps / wdt, "unit": "verifications/s", "ms_per_step": wdt / a.steps * 1e3,
                "note": f"from {wl['wire_len']}-byte wire-format proofs + 32-byte compressed commitments: R1CSProof::from_bytes, point "
                        "decompression, transcript replay and verification on the GPU; per-proof accept bits"}

        # ---- secondary: combined batch check (BASELINE configs[1] 'single big MSM': sum_p rho_p * check_p as ONE point per
        # GPU; RCCL all-gather of the 64-byte partials + local add when n_gpus > 1).
        import random
        rnd = random.Random(0xC0B1 + rank)   # verifier-chosen weights: any scalars < 2^250 < n
        d_rho = gpu.to_device(b"".join(rnd.getrandbits(250).to_bytes(32, "little") for _ in range(nb)))
        d_parts = [gpu.malloc(64) for _ in ctxs]

        def cstep(i):
            ctxs[i % len(ctxs)].r1cs_verify_combined_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_rho, d_parts[i % len(ctxs)])

        for i in range(len(ctxs)):
            cstep(i)
        sync_all()
        for c in ctxs:
            c.profile_read()

        def comb_on():
            for i, c in enumerate(ctxs):
                c.profile_enable(not noprof and i % prof_every == 0)

        cdt = timed(cstep, a.steps, before=comb_on)
        cprof = {}
        for c in ctxs:
            for name, (ms, cnt) in c.profile_read().items():
                if name.startswith("combined") and cnt:
                    pm, pc = cprof.get(name, (0.0, 0))
                    cprof[name] = (pm + ms, pc + cnt)
            c.profile_enable(False)
        part = ctxs[0].download(d_parts[0], 64)
        if world > 1:
            from mpc_bulletproof_amd import sharding
            part = sharding.combine_partial_points(part, gpu.points_sum)
        assert part == bytes(64), "combined batch check must be the identity for valid proofs"
        comb = {"value": world * nb * a.steps / cdt, "unit": "verifications/s", "ms_per_step": cdt / a.steps * 1e3,
                "vs_per_proof_value": (world * nb * a.steps / cdt) / (world * nb * a.steps / dt),
                "stage_ms_per_step": {n_: v[0] / max(v[1], 1) for n_, v in cprof.items()},
                "note": "sum_p rho_p*mega_check_p == identity (single accept bit per batch; one fixed-base MSM over the 130 "
                        f"generators + one {nb * (11 + m + 2 * k)}-term bucket-method MSM per GPU; partial points all-gathered "
                        "over RCCL when n_gpus > 1)"}

    # ---- secondary: the other half of BASELINE.json's metric, R1CS constraints/s of the prover, on configs[2]'s
    # shape: 256 provers in lock-step, each range-proving 16 x 64-bit values in one constraint system (n = 1024
    # multipliers, q = 2064 constraints), through the C++ host mirror (wall clock incl. circuit building,
    # transcripts and packing on the host).  tools/bench_prove.py checks the proof bytes against the oracle.
    prove = None
    if world == 1 and not a.no_prover:
        import ctypes as C
        host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
        pnb, nvals = 256, 16
        pq, pn = nvals * (2 * N_BITS + 1), nvals * N_BITS
        vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(pnb) for i in range(nvals)]
        arr = (C.c_uint64 * len(vals))(*vals)
        lab = (C.c_uint8 * len(LABEL)).from_buffer_copy(LABEL)
        pout, pcom, plen_ = (C.c_uint8 * (pnb * 4096))(), (C.c_uint8 * (pnb * nvals * 64))(), C.c_size_t(0)
        best = None
        for rep in range(6):
            t0 = time.perf_counter()
            rc = host.bph_range_prove_batch(C.c_size_t(pnb), C.c_size_t(nvals), C.c_size_t(N_BITS), lab, C.c_size_t(len(LABEL)),
                                            arr, C.c_uint64((1 << 64) - 1), C.c_size_t(pn), pout, C.byref(plen_), pcom)   # all ones = OsRng
            pdt = time.perf_counter() - t0
            assert rc == 0, f"bph_range_prove_batch rc={rc}"
            best = pdt if best is None or (rep and pdt < best) else best
        prove = {"value": pnb * pq / best, "unit": "R1CS constraints/s", "proofs_per_s": pnb / best, "ms_per_batch": best * 1e3,
                 "workload": f"{pnb} provers x ({nvals} x 64-bit range gadgets in one constraint system: n = {pn}, q = {pq}, m = {nvals})",
                 "note": "wall clock of Prover::prove_batch incl. host circuit building, transcripts and packing; blinding factors from the "
                         "default RNG (OsRng: getrandom(2)-keyed Keccak sponge), best of batches 2-6 (the 1st builds the generator tables; the host phases share the box's cores with the other tenants)"}

    if rank == 0:
        nvar = 11 + m + 2 * k
        nterms = 13 + m + 2 * (1 << k) + 2 * k
        W = 252 // a.window_bits + 1
        # The kernels of the window-parallel chain that consume algorithmic bytes (96 B per MSM term = 64 B point +
        # 32 B scalar) and what each reads of them; the dominant kernel = the one with the largest summed duration.
        lpm = 16 if nb >= 1024 else 32
        names = {"verify_back": f"k_verify_back<{a.window_bits},{lpm}>", "verify_front": "k_verify_front<4>",
                 "verify_windows": "k_verify_windows", "verify_scalars": "k_verify_scalars"}
        bytes_per_proof = {"verify_back": (nterms - nvar) * 96, "verify_front": nvar * 64, "verify_windows": nvar * 32,
                           "verify_scalars": (6 + k + 5) * 32 + nterms * 32}
        timed_names = [n_ for n_ in names if prof.get(n_, (0.0, 0))[1] > 0]
        pmc = None
        if os.path.exists(PMC_FILE):
            with open(PMC_FILE) as f:
                pmc = json.load(f)
        lib_hash = _lib_hash()
        roof = None
        if timed_names:
            dom = max(timed_names, key=lambda n_: prof[n_][0])
            # The kernel's launches overlap with those of the other steps in flight, so its "duration" depends on how many: a
            # 20-step burst holds twenty of them at once (replay below), the steady state of the warm-up and of a long timed
            # region about five.  The roofline line uses what `rocprofv3 --kernel-trace --stats` of this same command averages
            # over -- every phase of the run in proportion -- i.e. the one-step-in-21 sample of the warm-up AND the timed region;
            # the every-launch replay of the timed steps is quoted beside it.
            pooled_ms = warm_sample.get(dom, (0.0, 0))[0] + prof_sample.get(dom, (0.0, 0))[0]
            pooled_cnt = warm_sample.get(dom, (0.0, 0))[1] + prof_sample.get(dom, (0.0, 0))[1]
            rep_ms, rep_cnt = prof[dom]
            ms, cnt = (pooled_ms, pooled_cnt) if pooled_cnt >= 8 else (rep_ms, rep_cnt)
            avg_s = ms / cnt / 1e3
            alg_bytes = nb * bytes_per_proof[dom]
            achieved = alg_bytes / avg_s / 1e9
            traffic = (pmc or {}).get("traffic_bytes_per_launch", {}).get(dom)
            roof = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_s * 1e3,
                    "algorithmic_bytes_per_launch": alg_bytes, "launches": cnt,
                    "timed": (f"HIP-event pairs on the launch stream around this kernel in one step of {prof_every} (rotating over the "
                              f"contexts) throughout the {warm_s:.1f} s warm-up and the timed region" if pooled_cnt >= 8 else
                              "every launch of the replay"),
                    "burst_replay": ({"avg_launch_ms": rep_ms / rep_cnt, "launches": rep_cnt, "ms_per_step": replay_dt / replay_steps * 1e3,
                                      "what": f"the same {replay_steps} steps again, HIP-event pairs around every launch of every context"}
                                     if replay_dt and rep_cnt else None),
                    "under_rocprofv3": ((pmc or {}).get("profiled_run_bench20") if (pmc or {}).get("lib_sha256_16") == lib_hash else None),
                    "timed_region_sample": ({"avg_launch_ms": prof_sample[dom][0] / prof_sample[dom][1], "launches": prof_sample[dom][1],
                                             "steps_sampled": f"1 in {prof_every} (rotating over the contexts)"} if prof_sample.get(dom, (0, 0))[1] else None),
                    "note": "achieved = algorithmic bytes of the dominant kernel / its average launch duration (HIP events on the "
                            "launch stream).  Twenty steps are in flight, so a launch shares the chip with its neighbours and lasts "
                            "longer than alone (0.50 ms) -- and longer than under rocprofv3, whose tracing thins the overlap: "
                            "`under_rocprofv3` holds that run's rocprofv3 and HIP-event averages (they agree with each other).  The "
                            "path is VALU-integer bound, not HBM bound: 252-bit modular arithmetic spends ~1 650 instructions per 96 "
                            "algorithmic bytes; see roofline_valu_issue"}
        step_s = dt / a.steps
        # integer roofline: algorithmic F_p multiplications x 94 limb MADs each (csrc/fe29.cuh), per step.  Per non-identity proof
        # point 7 table additions + 60 window additions (mixed, 11 mul), one inversion per 4 points (~310), per proof 252
        # doublings (9) + 64 additions (16) in the Horner passes; fixed-base: one mixed addition per (generator, window) + the
        # 16-lane butterfly.  A_I2, A_O2, S2 are the identity in 1-phase proofs and are skipped.
        fp_var = nb * ((nvar - 3) * (7 + 60) * 11 + ((nvar + 3) // 4) * 310 + 252 * 9 + 64 * 16)
        fp_fixed = nb * ((nterms - nvar) * W * 11 + 15 * 16)
        valu = None
        if pmc and nb == 1024 and a.window_bits == 20:
            instr = pmc["valu_wave_instr_per_step_1024"]
            valu = {"bound": "VALU issue slots (mix-weighted)", "achieved": instr / step_s, "peak": VALU_ISSUE_PEAK_MIX,
                    "unit": "wave-instr/s", "frac": instr / step_s / VALU_ISSUE_PEAK_MIX,
                    "valu_wave_instr_per_step": instr, "measured_on_lib_sha256_16": pmc.get("lib_sha256_16"),
                    "this_lib_sha256_16": lib_hash, "binary_matches": pmc.get("lib_sha256_16") == lib_hash,
                    "note": "instructions per step: rocprofv3 --pmc SQ_INSTS_VALU of a solo run (" + pmc.get("source", "profiles/") +
                            "); peak = 1024 SIMDs x 2.4 GHz / 4.25 cycles per wave64 instruction, the issue rate of THIS instruction "
                            "mix (74 % v_mad_i64_i32) in the doubling / addition micro-benchmarks"}
        out = {
            "metric": "range-proof verifications/sec (64-bit, m=1)",
            "value": world * nb * a.steps / dt, "unit": "verifications/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 252-bit prime fields)", "data": "synthetic",
            "config": {"workload": f"batch verify {nb} x 64-bit range-gadget R1CS proofs (m=1, n=64, 154-term "
                                   f"mega_check MSM per proof, per-proof accept bits) per GPU",
                       "window_bits": a.window_bits, "proofs_per_step_per_gpu": nb, "steps_in_flight": len(ctxs),
                       "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "untimed_before_the_timed_region": f"max(warmup, steps_in_flight) steps, result check, {warm_s:.1f} s of further untimed "
                                                          "steps (clocks, event pools), one hand-over step per context"},
            "roofline": roof,
            "roofline_int": {"bound": "valu_int (v_mad_u64_u32)", "scope": "variable-base + fixed-base halves of one step's mega_check MSMs / wall time per step",
                             "achieved": (fp_var + fp_fixed) * 94 / step_s / 1e12, "peak": MAD_PEAK_TOPS, "unit": "Tmad/s",
                             "frac": (fp_var + fp_fixed) * 94 / step_s / 1e12 / MAD_PEAK_TOPS},
            "roofline_valu_issue": valu,
            "kernel_ms_per_launch": {n_: (v[0] / v[1]) for n_, v in prof.items() if v[1]},
            "kernel_launches_timed": {n_: v[1] for n_, v in prof.items() if v[1]},
            "cpu_baseline": cpu,
            "cpu_baseline_1t": cpu1,
            "single_batch": single,
            "h2d_inclusive": h2d,
            "with_device_transcript": fs,
            "from_wire_format": wire,
            "combined_batch_check": comb,
            "r1cs_prove": prove,
            "range_prove": ({"value": nb / wl["prove_seconds"], "unit": "64-bit range proofs/s", "ms_per_batch": wl["prove_seconds"] * 1e3,
                             "note": f"the {nb} proofs of this workload, proved in lock-step by the GPU prover while it was generated "
                                     "(wall clock incl. host circuit building and transcripts)"} if wl.get("prove_seconds") else None),
        }
        if cpu is None:
            out["cpu_baseline_reason"] = "--no-cpu-baseline"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gpu.gens_destroy(gens)
    gpu.circuit_destroy(circ)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
