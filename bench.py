#!/usr/bin/env python3
"""bench.py -- tCG Hess-vec throughput of the device-resident RTR/tCG hot path.

Workload (BASELINE.json configs[1]): MaxCut SDP of Gset G81 (n = 20000, 92 644 stored
nonzeros in C = -L/4) through the ManiSDP_onlyunitdiag path, oblique manifold, p = 32.
A *step* is one complete ``trustregions(problem, Y, opts)`` call (reference defaults
TR_maxiter = 40, TR_maxinner = 100) from the same seeded start point with all problem
data and the factor already resident in HBM.  ``value`` = Hess-vec products executed by
all ranks' job / wall time, i.e. whole-tCG throughput (Hess-vec + every vector update,
retraction and cost evaluation of the solve).

N > 1 (one process per GPU, RCCL): rows of the factor are sharded over the ranks with an
all-gather of the thin n x p direction before every S*U and an all-reduce of the partial
sums; the instance is the same toroidal-grid family scaled to n = 20000 * N rows so the
per-GPU work is fixed ("weak" scaling); value counts 20000-row-equivalent Hess-vecs.  The same K steps are then timed once
more with the halo exchange (option halo_exchange: only the rows a rank's rows of C reference travel, bit-identical results);
both figures are in the line under "row_exchange".  Round 6: `value` at N > 1 is quoted on the cross-rank persistent kernels
(--multi-gpu-path xr, the default: one process group over HIP IPC / peer access, two-level grid reductions, no collective per trip or TR
iteration; "cross_rank_persistent" in the line) where the node offers peer access between the ranks' devices, and on the RCCL leg
--row-exchange names otherwise (fixed by configuration, never best-of); config.multi_gpu_path says which.

One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# every cpu_baseline entry is the build's own CPU restatement of the reference's .m logic: MATLAB / Octave exist neither here
# nor on the GPU box, so the reference's interpreter path itself cannot be timed
PORT_DETAIL = "port (C/OpenMP restatement, oracle/oracle_core.c; MATLAB unavailable)"
PORT_DETAIL_NUMPY = "port (NumPy/SciPy restatement, oracle/manisdp_ref.py; MATLAB unavailable)"


MFMA_F64_PEAK_TFLOPS = 78.6    # MI355X_MICROARCH.md: dense fp64 matrix peak
RIDGE_P = 39                   # dense tall-skinny contraction is HBM bound up to p ~ 4 * (78.6 TF / 8 TB/s)


def secondary_roofline(kernel, us, algo_bytes, algo_flops, pmc_names=(), bound=None, per="Hess-vec"):
    """The roofline object of the headline line for a secondary workload: achieved = algorithmic bytes (or flops) of SURVEY.md
    8(d) / the measured time of one Hess-vec (HIP events over graph replays, steady state); traffic = HBM bytes per Hess-vec of
    the committed rocprofv3 --pmc summary (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), with the file it came from."""
    if bound is None:
        bound = "hbm"
    rec, src = None, None
    for name in pmc_names:
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            rec, src = json.load(open(f)), "profiles/" + name
            break
    traffic = None if rec is None else rec.get("hbm_bytes_per_hessvec", rec.get("hbm_bytes_per_launch"))
    # (a summary carries the hash of the kernel sources it was measured on; one that does not, or whose sources changed since, is marked)
    stale = None if rec is None else (not rec.get("sources_sha16") or rec["sources_sha16"] != sources_sha16(rec.get("sources", [])))
    if bound == "mfma":
        achieved, peak, unit = algo_flops / (us * 1e-6) / 1e12, MFMA_F64_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = algo_bytes / (us * 1e-6) / 1e9, HBM_PEAK_GBS, "GB/s"
    return {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": traffic,
            "traffic_source": src, "traffic_stale": stale, "algorithmic_bytes": algo_bytes, "algorithmic_flops": algo_flops, "kernel": kernel,
            "kernel_us": us, "per": per,
            "frac_hbm_peak": algo_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_mfma_f64_peak": algo_flops / (us * 1e-6) / 1e12 / MFMA_F64_PEAK_TFLOPS}


def sources_sha16(files):
    """Hash of the named source files (paths relative to the repository) -- what a committed counter summary was measured on."""
    import hashlib
    hh = hashlib.sha256()
    for f in files:
        try:
            hh.update(open(os.path.join(ROOT, f), "rb").read())
        except OSError:
            return None
    return hh.hexdigest()[:16] if files else None


def cpu_baseline(C, Y0, budget_s=15.0):
    """Oracle ("port": oracle/oracle_core.c, plain C + OpenMP over rows) timed on the host cores of this
    box on the SAME step (one full RTR call from Y0), repeated until ~budget_s of CPU work is done."""
    from oracle import core
    # all host cores; one untimed call at that count and one at the parity tests' default of 16 decide which the
    # bounded sample runs on -- a row-parallel OpenMP loop over 20000 rows does not always gain from 100+ threads
    ncpu = core.host_cpus()
    trial = {}
    for nt in sorted({ncpu, min(16, ncpu)}):
        core.set_threads(nt)
        t1 = time.time()
        _, st = core.rtr_onlyunitdiag(C, Y0, 40, 100, 1e-8)
        trial[nt] = st.hessvecs / (time.time() - t1)
    core.set_threads(max(trial, key=trial.get))
    t0 = time.time()
    hv = 0
    reps = 0
    while time.time() - t0 < budget_s and reps < 20:
        t1 = time.time()
        _, st = core.rtr_onlyunitdiag(C, Y0, 40, 100, 1e-8)
        hv += st.hessvecs
        reps += 1
        if time.time() - t1 > budget_s / 2:       # one call already uses most of the budget
            break
    dt = time.time() - t0
    return {"value": hv / dt, "unit": "Hess-vec/s", "cores": core.num_threads(), "host_cpus": ncpu,
            "threads_tried_hessvec_per_s": {str(k): v for k, v in trial.items()}, "kind": "port", "kind_detail": PORT_DETAIL,
            "sample": f"{reps} full RTR calls ({hv} Hess-vecs incl. all tCG vector work, retractions and cost "
                      f"evaluations) of the same G81 p={Y0.shape[1]} step in the C/OpenMP oracle on {core.num_threads()} of "
                      f"{ncpu} host CPUs (the faster of all-cores / 16 threads), {dt:.1f} s"}


def cpu_dense_hessvec(n, p, budget_s=3.0):
    """The oracle's dense-C Hess-vec (ManiSDP_onlyunitdiag.m:127-130 restated in NumPy: one dgemm + the projection
    terms) on this box's host cores, same shape, random symmetric C."""
    from oracle import manisdp_ref
    rng = np.random.default_rng(0)
    C = rng.standard_normal((n, n)); C = (C + C.T) / (2.0 * np.sqrt(n))
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    U = rng.standard_normal((n, p))
    manisdp_ref.hessvec_onlyunitdiag(C, Y, U)
    t0 = time.time(); reps = 0
    while time.time() - t0 < budget_s:
        manisdp_ref.hessvec_onlyunitdiag(C, Y, U)
        reps += 1
    dt = (time.time() - t0) / reps
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:                                     # pragma: no cover
        cores = os.cpu_count()
    return {"value": 1.0 / dt, "unit": "Hess-vec/s", "cores": cores, "kind": "port", "kind_detail": PORT_DETAIL_NUMPY,
            "sample": "%d NumPy (BLAS dgemm) Hess-vecs of the oracle, dense C n=%d p=%d" % (reps, n, p)}


def affine_shapes(_lib, problems, with_cpu):
    """Hess-vec of the two affine configurations of BASELINE.json at their sizes, p = 32, with the oracle's closures timed
    beside them on the host cores (bounded sample): config 3 = BQP d = 60 through ManiSDP_unitdiag (n = 1831,
    m = 1 155 281), config 4 = theta-like unit-trace problem (n = 5000, m = 25 139)."""
    gold = os.path.join(ROOT, "tests", "golden")
    out = []

    def cpu(prob, Y, U, label, budget_s=4.0):
        prob.cost(Y); prob.grad(Y)
        if hasattr(prob, "on_accept"):
            prob.on_accept()
        prob.hess(Y, U)
        t0 = time.time(); reps = 0
        while time.time() - t0 < budget_s:
            prob.hess(Y, U); reps += 1
        dt = (time.time() - t0) / reps
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
        except Exception:                                     # pragma: no cover
            cores = os.cpu_count()
        return {"value": 1.0 / dt, "unit": "Hess-vec/s", "cores": cores, "kind": "port", "kind_detail": PORT_DETAIL_NUMPY,
                "sample": "%d Hess-vecs of the oracle's %s closures (NumPy BLAS threads; the SciPy sparse products are serial)" % (reps, label)}

    for name in ("bqp60", "theta5000"):
        try:
            if name == "bqp60":
                Q = np.loadtxt(os.path.join(gold, "bqp_Q_60_1.txt.gz"), delimiter=",")
                e = np.loadtxt(os.path.join(gold, "bqp_e_60_1.txt.gz"), delimiter=",")
                At, b, c, K = problems.bqpmom(60, Q, e)
                kind, label = _lib.KIND_UNITDIAG, "ManiSDP_unitdiag"
            else:
                At, b, c, K = problems.theta_problem(5000, ndraws=50000, seed=1)
                kind, label = _lib.KIND_UNITTRACE, "ManiSDP_unittrace"
            c = np.asarray(c.todense()).ravel() if hasattr(c, "todense") else np.asarray(c, float).ravel()
            b = np.asarray(b.todense()).ravel() if hasattr(b, "todense") else np.asarray(b, float).ravel()
            n, p = int(K["s"]), 32
            rng = np.random.default_rng(0)
            Y = rng.standard_normal((n, 64))
            h = _lib.Handle.affine(kind, At, b, c, n, pcap=64)
            h.set_multipliers(np.zeros(b.size), 1.0)
            sweep = {}
            for pp in (8, 16, 64, 32):                            # Hess-vec at p in {8, 16, 32} (+ 64: VERDICT round 4); p = 32 last = the headline entry
                h.set_point(np.ascontiguousarray(Y[:, :pp] / (np.linalg.norm(Y[:, :pp], axis=1, keepdims=True) if kind == _lib.KIND_UNITDIAG
                                                                else np.linalg.norm(Y[:, :pp]))))
                for _ in range(2):
                    ms, aby, afl = h.bench_hessvec(100)
                sweep["p%d" % pp] = ms * 1e3
            h.set_option("affine_overlap", 1)                     # A/B: 2*eS*U forked onto a second stream (default off: slower)
            for _ in range(2):
                ms1, _, _ = h.bench_hessvec(100)
            h.close()
            ent = {"workload": name, "entry_point": label, "n": n, "m": int(b.size), "nnz_At": int(At.nnz), "p": p, "hessvec_us": ms * 1e3,
                   "hessvec_per_s": 1e3 / ms, "hessvec_us_by_p": sweep, "hessvec_us_two_streams": ms1 * 1e3,
                   "roofline": secondary_roofline(
                       ("affine Hess-vec chain (Gram matrix -> B route A'(A(.)) -> two-matrix contraction -> epilogue)" if name == "bqp60" else
                        "affine Hess-vec chain (contraction with the SDDMM as a side job -> fused sparse A'(w)*Y + sphere epilogue)"), ms * 1e3, aby, afl,
                       ("r6_pmc_%s_p32.json" % name, "r4_pmc_%s_p32.json" % name, "r3_pmc_%s_p32.json" % name))}
            if with_cpu:
                from oracle import manisdp_ref
                U = rng.standard_normal((n, p))
                Y = Y[:, :p] / (np.linalg.norm(Y[:, :p], axis=1, keepdims=True) if kind == _lib.KIND_UNITDIAG else np.linalg.norm(Y[:, :p]))
                prob = (manisdp_ref._UnitDiagProblem if kind == _lib.KIND_UNITDIAG else manisdp_ref._UnitTraceProblem)(At, b, c, n, p)
                prob.y, prob.sigma = np.zeros(b.size), 1.0
                ent["cpu_baseline"] = cpu(prob, Y, U, label)
            out.append(ent)
        except Exception as e:  # noqa: BLE001 -- secondary figures never cost the headline line
            out.append({"workload": name, "error": "%s: %s" % (type(e).__name__, e)})
    return out


def cross_rank_trip(_lib, problems, N=2, p=32):
    """Two in-process ranks on ONE GPU (msdp_comm_init_local), 20 000 rows each (the weak-scaled G81 family of --gpus N): the tCG
    trip of the cross-rank persistent kernel (one combined launch, grid reductions and row exchange through shared uncached
    memory, no collective per trip; msdp_persist.hip XR) against the lock-step chunked trips (one exchange + one all-reduce per
    trip through the in-process stand-in of the communicator).  What a box with one GPU can say about the multi-GPU tCG."""
    import threading
    C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    res = {}
    gid = [9100]

    def run(xp):
        gid[0] += 1
        out, err = [None] * N, [None] * N

        def body(r):
            try:
                h = _lib.Handle.onlyunitdiag(C, pcap=p)
                h.comm_init_local(N, r, gid[0])
                h.set_option("xpersist", xp)
                h.set_point(Y)
                c0 = h.collective_calls()
                t = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
                out[r] = (t, h.tcg_path(), h.collective_calls() - c0)
                h.close()
            except BaseException as e:  # noqa: BLE001
                err[r] = e
        th = [threading.Thread(target=body, args=(r,)) for r in range(N)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        for e in err:
            if e is not None:
                raise e
        return out
    a, b = run(1), run(0)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.set_point(Y)
    one = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
    h.close()
    return {"workload": "toroidal grid MaxCut, %d in-process ranks x 20000 rows on one GPU, p = %d" % (N, p), "n": n, "p": p, "ranks": N,
            "trip_us_cross_rank_persistent": max(q[0] for q in a), "tcg_path": a[0][1], "collective_calls_per_768_trips": a[0][2],
            "trip_us_lockstep_chunks": max(q[0] for q in b), "collective_calls_per_768_trips_lockstep": b[0][2],
            "trip_us_one_unsharded_handle": one,
            "note": "in-process ranks: the members' workgroups (2 x 128) run in ONE launch; members in different processes / on different GPUs: process_rank_trip"}


def process_rank_worker(argv):
    """bench.py --ipc-worker rank N name p out.json: one member of a group of PROCESSES sharing the GPU (msdp_comm_init_ipc)."""
    rank, N, name, p, out = int(argv[0]), int(argv[1]), argv[2], int(argv[3]), argv[4]
    grid_rows = int(argv[5]) if len(argv) > 5 else 100            # grid rows of 200 vertices per rank
    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    C = problems.toroidal_grid_maxcut(grid_rows * N, 200, seed=81)
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    h.comm_init_ipc(N, rank, name)
    h.set_point(Y)
    trip = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)
    res = {}
    for xtail in (1, 0):
        h.set_option("xtail", xtail)
        best, hv, calls = 1e9, 0, 0
        for _ in range(3):
            h.set_point(Y)
            c0 = h.collective_calls()
            t0 = time.perf_counter(); st = h.rtr(opts); dt = time.perf_counter() - t0
            best, hv, calls = min(best, dt), st.hessvecs, h.collective_calls() - c0
        res[xtail] = (best, hv, calls, st.iters)
    path = h.tcg_path()
    # the same with the two-level grid reductions (msdp_psync.h psync2: what ranks on different GPUs run; here both members share the device)
    h.set_option("xtail", 1)
    h.set_option("xr_twolevel", 1)
    h.set_point(Y)
    trip2 = min(h.bench_tcg_trip(256) for _ in range(3)) * 1e3
    best2 = 1e9
    for _ in range(3):
        h.set_point(Y)
        t0 = time.perf_counter(); st2 = h.rtr(opts); best2 = min(best2, time.perf_counter() - t0)
    json.dump({"trip_us": trip, "tcg_path": path, "rtr_seconds": res[1][0], "hessvecs": res[1][1], "collective_calls_per_rtr_call": res[1][2],
               "iters": res[1][3], "rtr_seconds_with_per_iteration_collectives": res[0][0], "collective_calls_with_them": res[0][2],
               "trip_us_two_level": trip2, "rtr_seconds_two_level": best2, "hessvecs_two_level": st2.hessvecs}, open(out, "w"))
    h.close()


def process_rank_trip(N=2, p=32, grid_rows=100):
    """N PROCESSES on ONE GPU (msdp_comm_init_ipc), 20 000 rows each: the group's slot regions live in one fine-grained device
    block of rank 0, every member's exchange buffer (its rows + a slot per foreign row it references) in its own memory, all
    exported / mapped through HIP IPC (the mapping goes over peer access when the ranks own different devices: the same code
    path); the owner of a boundary row stores it into the neighbour's buffer, gathers are local.  Every process launches its own workgroups of the cross-rank persistent tCG and of the cross-rank TR
    tail -- a trustregions() call issues no collective per trip and none per iteration."""
    import shutil, subprocess, tempfile
    name = "/msdp_bench_%d" % os.getpid()
    tmp = tempfile.mkdtemp()
    procs, logs = [], []
    try:
        # (every member's stderr goes to a file of its own: a pipe that nobody drains while the members wait for each other in the
        # group's barrier would stall all of them -- ADVICE round 5)
        for r in range(N):
            log = open(os.path.join(tmp, "r%d.err" % r), "w")
            logs.append(log)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ipc-worker", str(r), str(N), name, str(p), os.path.join(tmp, "r%d.json" % r), str(grid_rows)],
                                          stdout=subprocess.DEVNULL, stderr=log))
        for pr in procs:
            try:
                pr.wait(timeout=240)
            except subprocess.TimeoutExpired:
                pr.kill(); pr.wait()
        for log in logs:
            log.close()
        if any(pr.returncode != 0 for pr in procs):
            errs = [open(os.path.join(tmp, "r%d.err" % r)).read() for r in range(N)]
            raise RuntimeError("a member failed: " + " | ".join(e[-300:] for e in errs if e))
        res = [json.load(open(os.path.join(tmp, "r%d.json" % r))) for r in range(N)]
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill(); pr.wait()
        for log in logs:
            if not log.closed:
                log.close()
        shutil.rmtree(tmp, ignore_errors=True)
        try:
            os.unlink("/dev/shm" + name)          # (rank 0 unlinks it when it closes its handle; this covers a member that was killed)
        except OSError:
            pass
    hv, sec, sec_c = res[0]["hessvecs"], max(q["rtr_seconds"] for q in res), max(q["rtr_seconds_with_per_iteration_collectives"] for q in res)
    return {"workload": "toroidal grid MaxCut, %d process ranks x %d rows on one GPU (HIP IPC), p = %d" % (N, 200 * grid_rows, p), "ranks": N, "p": p,
            "trip_us_cross_rank_persistent": max(q["trip_us"] for q in res), "tcg_path": res[0]["tcg_path"],
            "trustregions_us_per_hessvec": sec * 1e6 / hv, "hessvec_per_s": hv / sec, "hessvecs": hv, "tr_iterations": res[0]["iters"],
            "collective_calls_per_trustregions_call": res[0]["collective_calls_per_rtr_call"],
            "trustregions_us_per_hessvec_with_per_iteration_collectives": sec_c * 1e6 / hv,
            "collective_calls_with_per_iteration_collectives": res[0]["collective_calls_with_them"],
            "two_level_reductions": {"trip_us": max(q["trip_us_two_level"] for q in res),
                                     "trustregions_us_per_hessvec": max(q["rtr_seconds_two_level"] for q in res) * 1e6 / max(res[0]["hessvecs_two_level"], 1),
                                     "hessvecs": res[0]["hessvecs_two_level"],
                                     "note": "the N-GPU form of the reductions (each member over its own grid, the members' sums pushed into every member's block) "
                                             "run by two processes on ONE device; on N devices the second level crosses xGMI"}}


def xr_unavailable(N, local_rank):
    """None when the process-group path can serve this job: every rank on this node, one device each, peer access between them."""
    try:
        import torch
        if int(os.environ.get("LOCAL_WORLD_SIZE", str(N))) != N:
            return "the ranks span more than one node"
        if N > 8:
            return "more than 8 ranks"
        if torch.cuda.device_count() < N:
            return "fewer than %d devices visible" % N
        for q in range(N):
            if q != local_rank and not torch.cuda.can_device_access_peer(local_rank, q):
                return "no peer access between devices %d and %d" % (local_rank, q)
        return None
    except Exception as e:  # noqa: BLE001
        return "%s: %s" % (type(e).__name__, e)


def main():
    # stdout carries exactly ONE line (the JSON result of rank 0).  Native libraries print there too (RCCL's version
    # banner with NCCL_DEBUG=VERSION arrives from C stdio at exit, i.e. after the JSON line), so file descriptor 1 is
    # pointed at stderr for the whole run and the result goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--p", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kkt", action="store_true", help="skip the full G81 solve to KKT 1e-8")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-C (fp64 MFMA) Hess-vec figure")
    ap.add_argument("--no-large-sparse", action="store_true", help="skip the n = 10^6 chunked-trip figure")
    ap.add_argument("--no-xrank", action="store_true", help="skip the two-ranks-on-one-GPU cross-rank persistent trip")
    ap.add_argument("--no-affine", action="store_true", help="skip the Hess-vec figures of the affine configurations (BQP d = 60, theta n = 5000)")
    ap.add_argument("--row-exchange", choices=("allgather", "halo"), default="allgather",
                    help="N > 1: which exchange in front of S*U `value` is quoted on (both legs are timed and reported): the "
                         "all-gather of the whole direction north_star prescribes (default), or the halo exchange")
    ap.add_argument("--multi-gpu-path", choices=("xr", "rccl"), default="xr",
                    help="N > 1: the path `value` is quoted on -- xr: the cross-rank persistent kernels over HIP IPC / xGMI where the node "
                         "offers peer access between the ranks' devices (RCCL lock-step trips otherwise); rccl: the RCCL lock-step trips")
    ap.add_argument("--force-comm", action="store_true",
                    help="diagnostic: run the N = 1 workload through the RCCL code path of the multi-GPU run "
                         "(size-1 communicator: all-gather + all-reduces per trip, chunked tCG)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    N = args.gpus
    if world != N and N > 1:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}: launch with torch.distributed.run")

    dist = None
    if N > 1:
        # A multi-rank run that stops making progress (a collective that one rank never enters) must end by itself with
        # a message instead of sitting in the launcher until an outer time limit: 20 minutes is ten times the expected run
        import threading

        def _stuck():
            sys.stderr.write("bench.py: rank %d made no progress for 1200 s -- giving up\n" % rank)
            sys.stderr.flush()
            os._exit(3)

        _dog = threading.Timer(1200.0, _stuck)
        _dog.daemon = True
        _dog.start()
    if N > 1 or args.force_comm:
        # torch (its bundled HIP runtime + RCCL) must come up BEFORE libmanisdp_hip.so pulls in the system HIP
        # runtime: the other order leaves torch with "No HIP GPUs are available" (seen on the MI355X box)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        torch.cuda.init()
        if N == 1:                               # --force-comm outside a launcher: a one-rank rendezvous of its own
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
                os.environ.setdefault(k, v)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from manisdp_matlab_amd import _lib, problems
    _lib.load()
    _lib.set_device(local_rank)

    p = args.p
    g81 = os.path.join(ROOT, "tests", "golden", "G81.txt.gz")
    if N == 1 and os.path.exists(g81):
        C = problems.maxcut_cost_matrix(g81)
        workload = "Gset G81 MaxCut, ManiSDP_onlyunitdiag RTR call, n=20000, p=%d" % p
        data_kind = "Gset G81 (public instance shipped with the reference; fixture copy)"
    else:
        C = problems.toroidal_grid_maxcut(100 * N, 200, seed=81)
        workload = "G81-family toroidal grid MaxCut (%dx200, +-1 weights), n=%d, p=%d, rows sharded over %d GPUs" % (
            100 * N, 20000 * N, p, N)
        data_kind = "synthetic"
    n = C.shape[0]
    rng = np.random.default_rng(0)
    Y0 = rng.standard_normal((n, p))
    Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)

    h = _lib.Handle.onlyunitdiag(C, pcap=p)
    if N > 1 or args.force_comm:
        import torch
        uid = [_lib.Handle.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        h.comm_init(N, rank, uid[0])
    opts = _lib.default_opts(maxiter=40, maxinner=100, tolgradnorm=1e-8)

    h.set_point(Y0)
    h.point_snapshot()           # the start point stays resident in HBM: every step restarts from this device copy

    def step():
        h.point_restore()
        return h.rtr(opts)

    def sync():
        if N > 1 or args.force_comm:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def join(hh):                                   # a fresh communicator over the same ranks for a fresh handle
        ident = [_lib.Handle.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        hh.comm_init(N, rank, ident[0])

    def allmax(x):
        import torch
        tt = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    hv = 0
    rtr_s = 0.0
    dev_ms = []                                     # the step's device time by HIP events on the library's stream (msdp_debug_last_rtr_device_ms)
    for _ in range(args.steps):
        st = step()
        hv += st.hessvecs
        rtr_s += st.seconds
        dev_ms.append(h.last_rtr_device_ms())
    sync()
    dt = time.perf_counter() - t0
    if N > 1:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Kernel-level numbers, timed with HIP events on the library's stream.
    #  * the S*U (ehess) kernel alone: algorithmic bytes of SURVEY.md 8(d) / average launch time;
    #  * one whole tCG trip (S*U + every vector update + the three reductions).  On one GPU at this size the
    #    trips run inside the persistent kernel k_tcg_persist_obl (working set resident in registers/LDS), which
    #    is where >90% of the step's device time goes: it is the dominant kernel of the roofline object
    #    (profiles/r2_bench_kernel_stats.csv: k_tcg_persist_obl<16,5,3,false> = the 512 timed trips of bench_tcg_trip).
    h.set_point(Y0)
    ms, abytes, aflops = h.bench_hessvec(200)
    trip_ms = h.bench_tcg_trip(512)
    persistent = (N == 1 and h.tcg_path() == 1)
    # round 5: the persistent kernel's default trip has ONE grid reduction (msdp_pipe.h); the two-reduction trip timed beside it
    one_reduction = persistent and h.persist_form() == 2
    trip2_ms = None
    if one_reduction:
        h.set_option("persist_pipe", 0)
        trip2_ms = h.bench_tcg_trip(512)
        h.set_option("persist_pipe", 1)
    hess_achieved = abytes / (ms * 1e-3) / 1e9
    # the algorithmic traffic of the path is that of the S*U (ehess) product, `abytes` per Hess-vec;
    # one tCG trip = one Hess-vec.  For the persistent kernel (one launch = all trips of a solve) bytes and time are
    # quoted per trip: achieved = abytes / trip time.  The streaming three-kernel formulation of a trip would move
    # abytes + ~10 n*p*8 (tCG.m:166-287); that figure is kept as a labelled extra, it is NOT the roofline number.
    trip_achieved = abytes / (trip_ms * 1e-3) / 1e9
    streaming_trip_bytes = abytes + 10.0 * n * p * 8

    def pmc(*names):
        """A committed rocprofv3 --pmc summary (profiles/), with a mark when the kernel's sources changed after it was taken: a summary
        carries the hash of the sources it was measured on (tools/pmc_to_json.py --sources); one without is of unknown generation."""
        for name in names:
            f = os.path.join(ROOT, "profiles", name)
            if N == 1 and p == 32 and os.path.exists(f):
                rec = json.load(open(f))
                rec["_source"] = "profiles/" + name
                rec["_stale"] = not rec.get("sources_sha16") or rec["sources_sha16"] != sources_sha16(rec.get("sources", []))
                return rec
        return None
    # HBM traffic comes from rocprofv3 --pmc passes (their own runs: counters cannot be collected inside a timed run);
    # the line carries the committed summary's value together with the file it was read from
    pm_h = pmc("r6_pmc_hess_g81_p32.json", "r4_pmc_hess_g81_p32.json", "r3_pmc_hess_g81_p32.json")
    pm_t = pmc("r6_pmc_pipe_g81_p32.json", "r5_pmc_pipe_g81_p32.json") if one_reduction else pmc("r6_pmc_persist_g81_p32.json", "r5_pmc_persist_g81_p32.json")
    pm_f = pmc("r6_pmc_fused_g81_p32.json")
    if persistent:
        # The DOMINANT kernel of the timed step: on the fused path ONE launch of k_tcg_pipe_obl<.., FUSE> runs the whole trustregions() call
        # (every trip, retraction, cost evaluation, decision) -- roofline.achieved = algorithmic bytes of that launch (its Hess-vecs x
        # SURVEY.md 8(d)'s bytes per Hess-vec; the 40 cost / gradient evaluations it also makes are not counted) / its duration by HIP
        # events on the library's stream over the timed steps.  `trip_only` = the trip microbenchmark (bench_tcg_trip: the per-iteration
        # instance, 512 trips, exits disabled) that earlier rounds quoted as `frac`; `frac_of_value` = the same bytes x `value` / peak,
        # i.e. with the host side of the step inside.
        hv_step = hv / args.steps
        fused_ms = sum(dev_ms) / len(dev_ms) if dev_ms and min(dev_ms) > 0 else None
        trip_only = {"kernel": "k_tcg_pipe_obl" if one_reduction else "k_tcg_persist_obl", "kernel_us": trip_ms * 1e3, "per": "tCG trip (one Hess-vec)",
                     "achieved": trip_achieved, "frac": trip_achieved / HBM_PEAK_GBS,
                     "traffic": (pm_t or {}).get("hbm_bytes_per_trip"), "traffic_source": (pm_t or {}).get("_source"),
                     "traffic_stale": (pm_t or {}).get("_stale"),
                     "two_reduction_trip_us": None if trip2_ms is None else trip2_ms * 1e3}
        if fused_ms is not None:
            achieved = hv_step * abytes / (fused_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": (pm_f or {}).get("hbm_bytes_per_launch"), "traffic_source": (pm_f or {}).get("_source"),
                        "traffic_stale": (pm_f or {}).get("_stale"),
                        "traffic_note": "replayed from the committed rocprofv3 --pmc summary of the same kernel and workload, not measured in this run",
                        "kernel": ("k_tcg_pipe_obl<.., FUSE>" if one_reduction else "k_tcg_persist_obl<.., FUSE>") + " (one launch = one trustregions() call)",
                        "kernel_us": fused_ms * 1e3, "per": "launch (%d Hess-vecs)" % round(hv_step),
                        "algorithmic_bytes_per_launch": hv_step * abytes, "algorithmic_bytes_per_hessvec": abytes,
                        "frac_is": "algorithmic bytes of the fused launch / its HIP-event duration / peak",
                        "frac_of_value": hv / dt * abytes / 1e9 / HBM_PEAK_GBS,
                        "us_per_hessvec_in_the_launch": fused_ms * 1e3 / hv_step,
                        "trip_only": trip_only}
        else:
            roofline = {"bound": "hbm", "achieved": trip_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": trip_achieved / HBM_PEAK_GBS,
                        "traffic": trip_only["traffic"], "traffic_source": trip_only["traffic_source"], "traffic_stale": trip_only["traffic_stale"],
                        "kernel": trip_only["kernel"], "kernel_us": trip_ms * 1e3, "per": "tCG trip (one Hess-vec)",
                        "algorithmic_bytes_per_launch": abytes, "frac_is": "algorithmic bytes of one trip / trip time of the trip microbenchmark / peak",
                        "frac_of_value": hv / dt * abytes / 1e9 / HBM_PEAK_GBS}
        roofline.update({
                    "streaming_formulation_bytes_per_trip": streaming_trip_bytes,
                    "note": "the working set is register / LDS resident and a trip is LATENCY, not HBM (n*p*8 = 5 MB per vector: the fraction of the "
                            "HBM roofline is low by construction): one grid reduction per trip (msdp_pipe.h) -- profiles/r6_fused_timeline_p32.md: "
                            "products + partial sums 1.4 us (fp64 VALU at two waves per SIMD), the reduction chain 3.3 us (wave butterfly, store "
                            "drain + barrier, post + back-off, poll, sums), alpha .. new direction 0.5 us; a TR iteration costs 13 us outside "
                            "its trips (round 5: 14; retraction, slot-line barrier, gather of the proposal rows through the fine-grained exchange "
                            "buffer, the iteration's reduction, decision with an LDS role swap)"})
    else:
        roofline = {"bound": "hbm", "achieved": hess_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hess_achieved / HBM_PEAK_GBS, "traffic": (pm_h or {}).get("hbm_bytes_per_launch"),
                    "traffic_source": (pm_h or {}).get("_source"), "traffic_stale": (pm_h or {}).get("_stale"),
                    "kernel": "k_hess_ell_obl", "kernel_us": ms * 1e3, "algorithmic_bytes_per_launch": abytes}
    hess_kernel = {"kernel": "k_hess_ell_obl", "kernel_us": ms * 1e3, "algorithmic_bytes_per_launch": abytes,
                   "achieved_GBps": hess_achieved, "frac_of_hbm_peak": hess_achieved / HBM_PEAK_GBS,
                   "traffic": (pm_h or {}).get("hbm_bytes_per_launch"), "traffic_source": (pm_h or {}).get("_source"),
                   "traffic_stale": (pm_h or {}).get("_stale")}

    out = {
        "metric": "tCG Hess-vec prods/sec (n,p), G81 MaxCut",
        "value": hv * N / dt,
        "unit": "Hess-vec/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": data_kind,
        "config": {"workload": workload, "n": n, "p": p, "nnz_C": int(C.nnz),
                   "TR_maxiter": 40, "TR_maxinner": 100, "hessvecs_per_step": hv / args.steps,
                   "parallelism": "rows%d" % N,
                   "tcg_path": ("persistent single-launch kernel, one grid reduction per trip" if one_reduction else "persistent single-launch kernel") if persistent else
                               "chunked hipGraph, two launches per trip (linear-product trip, msdp_trip1.hip)"},
        "roofline": roofline,
        "hessvec_kernel": hess_kernel,
        "tcg_trip_us": trip_ms * 1e3,
        "hessvec_per_s_kernel_only": 1e3 / ms,           # stand-alone S*U kernel back to back (both figures)
        "hessvec_per_s_whole_tcg": 1e3 / trip_ms,        # one Hess-vec + all vector work and reductions of a tCG trip
        "hessvec_per_s_in_rtr": hv / rtr_s if rtr_s > 0 else None,
    }
    if N == 1 and rank == 0 and not args.force_comm:
        # the Hess-vec microbenchmark at p in {8, 16, 32} -- stand-alone S*U kernel and whole tCG trip
        sweep = {}
        for pp in (8, 16, 32):
            try:
                hp = _lib.Handle.onlyunitdiag(C, pcap=pp)
                hp.set_point(np.ascontiguousarray(Y0[:, :pp] / np.linalg.norm(Y0[:, :pp], axis=1, keepdims=True)) if pp <= p else
                             (lambda Z: Z / np.linalg.norm(Z, axis=1, keepdims=True))(np.random.default_rng(pp).standard_normal((n, pp))))
                hms, hby, _ = hp.bench_hessvec(200)
                tms = hp.bench_tcg_trip(512)
                sweep["p%d" % pp] = {"hessvec_kernel_us": hms * 1e3, "hessvec_kernel_GBps": hby / (hms * 1e-3) / 1e9, "tcg_trip_us": tms * 1e3,
                                     "hessvec_per_s_whole_tcg": 1e3 / tms}
                hp.close()
            except Exception as e:  # noqa: BLE001
                sweep["p%d" % pp] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["hessvec_by_p"] = sweep
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(C, Y0) if N == 1 else None
    if not args.no_kkt and N == 1 and rank == 0 and data_kind.startswith("Gset"):
        # second half of the metric: wall-clock to KKT 1e-8 on G81 with the reference's own example
        # setting options.p0 = 40 (example/example_maxcut.m:32), everything else default
        from manisdp_matlab_amd import solvers
        # two solves, each on a fresh handle: the first pays the process's one-time costs (code objects of the escape
        # kernels, first allocation of the block eigen-solver's 0.35-GB workspace); the second is the figure, like the
        # warmed-up steps above
        t1 = time.perf_counter()
        solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
        first_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        _, obj, data = solvers.ManiSDP_onlyunitdiag(C, {"p0": 40}, verbose=False)
        out["g81_kkt"] = {"seconds_to_dinf_1e-8": time.perf_counter() - t1, "first_solve_in_process_seconds": first_s,
                          "obj": obj, "dinf": data["dinf"],
                          "status": data["status"], "AL_iters": data["iters"], "hessvecs": data["hessvecs"],
                          "rtr_seconds": data["rtr_seconds"], "escape_seconds": data["eig_seconds"],
                          "independent_lambda_min_checks": data.get("eig_verifications", 0),
                          "options": {"p0": 40},
                          "note": "dinf is confirmed by a plain Lanczos run (no deflation, nothing reused from earlier calls) before the solve "
                                  "stops; that run is inside seconds_to_dinf_1e-8 and escape_seconds"}
        # BASELINE config 0 beside it: Gset G1 (n = 800, rows of ~48 entries: CSR rows in the persistent tCG), the reference's default options
        g1 = os.path.join(ROOT, "tests", "golden", "G1.txt.gz")
        if os.path.exists(g1):
            try:
                C1 = problems.maxcut_cost_matrix(g1)
                solvers.ManiSDP_onlyunitdiag(C1, {}, verbose=False)
                t1 = time.perf_counter()
                _, obj1, d1 = solvers.ManiSDP_onlyunitdiag(C1, {}, verbose=False)
                sec1 = time.perf_counter() - t1
                h1 = _lib.Handle.onlyunitdiag(C1, pcap=16)
                r1 = np.random.default_rng(0)
                Y1 = r1.standard_normal((C1.shape[0], 16)); Y1 /= np.linalg.norm(Y1, axis=1, keepdims=True)
                h1.set_point(Y1)
                trip1 = min(h1.bench_tcg_trip(256) for _ in range(3)) * 1e3
                h1.close()
                out["g1_config0"] = {"seconds_to_dinf_1e-8": sec1, "obj": obj1, "dinf": d1["dinf"], "status": d1["status"], "AL_iters": d1["iters"],
                                     "hessvecs": d1["hessvecs"], "rtr_seconds": d1["rtr_seconds"], "escape_seconds": d1["eig_seconds"],
                                     "independent_lambda_min_checks": d1.get("eig_verifications", 0), "tcg_trip_us_p16": trip1,
                                     "note": "default options (p0 = 2); the saddle escape runs on the device from n = 601 on (options['eig'] = 'host': the "
                                             "reference's own eig(full(S)) on the host, 0.12 s for this solve)"}
            except Exception as e:  # noqa: BLE001 -- secondary figure
                out["g1_config0"] = {"error": "%s: %s" % (type(e).__name__, e)}
    h.close()
    _lib.release_cache()         # the parked escape workspace: the dense shapes below allocate up to 160 GB of their own
    if not args.no_dense and N == 1 and rank == 0 and not args.force_comm:
        # The dense tall-skinny contraction S*U is the one place the path uses the matrix cores (north_star): report its
        # fp64 MFMA and HBM fractions on the dense-C shapes of BASELINE configs 4 / 5 (synthetic symmetric C, seed 0).
        MFMA_F64_TFLOPS = 78.6                      # MI355X_MICROARCH.md: dense fp64 matrix peak
        dense = []
        # n = 20000, p = 32 is the shape north_star's ">= 60 % of the HBM roofline on the S*Y / ehess kernel" names
        for (dn, dp) in ((5000, 32), (5000, 64), (20000, 16), (20000, 32), (20000, 64)):
            hd = _lib.Handle.dense_synthetic(dn, 0, pcap=dp)
            rngd = np.random.default_rng(0)
            Yd = rngd.standard_normal((dn, dp)); Yd /= np.linalg.norm(Yd, axis=1, keepdims=True)
            hd.set_point(Yd)
            for _ in range(2):
                msd, byd, fld = hd.bench_hessvec(100)
            hd.close()
            # symmetric C, p <= 32, n >= 8192: the upper-triangle contraction of msdp_densesym.hip (k_dense_sym + k_sym_fold) takes the product
            sym = dp <= 32 and dn >= 8192
            kname = ("k_dense_sym + k_sym_fold + k_dense_hess_epi_obl" if sym else "k_dense_partial3 + k_dense_hess_epi_obl")
            survey_bytes = byd                       # 8 n^2 + 3 * 8 n p (the whole matrix once)
            if sym:
                # what the symmetric algorithm must move: the upper triangle incl. the diagonal once + three panels.  The bound follows
                # from the arithmetic intensity against the ridge (fp64 MFMA peak / HBM peak = 9.8 flop/B), not from p alone
                byd = 8.0 * dn * (dn + 1) / 2 + 3 * 8.0 * dn * dp
            ridge = MFMA_F64_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            bound = "mfma" if fld / byd > ridge else "hbm"
            ent = {"n": dn, "p": dp, "kernel": kname, "hessvec_us": msd * 1e3,
                   "algorithmic_bytes": byd, "algorithmic_flops": fld, "arithmetic_intensity_flop_per_byte": fld / byd,
                   "TFLOPs_f64": fld / msd / 1e9, "frac_mfma_f64_peak": fld / msd / 1e9 / MFMA_F64_TFLOPS,
                   "GBps": byd / msd / 1e6, "frac_hbm_peak": byd / msd / 1e6 / HBM_PEAK_GBS,
                   "roofline": secondary_roofline(kname, msd * 1e3, byd, fld,
                                                  ("r6_pmc_dense%d_p%d.json" % (dn, dp), "r5_pmc_dense%d_p%d.json" % (dn, dp), "r4_pmc_dense%d_p%d.json" % (dn, dp), "r3_pmc_dense%d_p%d.json" % (dn, dp), "r2_pmc_dense%d_p%d.json" % (dn, dp)),
                                                  bound=bound)}
            if sym:
                ent["survey_8d_bytes"] = survey_bytes
                ent["roofline"]["note"] = ("symmetric route: algorithmic_bytes = 8 n (n + 1) / 2 + 3 * 8 n p (upper triangle once + three panels); "
                                           "SURVEY.md 8(d)'s whole-matrix figure is kept as survey_8d_bytes and is not a bound for this kernel")
            if dn == 5000 and not args.no_cpu_baseline:
                ent["cpu_baseline"] = cpu_dense_hessvec(dn, dp)
            dense.append(ent)
        # config 5 per-GPU shard: rank 0 of 8 of the n = 100 000 problem (12 500 x 100 000 rows of C = 10 GB, generated on the
        # device), p = 64; the full direction is supplied locally instead of by the all-gather
        try:
            hk = _lib.Handle.dense_synthetic(100000, 0, nranks=8, rank=0, pcap=64)
            rngk = np.random.default_rng(0)
            Yk = rngk.standard_normal((100000, 64)); Yk /= np.linalg.norm(Yk, axis=1, keepdims=True)
            hk.set_point(Yk)
            hk.debug_set_full_rows(Yk)
            for _ in range(2):
                msk, byk, flk = hk.bench_hessvec(50)
            hk.close()
            dense.append({"n": 100000, "rows_on_this_gpu": 12500, "p": 64, "shard": "rank 0 of 8 (BASELINE config 5)",
                          "kernel": "k_dense_partial3 + k_dense_hess_epi_obl", "hessvec_us": msk * 1e3,
                          "TFLOPs_f64": flk / msk / 1e9, "frac_mfma_f64_peak": flk / msk / 1e9 / MFMA_F64_TFLOPS,
                          "GBps": byk / msk / 1e6, "frac_hbm_peak": byk / msk / 1e6 / HBM_PEAK_GBS,
                          "roofline": secondary_roofline("k_dense_partial3 + k_dense_hess_epi_obl", msk * 1e3, byk, flk,
                                                         ("r3_pmc_k5shard_p64.json",), bound="mfma")})
        except Exception as e:  # noqa: BLE001 -- secondary figure
            dense.append({"n": 100000, "p": 64, "error": "%s: %s" % (type(e).__name__, e)})
        out["dense_mfma"] = dense
    if not args.no_large_sparse and N == 1 and rank == 0 and not args.force_comm:
        # The chunked tCG path -- what every sparse problem beyond the persistent kernel's reach takes -- at n = 10^6, p = 32
        # (toroidal grid MaxCut, the G81 family scaled up): the two launches of msdp_trip1.hip move 14 vectors per trip.
        try:
            ln, lp = 1000, 32
            Cl = problems.toroidal_grid_maxcut(ln, ln, seed=3)
            rngl = np.random.default_rng(0)
            Yl = rngl.standard_normal((ln * ln, lp)); Yl /= np.linalg.norm(Yl, axis=1, keepdims=True)
            hl = _lib.Handle.onlyunitdiag(Cl, pcap=lp)
            hl.set_point(Yl)
            trip_us = min(hl.bench_tcg_trip(64) for _ in range(3)) * 1e3
            msl, byl, fll = hl.bench_hessvec(50)
            hl.set_option("window", 0)
            msl0, _, _ = hl.bench_hessvec(50)
            hl.set_option("window", 1)
            path = hl.tcg_path()
            hl_passes = 14
            hl.close()
            vec = ln * ln * lp * 8.0
            npass = hl_passes
            formulation = npass * vec + Cl.nnz * 12.0    # what the two launches stream: their vector passes + the (index, value) slices of C
            # the ALGORITHMIC traffic of a trip is that of its one Hess-vec -- nnz*12 + 4(n+1) + 3*8*n*p + 8n
            algo = Cl.nnz * 12.0 + 4.0 * (ln * ln + 1) + 3.0 * vec + 8.0 * ln * ln
            out["large_sparse_trip"] = {
                "workload": "toroidal grid MaxCut n=%d, p=%d, one tCG trip (tCG.m:160-287) of the chunked path" % (ln * ln, lp),
                "n": ln * ln, "p": lp, "tcg_path": path, "trip_us": trip_us, "hessvec_kernel_us": msl * 1e3,
                "vector_passes_per_trip": npass,
                "roofline": secondary_roofline("k_tcg1_upd + k_tcg1_head", trip_us, algo, 0.0,
                                               ("r6_pmc_linear_n1e6_p32.json", "r4_pmc_linear_n1e6_p32.json", "r3_pmc_linear_n1e6_p32.json"), bound="hbm", per="tCG trip")}
            # the stand-alone S*U launch of the same handle: k_hess_win_obl (round 5: the rows of U a breadth-first patch of rows touches
            # staged once per workgroup in LDS), and beside it the direct gathers of k_hess_ell_obl (option window = 0)
            out["large_sparse_trip"]["standalone_hessvec"] = {
                "kernel_us": msl * 1e3, "kernel_us_direct_gathers": msl0 * 1e3,
                "roofline": secondary_roofline("k_hess_win_obl", msl * 1e3, byl, fll, ("r6_pmc_hess_win_n1e6_p32.json", "r5_pmc_hess_win_n1e6_p32.json", "r4_pmc_hess_n1e6_p32.json"), bound="hbm", per="launch")}
            rec = out["large_sparse_trip"]["roofline"]
            rec["formulation_bytes"] = formulation
            rec["formulation_GBps"] = formulation / (trip_us * 1e-6) / 1e9
            rec["formulation_frac_of_hbm_peak"] = rec["formulation_GBps"] / HBM_PEAK_GBS
            rec["note"] = ("algorithmic_bytes = the Hess-vec of SURVEY.md 8(d) (one per trip); formulation_bytes = the %d vector passes "
                           "the two launches of the trip actually stream -- the gap between the two fractions is the pass count, not the kernels" % npass)
            if rec.get("traffic") is None and rec.get("traffic_source"):
                rec["traffic"] = json.load(open(os.path.join(ROOT, rec["traffic_source"]))).get("hbm_bytes_per_trip")
        except Exception as e:  # noqa: BLE001 -- secondary figure
            out["large_sparse_trip"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not args.no_affine and N == 1 and rank == 0 and not args.force_comm:
        out["affine_hessvec"] = affine_shapes(_lib, problems, not args.no_cpu_baseline)
    if not args.no_xrank and N == 1 and rank == 0 and not args.force_comm:
        try:
            out["cross_rank_trip"] = cross_rank_trip(_lib, problems)
        except Exception as e:  # noqa: BLE001 -- secondary figure
            out["cross_rank_trip"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["process_rank_trip"] = process_rank_trip()
        except Exception as e:  # noqa: BLE001 -- secondary figure
            out["process_rank_trip"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            # 2 x 10 000 rows: 79 rows per workgroup on the members' 128 workgroups each -- the shape every member of an N-GPU run has on
            # its own 216 (20 000 rows), where the cross-rank kernels run the ONE-reduction trip (msdp_pipe.h XRM; round 6); at 2 x 20 000
            # rows on one device a workgroup owns 157 rows and the two-reduction trip runs
            out["process_rank_trip_10000"] = process_rank_trip(grid_rows=50)
            hq = _lib.Handle.onlyunitdiag(problems.toroidal_grid_maxcut(100, 200, seed=81), pcap=p)
            rq = np.random.default_rng(0)
            Yq = rq.standard_normal((20000, p)); Yq /= np.linalg.norm(Yq, axis=1, keepdims=True)
            hq.set_point(Yq)
            out["process_rank_trip_10000"]["trip_us_one_unsharded_handle"] = min(hq.bench_tcg_trip(256) for _ in range(3)) * 1e3
            hq.close()
        except Exception as e:  # noqa: BLE001 -- secondary figure
            out["process_rank_trip_10000"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if N > 1 or args.force_comm:
        # BASELINE config 5 next to the headline metric: synthetic dense C generated per shard on the device
        # (12 500 rows per GPU, n = 12 500 * N, so N = 8 is exactly n = 100 000), p = 64, RCCL all-gather of the
        # direction before every S*U.  A watchdog keeps a failure here from costing the headline line.
        import threading
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                out.setdefault("k5_dense_sharded", {"error": "no result within 240 s"})
                result_out.write(json.dumps(out) + "\n")
                result_out.flush()
            os._exit(0)

        dog = threading.Timer(240.0, bail)
        dog.daemon = True
        dog.start()
        if N > 1:
            # The same K steps with the halo exchange (option halo_exchange: only the rows a rank's rows of C reference
            # travel before S*U -- 400 of 20000*N rows on this family -- instead of the all-gather of the whole direction).
            # Results are bit-identical (tests/test_gpu_local_ranks.py); the all-gather figure above was taken first and
            # stays in the line, so a failure of this leg costs nothing.
            # Which of the two is `value` is fixed by --row-exchange, never chosen after the fact.
            hl = halo_leg(_lib, join, sync, allmax, N, rank, C, Y0, p, opts, args.steps, args.warmup)
            out["row_exchange"] = {"all_gather": {"value": out["value"], "ms_per_step": out["ms_per_step"]}, "halo": hl}
            out["config"]["row_exchange"] = "all-gather"
            if args.row_exchange == "halo" and "error" not in hl and hl["hessvecs"] == hv:
                out["value"], out["ms_per_step"] = hl["value"], hl["ms_per_step"]
                out["config"]["row_exchange"] = "halo (option halo_exchange): grouped ncclSend/ncclRecv of the referenced rows"
            # Round 6: the N-GPU form of the persistent kernels -- one PROCESS group over HIP IPC / peer access, the cross-rank tCG
            # and TR tail with two-level grid reductions (each member over its own grid, eight sums per member over xGMI), boundary
            # rows pushed into the neighbours' buffers: no collective per trip, none per iteration.  `value` is quoted on THIS path
            # when --multi-gpu-path xr (the default) and it ran; the RCCL legs above stay in the line beside it.  Whether the path is
            # available is a property of the node (every rank on it, peer access between the devices), decided before any timing.
            out["config"]["multi_gpu_path"] = "rccl (%s)" % out["config"]["row_exchange"]
            why_not = xr_unavailable(N, local_rank)
            if args.multi_gpu_path != "xr":
                out["cross_rank_persistent"] = {"skipped": "--multi-gpu-path rccl"}
            elif allmax(0.0 if why_not is None else 1.0) > 0.0:
                out["cross_rank_persistent"] = {"skipped": why_not or "unavailable on another rank"}
            else:
                os.environ.setdefault("MSDP_LOCAL_BARRIER_TIMEOUT", "60")
                ident = ["/msdp_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid()) if rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                xl = xr_leg(_lib, lambda hh: hh.comm_init_ipc(N, rank, ident[0]), sync, allmax, N, rank, C, Y0, p, opts, args.steps, args.warmup)
                out["cross_rank_persistent"] = xl
                if "error" not in xl and xl.get("tcg_path") == 2 and xl["hessvecs"] == hv:
                    out["value"], out["ms_per_step"] = xl["value"], xl["ms_per_step"]
                    out["config"]["multi_gpu_path"] = ("cross-rank persistent tCG + TR tail over HIP IPC / xGMI, two-level grid reductions "
                                                       "(msdp_comm_init_ipc); RCCL legs: row_exchange")
                    out["config"]["tcg_path"] = "persistent kernels spanning the ranks: no collective per trip or TR iteration"
        if not args.no_dense:
            try:
                out["k5_dense_sharded"] = k5_dense_sharded(_lib, join, sync, allmax, N, rank)
            except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
                out["k5_dense_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dist.barrier()
        finished.set()
        dog.cancel()
        dist.destroy_process_group()
    if rank == 0:
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()


def _sharded_leg(_lib, join, sync, allmax, N, rank, C, Y0, p, opts, steps, warmup, options, what, probe=None):
    """The timed region of main() once more on a fresh handle.  `join(h)` makes the handle a member of the job's group (RCCL
    communicator or process group over HIP IPC), `sync()` is the barrier + device synchronisation of main(), `allmax(x)` the
    maximum of x over the ranks (closures, so that tests can drive this on in-process ranks).  options: set on the handle behind
    the join; probe(h): extra figures taken behind the timed loop (a collective call: every rank makes it)."""
    # Set-up may fail on ONE rank only: every rank reports its outcome and all of them skip the timed region together -- a rank
    # that raised alone would leave the others waiting in a collective / a grid reduction.
    h, err = None, None
    try:
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
    except Exception as e:  # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    if allmax(0.0 if err is None else 1.0) > 0.0:           # group set-up is a collective itself: only if everybody has a handle
        if h is not None:
            h.close()
        return {"error": err or "another rank failed to create its handle"}
    try:
        join(h)
        for k, v in options.items():
            h.set_option(k, v)
        h.set_point(Y0)
        h.point_snapshot()
    except Exception as e:  # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    if allmax(0.0 if err is None else 1.0) > 0.0:
        h.close()
        return {"error": err or "another rank failed to set %s up" % what}
    # (from here on a failure -- a grid reduction of the cross-rank kernels that times out -- is raised on every rank by the library
    # itself; the barriers and maxima below are still entered by everybody, in the same order)
    try:
        for _ in range(max(1, warmup)):
            h.point_restore()
            h.rtr(opts)
    except Exception as e:  # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    sync()
    t0 = time.perf_counter()
    hv = 0
    if err is None:
        try:
            for _ in range(steps):
                h.point_restore()
                hv += h.rtr(opts).hessvecs
        except Exception as e:  # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
    sync()
    dt = allmax(time.perf_counter() - t0)
    bad = allmax(0.0 if err is None else 1.0) > 0.0
    res = {"value": hv * N / dt, "ms_per_step": dt / steps * 1e3, "hessvecs": hv}
    if bad:
        res = {"error": err or "another rank failed in the timed region of %s" % what}
    elif probe is not None:
        try:
            res.update(probe(h))
        except Exception as e:  # noqa: BLE001
            res["probe_error"] = "%s: %s" % (type(e).__name__, e)
    try:
        h.close()
    except Exception:  # noqa: BLE001 -- a broken group refuses its collectives; the process ends soon
        pass
    return res


def halo_leg(_lib, join, sync, allmax, N, rank, C, Y0, p, opts, steps, warmup):
    """The K steps on a fresh handle with the halo exchange in front of S*U (RCCL: grouped ncclSend / ncclRecv of the referenced rows)."""
    return _sharded_leg(_lib, join, sync, allmax, N, rank, C, Y0, p, opts, steps, warmup, {"halo_exchange": 1}, "the halo exchange")


def xr_leg(_lib, join_ipc, sync, allmax, N, rank, C, Y0, p, opts, steps, warmup):
    """The K steps on a fresh handle whose group is a PROCESS group over HIP IPC (msdp_comm_init_ipc, one rank per GPU): the
    cross-rank persistent tCG and TR tail -- every member reduces over its own grid, the members' sums travel over xGMI (two-level
    reductions, msdp_psync.h), boundary rows are pushed into the neighbours' exchange buffers; no collective per trip or iteration."""
    def probe(h):
        path = h.tcg_path()
        return {"tcg_path": path, "collective_calls_total": h.collective_calls(),
                "trip_us_cross_rank_persistent": min(h.bench_tcg_trip(256) for _ in range(2)) * 1e3 if path == 2 else None}
    return _sharded_leg(_lib, join_ipc, sync, allmax, N, rank, C, Y0, p, opts, steps, warmup, {}, "the process group", probe)


def k5_dense_sharded(_lib, join, sync, allmax, N, rank, rows_per_gpu=12500, p=64):
    """One short RTR call (6 TR iterations, at most 8 inner trips each) on the row-sharded dense-C problem; every rank fills its
    own rows on the device.  Reports the time per S*X product (Hess-vecs + cost/gradient evaluations) and the
    aggregate fp64 rate 2 n^2 p per product.  join / sync / allmax: see halo_leg."""
    n = rows_per_gpu * N
    h = _lib.Handle.dense_synthetic(n, 0, nranks=N, rank=rank, pcap=p)
    join(h)
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((n, p))
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    h.set_point(Y)
    h.point_snapshot()
    opts = _lib.default_opts(maxiter=6, maxinner=8, tolgradnorm=1e-12)
    best = None
    for _ in range(3):
        h.point_restore()
        sync()
        t0 = time.perf_counter()
        st = h.rtr(opts)
        sync()
        dt = allmax(time.perf_counter() - t0)
        products = st.hessvecs + st.iters + 1
        if best is None or dt < best[0]:
            best = (dt, products, st.hessvecs)
    h.close()
    sec, products, hv = best
    return {"workload": "synthetic dense-C unit-diag SDP, n=%d, p=%d, rows sharded over %d GPUs (BASELINE config 5 at N=8)" % (n, p, N),
            "n": n, "p": p, "ranks": N, "rtr_seconds": sec, "hessvecs": hv, "S_times_X_products": products,
            "ms_per_product": sec / products * 1e3, "aggregate_TFLOPs_f64": 2.0 * n * n * p * products / sec / 1e12,
            "per_gpu_matrix_GB": rows_per_gpu * float(n) * 8 / 1e9}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--ipc-worker":
        process_rank_worker(sys.argv[2:])
    else:
        main()
