#!/usr/bin/env python3
"""Hot-path benchmark (driver contract: `python bench.py --gpus N --steps K --warmup W`).

`--gpus N > 1` without a torchrun environment: this process starts N fresh rank processes
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) BEFORE anything touches the GPU,
relays rank 0's JSON line and exits with the children's status.  Under torchrun (WORLD_SIZE set) it
is one rank.

Workload (BASELINE.json configs[3] / its per-GPU share, named in `config.workload`): SVO 3-band
Wannier Hamiltonian (aps_example/svo_hr.dat, 1331 R vectors), full-BZ PTR grid npt^3.
  Phase A (primary metric, "k-point evals/sec (H(k)+eig)"): one PASS = re-evaluate H(k) and its
     Hermitian eigenvalues at all npt^3 nodes from the device-resident coefficients into the
     device-resident rule (abz_rule_rebuild: contract, contract, eval+eig kernels); one STEP =
     `--passes-per-step` back-to-back passes (a step of one pass is 0.14 ms: 20 of them are below
     timer and placement noise), so the timed region of K steps is >= 0.3 s.
  Phase B (secondary, "DOS(omega) points/sec"): the DOS integrand scan+reduce over the cached rule
     for this rank's share of the 256-omega sweep (256/8 = 32 per GPU), fused in one pass.
Multi-GPU, primary number: every rank holds a replica of the coefficients and rebuilds its own rule
(no data-path collective) => scaling "weak".  Beside it, at every N, the FIXED jobs the north star
names (strong scaling, speed-up against the same job on rank 0 alone measured in the same run):
  * the 256-omega DOS sweep on the 150^3 grid, omega-sharded (replicated build + 256/N omega per rank +
    one all_gather, src/interfaces.jl:199-222) and k-sharded (1/N slab of the grid per rank + all 256
    omega + one all_reduce) -- scan results stay in HBM and feed the RCCL collective on the
    same stream (abz_rule_reduce_device);
  * a 432-omega IAI sweep on the cubic IBZ (the size of the reference demo's sweep) through
    batchsolve_sharded.
Inputs are resident in HBM when the timed region starts; all timing is bracketed by a barrier and a
device synchronize on both sides; kernel durations come from HIP events on the library's stream.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured copy)
F64_PEAK_TFLOPS = 78.6  # f64 vector = f64 matrix peak (AMD spec, not in the local guide)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--npt", type=int, default=150)
    ap.add_argument("--passes-per-step", type=int, default=128)
    ap.add_argument("--omegas-per-rank", type=int, default=32)
    ap.add_argument("--eta", type=float, default=0.1)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-iai", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="primary metric + sharded jobs only")
    ap.add_argument("--no-ref-layout", action="store_true",
                    help="skip the reference-layout and eigenvalues-only legs (profiling runs: the kernel trace then holds "
                         "one layout per instance of the Fourier-eval kernel)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise the process group and run the sharded jobs even at world size 1 "
                         "(under `python -m torch.distributed.run --nproc-per-node 1`): the RCCL calls on one GPU")
    ap.add_argument("--big-npt", type=int, default=400, help="grid of the fine-grid 256-omega job (eta = 0.01)")
    ap.add_argument("--no-big-job", action="store_true")
    ap.add_argument("--no-scaling-model", action="store_true", help="skip the 8-GPU scaling model from one-GPU shard timings (N = 1; ~40 s)")
    ap.add_argument("--no-c5-shard", action="store_true", help="skip config 5 as one sharded solve (N > 1 only; ~11 s for its N = 1 leg)")
    ap.add_argument("--c5-abstol", type=float, default=1e-3, help="config 5 (16-band IAI) tolerance; SURVEY 8d: 1e-3")
    return ap.parse_args()


def spawn_ranks(a):
    """Start N rank processes; this parent never imports torch.cuda nor creates an abz context."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"bench.py: starting {a.gpus} ranks: {' '.join(cmd[1:6])} ...", file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env)
    return p.returncode


def pmc_traffic(npt, compact=False):
    """HBM bytes per launch of the Fourier-eval kernel from the committed rocprofv3 PMC passes
    (profiles/r04_traffic_compact.json / r03_traffic.json (or an earlier round's), written by tools/collect_profiles.sh: WRITE_SIZE and
    FETCH_SIZE in separate --pmc runs, FETCH_SIZE doubled per the gfx950 correction), for the rule layout that was
    timed.  None if not collected for this grid size: counters cannot be read from inside an un-profiled bench run."""
    names = ("r05_traffic_compact.json", "r04_traffic_compact.json", "r03_traffic_compact.json") if compact else ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")
    for name in names:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            if int(t.get("npt", -1)) == int(npt):
                return float(t["hbm_bytes_per_launch"])
        except Exception:
            pass
    return None


def cpu_baseline(npt, s, eta, budget_s=25.0):
    """Time the C restatement of the reference's CPU path (oracle/abz_oracle.c, kind 'port') on the host
    cores, on the ACTUAL grid of the GPU workload (npt^3) and a bounded omega count."""
    lib_path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(lib_path):
        return None
    flags = "-march=x86-64-v3 (prebuilt)"
    try:  # a -march=native build made on the box that times it (building the checker is not using it)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=120)
        native = os.path.join(ROOT, "oracle", "_build", "liboracle_native.so")
        if os.path.exists(native):
            lib_path, flags = native, "-march=native (built on this box)"
    except Exception:
        pass
    lib = ctypes.CDLL(lib_path)
    lib.orc_num_threads.restype = ctypes.c_int
    aff = len(os.sched_getaffinity(0))
    from autobzcore.jl_amd.series import julia_coefficient_order
    coef = np.ascontiguousarray(julia_coefficient_order(s.c, 3))
    dims = np.array(s.dims, dtype=np.int32)
    first = np.array(s.first, dtype=np.int32)
    nk = npt**3
    vals = np.empty(nk * 9, dtype=np.complex128)
    eig = np.empty(nk * 3)
    P = ctypes.c_void_p
    args = (coef.ctypes.data_as(P), 3, dims.ctypes.data_as(P), first.ctypes.data_as(P), 3, npt,
            vals.ctypes.data_as(P), eig.ctypes.data_as(P))
    fn = getattr(lib, "orc_fourier_ptr3", None) or lib.orc_fourier_ptr
    # thread count: every core the process may use, unless fewer threads are faster (a 1-GPU box is a CPU-quota
    # share of a 256-thread host: 256 OpenMP threads get throttled there and run 8x slower than 16)
    tried = {}
    cands = [int(os.environ["ABZ_CPU_THREADS"])] if "ABZ_CPU_THREADS" in os.environ else \
        sorted({t for t in (aff, 128, 64, 32, 16, 8) if t <= aff}, reverse=True)
    for th in cands:  # SUSTAINED rate per candidate (>= 1 s each: a burst of 10 ms is not throttled by the CPU quota, a run is)
        lib.orc_set_threads(th)
        fn(*args)  # warm-up (page faults, thread pool)
        t0 = time.perf_counter()
        r = 0
        while r < 2 or time.perf_counter() - t0 < 1.0:
            fn(*args)
            r += 1
        tried[th] = nk * r / (time.perf_counter() - t0)
    want = max(tried, key=tried.get)
    lib.orc_set_threads(want)
    cores = lib.orc_num_threads()
    fn(*args)
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 0.3 * budget_s:
        fn(*args)
        reps += 1
    tA = (time.perf_counter() - t0) / reps
    scan = getattr(lib, "orc_dos_scan3", None) or lib.orc_dos_scan
    scan.argtypes = [P, ctypes.c_int64, ctypes.c_int, ctypes.c_double, P, ctypes.c_int, P]
    n_om = 32
    omegas = np.linspace(10, 15, 256)[:: 256 // n_om]
    out = np.empty(n_om)
    t0 = time.perf_counter()
    scan(vals.ctypes.data_as(P), nk, 3, eta, omegas.ctypes.data_as(P), n_om, out.ctypes.data_as(P))
    tB = time.perf_counter() - t0
    res = {"value": nk / tA, "unit": "k-point evals/s (H(k)+eig)", "cores": cores, "kind": "port",
           "nproc": os.cpu_count(), "affinity": aff, "kpoints_per_sec_by_thread_count": tried,
           "sample": f"SVO 3-band, PTR npt={npt} FBZ ({nk} k-points, the GPU workload's grid), {reps} builds + a "
                     f"{n_om}-omega scan of the cached H(k); C restatement of the reference loops (not Julia): closed-form "
                     f"3x3 Hermitian eigenvalues and adjugate inverse like StaticArrays, gcc -O3 {flags} "
                     f"-fcx-limited-range -fopenmp, {cores} threads = the fastest of {sorted(tried)} (affinity {aff}, {os.cpu_count()} logical CPUs)",
           "build_seconds": tA, "dos_kpoint_omega_per_sec": nk * n_om / tB, "scan_seconds_per_omega": tB / n_om,
           "dos_sample": f"{n_om} of the 256 omegas over the same {nk} cached H(k)"}
    lib.orc_set_threads(1)
    t0 = time.perf_counter()
    r1 = 0
    while r1 < 1 or time.perf_counter() - t0 < 3.0:
        fn(*args)
        r1 += 1
    res["value_1_thread"] = nk * r1 / (time.perf_counter() - t0)
    lib.orc_set_threads(want)
    return res


def cpu_baseline_legs(out, abz, s, npt, eta, cores, c5_abstol):
    """The C port (oracle/abz_oracle.c, kind 'port') beside the IAI and GGR legs of the line: the reference's nested
    GK(7,15) loop (src/fourier.jl:432-510) and get_ggr_data + sum_ggr (src/dos_ggr.jl:14-65), on bounded samples."""
    native = os.path.join(ROOT, "oracle", "_build", "liboracle_native.so")
    lib = ctypes.CDLL(native if os.path.exists(native) else os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    lib.orc_num_threads.restype = ctypes.c_int
    lib.orc_set_threads(cores)
    P = ctypes.c_void_p
    from autobzcore.jl_amd.series import julia_coefficient_order
    lib.orc_iai_dos3.restype = ctypes.c_double
    lib.orc_iai_dos3.argtypes = [P, P, P, ctypes.c_int, P, P, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_int64, P, P]

    def iai(series, eta_, omega, abstol):
        coef = np.ascontiguousarray(julia_coefficient_order(series.c, 3))
        dims, first = np.array(series.dims, dtype=np.int32), np.array(series.first, dtype=np.int32)
        lo, hi = np.zeros(3), np.ones(3)
        err, nev = ctypes.c_double(0), ctypes.c_int64(0)
        t0 = time.perf_counter()
        u = lib.orc_iai_dos3(coef.ctypes.data_as(P), dims.ctypes.data_as(P), first.ctypes.data_as(P), series.n, lo.ctypes.data_as(P),
                             hi.ctypes.data_as(P), eta_, omega, abstol, -1.0, 2**62, ctypes.byref(err), ctypes.byref(nev))
        return u, nev.value, time.perf_counter() - t0

    what = ("C port of the reference's nested adaptive GK(7,15) loop (depth first, scalar refinement, abstol / len per level; the "
            "nodes of an outermost round dealt to threads like NestedBatchIntegrand's workers), gcc -O3 -fopenmp")
    ex = out.get("iai_example", {}).get("FBZ") if isinstance(out.get("iai_example"), dict) else None
    if isinstance(ex, dict) and "numevals" in ex:
        try:  # the reference example's own solve (aps_example/aps_example.jl:29-34: eta = 0.01 eV, abstol 1e-3) in full
            j = abs(np.linalg.det(2 * np.pi * np.linalg.inv(3.85856 * np.eye(3)).T))
            u, nev, dt = iai(s, 0.01, ex.get("omega", 12.5), 1e-3 / j)
            ex["cpu_baseline"] = {"kind": "port", "cores": cores, "seconds": dt, "numevals": nev, "nodes_per_sec": nev / dt, "u": u * j,
                                  "same_numevals_as_gpu": bool(nev == ex["numevals"]), "gpu_over_cpu": dt / ex["seconds"],
                                  "sample": "the whole solve; " + what}
        except Exception as e:
            ex["cpu_baseline"] = {"error": repr(e)}
    c5 = out.get("iai_config5")
    if isinstance(c5, dict) and "nodes_per_sec" in c5:
        try:  # config 5 at a looser tolerance (the node rate does not depend on it): a bounded sample
            s16 = abz.synthetic_wannier()
            j16 = (2 * np.pi) ** 3  # |det B| of load_bz(FBZ, I): do_solve hands abstol / |det B| to the nested quadrature
            atol = 64.0  # tightened until the sample lasts a few seconds (never beyond the GPU leg's own tolerance)
            lib.orc_set_deadline.argtypes = [ctypes.c_double]
            lib.orc_deadline_passed.restype = ctypes.c_int
            cut = False
            while True:
                atol = max(atol / 2, c5_abstol)
                lib.orc_set_deadline(20.0)  # (a halving can cost 30x: the port stops refining at the deadline, its count stays valid)
                u, nev, dt = iai(s16, 0.05, 0.2, atol / j16)
                cut = bool(lib.orc_deadline_passed())
                lib.orc_set_deadline(0.0)
                if dt >= 3.0 or atol <= c5_abstol:  # (a sample of 3 ... 20 s; shorter ones leave most threads idle)
                    break
            c5["cpu_baseline"] = {"kind": "port", "cores": cores, "abstol_of_the_sample": atol, "stopped_at_the_20_s_budget": cut, "seconds": dt, "numevals": nev,
                                  "nodes_per_sec": nev / dt, "gpu_over_cpu": c5["nodes_per_sec"] / (nev / dt),
                                  "sample": f"config 5 (synthetic 16-band IAI on the FBZ) at abstol {atol:g} instead of {c5_abstol:g}: nodes/s "
                                            "against nodes/s; " + what + "; Gauss-Jordan inverse with partial pivoting per node"}
        except Exception as e:
            c5["cpu_baseline"] = {"error": repr(e)}
    g = out.get("ggr")
    if isinstance(g, dict) and "build_seconds" in g:
        try:
            coef = np.ascontiguousarray(julia_coefficient_order(s.c, 3))
            dims, first, per = np.array(s.dims, dtype=np.int32), np.array(s.first, dtype=np.int32), np.ones(3)
            nk = npt**3
            eig, vel = np.empty(nk * 3), np.empty(nk * 9)
            args = (coef.ctypes.data_as(P), dims.ctypes.data_as(P), first.ctypes.data_as(P), 3, npt, per.ctypes.data_as(P),
                    eig.ctypes.data_as(P), vel.ctypes.data_as(P))
            lib.orc_ggr_data(*args)
            t0 = time.perf_counter()
            reps = 0
            while reps < 2 or time.perf_counter() - t0 < 4.0:
                lib.orc_ggr_data(*args)
                reps += 1
            tb = (time.perf_counter() - t0) / reps
            nE = 32
            Es = np.linspace(10.0, 15.0, 256)[:: 256 // nE]
            o = np.empty(nE)
            lib.orc_sum_ggr3.argtypes = [ctypes.c_int, P, ctypes.c_int, ctypes.c_int64, ctypes.c_int, P, P, P]
            t0 = time.perf_counter()
            lib.orc_sum_ggr3(npt, Es.ctypes.data_as(P), nE, nk, 3, eig.ctypes.data_as(P), vel.ctypes.data_as(P), o.ctypes.data_as(P))
            tsc = (time.perf_counter() - t0) / nE
            g["cpu_baseline"] = {"kind": "port", "cores": cores, "build_seconds": tb, "build_kpoints_per_sec": nk / tb,
                                 "scan_seconds_per_energy": tsc, "scan_kE_per_sec": nk / tsc,
                                 "gpu_over_cpu": tb / g["build_seconds"], "gpu_over_cpu_scan_256_energies": 256 * tsc / g["scan_seconds"],
                                 "sample": f"get_ggr_data on the same {npt}^3 grid ({reps} builds) + sum_ggr for {nE} of the 256 energies; C port: "
                                           "JacobianSeries contracted hierarchically, closed-form 3x3 eigenvalues, eigenvectors from cross "
                                           "products (StaticArrays' route), threads over the outermost grid index and over the nodes of a scan "
                                           "(the reference's own eigen loop and sum_ggr are serial)"}
        except Exception as e:
            g["cpu_baseline"] = {"error": repr(e)}


def rank_main(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    rehearsal = ndev < world  # several ranks share one card: collectives through gloo (RCCL refuses duplicate devices)
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    backend = os.environ.get("ABZ_DIST_BACKEND", "gloo" if rehearsal else "nccl")  # nccl = RCCL over xGMI
    distributed = world > 1 or a.force_dist
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    n_ranks_seen = dist.get_world_size() if distributed else 1
    assert n_ranks_seen == a.gpus, (n_ranks_seen, a.gpus)
    cdev = "cuda" if (world == 1 or backend == "nccl") else "cpu"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    import autobzcore.jl_amd as abz
    from autobzcore.jl_amd import _lib as L
    # the library works on a stream owned by torch: its launches and the RCCL collectives are ordered there
    st = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(st)
    ctx = abz.Context(local, stream=st.cuda_stream)
    abz.Context._default = ctx
    s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
    dev = s.device(ctx)
    npt = a.npt
    nk = npt**3
    n = 3
    # inputs + rule buffers resident before timing.  dev.rule() is the host mirror's own call: rules of a Hermitian series
    # keep H(k) as its upper triangle (ABZ_WANT_H_COMPACT, 96 B per k-point with the eigenvalues instead of the
    # reference layout's 168); ABZ_RULE_COMPACT=0 times the reference layout as the primary number instead
    rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
    WANT = rule.want
    compact = bool(WANT & L.WANT_H_COMPACT)
    bpk = 8 * (n * n + n) if compact else 16 * n * n + 8 * n  # bytes one k-point leaves in HBM (H + eigenvalues)
    base_addr, rule_bytes = rule.values_ptr()
    n_total = a.omegas_per_rank * world
    omegas_all = np.linspace(10.0, 15.0, n_total)
    mine = omegas_all[rank::world]
    P = max(1, a.passes_per_step)

    # ---------------- Phase A: K steps of P rebuilds
    ctx.prof_enable(True, kernels=[L.K_CONTRACT])
    ctx.prof_reset()
    for _ in range(max(a.warmup, 1) * P):
        rule.rebuild()
    ctx.sync()
    con_ms, con_n = ctx.prof_read(L.K_CONTRACT)
    ctx.prof_enable(True, kernels=[L.K_EVAL])  # timed region: HIP events only around the dominant (Fourier-eval) kernel
    ctx.prof_reset()
    block_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for _ in range(P):
            rule.rebuild()
        if world == 1:  # per-step spread (the sync costs ~10 us per 18 ms step)
            ctx.sync()
            block_ms.append(time.perf_counter())
    ctx.sync()
    barrier()
    tA = time.perf_counter() - t0
    eval_ms, eval_n = ctx.prof_read(L.K_EVAL)
    ctx.prof_enable(False)
    step_ms = np.diff([t0] + block_ms) * 1e3 if block_ms else np.array([tA * 1e3 / a.steps])

    # the same passes with H(k) in the reference's full SMatrix layout (168 B per k-point): the numbers of rounds 1-2
    ref_layout = None
    if compact and world == 1 and not a.force_dist and not a.no_ref_layout:
        rfull = abz.DeviceRule(dev, npt, None, L.WANT_H | L.WANT_EIG)
        for _ in range(max(a.warmup, 1) * 8):
            rfull.rebuild()
        ctx.sync()
        ctx.prof_enable(True, kernels=[L.K_EVAL])
        ctx.prof_reset()
        nrep = max(a.steps * P // 4, 8)
        t0 = time.perf_counter()
        for _ in range(nrep):
            rfull.rebuild()
        ctx.sync()
        tF = time.perf_counter() - t0
        msF, nF = ctx.prof_read(L.K_EVAL)
        ctx.prof_enable(False)
        fb = 16 * n * n + 8 * n
        ref_layout = {"what": "the same rebuild passes with H(k) stored as the reference does (full n x n complex matrix per node, "
                              "FourierPTR's vals, src/fourier.jl:127-174): 168 B per k-point",
                      "value": nk * nrep / tF, "ms_per_pass": 1e3 * tF / nrep, "avg_launch_ms": msF / max(nF, 1), "launches": nF,
                      "algorithmic_bytes_per_kpoint": fb, "achieved_GBs": nk * fb / ((msF / max(nF, 1)) * 1e-3) / 1e9,
                      "frac": nk * fb / ((msF / max(nF, 1)) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(npt, False),
                      "rule_bytes": rfull.values_ptr()[1]}
        rfull.close()
        reig = abz.DeviceRule(dev, npt, None, L.WANT_EIG)  # the kernel without its H stores: the compute floor
        for _ in range(max(a.warmup, 1) * 8):
            reig.rebuild()
        ctx.sync()
        ctx.prof_enable(True, kernels=[L.K_EVAL])
        ctx.prof_reset()
        for _ in range(nrep):
            reig.rebuild()
        ctx.sync()
        msE, nE = ctx.prof_read(L.K_EVAL)
        ctx.prof_enable(False)
        reig.close()
        ref_layout["eigenvalues_only_avg_launch_ms"] = msE / max(nE, 1)

    # ---------------- Phase B: fused sweeps over this rank's omegas (matrix-cached, reference-faithful)
    nB = a.steps * 50
    # warm-up: the scan is f64-VALU bound while Phase A is HBM-write bound, and the GPU's clocks take ~0.1 s of the new
    # load to settle (the first ~1000 calls after Phase A average 0.157 ms, every later thousand 0.112 ms)
    for _ in range(max(1, a.warmup) * 400):
        rule.reduce(L.F_DOS, [a.eta], mine)
    barrier()
    t0 = time.perf_counter()
    for _ in range(nB):
        dos = rule.reduce(L.F_DOS, [a.eta], mine)[:, 0].real
    ctx.sync()
    barrier()
    tB = (time.perf_counter() - t0) / nB
    ctx.prof_enable(True, kernels=[L.K_REDUCE])  # kernel time of the same sweep by HIP events, outside the timed calls
    ctx.prof_reset()
    for _ in range(max(nB // 10, 20)):
        rule.reduce(L.F_DOS, [a.eta], mine)
    ctx.sync()
    red_ms, red_n = ctx.prof_read(L.K_REDUCE)
    ctx.prof_enable(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(nB):  # eigenvalue-cached variant of the same sweep (24 B per k instead of 144 B)
        dos_e = rule.reduce(L.F_DOS_EIG, [a.eta], mine)[:, 0].real
    ctx.sync()
    barrier()
    tBe = (time.perf_counter() - t0) / nB
    assert np.abs(dos - dos_e).max() < 1e-9 * np.abs(dos).max()

    # ---------------- the fixed 256-omega job, strong scaling
    om256 = np.linspace(10.0, 15.0, 256)
    om_dev = torch.from_numpy(om256).to(f"cuda:{local}")
    out_dev = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")
    REPS = 40

    def job_n1():  # rank 0 alone: build + one fused 256-omega scan; the sums come back over PCIe
        rule.rebuild()
        rule.reduce_device(L.F_DOS, [a.eta], om_dev.data_ptr(), 256, out_dev.data_ptr())
        return out_dev[:, 0].cpu().numpy()

    def timed(fn, reps, everyone=True):
        fn()
        if everyone:
            barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        if everyone:
            barrier()
        return (time.perf_counter() - t0) / reps, r

    t_n1, ref256 = (None, None)
    if rank == 0:
        t_n1, ref256 = timed(job_n1, REPS, everyone=False)
    barrier()
    jobs = None
    if distributed:
        # (1) omega-sharded: replicated build, 256/N omega per rank (round-robin like batchparam), one all_gather
        idx = np.arange(256)[rank::world]
        om_mine = torch.from_numpy(om256[idx]).to(f"cuda:{local}")
        part = torch.zeros(len(idx), 2, dtype=torch.float64, device=f"cuda:{local}")
        gathered = [torch.zeros_like(part) for _ in range(world)] if 256 % world == 0 else None

        def job_omega():
            rule.rebuild()
            rule.reduce_device(L.F_DOS, [a.eta], om_mine.data_ptr(), len(idx), part.data_ptr())
            if gathered is None:
                raise SystemExit("256 omegas do not divide over this world size")
            if cdev == "cuda":
                dist.all_gather(gathered, part)
                full = torch.stack(gathered, 1).reshape(-1, 2)[:, 0]  # [256/N, N] -> omega order (round-robin)
                return full.cpu().numpy()
            pc = part.cpu()
            g = [torch.zeros_like(pc) for _ in range(world)]
            dist.all_gather(g, pc)
            return torch.stack(g, 1).reshape(-1, 2)[:, 0].numpy()

        # (2) k-sharded: a slab of the outermost grid variable per rank, all 256 omega, one all_reduce(sum)
        dev.kshard = (rank, world)
        rule_k = abz.DeviceRule(dev, npt, None, WANT)
        dev.kshard = None
        acc = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")

        def job_k():
            rule_k.rebuild()
            rule_k.reduce_device(L.F_DOS, [a.eta], om_dev.data_ptr(), 256, acc.data_ptr())
            if cdev == "cuda":
                dist.all_reduce(acc)
                return acc[:, 0].cpu().numpy()
            c = acc.cpu()
            dist.all_reduce(c)
            return c[:, 0].numpy()

        t_om, r_om = timed(job_omega, REPS)
        t_k, r_k = timed(job_k, REPS)
        tt = torch.tensor([t_om, t_k], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_om, t_k = (float(v) for v in tt.cpu())
        if rank == 0:
            err_om = float(np.abs(r_om - ref256).max() / np.abs(ref256).max())
            err_k = float(np.abs(r_k - ref256).max() / np.abs(ref256).max())
            assert err_om < 1e-12 and err_k < 1e-11, (err_om, err_k)
            jobs = {"omega_sharded": {"seconds": t_om, "speedup_vs_n1": t_n1 / t_om, "max_rel_diff_vs_n1": err_om,
                                      "collective": "all_gather of 256/N complex sums per rank"},
                    "k_sharded": {"seconds": t_k, "speedup_vs_n1": t_n1 / t_k, "max_rel_diff_vs_n1": err_k,
                                  "collective": "all_reduce(sum) of 256 complex partial sums",
                                  "slab_planes": [int((npt * (r + 1)) // world - (npt * r) // world) for r in range(world)]}}
        rule_k.close()

    # ---------------- the same sweep at the reference example's eta = 0.01 eV on a grid that resolves it (k-sharded)
    big_job = None
    if not a.no_big_job:
        try:
            big_job = big_dos_job(a, abz, L, torch, dist, dev, rank, world, local, distributed, cdev, barrier, timed, WANT)
        except Exception as e:
            big_job = {"error": str(e)}

    # ---------------- a sweep big enough to shard by omega: 432 IAI solves on the cubic IBZ
    iai_job = None
    if not a.no_iai:
        fiai = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
        bzc = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
        sol_iai = abz.IntegralSolver(fiai, bzc, abz.IAI(), abstol=1e-3)
        om432 = np.linspace(10.0, 15.0, 432)
        t1, r1 = (None, None)
        if rank == 0:
            abz.batchsolve(sol_iai, om432)  # warm-up at full size: staging buffers and pools reach their final sizes
            t0 = time.perf_counter()
            r1 = abz.batchsolve(sol_iai, om432)
            t1 = time.perf_counter() - t0
        barrier()
        iai_job = {"n_solves": 432, "seconds_n1": t1}
        if distributed:
            abz.batchsolve_sharded(sol_iai, om432, device=cdev)  # warm-up at full size
            barrier()
            t0 = time.perf_counter()
            rN = abz.batchsolve_sharded(sol_iai, om432, device=cdev)
            barrier()
            tN = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
            dist.all_reduce(tN, op=dist.ReduceOp.MAX)
            if rank == 0:
                assert np.array_equal(np.asarray(rN, dtype=float), np.asarray(r1, dtype=float)), "sharded IAI sweep differs"
                iai_job.update({"seconds": float(tN.cpu()[0]), "speedup_vs_n1": t1 / float(tN.cpu()[0]),
                                "bit_identical_to_n1": True})

    # ---------------- config 5 as ONE solve on all GPUs: its inner integrals dealt to the ranks every round
    c5_job = None
    if distributed and not a.no_iai and not a.no_c5_shard:
        try:
            s16 = abz.synthetic_wannier()
            f16 = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
            prob16 = abz.IntegralProblem(f16, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
            abz.solve(prob16, abz.IAI(), abstol=10.0, reltol=0.0)  # warm-up
            t1, r1 = (None, None)
            if rank == 0:
                t0 = time.perf_counter()
                r1 = abz.solve(prob16, abz.EvalCounter(abz.IAI()), abstol=a.c5_abstol, reltol=0.0)
                t1 = time.perf_counter() - t0
            barrier()
            with abz.iaishard(s16, device=cdev, force=a.force_dist) as sh:
                abz.solve(prob16, abz.IAI(), abstol=10.0, reltol=0.0)
                barrier()
                t0 = time.perf_counter()
                rN = abz.solve(prob16, abz.EvalCounter(abz.IAI()), abstol=a.c5_abstol, reltol=0.0)
                barrier()
                tN = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
                nex = sh.rounds
            dist.all_reduce(tN, op=dist.ReduceOp.MAX)
            if rank == 0:
                assert rN.u == r1.u and rN.numevals == r1.numevals, (rN, r1)
                c5_job = {"what": "config 5 (synthetic 16-band IAI on the FBZ) as one solve sharded over the ranks: every round's innermost "
                                  "integrals dealt out in blocks of 64 nodes, one all-gather per round", "abstol": a.c5_abstol,
                          "numevals": rN.numevals, "seconds_n1": t1, "seconds": float(tN.cpu()[0]), "speedup_vs_n1": t1 / float(tN.cpu()[0]),
                          "bit_identical_to_n1": True, "exchanges": nex}
        except Exception as e:
            c5_job = {"error": repr(e)}

    # max-over-ranks timing of the primary legs
    times = torch.tensor([tA, tB, tBe], dtype=torch.float64, device=cdev)
    if distributed:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    tA, tB, tBe = (float(v) for v in times.cpu())

    if rank == 0:
        kps = world * nk * P * a.steps / tA
        eval_avg_s = (eval_ms / max(eval_n, 1)) * 1e-3
        achieved = nk * bpk / eval_avg_s / 1e9 if eval_avg_s > 0 else 0.0
        build_ms = 1e3 * tA / (a.steps * P)
        out = {
            "metric": "k-point evals/sec (H(k)+eig)", "value": kps, "unit": "k-points/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * tA / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic PTR grid over the reference's svo_hr.dat example coefficients (device resident)",
            "config": {"workload": f"BASELINE configs[3] per-GPU share: SVO 3-band Wannier H(k)+eig on a {npt}^3 PTR grid (FBZ) "
                                   f"+ fused DOS sweep of {a.omegas_per_rank} omega per GPU (256 at 8 GPUs), eta={a.eta}; "
                                   f"one step = {P} passes over the grid",
                       "npt": npt, "nk_per_gpu": nk, "passes_per_step": P, "n_bands": 3, "n_R": 1331,
                       "omegas_per_gpu": a.omegas_per_rank,
                       "rule_layout": ("hermitian-compact: upper triangle of H(k) + eigenvalues, 96 B per k-point (the host mirror's default "
                                       "for Hermitian series; exports and rule sums bit-identical to the full layout)") if compact else
                                      "reference: full H(k) + eigenvalues, 168 B per k-point",
                       "parallelism": f"omega-sharded x{world}, coefficient+rule replicas"},
            "n_ranks_seen": n_ranks_seen, "backend": ("rccl(nccl)" if backend == "nccl" else backend) if distributed else None,
            "rehearsal_ranks_share_one_gpu": bool(rehearsal),
            "value_reference_layout": ref_layout["value"] if ref_layout else None,  # the same metric with the full 168-B layout of rounds 1-2
            "timed_region_seconds": tA, "ms_per_pass": build_ms,
            "ms_per_step_min_median_max": [float(step_ms.min()), float(np.median(step_ms)), float(step_ms.max())],
            "rule_buffer": {"base_address": hex(base_addr), "bytes": rule_bytes, "base_mod_2MiB": base_addr % (2 << 20),
                            "base_mod_4KiB": base_addr % 4096},
            "dos_points_per_sec": world * len(mine) / tB,
            "dos_points_per_sec_eigcached": world * len(mine) / tBe,
            "ms_per_sweep": 1e3 * tB,
            "kpoint_omega_per_sec": world * len(mine) * nk / tB,
            "job_seconds_256_omega": min([t_n1] + ([jobs["omega_sharded"]["seconds"], jobs["k_sharded"]["seconds"]] if jobs else [])),
            "speedup_vs_n1": (t_n1 / min(jobs["omega_sharded"]["seconds"], jobs["k_sharded"]["seconds"])) if jobs else 1.0,
            "job_256_omega": {"what": f"fixed job (strong scaling): rule build on the {npt}^3 grid + DOS at 256 omega + the collective + "
                                      "results on the host; n1 = the same job on rank 0 alone in this run",
                              "seconds_n1": t_n1, "sharded": jobs,
                              "amdahl_note": f"omega-sharding replicates the {build_ms:.3f} ms build on every rank: speed-up <= "
                                             f"{1e3 * t_n1 / build_ms:.1f}x whatever N; k-sharding divides build and scan alike and is bounded "
                                             "by the all_reduce latency only"},
            "job_256_omega_fine_grid": big_job,
            "speedup_vs_n1_fine_grid": (big_job or {}).get("k_sharded", {}).get("speedup_vs_n1") if isinstance(big_job, dict) else None,
            "iai_sweep_432_omega": iai_job,
            "iai_config5_sharded": c5_job,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(npt, compact),
                         "compute_floor_note": ("the same kernel storing eigenvalues only (no H planes) takes "
                                                f"{ref_layout['eigenvalues_only_avg_launch_ms']:.4f} ms: with the compact layout the H stores are almost "
                                                "hidden behind the Fourier sums and the eigensolves, the kernel sits between its f64-VALU floor and its HBM-write floor")
                                               if ref_layout else None,
                         "store_pattern_note": "reference layout: bare store pattern of this kernel's tiled planar layout without any compute "
                                               "(tools/micro/placement.hip, profiles/r02_placement_microbench.txt): 6.2 TB/s = 5.83 TB/s of useful bytes with temporal stores, "
                                               "5.2-5.6 TB/s with the non-temporal stores the kernel needs (allocation-dependent); a linear memset 8.2 TB/s",
                         "traffic_note": f"HBM bytes per launch from rocprofv3 PMC passes (profiles/{'r05_traffic_compact' if compact else 'r03_traffic'}.json); "
                                         f"algorithmic bytes per launch = nk*{bpk}",
                         "reference_layout": ref_layout,
                         # flat copies: the driver's parsed view keeps scalars of this block and drops nested ones
                         "reference_layout_frac": ref_layout["frac"] if ref_layout else None,
                         "reference_layout_avg_launch_ms": ref_layout["avg_launch_ms"] if ref_layout else None,
                         "reference_layout_achieved_GBs": ref_layout["achieved_GBs"] if ref_layout else None,
                         "reference_layout_bytes_per_kpoint": ref_layout["algorithmic_bytes_per_kpoint"] if ref_layout else None,
                         "survey_8d_accounting_note": ("SURVEY 8(d) prices a k-point at 168 B because the reference stores the full matrix; this "
                                                       "launch writes the 96 B a Hermitian matrix + its eigenvalues need, so `achieved` and `frac` "
                                                       "use 96 B (168 B x this rate would exceed the HBM peak and is not claimed); the reference "
                                                       "layout's own time and fraction are in `reference_layout`") if compact else None,
                         "kernel": "eval_grid_kernel<3> (Fourier-eval + fused eig)",
                         "algorithmic_bytes_per_kpoint": bpk,
                         "avg_launch_ms": eval_ms / max(eval_n, 1), "launches": eval_n,
                         "contract_avg_ms": con_ms / max(con_n, 1),
                         "reduce_avg_ms": red_ms / max(red_n, 1),
                         # Hermitian rule: the scan reads the upper triangle only, n^2 doubles per k-point
                         "reduce_read_GBs": nk * (8 * n * n) / ((red_ms / max(red_n, 1)) * 1e-3) / 1e9 if red_n else None},
        }
        if world > 1:
            # the measured speed-ups of this run beside the one-GPU model's predictions for 8 ranks (profiles/scaling_model_latest.json,
            # written by tools/save_scaling_model.py from an N = 1 line): the first real multi-GPU run reads as a check of the model
            try:
                with open(os.path.join(ROOT, "profiles", "scaling_model_latest.json")) as fh:
                    sm = json.load(fh)
                meas = {"job_256_omega_k_sharded": (jobs or {}).get("k_sharded", {}).get("speedup_vs_n1"),
                        "job_256_omega_fine_grid_k_sharded": ((big_job or {}).get("k_sharded") or {}).get("speedup_vs_n1") if isinstance(big_job, dict) else None,
                        "iai_config5_one_solve_sharded": (c5_job or {}).get("speedup_vs_n1") if isinstance(c5_job, dict) else None,
                        "iai_sweep_432_omega_sharded": (iai_job or {}).get("speedup_vs_n1") if isinstance(iai_job, dict) else None}
                out["scaling_model_check"] = {"world": world, "note": "model = shares of 8 virtual ranks timed on ONE GPU (MODEL, NOT MEASUREMENT); it "
                                              "predicts world = 8 only -- at other sizes the columns are not comparable",
                                              "jobs": {k: {"measured_speedup_vs_n1": v, "model_predicted_speedup_8_at_30us": (sm.get(k) or {}).get("predicted_speedup_8_at_30us"),
                                                           "model_predicted_speedup_8_at_measured_1rank_latency": (sm.get(k) or {}).get("predicted_speedup_8_at_measured_1rank_latency")}
                                                       for k, v in meas.items()}}
            except Exception as e:
                out["scaling_model_check"] = {"error": repr(e)}
        if world == 1 and not a.no_extras and not a.force_dist:
            extras(a, abz, L, s, ctx, out, nk)
            if not a.no_scaling_model:
                try:
                    dev.drop_rules()  # the model's jobs build their own rules (the 400^3 one needs the room)
                    out["scaling_model"] = scaling_model(a, abz, L, torch, s, ctx, dev, npt, local)
                except Exception as e:
                    out["scaling_model"] = {"error": repr(e)}
        if not a.no_cpu and world == 1 and not a.force_dist:  # the CPU leg is timed at N = 1 only
            try:
                cb = cpu_baseline(npt, s, a.eta)
                out["cpu_baseline"] = cb
                if cb:
                    cpu_job = cb["build_seconds"] + 256 * cb["scan_seconds_per_omega"]
                    out["dos_sweep_256_omega"] = {"gpu_seconds": t_n1, "cpu_port_seconds": cpu_job, "speedup": cpu_job / t_n1,
                                                  "note": f"rule build + 256-omega scan of the same {npt}^3 grid: GPU measured end to end; CPU port "
                                                          f"build measured on this grid, its scan measured for 32 of the 256 omegas (per-omega cost is constant) x 8"}
                    if not a.no_extras:
                        cpu_baseline_legs(out, abz, s, npt, a.eta, cb["cores"], a.c5_abstol)
            except Exception as e:  # the baseline never blocks the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(flat_first(out)), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def flat_first(out):
    """The line with the contract's keys first and, right after them, flat scalar copies of the numbers a reader of the
    driver's record needs (its parsed view keeps a limited list of top-level keys and no nested block): the Section-8(d)
    layout's rate and fraction, the 8-GPU model's predictions, the CPU ratios of the other legs."""
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"]
    flat = {}
    rl = (out.get("roofline") or {}).get("reference_layout") or {}
    flat["value_reference_layout"] = out.get("value_reference_layout")
    flat["reference_layout_frac"] = rl.get("frac")
    flat["reference_layout_avg_launch_ms"] = rl.get("avg_launch_ms")
    sm = out.get("scaling_model") or {}
    for key, name in (("job_256_omega_k_sharded", "predicted_speedup_8_job256"), ("job_256_omega_fine_grid_k_sharded", "predicted_speedup_8_fine_grid"),
                      ("iai_config5_one_solve_sharded", "predicted_speedup_8_config5"), ("iai_sweep_432_omega_sharded", "predicted_speedup_8_iai_sweep")):
        m = sm.get(key) or {}
        flat[name] = m.get("predicted_speedup_8_at_30us")
    for blk, name in (("ggr", "ggr_build_frac_of_hbm"),):
        flat[name] = ((out.get(blk) or {}).get("build_roofline") or {}).get("frac") if isinstance(out.get(blk), dict) else None
    for blk in ("iai_config5", "ggr"):
        cb = (out.get(blk) or {}).get("cpu_baseline") if isinstance(out.get(blk), dict) else None
        if isinstance(cb, dict):
            flat[f"{blk}_speedup_vs_cpu_port"] = cb.get("gpu_over_cpu")
    try:  # the round-5 legs as flat scalars too
        gb = out.get("ggr_bands_5_to_32") or {}
        flat["ggr_bands16_seconds_24cubed"] = (gb.get("bands16_M7") or {}).get("seconds")
        flat["ggr_bands32_seconds_24cubed"] = (gb.get("bands32_M5") or {}).get("seconds")
        flat["bands48_rule_H_and_eig_seconds_24cubed"] = ((out.get("bands48_64_fixed_grids") or {}).get("bands48") or {}).get("rule_24cubed_H_and_eig_seconds")
        flat["bands48_ggr_build_seconds_24cubed"] = ((out.get("bands48_64_fixed_grids") or {}).get("bands48") or {}).get("ggr_build_24cubed_seconds")
        ex = ((out.get("iai_example") or {}).get("FBZ") or {})
        flat["iai_example_fbz_seconds"] = ex.get("seconds")
        flat["iai_example_speedup_vs_cpu_port"] = (ex.get("cpu_baseline") or {}).get("gpu_over_cpu")
        flat["iai_example_same_numevals_as_cpu_port"] = (ex.get("cpu_baseline") or {}).get("same_numevals_as_gpu")
        flat["ggr_build_kernel_ms"] = (out.get("ggr") or {}).get("build_kernel_ms")
    except Exception:
        pass
    res = {k: out[k] for k in head if k in out}
    res.update({k: v for k, v in flat.items()})
    res.update({k: v for k, v in out.items() if k not in res})
    return res


def big_dos_job(a, abz, L, torch, dist, dev, rank, world, local, distributed, cdev, barrier, timed, WANT=None):
    """Fixed job with enough work to divide by 8: the 256-omega DOS sweep at eta = 0.01 eV (the value of
    aps_example/aps_example.jl:29) on the --big-npt^3 full-BZ grid (400^3 = 6.4e7 k-points, 10.8 GB of rule values:
    a PTR grid that resolves eta = 0.01), k-sharded: 1/N slab per rank, all 256 omega, one all_reduce(sum).
    N = 1 reference: rank 0 builds and scans the whole grid alone in the same run."""
    npt, eta = a.big_npt, 0.01
    om256 = np.linspace(10.0, 15.0, 256)
    om_dev = torch.from_numpy(om256).to(f"cuda:{local}")
    reps = 5
    WANT = WANT or (L.WANT_H | L.WANT_EIG)
    res = {"what": f"rule build on the {npt}^3 grid + DOS at 256 omega, eta = {eta} + collective + results on the host",
           "npt": npt, "nk": npt**3, "eta": eta, "rule_bytes_n1": npt**3 * (96 if WANT & L.WANT_H_COMPACT else 168)}
    ref = None
    if rank == 0:
        full = abz.DeviceRule(dev, npt, None, WANT)
        out = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")

        def job1():
            full.rebuild()
            full.reduce_device(L.F_DOS, [eta], om_dev.data_ptr(), 256, out.data_ptr())
            return out[:, 0].cpu().numpy()
        t1, ref = timed(job1, reps, everyone=False)
        full.close()
        res["seconds_n1"] = t1
        res["kpoint_omega_per_sec_n1"] = npt**3 * 256 / t1
    barrier()
    if distributed:
        dev.kshard = (rank, world)
        rk = abz.DeviceRule(dev, npt, None, WANT)
        dev.kshard = None
        acc = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")

        def jobk():
            rk.rebuild()
            rk.reduce_device(L.F_DOS, [eta], om_dev.data_ptr(), 256, acc.data_ptr())
            if cdev == "cuda":
                dist.all_reduce(acc)
                return acc[:, 0].cpu().numpy()
            c = acc.cpu()
            dist.all_reduce(c)
            return c[:, 0].numpy()
        tk, rkv = timed(jobk, reps)
        tt = torch.tensor([tk], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rk.close()
        if rank == 0:
            tk = float(tt.cpu()[0])
            err = float(np.abs(rkv - ref).max() / np.abs(ref).max())
            assert err < 1e-11, err
            res["k_sharded"] = {"seconds": tk, "speedup_vs_n1": res["seconds_n1"] / tk, "max_rel_diff_vs_n1": err,
                                "kpoint_omega_per_sec": npt**3 * 256 / tk}
    return res if rank == 0 else None


def virtual_rank_iai(abz, L, W, make_series, eta, omega, abstol, lims):
    """ONE IAI solve sharded over W virtual ranks on ONE GPU, the ranks' shares computed one after another.

    Every virtual rank is a host thread with its own device context and its own copy of the series, wired to the library's
    exchange hook (abz_iai_set_exchange(rank = r, world = W)) exactly like a rank process of `dist.iaishard`; the all-gather
    is a shared host buffer.  A token serialises the ranks: rank r holds it from the moment a round's gather is complete
    until it has computed its own share of the next round and written it -- so the wall time a rank holds the token is its
    per-round cost (replicated host bookkeeping + its kernels) with the GPU to itself.  Returns per-rank seconds, exchanges,
    the value and numevals (identical on every rank by construction)."""
    import ctypes as C
    import threading
    from autobzcore.jl_amd import solver as S
    cond = threading.Condition()
    st = {"turn": 0, "round": 0, "written": 0, "shared": None, "err": None}
    held = [0.0] * W
    t_got = [0.0] * W
    rounds = [0] * W
    results = [None] * W

    def take(r):
        with cond:
            cond.wait_for(lambda: st["turn"] == r or st["err"])
            t_got[r] = time.perf_counter()

    def make_cb(r):
        def cb(user, buf, per):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(W * per,))
                with cond:
                    if st["shared"] is None or len(st["shared"]) != W * per:
                        st["shared"] = np.zeros(W * per)
                    st["shared"][r * per:(r + 1) * per] = arr[r * per:(r + 1) * per]
                    st["written"] += 1
                    held[r] += time.perf_counter() - t_got[r]
                    rounds[r] += 1
                    my_round = st["round"]
                    if st["written"] == W:  # the last rank of the round completes the gather
                        st["written"] = 0
                        st["round"] += 1
                        st["full"] = st["shared"].copy()
                        st["shared"] = None
                        st["turn"] = 0
                    else:
                        st["turn"] = r + 1
                    cond.notify_all()
                    cond.wait_for(lambda: st["round"] > my_round or st["err"])
                    arr[:] = st["full"]
                    cond.wait_for(lambda: st["turn"] == r or st["err"])  # my turn for the next round
                    t_got[r] = time.perf_counter()
                    if st["err"]:
                        return 1
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                with cond:
                    st["err"] = repr(e)
                    cond.notify_all()
                return 1
        return cb

    bar = threading.Barrier(W)

    def worker(r):
        try:
            ctx = L.Context()
            sr = make_series()
            dev = sr.device(ctx)
            cb = L.EXCHANGE_FN(make_cb(r))
            L.check(L.lib().abz_iai_set_exchange(dev.h, cb, None, r, W))
            f = abz.FourierIntegrand(abz.DOSIntegrand(), sr, eta)
            for rep in range(2):
                bar.wait()
                if r == 0:
                    with cond:
                        st.update({"turn": 0, "round": 0, "written": 0, "shared": None})
                bar.wait()
                held[r] = 0.0
                rounds[r] = 0
                take(r)
                # (the first pass, at a looser tolerance, sizes the pools and staging buffers of the rank's context)
                u, err, nev, _ = S._iai_device(f, dev, lims, f.f.p.merge(abz.MixedParameters(omega)), abstol * (30.0 if rep == 0 else 1.0), 0.0, 2**62)
                with cond:  # the tail after the last exchange (every rank finishes the solve on its own)
                    held[r] += time.perf_counter() - t_got[r]
                    st["turn"] = r + 1
                    cond.notify_all()
                results[r] = (u, err, nev)
            L.check(L.lib().abz_iai_set_exchange(dev.h, None, None, 0, 1))
            dev.close()
            ctx.close()
        except Exception as e:
            with cond:
                st["err"] = repr(e)
                cond.notify_all()
            try:
                bar.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if st["err"]:
        raise RuntimeError(st["err"])
    assert all(res == results[0] for res in results), "virtual ranks disagree"
    return {"per_rank_s": held, "exchanges": rounds[0], "u": results[0][0], "numevals": results[0][2]}


def scaling_model(a, abz, L, torch, s, ctx, dev, npt, local):
    """An 8-GPU scaling MODEL from one-GPU shard timings: every virtual rank's share of each sharded job
    is timed alone on this GPU, one after another; predicted time at W = 8 = the slowest share + collectives x latency."""
    W = 8
    out = {"label": "MODEL, NOT MEASUREMENT: shares of W = 8 virtual ranks timed one after another on ONE MI355X; predicted_s = "
                    "max(per_rank_s) + collectives * latency; predicted_speedup_8 = n1_s / predicted_s.  The collective latency of an "
                    "8-GPU xGMI node was not measurable here: the model is evaluated for the RCCL call cost measured at ONE rank (a "
                    "lower bound) and for an assumed 30 us per small-message collective", "world": W}
    # --- cost of one small RCCL collective at ONE rank (call + kernel on the stream; no wire)
    lat = None
    try:
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{local}"))
        t = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")
        g = torch.zeros(32, 2, dtype=torch.float64, device=f"cuda:{local}")
        go = [torch.zeros_like(g)]
        for _ in range(20):
            dist.all_reduce(t)
            dist.all_gather(go, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(t)
            torch.cuda.synchronize()
        lat_ar = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_gather(go, g)
            torch.cuda.synchronize()
        lat_ag = (time.perf_counter() - t0) / 200
        lat = {"all_reduce_4KB_s": lat_ar, "all_gather_512B_s": lat_ag, "ranks": 1,
               "note": "RCCL at world size 1 incl. the stream synchronisation that follows it in the jobs"}
        if own:
            dist.destroy_process_group()
    except Exception as e:
        lat = {"error": repr(e)}
    out["rccl_latency_s"] = lat
    lat1 = (lat or {}).get("all_reduce_4KB_s") or 20e-6
    ASSUMED = 30e-6

    def predict(n1, per, ncoll):
        mx, mean = max(per), sum(per) / len(per)
        return {"n1_s": n1, "per_rank_s": per, "imbalance_max_over_mean": mx / mean, "collectives": ncoll,
                "predicted_s_at_measured_1rank_latency": mx + ncoll * lat1, "predicted_speedup_8_at_measured_1rank_latency": n1 / (mx + ncoll * lat1),
                "predicted_s_at_30us": mx + ncoll * ASSUMED, "predicted_speedup_8_at_30us": n1 / (mx + ncoll * ASSUMED)}

    om256 = np.linspace(10.0, 15.0, 256)
    om_dev = torch.from_numpy(om256).to(f"cuda:{local}")
    acc = torch.zeros(256, 2, dtype=torch.float64, device=f"cuda:{local}")
    WANT = L.WANT_H | L.WANT_EIG

    def dos_job(grid, eta, reps):
        def run(rule):
            def job():
                rule.rebuild()
                rule.reduce_device(L.F_DOS, [eta], om_dev.data_ptr(), 256, acc.data_ptr())
                return acc[:, 0].cpu().numpy()
            for _ in range(3):
                job()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = job()
            return (time.perf_counter() - t0) / reps, r
        full = dev.rule(grid, None, WANT)
        n1, ref = run(full)
        dev.drop_rules()
        per, tot = [], np.zeros(256)
        for r in range(W):
            dev.kshard = (r, W)
            rk = abz.DeviceRule(dev, grid, None, WANT)
            dev.kshard = None
            t, v = run(rk)
            rk.close()
            per.append(t)
            tot += v
        assert np.abs(tot - ref).max() <= 1e-11 * np.abs(ref).max()
        return predict(n1, per, 1)

    try:
        m = dos_job(npt, a.eta, 40)
        m["what"] = f"(a) 256-omega DOS job on the {npt}^3 grid, k-sharded (slab of the outermost variable per rank, one all_reduce of 256 sums)"
        out["job_256_omega_k_sharded"] = m
    except Exception as e:
        out["job_256_omega_k_sharded"] = {"error": repr(e)}
    if not a.no_big_job:
        try:
            m = dos_job(a.big_npt, 0.01, 4)
            m["what"] = f"(b) the same at eta = 0.01 on the {a.big_npt}^3 grid"
            out["job_256_omega_fine_grid_k_sharded"] = m
        except Exception as e:
            out["job_256_omega_fine_grid_k_sharded"] = {"error": repr(e)}
    if not a.no_iai:
        # (c) config 5 as one solve sharded over the ranks: the driver's dealing rule, per-round exchange
        try:
            s16 = abz.synthetic_wannier()
            f16 = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
            bz16 = abz.load_bz(abz.FBZ(), np.eye(3))
            lims = bz16.lims
            at16 = a.c5_abstol / abs(np.linalg.det(bz16.B))  # what do_solve hands the nested quadrature (src/brillouin.jl:340-342)
            from autobzcore.jl_amd import solver as S
            pm16 = f16.f.p.merge(abz.MixedParameters(0.2))
            S._iai_device(f16, s16.device(), lims, pm16, 10.0, 0.0, 2**62)
            t0 = time.perf_counter()
            u1, _, nev1, _ = S._iai_device(f16, s16.device(), lims, pm16, at16, 0.0, 2**62)
            n1 = time.perf_counter() - t0
            vr = virtual_rank_iai(abz, L, W, abz.synthetic_wannier, 0.05, 0.2, at16, lims)
            assert vr["u"] == u1 and vr["numevals"] == nev1, (vr["u"], u1)
            m = predict(n1, vr["per_rank_s"], vr["exchanges"])
            m.update({"what": "(c) config 5 (synthetic 16-band IAI, abstol %g) as ONE solve: every round's innermost integrals dealt to the "
                              "ranks in blocks of 64 nodes (owner(t) = (t >> 6) mod W), one all-gather per round; a rank's seconds = its "
                              "replicated host bookkeeping + its own kernels, alone on the GPU" % a.c5_abstol,
                      "exchanges": vr["exchanges"], "numevals": vr["numevals"], "bit_identical_to_n1": True})
            out["iai_config5_one_solve_sharded"] = m
        except Exception as e:
            out["iai_config5_one_solve_sharded"] = {"error": repr(e)}
        # (d) the 432-omega IAI sweep of the reference's example, round-robin like batchparam
        try:
            fiai = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
            bzc = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
            sol_iai = abz.IntegralSolver(fiai, bzc, abz.IAI(), abstol=1e-3)
            om432 = np.linspace(10.0, 15.0, 432)
            abz.batchsolve(sol_iai, om432)
            t0 = time.perf_counter()
            r1 = abz.batchsolve(sol_iai, om432)
            n1 = time.perf_counter() - t0
            per = []
            got = np.zeros(432)
            for r in range(W):
                abz.batchsolve(sol_iai, om432[r::W])
                t0 = time.perf_counter()
                got[r::W] = abz.batchsolve(sol_iai, om432[r::W])
                per.append(time.perf_counter() - t0)
            assert np.array_equal(got, np.asarray(r1, dtype=float))
            m = predict(n1, per, 1)
            m["what"] = "(d) 432-omega IAI sweep on the cubic IBZ, rank r solves omega[r::8] (batchparam), one all_gather of 54 results"
            out["iai_sweep_432_omega_sharded"] = m
        except Exception as e:
            out["iai_sweep_432_omega_sharded"] = {"error": repr(e)}
    return out


def extras(a, abz, L, s, ctx, out, nk):
    """N = 1 only, outside every timed region of the primary metric: end-to-end times of the other configs."""
    if not a.no_iai:
        # one IAI solve of the reference's own example (aps_example/aps_example.jl:29-34, eta = 0.01 eV, abstol 1e-3)
        try:
            fiai = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
            out["iai_example"] = {}
            for kind, bzk in (("CubicSymIBZ", abz.CubicSymIBZ()), ("FBZ", abz.FBZ())):
                bz = abz.load_bz(bzk, 3.85856 * np.eye(3))
                prob = abz.IntegralProblem(fiai, bz, abz.MixedParameters(12.5))
                abz.solve(prob, abz.IAI(), abstol=1e-3)  # warm-up (allocations)
                t0 = time.perf_counter()
                sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1e-3)
                dt = time.perf_counter() - t0
                out["iai_example"][kind] = {"u": sol.u, "resid": sol.resid, "numevals": sol.numevals, "seconds": dt,
                                            "nodes_per_sec": sol.numevals / dt}
        except Exception as e:
            out["iai_example"] = {"error": str(e)}
    # GGR (north star: "dos_ggr.jl's eigenvalue sweep"; ref src/dos_ggr.jl:14-65): the fused eigenvalue + velocity build on the
    # bench grid and the scan of 256 energies over it, each with its own HIP-event kernel time and roofline fraction
    try:
        dev0 = s.device()
        n, d = s.n, s.d
        rg = abz.DeviceRule(dev0, a.npt, None, L.WANT_EIG | L.WANT_VEL)
        for _ in range(5):
            rg.rebuild()
        ctx.sync()
        ids = [L.K_CONTRACT, L.K_EVAL, L.K_EIG, L.K_GGRBUILD]
        ctx.prof_enable(True, kernels=ids)
        ctx.prof_reset()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            rg.rebuild()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        parts = {k: ctx.prof_read(k) for k in ids}
        ctx.prof_enable(False)
        kb_ms = parts[L.K_GGRBUILD][0] / max(parts[L.K_GGRBUILD][1], 1)
        bytes_k = 8 * n * (1 + d)  # e and v out per k-point; + 8 (weight) only on symmetric rules
        g = {"npt": a.npt, "kpoints": nk, "build_seconds": dt, "build_kpoints_per_sec": nk / dt,
             "build_kernel_ms": kb_ms, "build_launches_per_rebuild": {int(k): parts[k][1] // reps for k in ids if parts[k][1]},
             "build_contract_ms": parts[L.K_CONTRACT][0] / reps,
             "build_algorithmic_bytes_per_kpoint": bytes_k,
             "build_roofline": {"bound": "hbm (write) / f64 VALU", "achieved_GBs": nk * bytes_k / (kb_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                                "frac": nk * bytes_k / (kb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "VALU-bound in practice (profiles/r03_ggr_pmc_summary.json: ~1000 f64 instructions per k-point, "
                                        "stores overlap); counter traffic 1.05x algorithmic"},
             "unfused_round2_build_seconds": 1.37e-3}
        Es = np.linspace(10.0, 15.0, 256)
        rg.ggr(Es)
        ctx.prof_enable(True, kernels=[L.K_GGR])
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(20):
            dos = rg.ggr(Es)
        dts = (time.perf_counter() - t0) / 20
        ms, nl = ctx.prof_read(L.K_GGR)
        ctx.prof_enable(False)
        ks_ms = ms / 20
        bytes_scan = 8 * n * (1 + d) + 8
        g.update({"scan_energies": len(Es), "scan_seconds": dts, "scan_kE_per_sec": nk * len(Es) / dts, "scan_kernel_ms": ks_ms,
                  "scan_algorithmic_bytes_per_kpoint": bytes_scan,
                  "scan_roofline": {"bound": "hbm (read, one pass over the rule for all energies)",
                                    "achieved_GBs": nk * bytes_scan / (ks_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                                    "frac": nk * bytes_scan / (ks_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                  "all_pairs_round2_scan_seconds": 6.1e-3, "dos_mid": float(dos[128])})
        rg.close()
        out["ggr"] = g
    except Exception as e:
        out["ggr"] = {"error": str(e)}
    # configs 2 and 3 end to end (host loop + kernels + transfers; not the primary metric)
    try:
        cfg = {}
        tb = abz.tb_integer(3)
        sol2 = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), tb, 0.1),
                                  abz.load_bz(abz.FBZ(), np.eye(3)), abz.PTR(npt=64))
        sol2(0.5)
        t0 = time.perf_counter()
        for _ in range(10):
            tb.device().drop_rules()  # cold: build the 64^3 rule every time
            u2 = sol2(0.5)
        cfg["config2_tb1band_64cubed_ptr"] = {"u": u2, "seconds_cold": (time.perf_counter() - t0) / 10}
        t0 = time.perf_counter()
        for _ in range(10):
            u2 = sol2(0.5)
        cfg["config2_tb1band_64cubed_ptr"]["seconds_cached_rule"] = (time.perf_counter() - t0) / 10
        for kind, bzk in (("FBZ", abz.FBZ()), ("CubicSymIBZ", abz.CubicSymIBZ())):
            sol3 = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, a.eta),
                                      abz.load_bz(bzk, 3.85856 * np.eye(3)), abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
            s.device().drop_rules()
            t0 = time.perf_counter()
            r3 = sol3.solve_p(abz.MixedParameters(12.5))
            tfirst = time.perf_counter() - t0  # the first solve of its kind in the process (symmetric tables, staging blocks)
            s.device().drop_rules()
            t0 = time.perf_counter()
            r3 = sol3.solve_p(abz.MixedParameters(12.5))
            tc = time.perf_counter() - t0
            # cached: the rules of the grid sequence are resident (a grid beyond `keepmost` is kept from its second visit on)
            for _ in range(3):
                r3 = sol3.solve_p(abz.MixedParameters(12.5))
            tcs = []
            for _ in range(50):
                t0 = time.perf_counter()
                r3 = sol3.solve_p(abz.MixedParameters(12.5))
                tcs.append(time.perf_counter() - t0)
            cfg["config3_svo_autoptr_" + kind] = {"u": r3.u, "resid": r3.resid, "numevals": r3.numevals, "grids": r3.extra.get("npt"),
                                                 "seconds_first_in_process": tfirst, "seconds_cold": tc,
                                                 "seconds_cached_rules": sorted(tcs)[len(tcs) // 2],
                                                 "note": "one abz_autoptr_solve_many call per solve (grid sequence, error test, numevals in "
                                                         "the library); first = incl. the symmetric-rule tables and staging blocks of the "
                                                         "context; cold = every rule dropped, tables cached; cached = median of 50"}
        out["configs_end_to_end"] = cfg
    except Exception as e:
        out["configs_end_to_end"] = {"error": str(e)}
    # store-free rule values (abz_ptr_sum): the 1000^3 grid (10^9 k-points, 168 GB if it were stored)
    try:
        dev0 = s.device()
        dev0.ptr_sum(300, L.F_DOS, [a.eta], [12.5])
        sf = {}
        for nw in (1, 8):
            om = np.linspace(12.0, 13.0, nw)
            t0 = time.perf_counter()
            dev0.ptr_sum(1000, L.F_DOS, [a.eta], om)
            dt = time.perf_counter() - t0
            sf[f"seconds_{nw}_omega"] = dt
            sf[f"kpoints_per_sec_{nw}_omega"] = 1e9 / dt
        sol_s = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01), abz.load_bz(abz.FBZ(), 3.85856 * np.eye(3)),
                                   abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
        t0 = time.perf_counter()
        r_s = sol_s.solve_p(abz.MixedParameters(12.5))
        sf["autoptr_fbz_eta0.01"] = {"u": r_s.u, "resid": r_s.resid, "numevals": r_s.numevals,
                                     "seconds": time.perf_counter() - t0}
        sf["note"] = "DOS on the 1000^3 full-BZ grid without materialising H(k): Fourier evaluation feeds the integrand"
        out["store_free_1000cubed"] = sf
    except Exception as e:
        out["store_free_1000cubed"] = {"error": str(e)}
    s16 = abz.synthetic_wannier()
    if not a.no_iai:
        # config 5: synthetic 16-band model, IAI on the full BZ
        try:
            f16 = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
            prob = abz.IntegralProblem(f16, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
            abz.solve(prob, abz.IAI(), abstol=10.0, reltol=0.0)  # warm-up
            t0 = time.perf_counter()
            sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=a.c5_abstol, reltol=0.0)
            dt = time.perf_counter() - t0
            # flops per inner node.  SURVEY 8d counts the series as 8 n^2 M and the elimination as 8 n^3; the round-3 kernel folds
            # +f with -f (8 n^2 F for the series) and still inverts fully; the DOS needs the trace of the inverse only
            # (~2/3 of an elimination by LU + selected inversion -- not cheaper in the row layout)
            fl_impl = 8 * 16 * 16 * 6 + 8 * 16**3
            fl_r2 = 8 * 16 * 16 * 13 + 8 * 16**3
            fl_need = 8 * 16 * 16 * 6 + (2 * 8 * 16**3) // 3
            tf = lambda fl: sol.numevals * fl / dt / 1e12
            out["iai_config5"] = {"model": f"synthetic 16-band, 2197 R (seed 20240601), DOS eta=0.05 omega=0.2, IAI on the FBZ",
                                  "abstol": a.c5_abstol, "abstol_stated_by_SURVEY_8d": 1e-3,
                                  "u": sol.u, "resid": sol.resid, "numevals": sol.numevals, "seconds": dt,
                                  "nodes_per_sec": sol.numevals / dt,
                                  "flops_per_node": {"implemented": fl_impl, "survey_8d_accounting": fl_r2, "algorithmic_need": fl_need},
                                  "f64_tflops": tf(fl_impl),
                                  "frac_of_f64_peak": tf(fl_impl) / F64_PEAK_TFLOPS,
                                  "frac_of_f64_peak_survey_8d_accounting": tf(fl_r2) / F64_PEAK_TFLOPS,
                                  "frac_of_f64_peak_algorithmic_need": tf(fl_need) / F64_PEAK_TFLOPS}
        except Exception as e:
            out["iai_config5"] = {"error": str(e)}
    # the same 16-band model on fixed grids: store-free PTR sums, a cached rule with eigenvalues, its scan
    try:
        dev16 = s16.device()
        om16 = np.linspace(-1.0, 1.0, 16)
        b16 = {}
        dev16.ptr_sum(96, L.F_DOS, [0.05], om16)
        t0 = time.perf_counter()
        dev16.ptr_sum(96, L.F_DOS, [0.05], om16)
        dt = time.perf_counter() - t0
        b16["store_free_96cubed_16_omega"] = {"seconds": dt, "kpoint_omega_per_sec": 96**3 * 16 / dt,
                                              "algorithm": "one Householder tridiagonalisation per k-point, tr G = p'/p by the three-term "
                                                           "recurrence per omega (round 1-2a: a 16x16 complex Gauss-Jordan inversion per (k, omega), "
                                                           "0.60 G (k,omega)/s = 0.27 of the f64 peak)"}
        t0 = time.perf_counter()
        dev16.ptr_sum(96, L.F_DOS, [0.05], om16[:1])
        dt1 = time.perf_counter() - t0
        fl = 8 * 16 * 16 * 13 + 8 * 16**3  # one omega: series + one inversion per k-point
        b16["store_free_96cubed_1_omega"] = {"seconds": dt1, "kpoints_per_sec": 96**3 / dt1, "f64_tflops": 96**3 * fl / dt1 / 1e12,
                                             "frac_of_f64_peak": 96**3 * fl / dt1 / 1e12 / F64_PEAK_TFLOPS}
        def build_time(want):
            rr = abz.DeviceRule(dev16, 48, None, want)
            for _ in range(3):
                rr.rebuild()
            dev16.ctx.sync()
            t0_ = time.perf_counter()
            for _ in range(10):
                rr.rebuild()
            dev16.ctx.sync()
            return rr, (time.perf_counter() - t0_) / 10
        r16, dt = build_time(L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT)  # the host mirror's layout for a Hermitian series
        b16["rule_48cubed_H_and_eig"] = {"seconds": dt, "kpoints_per_sec": 48**3 / dt, "bytes_per_kpoint": 8 * (16 * 16 + 16),
                                         "layout": "upper triangle of H(k) + eigenvalues (ABZ_WANT_H_COMPACT), stores through an LDS tile"}
        for name, want in (("rule_48cubed_H_and_eig_reference_layout", L.WANT_H | L.WANT_EIG), ("rule_48cubed_H_only", L.WANT_H | L.WANT_H_COMPACT),
                           ("rule_48cubed_eig_only", L.WANT_EIG)):
            rx, dtx = build_time(want)
            rx.close()
            b16[name] = {"seconds": dtx, "kpoints_per_sec": 48**3 / dtx}
        r16.reduce(L.F_DOS, [0.05], om16)
        t0 = time.perf_counter()
        r16.reduce(L.F_DOS, [0.05], om16)
        dt = time.perf_counter() - t0
        b16["rule_scan_16_omega"] = {"seconds": dt, "kpoint_omega_per_sec": 48**3 * 16 / dt}
        r16.close()
        out["bands16_fixed_grids"] = b16
    except Exception as e:
        out["bands16_fixed_grids"] = {"error": str(e)}
    # GGR builds of 5...32 bands (round 5: one row-layout kernel, kernels_ggr_rows.hip; ref src/dos_ggr.jl:14-44 calls LAPACK there)
    try:
        gb = {"what": "rule build with eigenvalues + band velocities (abz_ptr_rule_build, ABZ_WANT_EIG | ABZ_WANT_VEL) of synthetic Hermitian "
                      "models on the 24^3 full-BZ grid: contraction chains of H and of the d derivative families + one fused kernel; "
                      "round 4 (eigenvectors and every dH/dk_j through HBM; wave-per-node above 16 bands): 16 bands 1.54 ms, 17 bands 13.8 ms, 24 bands 29.5 ms",
              "npt": 24}
        for nb, rmax in ((5, 2), (8, 2), (12, 2), (16, 3), (16, 6), (17, 2), (24, 2), (32, 2)):
            sg = abz.synthetic_wannier(n=nb, rmax=rmax, seed=7) if (nb, rmax) != (16, 6) else s16
            rr = abz.DeviceRule(sg.device(), 24, None, L.WANT_EIG | L.WANT_VEL)
            for _ in range(3):
                rr.rebuild()
            ctx.sync()
            ts_ = []
            for _ in range(10):
                t0 = time.perf_counter()
                rr.rebuild()
                ctx.sync()
                ts_.append(time.perf_counter() - t0)
            rr.close()
            if sg is not s16:
                sg.device().close() if hasattr(sg.device(), "close") else None
            gb[f"bands{nb}_M{2 * rmax + 1}"] = {"seconds": min(ts_), "nodes_per_sec": 24**3 / min(ts_)}
        out["ggr_bands_5_to_32"] = gb
    except Exception as e:
        out["ggr_bands_5_to_32"] = {"error": repr(e)}
    # 33...64 bands (round 5, kernels_big.hip: wave-per-node Householder in LDS; ABZ_MAX_BANDS was 32)
    try:
        b48 = {"what": "synthetic Hermitian models on the 24^3 full-BZ grid: rule build with H(k) + eigenvalues, a 16-omega store-free DOS sweep, GGR build (eigenvalues + band velocities)"}
        for nb in (48, 64):
            sg = abz.synthetic_wannier(n=nb, rmax=2, seed=7)
            dg = sg.device()
            rr = abz.DeviceRule(dg, 24, None, L.WANT_H | L.WANT_EIG)
            ts_ = []
            for _ in range(4):
                t0 = time.perf_counter()
                rr.rebuild()
                ctx.sync()
                ts_.append(time.perf_counter() - t0)
            rr.close()
            om = np.linspace(-1.0, 1.0, 16)
            dg.ptr_sum(24, L.F_DOS, [0.05], om)
            t0 = time.perf_counter()
            dg.ptr_sum(24, L.F_DOS, [0.05], om)
            dt_sum = time.perf_counter() - t0
            rg = abz.DeviceRule(dg, 24, None, L.WANT_EIG | L.WANT_VEL)  # GGR build: eigenvalues + band velocities (kernels_big_vec.hip)
            tg_ = []
            for _ in range(4):
                t0 = time.perf_counter()
                rg.rebuild()
                ctx.sync()
                tg_.append(time.perf_counter() - t0)
            rg.close()
            b48[f"bands{nb}"] = {"rule_24cubed_H_and_eig_seconds": min(ts_), "kpoints_per_sec": 24**3 / min(ts_),
                                 "store_free_16_omega_seconds": dt_sum, "ggr_build_24cubed_seconds": min(tg_)}
        b48["mfma_vs_fma"] = "profiles/r05_big_series_mfma_vs_fma.txt: the level-1 GEMM on v_mfma_f64_16x16x4_f64 is 0-25 % slower than the vector form"
        out["bands48_64_fixed_grids"] = b48
    except Exception as e:
        out["bands48_64_fixed_grids"] = {"error": repr(e)}
    # 17...32 bands: the same row kernels with two nodes per wave (round 4; wave-per-node kernels before: a 100x step at 17 bands)
    try:
        b24 = {}
        rng24 = np.random.default_rng(24)
        c24 = rng24.standard_normal((5, 5, 5, 24, 24)) + 1j * rng24.standard_normal((5, 5, 5, 24, 24))
        c24 = (c24 + np.conj(np.swapaxes(c24[::-1, ::-1, ::-1], -1, -2))) / 24.0
        s24 = abz.FourierSeries(c24, period=1.0, first=(-2, -2, -2))
        dev24 = s24.device()
        r24 = abz.DeviceRule(dev24, 32, None, L.WANT_H | L.WANT_EIG)
        for _ in range(2):
            r24.rebuild()
        dev24.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            r24.rebuild()
        dev24.ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        b24["rule_32cubed_H_and_eig"] = {"seconds": dt, "kpoints_per_sec": 32**3 / dt, "before_round4_seconds": 0.0548}
        r24.close()
        om24 = np.linspace(-1.0, 1.0, 16)
        dev24.ptr_sum(32, L.F_DOS, [0.05], om24)
        t0 = time.perf_counter()
        dev24.ptr_sum(32, L.F_DOS, [0.05], om24)
        dt = time.perf_counter() - t0
        b24["store_free_32cubed_16_omega"] = {"seconds": dt, "kpoint_omega_per_sec": 32**3 * 16 / dt, "before_round4_seconds": 0.0650}
        b24["model"] = "random Hermitian 24-band series, 5^3 coefficients"
        out["bands24_fixed_grids"] = b24
    except Exception as e:
        out["bands24_fixed_grids"] = {"error": str(e)}


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank_main(a)


if __name__ == "__main__":
    main()
