#!/usr/bin/env python3
"""bench.py — measures the MSM / inner-product hot path on MI355X and prints ONE JSON line.

  python bench.py --gpus N --steps K --warmup W [--workload headline|prove|verify|msm]

BASELINE.json's metric has two halves — R1CS constraints proved/s and batch verifies/s — so the default workload
("headline") runs both on the same box: the top-level fields are the prove half (cfg3: 2^20-constraint proofs), and the
"verify" object carries the batch-verify half (cfg4: 4096 proofs of 2^14 constraints per GPU) with its own value, roofline
and cpu_baseline.  The reference's own benchmark workload is benches/r1cs_secq256k1.rs:156-190 (prove) and :201-250 (verify).

A "step" is one pass of the hot path over one batch of synthetic input.  `--gpus N` with N > 1 and no WORLD_SIZE in the
environment starts N ranks itself (torch.distributed.run, one process per GPU, before this process touches the GPU); under a
launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Units are sharded across ranks (weak scaling); the only exchange is the
all-gather of one 64-byte partial point (and a status word) per rank and step — in the verify workload, where several batches are
in flight per GPU, gathered for all steps at once at the end of the timed region.

Documented multi-GPU commands (SCALE): `bench.py --gpus N` (prove replicas + proof-sharded verify),
`bench.py --gpus N --workload verify`, `bench.py --gpus N --workload prove --shard windows --logn 22` (cfg5: one proof at a
time, Pippenger windows and the IPA folds partitioned across the ranks), `bench.py --gpus N --workload msm --shard windows`.

Only the `cpu_baseline` legs touch oracle/ (the CPU restatement), as the timed CPU reference, after the timed regions.
"""
import argparse
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DTYPE = "u32x9 (256-bit modular integers, radix 2^29)"
# Integer-VALU ceiling for one Montgomery product of the kernels' field arithmetic (csrc/fp29.cuh), from the guide's issue rates
# (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz max clock, a wave64 VALU instruction occupies its SIMD for 2 cycles; the
# guide has no row for v_mad_u64_u32 — tools/ubench.hip measures 0.95 wave-instructions per clock and CU, i.e. one per 4 cycles
# and SIMD: quarter rate) and the instruction mix of one product in the shipped code object (162 v_mad_u64_u32 + ~100 full-rate
# carry / mask / move instructions, counted in the ISA):  cycles per wave-product and SIMD = 162 * 4 + 100 * 2 = 848.
VALU_MAD64_PER_PRODUCT, VALU_OTHER_PER_PRODUCT = 162, 100
VALU_CYCLES_PER_PRODUCT = VALU_MAD64_PER_PRODUCT * 4 + VALU_OTHER_PER_PRODUCT * 2
VALU_PEAK_GMODMUL = 256 * 4 * 2.4e9 / VALU_CYCLES_PER_PRODUCT * 64 / 1e9        # 185.5 G modmul/s at the maximum clock
VALU_MEASURED_GMODMUL = 169.0      # tools/ubench.hip on this GPU (profiles/r01_ubench_radix29.txt): the clock the chip holds under this load
VALU_INSTR_PER_PRODUCT = VALU_MAD64_PER_PRODUCT + VALU_OTHER_PER_PRODUCT
# The SCALAR fields of both curves are pseudo-Mersenne since round 4 (csrc/fp29.cuh fe_pm_product): a product there is 91 v_mad_u64_u32 +
# ~150 full-rate instructions (hipcc -S of one-product kernels: 378 VALU instructions with 91 multiply-adds against 399 / 160 for the
# Montgomery form of the base field, ~140 of either being the load / store / canonicalisation around the product).
VALU_PM_MAD64_PER_PRODUCT, VALU_PM_OTHER_PER_PRODUCT = 91, 150
VALU_PM_CYCLES_PER_PRODUCT = VALU_PM_MAD64_PER_PRODUCT * 4 + VALU_PM_OTHER_PER_PRODUCT * 2
VALU_PM_PEAK_GMODMUL = 256 * 4 * 2.4e9 / VALU_PM_CYCLES_PER_PRODUCT * 64 / 1e9    # 238 G modmul/s
MADD_PRODUCTS, JADD_PRODUCTS, DBL_PRODUCTS = 11, 16, 7        # modular products of a mixed addition / Jacobian addition / doubling (csrc/ec.cuh)


def valu_entry(products, seconds, note, scalar_field=False):
    """the integer-VALU view of a kernel (group): modular products it needs / its time, against the guide-derived ceiling
    (scalar_field: the kernel multiplies in Fr — pseudo-Mersenne products, cheaper per product, higher ceiling)"""
    rate = products / seconds / 1e9
    if scalar_field:
        return {"unit": "G modmul/s", "achieved": rate, "peak": VALU_PM_PEAK_GMODMUL, "frac": rate / VALU_PM_PEAK_GMODMUL, "products": products,
                "valu_instructions": products * (VALU_PM_MAD64_PER_PRODUCT + VALU_PM_OTHER_PER_PRODUCT),
                "peak_derivation": "256 CUs x 4 SIMDs x 2.4 GHz x 64 lanes / (91 v_mad_u64_u32 x 4 cyc + 150 full-rate x 2 cyc): the pseudo-Mersenne product of the scalar "
                                   "field (csrc/fp29.cuh fe_pm_product); the 4 cycles per v_mad_u64_u32 are MEASURED (tools/ubench.hip), the rest is the guide's", "note": note}
    return {"unit": "G modmul/s", "achieved": rate, "peak": VALU_PEAK_GMODMUL, "frac": rate / VALU_PEAK_GMODMUL,
            "measured_rate": VALU_MEASURED_GMODMUL, "frac_of_measured_rate": rate / VALU_MEASURED_GMODMUL,
            "products": products, "valu_instructions": products * VALU_INSTR_PER_PRODUCT,
            "peak_derivation": "256 CUs x 4 SIMDs x 2.4 GHz x 64 lanes / (162 v_mad_u64_u32 x 4 cyc + 100 full-rate x 2 cyc); the 4 cycles per "
                               "v_mad_u64_u32 are MEASURED here (tools/ubench.hip, profiles/r01_ubench_radix29.txt: 0.95 wave-instructions per clock and CU) — the "
                               "guide has no row for that instruction; clocks, SIMD count and the 2-cycle full-rate issue are the guide's", "note": note}


def kernel_entry(name, ms, launches, alg_bytes, traffic, products, note=""):
    """one row of roofline.kernels: everything per UNIT (one proof / one launch, stated in `per`)"""
    sec = ms * 1e-3
    e = {"kernel": name, "ms": ms, "launches": launches, "algorithmic_bytes": alg_bytes, "hbm_GBps": alg_bytes / sec / 1e9 if (alg_bytes and sec > 0) else None,
         "hbm_frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBS if (alg_bytes and sec > 0) else None, "counter_traffic_bytes": traffic,
         "traffic_ratio": (traffic / alg_bytes) if (traffic and alg_bytes) else None, "products": products,
         "valu_instructions": products * VALU_INSTR_PER_PRODUCT if products else None,
         "valu_G_modmul_per_s": products / sec / 1e9 if (products and sec > 0) else None,
         "valu_frac": products / sec / 1e9 / VALU_PEAK_GMODMUL if (products and sec > 0) else None}
    if note:
        e["note"] = note
    return e
CURVES = ["secq256k1", "zorro"]

# Eight HIP streams (the proofs in flight) share FOUR hardware queues by default: at most four kernels run at once and the GPU idles a
# quarter of the time (profiles/r03_gpu_busy_prove.txt).  Sixteen queues: prove +6 %, batch verification unchanged within its noise
# (eight queues cost it ~7 %).  Read by the HIP runtime when it initialises, so it is set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
COLL_DEVICE = "cuda"   # where the collectives' tensors live ("cuda" over RCCL; None = CPU tensors over gloo, rehearsal only)


def statement_seed(tag, k):
    """32-byte ChaCha20 seed of synthetic statement k of a run (tag separates warmup / timed / isolated statements);
    the same on every rank, valid for any k < 2^32"""
    return bytes([3 + (tag & 0x7F)]) + int(k).to_bytes(4, "little") + bytes([3]) * 27


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) from this process, which has not
    touched the GPU, and exit with their status.  Never re-execs a process that has initialised HIP."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dist_setup(n_gpus):
    """one process per GPU over RCCL.  Rehearsal on a one-GPU box: ARKBP_BENCH_REHEARSE=1 puts every rank on cuda:0 and runs the
    collectives over gloo, so the multi-rank control flow can be exercised without a second GPU (numbers are meaningless)."""
    global COLL_DEVICE
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("ARKBP_BENCH_REHEARSE"):
            local = 0
            COLL_DEVICE = None
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def barrier(world):
    import torch

    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dt, world):
    if world <= 1:
        return dt
    import torch
    import torch.distributed as dist

    tt = torch.tensor([dt], dtype=torch.float64, device=COLL_DEVICE or "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt[0].item())


def run_threads(targets):
    """starts one thread per (fn, args) and joins them all; the first exception raised in any of them is re-raised here
    (a worker failure must fail the bench, not surface three frames later as a None)"""
    errors = []

    def guard(fn, a):
        try:
            fn(*a)
        except BaseException as e:  # noqa: BLE001 — re-raised below
            errors.append(e)

    th = [threading.Thread(target=guard, args=(fn, a)) for fn, a in targets]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]


def synth_msm_inputs(eng, n, rank):
    """cfg2 inputs without the (slow, sequential) generator derivation: bases = k_i * G for a seeded k_i
    computed ON THE GPU by the engine's own scalar-mul kernel; scalars = seeded 256-bit values < r."""
    rng = np.random.default_rng(1234 + rank)
    gen = {0: (53718550993811904772965658690407829053653678808745171666022356150019200052646,
               28941648020349172432234515805717979317553499307621291159490218670604692907903,
               0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141),
           1: (2, 19711758720854384559191066596451394956860102304684364148268676039962145446511,
               57896044618658097711785492504343953927116110621106131396339151912985063395361)}[eng.curve]
    q = gen[2]
    R = 1 << 256

    def limbs(x):
        return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]

    g = np.array(limbs(gen[0] * R % q) + limbs(gen[1] * R % q), dtype=np.uint64)
    ks = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    ks[:, 3] >>= np.uint64(2)  # < 2^253 < r for both curves: canonical scalars
    bases = np.zeros((n, 8), dtype=np.uint64)
    step = 1 << 14
    for i in range(0, n, step):
        m = min(step, n - i)
        bases[i:i + m] = eng.debug_point_op(3, np.tile(g, (m, 1)), np.tile(g, (m, 1)), ks[i:i + m])
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(2)  # treated as Montgomery words of some scalar < r
    return bases, sc


def cpu_quota():
    """CPUs this process may use: the cgroup's cpu.max quota when there is one (the GPU boxes allow 16 of their 256 hardware
    threads per GPU), else the affinity mask"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cgroup_cpu_stat():
    """(usage_usec, nr_periods, nr_throttled, throttled_usec) of this cgroup, or None"""
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))
        return tuple(int(d.get(k, 0)) for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec"))
    except Exception:
        return None


def cgroup_cpu_delta(before, wall_s):
    """host CPU actually consumed over a timed region and how often the cgroup's quota stopped the process tree in it"""
    after = cgroup_cpu_stat()
    if not before or not after or wall_s <= 0:
        return None
    per = max(1, after[1] - before[1])
    return {"avg_cores_used": (after[0] - before[0]) / 1e6 / wall_s, "quota_cores": cpu_quota(), "periods": after[1] - before[1],
            "periods_throttled_frac": (after[2] - before[2]) / per, "throttled_s": (after[3] - before[3]) / 1e6}


def host_cpu_info():
    """CPU model of the host and whether the AVX-512 lockstep paths of the library (TranscriptRng x8, commitment appends x8) apply"""
    model, flags = None, ""
    try:
        for ln in open("/proc/cpuinfo"):
            if model is None and ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
            if ln.startswith("flags"):
                flags = ln
                break
    except Exception:
        pass
    return {"model": model, "avx512f": " avx512f" in flags}


def mem_limit_bytes():
    """host memory this process tree may use: the cgroup's memory.max when there is one, else MemAvailable"""
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            return int(v)
    except Exception:
        pass
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except Exception:
        pass
    return 64 << 30


def pmc_traffic(keys, applicable, fname, field="largest", raw_fetch=False):
    """HBM bytes from the committed rocprofv3 PMC passes (profiles/<fname>, made by tools/pmc_summary.py from FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same workload).  FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM (gfx950
    tallies 128-B requests at 64 B for 16-B-per-lane loads); WRITE_SIZE is taken as is.  field = "largest": the dispatch with the
    largest grid; "all_launches": summed over the run.  keys: one kernel key or a list (summed).  None when the shape differs."""
    if not applicable:
        return None
    try:
        cands = [os.path.join(ROOT, "profiles", fname.replace("r02_", r)) for r in ("r04_", "r03_", "r02_")]   # the newest round's passes of this workload
        d = json.load(open(next(c for c in cands if os.path.exists(c))))
        tot, found = 0.0, 0
        for k in ([keys] if isinstance(keys, str) else keys):
            # (a kernel the profiled configuration did not launch contributes nothing; "name<Curve>" also stands for the
            # instantiations with further template arguments, "name<Curve, false>")
            for kk in d:
                if kk == k or kk.startswith(k[:-1] + ","):
                    # raw_fetch: kernels whose reads are 64-byte random gathers (one affine point per lane): the guide's doubling is for
                    # wide coalesced 16-B-per-lane loads; tools/ubench.hip `gather64` calibrates FETCH_SIZE at 1.0x for this shape
                    # (profiles/r04_fetch_calibration.txt)
                    tot += ((1.0 if raw_fetch else 2.0) * d[kk]["fetch_KiB_" + field] + d[kk]["write_KiB_" + field]) * 1024.0
                    found += 1
        return tot if found else None
    except Exception:
        return None


def run_msm(args, rank, world, local):
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    n = args.terms
    eng = A.Engine(curve=args.curve, device=local)
    bases, sc = synth_msm_inputs(eng, n, 0 if args.shard == "windows" else rank)
    db, ds = eng.upload_points(bases), eng.upload_scalars(sc)
    for _ in range(args.warmup):
        eng.msm_dev(db, ds, n)
    eng.set_profiling(True)
    eng.reset_profiling()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.shard == "windows":
            # window-sharded MSM (north_star): every rank holds the same n terms and owns a range of Pippenger windows
            P.window_sharded_msm(args.curve, n, lambda lo, hi: eng.msm_dev_windows(db, ds, n, lo, hi), E.msm_window_count, E.host_points_sum,
                                 rank, world, device=COLL_DEVICE if world > 1 else None)
        else:
            # term-sharded MSM: local partial, all-gather of one 64-byte point per rank over RCCL, host point-reduce
            if world == 1:
                eng.msm_dev(db, ds, n)   # (one rank: the local MSM is the result)
            else:
                P.sharded_msm(args.curve, lambda: eng.msm_dev(db, ds, n), E.host_points_sum, device=COLL_DEVICE)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    acc_ms, acc_n = eng.kernel_time(0)
    fs_ms, fs_n = eng.kernel_time(9)            # the fixed-shape pipeline's accumulate has its own timer
    acc_ms, acc_n = acc_ms + fs_ms, acc_n + fs_n
    tot_ms, tot_n = eng.kernel_time(1)
    res = {
        "metric": "msm_terms_per_sec", "value": n * (1 if args.shard == "windows" else world) * args.steps / dt, "unit": "terms/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.shard == "windows" else "weak", "vs_baseline": None,
        "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "cfg2: 2^%d-term variable-base MSM, %s, inputs HBM-resident" % (int(np.log2(n)), CURVES[args.curve]),
                   "terms_per_gpu": n, "curve": CURVES[args.curve], "parallelism": "%s-sharded x%d" % (args.shard[:-1], world)},
    }
    if acc_n:
        avg_s = acc_ms / acc_n * 1e-3
        res["roofline"] = {"bound": "hbm", "kernel": "k_msm_accum_fs (k_msm_accum above 2^18 buckets)", "achieved": n * 96 / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": n * 96 / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("msm/k_msm_accum_fs<Secq>", n == 1 << 16 and args.curve == 0, "r02_pmc_msm_summary.json"),
                           "avg_kernel_ms": acc_ms / acc_n,
                           "msm_all_kernels_ms": tot_ms / max(tot_n, 1)}
        W, c = E.msm_window_count(args.curve, n)
        full_windows = [256, 255][args.curve] // c          # the remaining top window holds at most a few bits
        madds = n * full_windows * (1.0 - 0.5 ** c)          # non-zero signed digits of uniform scalars
        res["roofline"]["valu"] = valu_entry(madds * MADD_PRODUCTS, avg_s, "the kernel is integer-VALU-bound: gathered mixed Jacobian+affine additions (11 modular products "
                                             "each; tools/ubench.hip measures 12.9 G mixed adds/s = 142 G modmul/s for the addition as a whole)")
        agg_ms, agg_n = eng.kernel_time(10)
        res["roofline"]["kernels"] = [kernel_entry("k_msm_accum_fs | k_msm_accum", acc_ms / acc_n, 1, n * 96.0, res["roofline"]["traffic"], madds * MADD_PRODUCTS),
                                      kernel_entry("bucket reduction + aggregation (k_msm_reduce_fs, k_msm_marginals_fs)", agg_ms / max(agg_n, 1), 2, None, None, None,
                                                   "latency-bound trees: ~%d buckets" % (W * (1 << (c - 1))))]
        res["roofline"]["wall_ms_per_msm"] = dt / args.steps * 1e3
        res["roofline"]["kernels_over_wall"] = (tot_ms / max(tot_n, 1)) / (dt / args.steps * 1e3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is reported at N = 1 only
        res["cpu_baseline"] = cpu_baseline_msm(args, bases, sc)
    db.free()
    ds.free()
    eng.close()
    return res


class ProvePipeline:
    """The prove workload's three stages, all inside the timed region:

      build   (host pool)  Prover::new + commit + gadget for statement k (the reference's benchmark builds these inside its
                           timed closure too, benches/r1cs_secq256k1.rs:172-184).  At most `--window` statements are alive at once
                           (built, not yet proved): a 2^20 statement holds ~0.3 GB of witness + constraints.
      stage 1 (host pool)  the head of prove(): the TranscriptRng chain — 8 sequential Keccak-f per multiplier, the reference's
                           design — for groups of 8 consecutive statements in AVX-512 lockstep.
      stage 2 (P drivers)  the rest of prove() on the GPU, one ctx/stream each; a proved statement is freed at once.

    The chain of proof k+1 overlaps the GPU work of proof k.  Results come back in `out` (index = statement number)."""

    def __init__(self, E, engs, args, N):
        self.E, self.engs, self.args, self.N = E, engs, args, N
        self.keep_info = set()      # statement numbers whose public side (commitments, public values) is kept for a later verify
        self.infos = {}
        self.ordered = False        # prove the statements strictly in index order (window-sharded ranks must issue their collectives in the same order)

    def run(self, tag, count, out):
        E, args, engs = self.E, self.args, self.engs
        window = max(8, args.window)
        built = [None] * count
        built_ev = [threading.Event() for _ in range(count)]
        pre_ev = [threading.Event() for _ in range(count)]
        nxt_ordered = [0]
        slots = threading.Semaphore(window)
        stop = threading.Event()
        lock = threading.Lock()
        nxt_build, nxt_pre = [0], [0]
        ready = queue.Queue()
        waits = self.waits = {"build_wait_slot": 0.0, "build_busy": 0.0, "rng_wait_build": 0.0, "rng_busy": 0.0, "gpu_wait_ready": 0.0, "gpu_busy": 0.0}
        wlock = threading.Lock()

        def acct(key, dt):
            with wlock:
                waits[key] += dt

        def wait(ev_or_sem, is_sem=False):
            while not stop.is_set():
                if (ev_or_sem.acquire(timeout=0.2) if is_sem else ev_or_sem.wait(0.2)):
                    return True
            return False

        def guarded(fn):
            def w(*a):
                try:
                    fn(*a)
                except BaseException:
                    stop.set()
                    raise
            return w

        @guarded
        def build():
            while not stop.is_set():
                with lock:
                    k = nxt_build[0]
                    nxt_build[0] += 1
                if k >= count:
                    return
                t0 = time.perf_counter()
                if not wait(slots, True):
                    return
                t1 = time.perf_counter()
                built[k] = E.Statement(args.curve, E.SC_SQUARE_CHAIN, [self.N, 0], statement_seed(tag, k))
                built_ev[k].set()
                acct("build_wait_slot", t1 - t0)
                acct("build_busy", time.perf_counter() - t1)

        @guarded
        def stage1():
            while not stop.is_set():
                with lock:
                    i = nxt_pre[0]
                    nxt_pre[0] += 8
                if i >= count:
                    return
                grp = list(range(i, min(i + 8, count)))
                t0 = time.perf_counter()
                for g in grp:
                    if not wait(built_ev[g]):
                        return
                t1 = time.perf_counter()
                E.precompute_batch([built[g] for g in grp])   # 8 chains in lockstep in AVX-512 lanes (Keccak-f x8)
                for g in grp:
                    pre_ev[g].set()
                    if not self.ordered:
                        ready.put(g)
                acct("rng_wait_build", t1 - t0)
                acct("rng_busy", time.perf_counter() - t1)

        @guarded
        def stage2(k):
            while True:
                t0 = time.perf_counter()
                if self.ordered:
                    with lock:
                        i = nxt_ordered[0]
                        nxt_ordered[0] += 1
                    if i >= count or not wait(pre_ev[i]):
                        return
                else:
                    try:
                        i = ready.get(timeout=0.2)
                    except queue.Empty:
                        acct("gpu_wait_ready", time.perf_counter() - t0)
                        if stop.is_set():
                            return
                        continue
                    if i is None:
                        return
                t1 = time.perf_counter()
                if i in self.keep_info:
                    self.infos[i] = built[i].info(m_cap=8)[:2]
                out[i] = built[i].prove(engs[k])
                built[i].free()   # a consumed statement (witness + constraints, ~0.3 GB at 2^20) is released at once
                built[i] = None
                slots.release()
                acct("gpu_wait_ready", t1 - t0)
                acct("gpu_busy", time.perf_counter() - t1)

        def producers():
            run_threads([(build, ())] * args.build_threads + [(stage1, ())] * args.host_threads)
            for _ in engs:
                ready.put(None)

        try:
            run_threads([(producers, ())] + [(stage2, (k,)) for k in range(len(engs))])
        finally:
            stop.set()
            for s in built:
                if s is not None:
                    s.free()
        missing = [i for i, r in enumerate(out) if r is None]
        if missing:
            raise RuntimeError("prove pipeline: %d statements were not proved (first: %d)" % (len(missing), missing[0]))


def run_prove(args, rank, world, local):
    """cfg3: R1CS proofs of a 2^logn-multiplier circuit (square chain: 1 commitment, N multiply gates, q = 2N+1 linear
    constraints).  A step = one batch of `batch` independent proofs on this GPU; `inflight` of them are on the GPU at a time,
    each on its own ctx/stream with its own host thread, all reading one resident copy of the generator tables.
    `value` = padded multiplication gates proved / wall seconds.  N > 1 ranks: replicas (weak scaling), or with
    --shard windows every rank works on the SAME proof (cfg5 partition, strong scaling)."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E

    N = 1 << args.logn
    window_sharded = args.shard == "windows" and world > 1
    P = 1 if window_sharded else max(1, args.inflight)
    engs = [A.Engine(curve=args.curve, device=local) for _ in range(P)]
    t0 = time.perf_counter()
    engs[0].gens_derive(N)
    t_gens = time.perf_counter() - t0
    msm_tab_info = None
    if args.msm_tables and not window_sharded:
        # fixed-base rows of the generators for the MSMs over the tables themselves (commitments, first-round L / R)
        t0 = time.perf_counter()
        nbytes = engs[0].gens_msm_tables(N)
        msm_tab_info = {"GB": nbytes / 1e9, "build_s": time.perf_counter() - t0}
    tab_info = None
    if args.fold_tables:
        # fixed-base tables of the generators for the first fold round (one-time setup like the derivation above; HBM-resident)
        t0 = time.perf_counter()
        # HBM budget: what is free now, minus the fixed-base MSM rows (65 x 64 B per generator, two vectors), the per-proof
        # workspaces of the proofs in flight (~3 KB per constraint each, MSM sort and tree buffers included) and some slack
        import torch

        free_b, _ = torch.cuda.mem_get_info(local)
        budget = int(free_b) - P * 3000 * N - (12 << 30)        # (the fixed-base MSM rows are installed already)
        # ... and a cap on what the tables may take: the width sweep of round 4 (profiles/r04_fold_table_width_sweep.txt: w = 8 / 219 GB
        # 23.9 M constraints/s, w = 7 / 122 GB 23.6, w = 6 / 71 GB 23.0, w = 5 / 44 GB 22.8, w = 4 / 27 GB 21.9) puts the smallest width
        # within 2 % of the best at w = 7: 125 GB leaves more than half of the HBM to whatever shares the GPU (the batch verifier does)
        if args.fold_table_budget_gb > 0:
            budget = min(budget, int(args.fold_table_budget_gb * 1e9))
        # bases [0, 3N/4): the first TWO fold rounds come straight from the tables (bases [0, N/2) would serve the first round only)
        # (the index-cyclic slices of a sharded prover defer their first fold the same way since round 4: same tables)
        tab_count = N * 3 // 4 if args.fold_tables >= 2 else N // 2
        # (one proof partitioned across the ranks: every rank keeps only ITS slice of the tables — the generators rank + i * world that its
        # index-cyclic slice of the inner-product argument looks up: 1/world of the memory, so the budget buys wider windows)
        wbits, nbytes = engs[0].gens_fold_tables(tab_count, window_bits=args.fold_table_bits, budget_bytes=max(budget, 1 << 30),
                                                 rank=rank if window_sharded else 0, world=world if window_sharded else 1)
        tab_info = {"window_bits": wbits, "GB": nbytes / 1e9, "build_s": time.perf_counter() - t0, "bases": tab_count, "rounds_from_tables": 2 if tab_count > N // 2 else 1}
    for e in engs[1:]:
        e.share_gens_from(engs[0])
    if getattr(args, "freeze_len", 0):
        for e in engs:
            e.set_tuning(2, args.freeze_len)
    for e in engs:
        e.set_tuning(10, 30)     # BP_TUNE_WAIT_SLEEP: the GPU driver threads sleep 30 us between polls (HIP's waits burn a core each)
    if window_sharded:
        # north_star / cfg5 partition: all ranks prove the SAME statements; every MSM inside prove() accumulates the rank's Pippenger
        # windows and the partial points are summed over RCCL (strong scaling of one proof at a time: one proof in flight, because
        # the ranks must issue their per-MSM collectives in the same order)
        from ark_bulletproofs_amd import parallel as PP

        if COLL_DEVICE is not None and not os.environ.get("ARKBP_BENCH_PY_COLLECTIVES"):
            # the collectives inside the library: ncclAllGather on the ctx's stream (bp_ctx_rccl_init); this layer only carries the unique id
            PP.enable_native_sharding(engs[0], rank, world, device=COLL_DEVICE)
            native = True
        else:
            # rehearsal on one GPU (gloo) or A/B: the exchanges go through host callbacks into torch.distributed
            coll_stat = {"n": 0, "s": 0.0}

            def counted_allgather(arr, _ag=PP.allgather_words):
                t_c = time.perf_counter()
                out_c = _ag(arr, None, COLL_DEVICE)
                coll_stat["n"] += 1
                coll_stat["s"] += time.perf_counter() - t_c
                return out_c

            PP.enable_window_sharding(engs[0], args.curve, E.host_points_sum, rank, world, device=COLL_DEVICE, allgather=counted_allgather)
            native = False
    pipe = ProvePipeline(E, engs, args, N)
    pipe.ordered = window_sharded
    if args.warmup:
        pipe.run(100, args.batch * args.warmup, [None] * (args.batch * args.warmup))
    nproofs = args.batch * args.steps
    flat = [None] * nproofs
    pipe.keep_info = {0, nproofs - 1}
    barrier(world)
    cpu0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    pipe.run(0, nproofs, flat)                   # K steps x `batch` proofs, statement construction included
    barrier(world)
    host_cpu = cgroup_cpu_delta(cpu0, time.perf_counter() - t0)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    # The proofs that were timed are checked (after the timed region): the first and the last of the run go through the GPU verifier
    # — Verifier::verify, src/r1cs/verifier.rs:549-600 — and a copy with one flipped bit of t_x must be rejected; the run fails otherwise.
    # The precomputed tables the proofs came through are walked entry by entry (bp_gens_tables_check).
    verified = 0
    for i in sorted(pipe.keep_info):
        commits, pubs = pipe.infos[i]
        proof = flat[i][0]
        rc = engs[0].verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], proof, commits, pubs)
        if rc != 0:
            raise RuntimeError("bench: timed proof %d does not verify (status %d)" % (i, rc))
        bad = bytearray(proof)
        bad[11 * 33 + 9] ^= 1
        rc = engs[0].verify_scenario(E.SC_SQUARE_CHAIN, [N, 0], bytes(bad), commits, pubs)
        if rc != -4:
            raise RuntimeError("bench: a tampered copy of timed proof %d was not rejected (status %d)" % (i, rc))
        verified += 1
    t_chk = time.perf_counter()
    tables_bad = list(engs[0].gens_tables_check()) if (tab_info or msm_tab_info) else None
    t_chk = time.perf_counter() - t_chk
    if tables_bad and any(tables_bad):
        raise RuntimeError("bench: precomputed table entries fail the chain-rule check: %r" % (tables_bad,))
    coll_info = None
    if window_sharded:
        cn, cs = engs[0].collective_stats() if native else (coll_stat["n"], coll_stat["s"])
        coll_info = {"native_rccl_in_library": native, "count": int(cn), "per_proof": cn / max(1, nproofs + args.batch * args.warmup + 2 * len(pipe.keep_info)),
                     "avg_latency_us": cs / max(cn, 1) * 1e6, "total_s": cs,
                     "note": "all-gathers of one 64-byte partial point per MSM (+ one vector gather per proof for the index-cyclic IPA), counted since the ctx entered the sharded "
                             "mode (warm-up and the post-run verification included in `count`)"}
    # thread-seconds of every stage of the timed pipeline (busy / waiting for its input), as fractions of (threads x wall)
    pipe_util = {k: v / dt for k, v in pipe.waits.items()}
    stages = np.zeros(8)
    for (_, tm) in flat:
        stages += np.array(tm)
    # kernel durations for the roofline: ONE more proof, alone on the GPU, after the timed region — HIP-event times taken while
    # several streams share the GPU include the other streams' kernels
    engs[0].set_profiling(True)     # (HIP events only here: the timed pipeline above runs without them)
    engs[0].reset_profiling()
    iso = E.Statement(args.curve, E.SC_SQUARE_CHAIN, [N, 0], statement_seed(120, 0))
    t_iso = time.perf_counter()
    iso.precompute()                                 # the TranscriptRng head of prove(): sequential by the reference's construction, one core
    rng_head_ms = (time.perf_counter() - t_iso) * 1e3
    t_iso = time.perf_counter()
    iso.prove(engs[0])
    alone_ms = (time.perf_counter() - t_iso) * 1e3   # everything after the head, this proof alone on the GPU (with HIP-event profiling on)
    iso.free()
    fold_ms, fold_n = engs[0].kernel_time(3)
    acc_ms = engs[0].kernel_time(0)[0]
    msm_ms = engs[0].kernel_time(1)[0]
    names = ["prove_total", "-", "transcript_rng", "uploads", "commit_msms", "flatten_constraints", "poly_kernels", "ipa"]
    res = {
        "metric": "r1cs_constraints_proved_per_sec", "value": N * (1 if window_sharded else world) * nproofs / dt, "unit": "constraints/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if window_sharded else "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "cfg3: 2^%d-constraint R1CS prove (square-chain circuit, m=1, q=2N+1), %s, a step = %d independent proofs per GPU (%d GPU streams, %d host threads for the "
                               "TranscriptRng stage, %d for statement construction, which is inside the timed region)"
                               % (args.logn, CURVES[args.curve], args.batch, P, args.host_threads, args.build_threads),
                   "constraints_per_proof": N, "proofs_per_step": args.batch, "gpu_streams": P, "host_threads": args.host_threads, "build_threads": args.build_threads,
                   "curve": CURVES[args.curve], "parallelism": ("window-sharded x%d" if window_sharded else "replicas x%d") % world,
                   "verified": verified, "verified_note": "timed proofs 0 and %d verified on the GPU after the timed region, a tampered copy of each rejected" % (nproofs - 1),
                   "table_entries_failing_check": tables_bad, "tables_check_s": t_chk if tables_bad is not None else None, "collectives": coll_info,
                   "pipeline_thread_seconds_per_wall_second": pipe_util, "host_cpu_in_timed_region": host_cpu, "host_cpu": host_cpu_info(),
                   "in_pipeline_latency_ms": float(stages[0]) / nproofs * 1e3, "alone_ms_after_rng_head": alone_ms, "rng_head_ms": rng_head_ms,
                   "latency_note": "in_pipeline: mean prove() wall of a proof with %d proofs in flight; alone: one proof on an otherwise idle GPU after its "
                                   "TranscriptRng head; rng_head: that head (merlin's sequential sponge, prover.rs:483-513) on one host core" % P,
                   "gens_derive_s": t_gens, "first_round_fold_tables": tab_info, "fixed_base_msm_tables": msm_tab_info,
                   "per_proof_stage_ms": {k: float(v) / nproofs * 1e3 for k, v in zip(names, stages) if k != "-"}},
    }
    if fold_n:
        # Dominant kernel group: the IPA G/H fold (k_ipa_fold_tab [round 1], k_ipa_fold_glv | k_ipa_fold_uniform, k_ipa_fold_finish,
        # k_ipa_fold_ab).  Per proof it reads 4*64 B and writes 2*64 B per folded pair of points plus 4*32 + 2*32 B of scalars, over
        # sum_j n_j = N-1 pairs => 576 B * (N-1); the unit of `achieved`, `traffic` and the time is "all fold
        # launches of ONE proof".  Times: one proof run alone after the timed region, HIP events on the ctx's stream.
        per_proof_s = fold_ms * 1e-3
        # modular products per output point of a ladder round — secq256k1 (GLV): 130 doublings x 7 + ~88 mixed adds x 11 + ~70
        # (endomorphism, shared inversion, conversions); zorro: 257 x 8 + ~87 x 11 + ~70
        per_lane = 1950.0 if args.curve == 0 else 3080.0
        tab_ms, tab_n = engs[0].kernel_time(6)
        lad_ms, lad_n = engs[0].kernel_time(7)
        fin_ms, fin_n = engs[0].kernel_time(8)
        accfs_ms, accfs_n = engs[0].kernel_time(9)
        agg_ms, agg_n = engs[0].kernel_time(10)
        acc_n = engs[0].kernel_time(0)[1]
        if tab_info and tab_n:
            # round 1 (N of the 2(N-1) output points) goes through the fold tables: per point nwin look-ups per scalar half, each a
            # mixed add (11) — with the GLV halves 2 * nwin adds + nwin endomorphism products —, the final add; conversions
            wb = tab_info["window_bits"]
            nwin = (130 if args.curve == 0 else 256) // wb + 1
            per_mult = 2 * nwin * 11 + nwin if args.curve == 0 else nwin * 11        # one fixed-base multiplication: look-ups + mixed adds (+ endomorphism products)
            if tab_info.get("rounds_from_tables") == 2:
                # rounds 1 and 2 in one table kernel: N/2 output points, three fixed-base multiplications + the final add each; the ladders
                # start at round 3 (N/2 - 2 output points)
                prod_tab, prod_lad = (N / 2.0) * (3 * per_mult + 11), (N / 2.0 - 2.0) * (per_lane - 20)
            else:
                prod_tab, prod_lad = N * (per_mult + 11), (N - 2.0) * (per_lane - 20)
        else:
            prod_tab, prod_lad = 0.0, 2.0 * (N - 1) * (per_lane - 20)
        n_out = (N - 2.0) if (tab_info and tab_info.get("rounds_from_tables") == 2) else 2.0 * (N - 1)      # points that are materialised
        prod_fin = n_out * 20                   # shared inversion (450 / 8 per point) + 3 products to (X/Z^2, Y/Z^3) + canonical forms
        modmul = prod_tab + prod_lad + prod_fin
        secq20 = args.logn == 20 and args.curve == 0 and bool(args.fold_tables)
        pf = "r02_pmc_prove2p20_summary.json"

        def tr(keys, field="all_launches", raw_fetch=False):
            return pmc_traffic(keys, secq20, pf, field, raw_fetch=raw_fetch)

        # MSM terms of one proof: commitments (2n+1) + (n+1) + (2n+1), then L and R of every round: 2 * (2 n_j + 1)
        msm_terms = (5 * N + 3) + sum(2 * (2 * (N >> (j + 1)) + 1) for j in range(args.logn))
        cW = E.msm_window_count(args.curve, max(N, 64))
        madds_per_term = (255 if args.curve else 256) // cW[1]       # ordinary schedule; the fixed-base schedule needs fewer (c = 20)
        # the accumulate launches split the proof's MSM terms: the fixed-base / large-bucket kernel takes the commitments and the L / R of
        # the rounds whose MSMs have >= 2^18 buckets or run over the generator tables; the fixed-shape pipeline the mid-size rest
        two_rounds = bool(tab_info and tab_info.get("rounds_from_tables") == 2)
        lr_terms = [2 * (2 * (N >> (j + 1)) + 1) for j in range(args.logn)]
        big_rounds = 2 if two_rounds else 1
        terms_big = (5 * N + 3) + (lr_terms[0] + (2 * (5 * N // 4 + 1) if two_rounds else 0)) + sum(t for t in lr_terms[big_rounds:] if t // 2 >= (1 << 18))
        freeze_len = 8192
        terms_fs = sum((t if (N >> (j + 1)) >= freeze_len else 2 * (2 * freeze_len + 1)) for j, t in enumerate(lr_terms) if j >= big_rounds and t // 2 < (1 << 18))
        madds_big = terms_big * (13 if msm_tab_info else madds_per_term)
        madds_fs = (255 if args.curve else 256) // E.msm_window_count(args.curve, 1 << 16)[1]
        kernels = [
            kernel_entry("k_ipa_fold_tab | k_ipa_fold_tab2 (rounds 1-2: fixed-base table look-ups)", tab_ms, tab_n, 576.0 * (N // 2) * (1.5 if (tab_info and tab_info.get("rounds_from_tables") == 2) else 1.0),
                         tr(["prove2p20/k_ipa_fold_tab<Secq>", "prove2p20/k_ipa_fold_tab2<Secq>"]), prod_tab,
                         "traffic includes the table rows it streams (34 rows x 64 B per point at w = 8): deliberate, 0.3 ms at HBM speed for ~12 ms of ladder saved"),
            kernel_entry("k_ipa_fold_glv | k_ipa_fold_uniform (the later rounds: scalar-multiplication ladders)", lad_ms, lad_n,
                         576.0 * ((N // 4 - 1) if (tab_info and tab_info.get("rounds_from_tables") == 2) else (N // 2 - 1)),
                         tr("prove2p20/k_ipa_fold_glv<Secq>"), prod_lad, "a round below 2^16 lanes takes ~1.15 ms whatever its size: one lane's serial ladder"),
            kernel_entry("k_ipa_fold_finish (Jacobian -> affine, one inversion per 8 points)", fin_ms, fin_n, 2.0 * (N - 1) * (96 + 64), tr("prove2p20/k_ipa_fold_finish<Secq>"), prod_fin,
                         "re-reads the Jacobian results the ladders wrote (unfused: 0.58 GB per proof, 0.07 ms at HBM speed)"),
            kernel_entry("k_msm_accum (MSMs above 2^18 buckets and the fixed-base MSMs)", acc_ms, acc_n, 96.0 * terms_big, tr("prove2p20/k_msm_accum<Secq>", raw_fetch=True),
                         madds_big * MADD_PRODUCTS, "terms: the 3 commitment MSMs + L / R of the rounds served by these launches; mixed additions: 13 per term on the "
                         "fixed-base schedule (c = 20, one bucket set for all windows), %d on the ordinary one" % madds_per_term),
            kernel_entry("k_msm_accum_fs (fixed-shape pipeline: the mid-size L / R MSMs)", accfs_ms, accfs_n, 96.0 * terms_fs, tr("prove2p20/k_msm_accum_fs<Secq>", raw_fetch=True),
                         terms_fs * madds_fs * MADD_PRODUCTS, "terms: L / R MSMs of the rounds below 2^18 buckets (incl. the frozen tail's 2 x 16 K-term MSMs per round)"),
            kernel_entry("MSM bucket reduction + aggregation (k_msm_reduce*, k_msm_marginals*, k_msm_window_sums)", agg_ms, agg_n, None, None, None, "latency-bound trees"),
        ]
        # the two accumulate rows share the proof's MSM terms: algorithmic bytes and products for their sum
        acc_all_ms = acc_ms + accfs_ms
        # the dominant kernel group of one proof names the roofline: MSM accumulate (k_msm_accum + k_msm_accum_fs) or the IPA folds
        acc_dominant = acc_all_ms > fold_ms
        dom_bytes = 96.0 * (terms_big + terms_fs) if acc_dominant else 576.0 * (N - 1)
        dom_s = (acc_all_ms if acc_dominant else fold_ms) * 1e-3
        res["roofline"] = {"bound": "hbm",
                           "kernel": ("MSM bucket accumulation (k_msm_accum + k_msm_accum_fs), all MSMs of one proof" if acc_dominant else
                                      "IPA G/H fold (k_ipa_fold_tab / k_ipa_fold_tab2 [rounds from the tables] + k_ipa_fold_glv + k_ipa_fold_finish + k_ipa_fold_ab), all rounds of one proof"),
                           "per": "one 2^%d proof (all launches of the kernel group), run alone after the timed region" % args.logn,
                           "achieved": dom_bytes / dom_s / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_bytes / dom_s / 1e9 / HBM_PEAK_GBS,
                           "dominant_group_ms_per_proof": dom_s * 1e3, "algorithmic_bytes_of_the_group": dom_bytes,
                           "fold_group": {"algorithmic_bytes": 576.0 * (N - 1), "ms": fold_ms, "hbm_GBps": 576.0 * (N - 1) / per_proof_s / 1e9,
                                          "hbm_frac": 576.0 * (N - 1) / per_proof_s / 1e9 / HBM_PEAK_GBS},
                           # HBM bytes of all fold launches of ONE proof (same unit as `achieved`), from the committed PMC passes of this shape
                           "traffic": tr(["prove2p20/k_ipa_fold_glv<Secq>", "prove2p20/k_ipa_fold_tab<Secq>", "prove2p20/k_ipa_fold_tab2<Secq>", "prove2p20/k_ipa_fold_finish<Secq>", "prove2p20/k_ipa_fold_ab<Secq>"]),
                           "traffic_note": "traffic above the algorithmic bytes is deliberate here: the table rows streamed by the rounds that come from the tables (34 rows x 64 B per "
                                           "point and multiplier, instead of ~20 ms of ladder arithmetic) and the Jacobian results written by the ladders and re-read by "
                                           "k_ipa_fold_finish; together < 1 ms at HBM speed",
                           "avg_kernel_ms": fold_ms / max(fold_n, 1), "fold_ms_per_proof": fold_ms,
                           "msm_kernels_ms_per_proof": msm_ms, "msm_accum_ms_per_proof": acc_all_ms,
                           "valu": valu_entry(modmul, per_proof_s, "the path is integer-VALU-bound: this is the fraction that measures the kernels — modular products the fold "
                                              "launches of one proof need (round 1: fixed-base table look-ups; later rounds: GLV ladder, ~1950 per point) against the "
                                              "ceiling derived from the guide's issue rates; `measured_rate` is what tools/ubench.hip sustains on this GPU"),
                           "msm_accumulate": {"terms_per_proof": msm_terms, "algorithmic_bytes": 96.0 * msm_terms, "ms": acc_all_ms,
                                              "hbm_GBps": 96.0 * msm_terms / (acc_all_ms * 1e-3) / 1e9 if acc_all_ms else None,
                                              "products_upper": msm_terms * madds_per_term * MADD_PRODUCTS,
                                              "valu_frac_upper": msm_terms * madds_per_term * MADD_PRODUCTS / (acc_all_ms * 1e-3) / 1e9 / VALU_PEAK_GMODMUL if acc_all_ms else None,
                                              "note": "upper figures: every term costed at the ordinary schedule's %d mixed additions; 7/9 of the terms take the fixed-base schedule (13)" % madds_per_term},
                           "kernels": kernels}
    # ---- the same pipeline with the precomputed tables released: what the 150 GB of HBM buy ----
    if args.tables_off_steps > 0 and (tab_info or msm_tab_info) and not window_sharded:
        engs[0].set_profiling(False)
        engs[0].gens_fold_tables(0)
        engs[0].gens_msm_tables(0)
        for e in engs[1:]:
            e.share_gens_from(engs[0])
        pipe.keep_info = set()
        pipe.run(101, args.batch, [None] * args.batch)
        n_off = args.batch * args.tables_off_steps
        barrier(world)
        t0 = time.perf_counter()
        pipe.run(1, n_off, [None] * n_off)
        barrier(world)
        dt_off = max_over_ranks(time.perf_counter() - t0, world)
        res["config"]["tables_off"] = {"value": N * world * n_off / dt_off, "unit": "constraints/s", "steps": args.tables_off_steps, "ms_per_step": dt_off / args.tables_off_steps * 1e3,
                                       "hbm_spent_on_tables_GB": (tab_info["GB"] if tab_info else 0.0) + (msm_tab_info["GB"] if msm_tab_info else 0.0),
                                       "note": "same pipeline, same statements family, fold tables and fixed-base MSM rows released"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is reported at N = 1 only
        res["cpu_baseline"] = cpu_baseline_prove(args)
    for e in engs[1:] + engs[:1]:
        e.close()
    return res


def run_verify(args, rank, world, local):
    """cfg4: batch verification of `--proofs` R1CS proofs with 2^14 constraints each (256 x 64-bit range proofs in one
    circuit, m = 256: n = 16384, q = 33024).  The instance list is built from `--distinct` distinct proofs (proved on the GPU
    before the timed region) repeated round-robin; every instance is replayed, alpha-scaled and accumulated separately.
    A step = one batch_verify call per rank over its shard of whole proofs; ranks exchange one 64-byte check point."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from ark_bulletproofs_amd import parallel as P

    nval, nbits = 256, 64
    N = nval * nbits
    shuffle_k = args.shuffle_k
    if shuffle_k:
        # the reference's own verification benchmark circuit (benches/r1cs_secq256k1.rs:201-250): a k-shuffle, all of whose
        # 2(k-1) multipliers are randomized (phase-2) constraints carrying the instance's challenge; m = 2k commitments
        N = 1
        while N < 2 * (shuffle_k - 1):
            N *= 2
    eng = A.Engine(curve=args.curve, device=local)
    eng.gens_derive(max(N, 2))
    distinct = []
    for i in range(args.distinct):
        if shuffle_k:
            pr = eng.prove_scenario(E.SC_SHUFFLE, [shuffle_k], statement_seed(2, i), m_cap=2 * shuffle_k + 8)
            distinct.append((E.SC_SHUFFLE, [shuffle_k], pr.proof, pr.commitments, pr.publics))
        else:
            pr = eng.prove_scenario(E.SC_MULTI_RANGE, [nval, nbits, 0], statement_seed(1, i), m_cap=nval + 8)
            distinct.append((E.SC_MULTI_RANGE, [nval, nbits, 0], pr.proof, pr.commitments, pr.publics))
    strong = bool(getattr(args, "verify_strong", False))
    total = args.proofs if strong else args.proofs * world      # strong: BASELINE cfg4 as stated — ONE batch of `--proofs` sharded across the GPUs
    lo, hi = P.shard_range(total, rank, world)
    inst_list = [distinct[i % len(distinct)] for i in range(lo, hi)]
    inst = E.pack_instances(inst_list)   # the C ABI's flat arrays, marshalled once (a host in the reference's language owns them already)
    seed = bytes([5]) * 32
    # `--verify-inflight` batches in flight per GPU (like the prover's proofs in flight): each on its own ctx / stream with its own
    # host pool, sharing the resident generator tables.  One batch alone leaves the GPU idle while the host parses and replays the
    # first block and the host idle during the drain and the final MSM; a second batch fills those.
    nfl = max(1, args.verify_inflight)
    engs = [eng]
    for _ in range(nfl - 1):
        e2 = A.Engine(curve=args.curve, device=local)
        e2.share_gens_from(eng)
        engs.append(e2)
    if (nfl > 1 or world > 1) and not os.environ.get("ARKBP_HOST_THREADS"):
        # several pools share the cores this process may use (measured on a 16-CPU cgroup, one session: 1 x 32 threads 215 K proofs/s,
        # 2 x 12 250 K, 3 x 8 255-263 K, 3 x 10 270 K, 4 x 6 274 K, 4 x 8 252 K — a plateau; before the pools divided the cores two
        # batches in flight were no faster than one)
        per_pool = min(32, max(4, int(round(cpu_quota() / world * 1.5 / nfl))))   # (the ranks of a node share its quota)
        for e in engs:
            e.set_tuning(6, per_pool)   # BP_TUNE_HOST_THREADS
    if getattr(args, "verify_wait_us", 0):
        for e in engs:
            e.set_tuning(10, args.verify_wait_us)   # BP_TUNE_WAIT_SLEEP
    t_w = time.perf_counter()
    for _ in range(args.warmup):
        for e in engs:
            e.batch_verify(inst, seed, alpha_skip=lo)
    t_one = (time.perf_counter() - t_w) / max(1, args.warmup * len(engs))     # one batch alone on the GPU
    barrier(world)
    cpu0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    tms = np.zeros(5)
    ok = True
    # every rank verifies its shard of every step's batch (its own alphas through alpha_skip), `nfl` calls in flight; the ranks'
    # statuses and check points of ALL steps are exchanged in one all-gather at the end of the timed region (a 64-byte point per rank
    # and step: the exchange is not on any step's critical path) and each step is judged as parallel.sharded_batch_verify does:
    # valid iff every rank's status is 0, the sum of the points is the identity as a consistency check
    step_pts = np.zeros((args.steps, 9), dtype=np.uint64)
    nxt, lock, tml = [0], threading.Lock(), []

    # The batches in flight start a fraction of a batch apart.  Started together they stay in lockstep (equal jobs under the GPU's fair
    # sharing finish together), every batch then stages and uploads its inputs at the same moment, and the GPU idles ~3 ms per round
    # (kernel trace, profiles/r04_gpu_busy_verify.txt: 79 % busy, every idle gap between a batch's last MSM kernel and the next
    # batch's first memset); a service's requests do not arrive in phase either.
    def worker(e, k=0):
        if k and t_one > 0:
            time.sleep(min(0.05, t_one) * k / len(engs))
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= args.steps:
                return
            rc, tm, pt = e.batch_verify(inst, seed, alpha_skip=lo, want_point=True)
            step_pts[i, :8] = pt
            step_pts[i, 8] = 0 if rc == 0 else 1
            tml.append(tm)

    run_threads([(worker, (e, k)) for k, e in enumerate(engs)])
    allp = P.allgather_words(step_pts, device=COLL_DEVICE if world > 1 else None)      # (world, steps, 9)
    for i in range(args.steps):
        ok = ok and not allp[:, i, 8].any() and not E.host_points_sum(args.curve, np.ascontiguousarray(allp[:, i, :8])).any()
    ok = ok and len(tml) == args.steps
    tms = np.sum(np.array(tml), axis=0)
    barrier(world)
    host_cpu = cgroup_cpu_delta(cpu0, time.perf_counter() - t0)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    assert ok, "batch verification of valid proofs failed"
    vfe_dev = sum(e.vfe_stats()[0] for e in engs)
    vfe_fb = sum(e.vfe_stats()[1] for e in engs)
    for e in engs[1:]:
        e.close()
    # kernel times for the roofline: ONE batch run alone after the timed region (HIP events on the ctx's stream; with batches in
    # flight a launch's duration would include its neighbour's share of the GPU)
    eng.set_profiling(True)
    eng.reset_profiling()
    rc, _, _ = eng.batch_verify(inst, seed, alpha_skip=lo, want_point=True)
    assert rc == 0
    steps_profiled = [1]
    k = int(np.log2(N))
    m_commit = 2 * shuffle_k if shuffle_k else nval
    per_proof_bytes = 352 * N + 96 * (13 + m_commit + 2 * k)
    vs_ms, vs_n = eng.kernel_time(5)
    res = {
        "metric": "r1cs_batch_verifies_per_sec", "value": total * args.steps / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": ("batch_verify of %d k-shuffle proofs per GPU (k = %d: %d randomized multipliers padded to %d, m = %d; benches/r1cs_secq256k1.rs:201-250), %s"
                                % (args.proofs, shuffle_k, 2 * (shuffle_k - 1), N, m_commit, CURVES[args.curve])) if shuffle_k else
                               ("cfg4: batch_verify of %d R1CS proofs %s, 2^14 constraints each (256 x 64-bit range proofs, m=256), %s"
                                % (args.proofs, ("in ONE batch sharded across %d GPUs (%d per GPU)" % (world, hi - lo)) if strong else "per GPU", CURVES[args.curve])),
                   "proofs_per_gpu": hi - lo, "distinct_proofs": len(distinct), "constraints_per_proof": N, "parallelism": "proof-sharded x%d" % world,
                   "batches_in_flight": nfl, "host_cpu_in_timed_region": host_cpu,
                   "front_end": {"batches_on_the_device": int(vfe_dev), "batches_handed_to_the_host_replay": int(vfe_fb),
                                 "note": "per-proof codec, transcript replay (Keccak-f / STROBE / ChaCha20 -> Fr::rand) and challenge arithmetic as GPU kernels "
                                         "(csrc/vfe.hip); ARKBP_VFY_HOST=1 forces the host replay for an A/B"},
                   "stage_ms_per_step": {"whole_call": tms[0] / args.steps * 1e3, "host_replay_overlapped_with_gpu": tms[1] / args.steps * 1e3,
                                         "gpu_drain_and_tail_scaling": tms[2] / args.steps * 1e3, "final_msm": tms[3] / args.steps * 1e3,
                                         "decode": tms[4] / args.steps * 1e3}},
    }
    if vs_n:
        avg_s = vs_ms / vs_n * 1e-3
        # k_vfy_batch (one launch per block of proofs): per proof it stands for the reference's scalar generation (64*N B written, 96*N B of
        # wL/wR/wO read) — 160*N algorithmic bytes per proof; the fused kernel itself reads only the 3.3 KB parameter
        # block per proof and the shared CSC, and writes chunk partials
        nproofs_per_launch = inst.n * steps_profiled[0] / max(vs_n, 1)  # the batch goes through in blocks of 512 proofs, one k_vfy_batch launch each
        cfg4 = args.proofs == 4096 and args.curve == 0 and nproofs_per_launch == 512 and not shuffle_k
        traffic = pmc_traffic("verify4096/k_vfy_batch<Secq>", cfg4, "r02_pmc_verify4096_summary.json")
        # modular products per (proof, element) in k_vfy_batch: 4 split-table look-ups (alpha*a*s_i, b*s_{N-1-i}, alpha*y^-i, alpha*x*y^-i),
        # z^(q+1) per distinct constraint of the column (~3 for the range-proof gadget: 3.3 with its one non-unit coefficient per bit;
        # ~2.5 for the shuffle), x*w_L, the g and h products (3), the delta product and its weight: 9 + ~3.3 + 1
        prod_pe = 13.3 if not shuffle_k else 13.5
        products = prod_pe * N * nproofs_per_launch
        tb_ms, tb_n = eng.kernel_time(11)
        # the per-proof front end on the device (csrc/vfe.hip): one launch each per BATCH (all proofs at once)
        pt_ms, pt_n = eng.kernel_time(12)
        sp_ms, sp_n = eng.kernel_time(13)
        pp_ms, pp_n = eng.kernel_time(14)
        npts = 11 + 2 * k
        fe_rows = []
        if sp_n:
            msg_bytes = (m_commit + npts) * 65 + 3 * 32
            fe_rows = [
                kernel_entry("k_vfe_points (decompression + serialization of every point of the batch)", pt_ms / pt_n, 1,
                             float(inst.n) * (33 * npts + 64 * m_commit + 64 * (npts + m_commit) + 72 * (npts + m_commit + 3)),
                             pmc_traffic("verify4096/vfe::k_vfe_points<Secq>", cfg4, "r02_pmc_verify4096_summary.json", raw_fetch=True), float(inst.n) * (npts * 385.0 + m_commit * 8.0),
                             "VALU-bound in the %d square roots per proof (~370 products each)" % npts),
                kernel_entry("k_vfe_sponge (merlin / STROBE replay + ChaCha20 -> Fr::rand, one lane per proof)", sp_ms / sp_n, 1, float(inst.n) * msg_bytes,
                             pmc_traffic("verify4096/vfe::k_vfe_sponge<Secq>", cfg4, "r02_pmc_verify4096_summary.json"), None,
                             "a serial chain per proof: ~%d Keccak-f[1600] permutations of ~6 K VALU instructions (no modular products); %d waves for the whole batch — "
                             "latency-bound, overlaps with other batches' k_vfy_batch" % (msg_bytes // 166 + 2 * (6 + k) + 4, (inst.n + 63) // 64)),
                kernel_entry("k_vfe_consts + k_vfe_wv + k_vfe_sum2 (inversion, power tables, tail scalars)", pp_ms / pp_n, 3, None, None, float(inst.n) * (520.0 + 3 * k + 9.0 * m_commit), ""),
            ]
        res["roofline"] = {"bound": "hbm", "kernel": "k_vfy_batch (one launch per block of 512 proofs)", "per": "one launch (%d proofs x %d elements)" % (int(nproofs_per_launch), N),
                           "achieved": 160.0 * N * nproofs_per_launch / avg_s / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 160.0 * N * nproofs_per_launch / avg_s / 1e9 / HBM_PEAK_GBS,
                           "traffic": traffic,
                           "traffic_note": "below the algorithmic bytes: the 2N scalars of a proof are never materialised (fused into the alpha-weighted accumulation)",
                           "avg_kernel_ms": vs_ms / vs_n, "algorithmic_bytes_per_verify": per_proof_bytes,
                           "valu": valu_entry(products, avg_s, "k_vfy_batch is integer-VALU-bound: ~%.1f modular products per (proof, element) in the scalar field "
                                              "(pseudo-Mersenne since round 4) against the ceiling derived from the issue rates" % prod_pe, scalar_field=True),
                           "kernels": [kernel_entry("k_vfy_batch", vs_ms / vs_n, 1, 160.0 * N * nproofs_per_launch, traffic, products),
                                       kernel_entry("k_vfy_tables (per-proof split tables)", tb_ms / max(tb_n, 1), 1, None, None, None)] + fe_rows}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is reported at N = 1 only
        from oracle import pyoracle as O

        m = max(1, min(args.cpu_verify_proofs, total))
        sample = [distinct[i % len(distinct)] for i in range(m)]
        tim = []
        rc = O.batch_verify(args.curve, sample, N, seed, timing=tim)
        assert rc == 0
        res["cpu_baseline"] = {"value": m / tim[0], "unit": "proofs/s", "cores": 1, "kind": "port",
                               "sample": "batch_verify of the first %d instances of the same batch (%.1f s), reference algorithm restated in C++" % (m, tim[0])}
    eng.close()
    return res


def cpu_baseline_prove(args):
    """CPU restatement of the reference prover (oracle/protocol.hpp: per-element 2-term msm + into_affine in the
    folds, ark window schedule), 1 thread, bounded sample of the same circuit family."""
    from oracle import pyoracle as O

    logn = min(args.logn, args.cpu_logn)
    n = 1 << logn
    pr = O.r1cs_prove(args.curve, O.SC_SQUARE_CHAIN, [n, 0], bytes([3]) * 32, n, m_cap=8)
    assert pr.rc == 0
    out = {"value": n / pr.t_prove, "unit": "constraints/s", "cores": 1, "kind": "port",
           "sample": "square-chain circuit with 2^%d constraints (prove() only, %.1f s), reference algorithm restated in C++" % (logn, pr.t_prove)}
    # the reference's optional `parallel` feature (Cargo.toml:76: rayon over the Pippenger windows of every msm call, nothing else),
    # restated with OpenMP, on every core this process may use (SURVEY.md §8d asks for it beside the default-features figure)
    cores = max(1, min(cpu_quota(), 64))
    if cores > 1:
        O.set_msm_threads(cores)
        try:
            pp = O.r1cs_prove(args.curve, O.SC_SQUARE_CHAIN, [n, 0], bytes([3]) * 32, n, m_cap=8)
        finally:
            O.set_msm_threads(1)
        assert pp.rc == 0 and pp.proof == pr.proof
        out["all_cores"] = {"value": n / pp.t_prove, "unit": "constraints/s", "cores": cores, "kind": "port",
                            "sample": "the same statement with the msm windows of every call spread over %d threads (%.1f s): the 2-term msm of each folded "
                                      "generator dominates and has little to spread" % (cores, pp.t_prove)}
    return out


def cpu_baseline_msm(args, bases, sc):
    """CPU restatement of ark-ec's VariableBaseMSM (oracle/curve.hpp), 1 thread, same inputs (bounded sample)."""
    from oracle import pyoracle as O

    m = min(len(bases), 1 << 16)
    _, secs = O.msm(args.curve, bases[:m], sc[:m], timed=True)
    return {"value": m / secs, "unit": "terms/s", "cores": 1, "kind": "port",
            "sample": "%d-term MSM (same bases/scalars), ark window schedule, single thread" % m}


def run_shuffle_sweep(args, rank, world, local):
    """The reference's own benchmark (benches/r1cs_secq256k1.rs:152-189 `bench_kshuffle_prove`, :201-250 `bench_kshuffle_verify`;
    benches/r1cs_zorro.rs is the same with the zorro types): k-shuffle proof creation and verification for k = 2, 4, ..., 1024,
    `BulletproofGens::new(2048, 1)`, one proof at a time (criterion times single calls).  Per k and curve: GPU engine (statement
    construction + prove, as the reference's closure does; Verifier::new + commits + gadget + verify) against the CPU restatement
    of the reference algorithm on the host's cores.  The k = 2 row on secq256k1 is BASELINE cfg1."""
    import ark_bulletproofs_amd as A
    from ark_bulletproofs_amd import engine as E
    from oracle import pyoracle as O

    ks = getattr(args, "sweep_ks", None) or [1 << i for i in range(1, 11)]
    rows = []
    reps = max(3, args.steps)
    for curve in ([args.curve] if args.sweep_one_curve else [0, 1]):
        eng = A.Engine(curve=curve, device=local)
        eng.gens_derive(2048)
        for k in ks:
            m_cap = 2 * k + 8
            seed = statement_seed(7, k)
            for _ in range(max(1, args.warmup)):
                pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=m_cap)
                assert eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics) == 0
            # one call at a time, as criterion does; the MEDIAN of the calls is the row's figure (a single stall of the host — a page
            # fault storm, a collector pause — is tens of calls long at these sizes), the mean rides along
            tp, tv = [], []
            for r in range(reps):
                t0 = time.perf_counter()
                pr = eng.prove_scenario(E.SC_SHUFFLE, [k], seed, m_cap=m_cap)
                tp.append(time.perf_counter() - t0)
            for r in range(reps):
                t0 = time.perf_counter()
                rc = eng.verify_scenario(E.SC_SHUFFLE, [k], pr.proof, pr.commitments, pr.publics)
                tv.append(time.perf_counter() - t0)
            t_prove, t_verify = float(np.median(tp)), float(np.median(tv))
            assert rc == 0
            row = {"curve": CURVES[curve], "k": k, "multipliers": 2 * (k - 1), "gpu_prove_ms": t_prove * 1e3, "gpu_verify_ms": t_verify * 1e3,
                   "gpu_prove_ms_mean": float(np.mean(tp)) * 1e3, "gpu_verify_ms_mean": float(np.mean(tv)) * 1e3,
                   "gpu_prove_inside_prove_ms": pr.timing[0] * 1e3}
            if not args.no_cpu_baseline:
                ref = O.r1cs_prove(curve, O.SC_SHUFFLE, [k], seed, 2048, m_cap=m_cap)
                assert ref.rc == 0 and ref.proof == pr.proof, "GPU proof differs from the CPU restatement's (k = %d)" % k
                tim = []
                assert O.r1cs_verify(curve, O.SC_SHUFFLE, [k], 2048, ref.proof, ref.commitments, ref.publics, timing=tim) == 0
                row.update({"cpu_prove_ms": (ref.t_prove + ref.t_setup) * 1e3, "cpu_prove_inside_prove_ms": ref.t_prove * 1e3, "cpu_verify_ms": tim[0] * 1e3 if tim else None,
                            "proof_bytes_identical": True})
            rows.append(row)
        direct_msms, direct_cap = eng.direct_stats()
        for r in rows:
            if r["curve"] == CURVES[curve]:
                r["small_statement_path"] = bool(direct_cap)   # csrc/small.cuh: direct window tables, no generator folding (BP_TUNE_DIRECT_MAX)
        eng.close()
    big = [r for r in rows if r["k"] == max(ks) and r["curve"] == CURVES[args.curve]][0]
    return {"metric": "kshuffle_prove_ms (k = 1024; the whole sweep under \"rows\")", "value": big["gpu_prove_ms"], "unit": "ms", "n_gpus": world, "steps": reps, "warmup": args.warmup,
            "ms_per_step": big["gpu_prove_ms"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "the reference's criterion sweep: k-shuffle prove / verify, k = 2..1024, one proof at a time, BulletproofGens::new(2048, 1) "
                                   "(benches/r1cs_secq256k1.rs:152-250, benches/r1cs_zorro.rs)", "cfg1": "row k = 2, secq256k1"},
            "rows": rows,
            "cpu_baseline": None if args.no_cpu_baseline else {"value": big.get("cpu_prove_ms"), "unit": "ms", "cores": 1, "kind": "port",
                                                               "sample": "the same k = 1024 statement, reference algorithm restated in C++ (every row carries its own CPU column)"}}


def run_cfg5(args, rank, world, local):
    """BASELINE cfg5 — ONE large proof at a time across all ranks (strong scaling): Pippenger windows of every MSM partitioned
    across the GPUs with the RCCL point-reduce, the inner-product argument partitioned index-cyclically.  Every
    rank runs prove() on the same statements and ends with the identical proof."""
    import copy

    a5 = copy.copy(args)
    a5.shard, a5.logn = "windows", args.cfg5_logn
    a5.batch, a5.steps, a5.warmup = 8, max(1, min(args.steps, args.cfg5_steps)), 0       # 8 statements = one lockstep group of the TranscriptRng stage
    a5.inflight, a5.tables_off_steps, a5.msm_tables = 1, 0, 0
    a5.window = 8
    t0 = time.perf_counter()
    r = run_prove(a5, rank, world, local)
    out = {"metric": "r1cs_constraints_proved_per_sec", "value": r["value"], "unit": r["unit"], "n_gpus": world, "scaling": "strong", "steps": a5.steps,
           "ms_per_step": r["ms_per_step"], "ms_per_proof": r["ms_per_step"] / a5.batch, "proofs_per_step": a5.batch,
           "config": {"workload": "cfg5: 2^%d-constraint R1CS prove, Pippenger windows + index-cyclic IPA partitioned across %d GPUs, %s" % (a5.logn, world, CURVES[args.curve]),
                      "verified": r["config"]["verified"], "collectives": r["config"]["collectives"], "first_round_fold_tables": r["config"]["first_round_fold_tables"],
                      "in_pipeline_latency_ms": r["config"]["in_pipeline_latency_ms"], "alone_ms_after_rng_head": r["config"]["alone_ms_after_rng_head"],
                      "rng_head_ms": r["config"]["rng_head_ms"],
                      "prove_ms_excluding_rng_head": r["config"]["per_proof_stage_ms"].get("prove_total"),
                      "prove_note": "the TranscriptRng heads of the statements are computed by the pipeline's host threads BEFORE prove() (bp_stmt_precompute_batch, "
                                    "eight sponges in lockstep): prove_ms_excluding_rng_head is the mean wall time inside prove() with ONE proof partitioned across all "
                                    "ranks; rng_head_ms (one core, sequential by the reference's construction) is reported beside it, not inside",
                      "per_rank_table_GB": (r["config"].get("first_round_fold_tables") or {}).get("GB"),
                      "per_proof_stage_ms": r["config"]["per_proof_stage_ms"],
                      "pipeline_thread_seconds_per_wall_second": r["config"]["pipeline_thread_seconds_per_wall_second"]},
           "wall_s_including_setup": time.perf_counter() - t0}
    if "roofline" in r:
        out["roofline"] = {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "fold_ms_per_proof", "msm_kernels_ms_per_proof") if k in r["roofline"]}
    return out


def auto_window(logn, world):
    """statements alive at once in the prove pipeline: ~0.35 GB of host memory each at 2^20; a quarter of this rank's share of the limit"""
    per_stmt = 0.35e9 * max(1.0, (1 << logn) / float(1 << 20))
    return int(max(16, min(128, (mem_limit_bytes() * 0.25 / max(world, 1)) / per_stmt)))


def run_headline(args, rank, world, local):
    """both halves of BASELINE.json's metric on one box: prove (top level), then batch verify ("verify"); with N > 1 ranks also the
    north_star's partition of one large proof ("cfg5")"""
    import copy

    res = run_prove(args, rank, world, local)
    ver = run_verify(args, rank, world, local)
    res["metric"] = "r1cs_constraints_proved_per_sec (+ r1cs_batch_verifies_per_sec under \"verify\")"
    res["verify"] = ver
    res["hbm_spent_on_tables_GB"] = ((res["config"].get("first_round_fold_tables") or {}).get("GB", 0.0)) + ((res["config"].get("fixed_base_msm_tables") or {}).get("GB", 0.0))

    def leg(name, fn, **over):
        """one more configuration of BASELINE.json in the same line; a failing leg is recorded, the line still comes"""
        a = copy.copy(args)
        for k_, v_ in over.items():
            setattr(a, k_, v_)
        t_leg = time.perf_counter()
        try:
            out = fn(a, rank, world, local)
            out["wall_s_including_setup"] = time.perf_counter() - t_leg
            res[name] = out
        except Exception as exc:
            res[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:400])}
            res.setdefault("errors", []).append("%s: %s" % (name, res[name]["error"]))

    if world > 1:
        # cfg4 in its stated form beside the per-GPU batches above: ONE batch of 4096 proofs sharded across the ranks
        leg("verify_cfg4_strong", run_verify, verify_strong=True, no_cpu_baseline=True)
    if world == 1 and not args.headline_only:
        # the other configurations of BASELINE.json on this one GPU (every config in the driver's line)
        leg("cfg1", run_shuffle_sweep, sweep_ks=[2, 64, 1024], sweep_one_curve=True, steps=20, warmup=3)             # k = 2 shuffle (cfg1) and two more rows of the reference's sweep, GPU beside the CPU restatement
        leg("msm", run_msm, terms=1 << 16, steps=200, warmup=20, shard="terms")                                      # cfg2
        # cfg3 on the zorro curve: no endomorphism there, so the fold tables are twice the size per window width and worth more (w = 5 / 84 GB
        # 14.0 M constraints/s, w = 7 / 238 GB 19.1 M): this leg may take 150 GB (w = 6)
        leg("zorro", run_prove, curve=1, steps=10, warmup=2, tables_off_steps=0, no_cpu_baseline=True, fold_table_budget_gb=150.0)
        leg("prove_2p22", run_prove, logn=22, batch=8, steps=6, warmup=1, tables_off_steps=0, no_cpu_baseline=True, window=auto_window(22, world))  # cfg5's size on ONE GPU (its 8-GPU partition: "cfg5" with --gpus N)
    if world > 1 and args.cfg5_logn > 0:
        res["cfg5"] = guarded_cfg5(args, rank, world, local, res)
        if "error" in res["cfg5"]:
            res.setdefault("errors", []).append("cfg5: " + res["cfg5"]["error"])
    return res


def guarded_cfg5(args, rank, world, local, res):
    """The cfg5 leg is the one part of the headline run whose collectives no builder box could exercise with more than one GPU (a
    one-GPU box cannot host two RCCL ranks): a rank failing alone there would leave the others waiting inside a collective and the
    run without its JSON line.  So the leg runs under a timer: when it fires, rank 0 prints the headline line (prove + verify are
    complete at this point) with the failure recorded under "cfg5", and every rank leaves."""
    import threading

    limit = args.cfg5_timeout
    fallback_line = None
    if rank == 0:
        msg = "the cfg5 leg did not finish within %d s (a rank failed or a collective did not complete); prove and verify above are complete" % limit
        fallback_line = json.dumps(dict(res, cfg5={"error": msg}, errors=["cfg5: " + msg]))

    def bail():
        if rank == 0:
            print(fallback_line, flush=True)
        os._exit(3 if args.strict_exit else 0)

    timer = threading.Timer(limit, bail)
    timer.daemon = True
    timer.start()
    try:
        out = run_cfg5(args, rank, world, local)
    except Exception as exc:       # (this rank failed alone: the others leave through their timers)
        out = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:400]), "rank": rank}
    timer.cancel()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="headline", choices=["headline", "prove", "verify", "msm", "shuffle-sweep"])
    ap.add_argument("--proofs", type=int, default=4096, help="proofs per GPU per batch (verify workload)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct proofs generated for the verify workload")
    ap.add_argument("--verify-inflight", type=int, default=6, help="verify workload: batch_verify calls in flight per GPU (own ctx and host pool each; the pools divide the process's CPU quota)")
    ap.add_argument("--shuffle-k", type=int, default=0, help="verify workload: batches of k-shuffle proofs (the reference's two-phase benchmark circuit) instead of cfg4's range proofs")
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--cpu-logn", type=int, default=15, help="CPU baseline sample of the prove workload: 2^cpu_logn constraints (about 13 s)")
    ap.add_argument("--cpu-verify-proofs", type=int, default=256, help="CPU baseline sample of the verify workload (about 10 s)")
    ap.add_argument("--batch", type=int, default=16, help="independent proofs per step per GPU (prove workload)")
    ap.add_argument("--host-threads", type=int, default=0, help="host threads running the TranscriptRng head of prove() (0 = half of this rank's share of the CPU quota)")
    ap.add_argument("--build-threads", type=int, default=0, help="host threads constructing statements (Prover::new + commit + gadget) (0 = 6/16 of this rank's share of the CPU quota)")
    ap.add_argument("--inflight", type=int, default=8, help="independent proofs in flight per GPU (prove workload)")
    ap.add_argument("--window", type=int, default=0, help="statements alive at once in the prove pipeline (built, waiting for or in the TranscriptRng stage, "
                    "on the GPU); ~0.3 GB of host memory each at 2^20.  Little's law: a statement spends ~3 s in the pipeline, so 32 caps the rate at ~11 proofs/s and 64 at ~21.  0 = 128, or fewer when a quarter of this rank's share of the host memory limit holds fewer")
    ap.add_argument("--fold-tables", type=int, default=2, help="prove workload: fixed-base tables of the generators for the first fold rounds (0 = off, 1 = the first round, "
                    "2 = the first two rounds: tables over 3N/4 bases)")
    ap.add_argument("--msm-tables", type=int, default=1, help="prove workload: fixed-base rows of the generators for the MSMs over the tables themselves (0 = off)")
    ap.add_argument("--sweep-one-curve", action="store_true", help="shuffle-sweep workload: only --curve (default: secq256k1 and zorro)")
    ap.add_argument("--cfg5-logn", type=int, default=22, help="headline with --gpus N > 1: size of the window-sharded proofs of the cfg5 leg (0 = skip the leg)")
    ap.add_argument("--cfg5-steps", type=int, default=2, help="steps (of 8 proofs) of the cfg5 leg")
    ap.add_argument("--cfg5-timeout", type=int, default=600, help="seconds after which the cfg5 leg is given up and the headline line printed without it")
    ap.add_argument("--verify-wait-us", type=int, default=0, help="verify workload: BP_TUNE_WAIT_SLEEP of the batch ctxs in microseconds (0 = HIP's busy wait)")
    ap.add_argument("--freeze-len", type=int, default=0, help="prove workload: BP_TUNE_IPA_FREEZE_LEN of every ctx (0 = the library's default)")
    ap.add_argument("--tables-off-steps", type=int, default=4, help="prove workload: timed steps of the same pipeline with the precomputed tables released (0 = skip)")
    ap.add_argument("--fold-table-bits", type=int, default=0, help="window width of those tables (0 = the widest that fits the budget below)")
    ap.add_argument("--fold-table-budget-gb", type=float, default=125.0, help="HBM the fold tables may take (0 = whatever is free): 125 GB = w 7 at 2^20 on secq256k1, within 2 %% of w 8 at 219 GB")
    ap.add_argument("--terms", type=int, default=1 << 16)
    ap.add_argument("--shard", default="terms", choices=["terms", "windows"],
                    help="multi-GPU partition: msm workload: terms | Pippenger windows; prove workload: replicas (default) | windows = every rank proves the same statements with window-sharded MSMs")
    ap.add_argument("--curve", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="headline workload at --gpus 1: only prove (cfg3) + verify (cfg4), without the cfg1 / cfg2 / zorro / 2^22 legs")
    ap.add_argument("--verify-strong", action="store_true", help="verify workload with --gpus N: ONE batch of --proofs sharded across the ranks (BASELINE cfg4 as stated) instead of --proofs per GPU")
    ap.add_argument("--strict-exit", action="store_true", help="exit with status 3 (after printing the line) when a leg of the run failed, e.g. the multi-GPU cfg5 leg")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))        # this process has not touched the GPU
    rank, world, local = dist_setup(args.gpus)
    # host thread counts of the prove pipeline from this rank's share of the CPUs (the ranks of a node share the cgroup's quota;
    # on a 16-CPU share: 8 TranscriptRng + 6 statement builders beside the 8 GPU driver threads, which sleep while they wait
    # (BP_TUNE_WAIT_SLEEP).  Six TranscriptRng threads feed a 26 M constraints/s GPU on the usual boxes — the seventh and eighth
    # wait there and cost nothing — but on the occasional box whose cores run that stage 3x slower they are the difference.)
    share = max(1.0, cpu_quota() / max(world, 1))
    if args.host_threads <= 0:
        args.host_threads = max(1, int(round(share * 8 / 16)))
    if args.window <= 0:
        args.window = auto_window(args.logn, world)
    if args.build_threads <= 0:
        args.build_threads = max(2, int(round(share * 6 / 16)))
    res = {"msm": run_msm, "prove": run_prove, "verify": run_verify, "headline": run_headline, "shuffle-sweep": run_shuffle_sweep}[args.workload](args, rank, world, local)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if isinstance(res, dict) and isinstance(res.get("cfg5"), dict) and "error" in res["cfg5"]:
        # (the other ranks may still sit in a collective of the failed leg: no orderly teardown is possible.)  The failure is in the
        # line ("cfg5.error" and the top-level "errors"); --strict-exit also turns it into exit status 3 for launchers that only look
        # at the status — the default stays 0 so that a scaling run keeps its complete prove / verify halves.
        os._exit(3 if args.strict_exit else 0)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
