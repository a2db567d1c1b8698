#!/usr/bin/env python3
"""bench.py -- hash-join build+probe throughput on MI355X (the BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 30] [--dist uniform]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N fresh rank processes of itself
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU) BEFORE it touches the GPU in any way, waits for them and
relays rank 0's JSON line; it exits non-zero if a rank fails or fewer than N devices are visible. Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's N > 1 command) the ranks are
already there and each process is one of them; `--gpus` that disagrees with WORLD_SIZE is an error, never a silent N=1.

One "step" = one pass of the hot path over one batch: build(R) -> probe(S) (the build writes
every reachable table slot once, empties included: there is no separate clear) with R and S
already resident in HBM (DataGen inputs, generated on the host and copied
in before the timed region). N=1 workload: |R| = |S| = 2^30 uint32-key tuples, `uniform`
(BASELINE.json metric; configs[1]'s operator at the metric's size). For N > 1 (launched
with torch.distributed.run, one rank per GPU) every rank holds its own 2^log2n tuples of
R and of S ("weak" scaling; --strong splits the N=1 total instead); tuples are exchanged
by key radix with one all-to-all per relation over RCCL and joined locally
(htm_hashjoin_amd/sharded.py).

Rank 0 prints ONE JSON line. `value` = (|R|+|S|) summed over all ranks / max-over-ranks
time, in Mtuples/s. Extra objects: `roofline` (dominant kernel, HIP-event timed on the
launch stream), `cpu_baseline` (the oracle's threaded port on a bounded sample, rank 0,
N=1 only), `other_workloads` (local_shuffle and PRJ measured in the same process).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; default: WORLD_SIZE if a launcher set it, else 1. N > 1 without "
                         "WORLD_SIZE: bench.py starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=30, help="per-GPU |R| = |S| = 2^log2n")
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--shuffle-range", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other_workloads legs")
    ap.add_argument("--cpu-sample-log2n", type=int, default=27)
    ap.add_argument("--build-variant", type=int, default=0,
                    help="0 auto, 1 global atomics, 2 workgroup LDS window, 3 wavefront LDS rings")
    ap.add_argument("--strong", action="store_true", help="N>1: split the N=1 total instead of 2^log2n per GPU")
    ap.add_argument("--no-other-split", action="store_true",
                    help="N>1: skip the few extra steps with the split that was NOT chosen (reported as other_split)")
    ap.add_argument("--split", default="low", choices=["auto", "low", "high"],
                    help="N>1: destination = low key bits (default: BASELINE config 4's all-to-all, (G-1)/G of the tuples "
                         "cross xGMI), high key bits (range split), or auto = high when >= 3/4 of every rank's tuples stay "
                         "put under it (sharded.py)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "a2a"],
                    help="N>1: batch of pairwise isend/irecv per relation, or one all_to_all_single per relation "
                         "(falls back to p2p for a step whose largest per-peer message is >= 512 MiB); auto = whichever "
                         "form the pre-flight exchange (sharded.exchange_selfcheck: both forms at 1 MiB per peer and at "
                         "the step's own size, every element checked) delivered correctly and faster")
    a = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus is None:
        a.gpus = int(env_world) if env_world else 1
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    return a


def kernel_source_hash():
    """sha256 over the kernel sources (htm-hashjoin_amd/csrc/*.hip, *.h, *.cpp): profiles/pmc_traffic.json records the
    hash of the sources its PMC passes were taken on (tools/summarize_prof.py traffic), and roofline.traffic is reported
    only while the sources are still those -- a committed byte count never describes a kernel that has changed since."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "htm-hashjoin_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def committed_traffic(kernel, workload_ok):
    """HBM bytes per launch of `kernel` from profiles/pmc_traffic.json, or (None, why)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, "no profiles/pmc_traffic.json"
    if not workload_ok:
        return None, "the PMC passes were taken on another workload"
    d = json.load(open(pmc))
    if d.get("_source_hash") != kernel_source_hash():
        return None, "kernel sources changed since the PMC passes (profiles/pmc_traffic.json: _source_hash)"
    return d.get(kernel.split("<")[0]), None


def to_device(np_u64, torch, dev):
    t = torch.from_numpy(np_u64.view("int64"))
    return t.to(dev, non_blocking=False)


def time_steps(torch, dist_mod, world, fn, steps, warmup):
    """W untimed + exactly K timed steps between barrier+synchronize; returns max-over-ranks seconds."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def oa_leg(torch, hj, ctx, n, dist, window, steps, warmup, S_dev=None, variant=0, keep=None):
    """Open-addressing build+probe on one GPU. Returns (result dict, S_dev). keep: a list that receives the host R
    (the CPU baseline joins the same relation afterwards)."""
    R = hj.generate_data(dist, n, n, window)
    R_dev = to_device(R, torch, "cuda")
    if keep is not None:
        keep.append(R)
    del R
    if S_dev is None:
        S_dev = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")   # generate_data("sorted"), main.cpp:93
    ctx.reserve("atomic", n, n, buildVariant=variant)
    kernel_us = {"clear_us": [], "build_us": [], "probe_us": [], "buildPhaseA_us": []}

    def step():
        ctx.build(R_dev.data_ptr(), n)
        ctx.probe(S_dev.data_ptr(), n)

    dt = time_steps(torch, None, 1, step, steps, warmup)
    # per-kernel device times of the same launches, from HIP events on the launch stream
    for _ in range(min(steps, 5)):
        step()
        r = ctx.fetch()
        for k in kernel_us:
            kernel_us[k].append(r[k])
    ctx.checksums()
    res = ctx.fetch()
    avg = {k: sum(v) / len(v) for k, v in kernel_us.items()}
    out = {
        "dist": dist, "shuffleRange": window, "rSize": n, "sSize": n,
        "ms_per_step": dt / steps * 1e3,
        "mtuples_per_s": 2 * n / (dt / steps) / 1e6,
        "kernel_us": avg,
        "conflicts": res["conflicts"], "totalMatches": res["totalMatches"], "inputSum": res["inputSum"],
        "buildVariant": res["buildVariant"], "buildDeferred": res["buildDeferred"], "compactFallback": res["compactFallback"],
        "checks": {
            "matches_plus_conflicts_eq_rSize": res["totalMatches"] + res["conflicts"] == n,
            "tableSum_plus_conflictSum_eq_inputSum": res["tableSumFull"] + res["conflictSum"] == res["inputSum"],
        },
    }
    del R_dev
    return out, S_dev


def prj_traffic(n, dist, window, path):
    """HBM bytes of the pass-1 scatter of R from the committed PMC passes (profiles/pmc_traffic.json), for the one workload
    they were taken on; None otherwise."""
    return committed_traffic("k_radix_scatter_frag_pass1_R_local_shuffle_1024",
                             n == 1 << 30 and dist == "local_shuffle" and window == 1024 and path == 1)[0]


def prj_leg(torch, hj, ctx, n, dist, window, steps, warmup, S_dev):
    R = hj.generate_data(dist, n, n, window)
    R_dev = to_device(R, torch, "cuda")
    del R
    ctx.reserve("prj", n, n)

    def step():
        ctx.prj_join(R_dev.data_ptr(), n, S_dev.data_ptr(), n)

    dt = time_steps(torch, None, 1, step, steps, warmup)
    step()
    res = ctx.fetch()
    return {
        "algo": "prj", "dist": dist, "shuffleRange": window, "rSize": n, "sSize": n,
        "radixBits": res["radixBits"], "ms_per_step": dt / steps * 1e3,
        "mtuples_per_s": 2 * n / (dt / steps) / 1e6,
        "partition_us": res["partition_us"], "join_us": res["join_us"],
        "totalMatches": res["totalMatches"],
        # 1 = the histogram-free passes held (12 + 8 + 4 = 24 B per tuple), 2 = they overflowed and the exact passes
        # (32 B per tuple: every pass reads its input twice) redid the join, 0 = exact passes only
        "prjPath": res["prjPath"],
        "hbm_frac_of_32B_per_tuple": 32.0 * 2 * n / (dt / steps) / 1e9 / HBM_PEAK_GBPS,
        "hbm_frac_of_24B_per_tuple": 24.0 * 2 * n / (dt / steps) / 1e9 / HBM_PEAK_GBPS,
        # dominant kernel of the radix join: the pass-1 scatter (tuples in, keys out: 8 B read + 4 B written per tuple),
        # HIP-event timed on the launch stream for R's launch (hj_result.prjScatterPass1R_us)
        "roofline": {"bound": "hbm", "kernel": ("k_radix_scatter_frag<false>" if res["prjPath"] == 1 else "k_radix_scatter") + " (pass 1, R)", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                     "algorithmic_bytes_per_launch": 12.0 * n, "launch_us": res["prjScatterPass1R_us"],
                     "achieved": 12.0 * n / (res["prjScatterPass1R_us"] * 1e-6) / 1e9 if res["prjScatterPass1R_us"] else None,
                     "frac": 12.0 * n / (res["prjScatterPass1R_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS if res["prjScatterPass1R_us"] else None,
                     "traffic": prj_traffic(n, dist, window, res["prjPath"])},
        "checks": {"matches_eq_n": res["totalMatches"] == n},
    }


def effective_cpus():
    """CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota (a GPU box hands a
    container 16 of the host's 64+ CPUs; os.cpu_count() would report the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, -(-int(quota) // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(hj, log2n, dist, window, R=None):
    """The oracle's threaded port (oracle/hj_oracle.c: orc_build_probe_mt) on the host cores: the SAME relation the
    GPU leg joined when R is handed over (the metric's 2^30), else the same distribution at 2^log2n; 64 chunks as in
    parallel_for(blocked_range(0, rSize, rSize/64)), min(64, effective CPUs) threads. A few seconds of CPU work."""
    from oracle import oracle   # checker/baseline only
    cores = effective_cpus()
    threads = min(64, cores)
    if R is None:
        n = 1 << log2n
        R = hj.generate_data(dist, n, n, window)
    n = R.size
    log2n = n.bit_length() - 1
    import numpy as np
    S = np.arange(1, n + 1, dtype=np.uint64)              # generate_data("sorted"), main.cpp:93
    best = None
    for _ in range(3):
        r = oracle.build_probe_mt(R, S, 4, 64, threads, atomic=True)
        us = r["build_us"] + r["probe_us"]
        best = us if best is None else min(best, us)
    rn = oracle.build_probe_mt(R, S, 4, 64, threads, atomic=False)
    # the same loops on unique keys (what the reference's sweep runs, and what tools/sweep.py's CPU legs time): no retries,
    # no failed CAS -- several times the rate of the duplicate-heavy `uniform`; measured at 2^27 so that it costs a blink.
    # (Round 2 read this difference as a slowdown with size: the rate on `uniform` is 2.8-3.0 Gtuples/s at every size from
    # 2^26 to 2^30, first-touched by the caller or by the worker threads alike -- tests/dev/cpu_baseline_sizes.py.)
    nu = 1 << min(27, log2n)
    Ru = hj.generate_data("local_shuffle", nu, nu, 16)
    Su = np.arange(1, nu + 1, dtype=np.uint64)
    ru = min((oracle.build_probe_mt(Ru, Su, 4, 64, threads, atomic=True) for _ in range(2)), key=lambda r_: r_["build_us"] + r_["probe_us"])
    unique_rate = 2 * nu / (ru["build_us"] + ru["probe_us"])
    del Ru, Su
    return {
        "unique_keys_sample": {"value": unique_rate, "unit": "Mtuples/s",
                               "sample": f"the same CAS loops on local_shuffle W=16 (unique keys), |R|=|S|=2^{nu.bit_length() - 1}, best of 2"},
        "value": 2 * n / best, "unit": "Mtuples/s", "cores": threads, "kind": "port",
        "sample": f"atomic (CAS) build+probe on the relation the GPU leg joined, {dist} W={window}, |R|=|S|=2^{log2n}, "
                  f"best of 3, {threads} threads on {cores} effective CPUs (affinity + cgroup quota; os.cpu_count() = "
                  f"{os.cpu_count()}); nocc (racy store) same input: {2 * n / (rn['build_us'] + rn['probe_us']):.0f} Mtuples/s",
    }


def cpu_baseline_prj(log2n):
    """The reference's own radix join, mc PRO, compiled from its sources (oracle/_ref/mchashjoins, oracle/Makefile), on
    the host cores: 2^log2n tuples per relation as in experiments/motivation.sh. In this fork PRO partitions R and S and
    builds R's tables but the probe is commented out (parallel_radix_join.c:259-276), so its time is a lower bound for
    a full join. None when the binary is not there (it is built where /root/reference is)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "mchashjoins")
    if not os.path.exists(exe):
        return None
    n = 1 << log2n
    threads = min(64, effective_cpus())
    try:
        us = None
        for _ in range(2):       # best of 2: the host share of a GPU box is a noisy place
            out = subprocess.run([exe, "--algo=PRO", f"--nthreads={threads}", f"--r-size={n}", f"--s-size={n}"],
                                 capture_output=True, text=True, timeout=90).stdout
            t = float(re.search(r"TOTAL-TIME-USECS[^\n]*\n\s*([0-9.]+)", out).group(1))
            us = t if us is None else min(us, t)
    except Exception as e:   # noqa: BLE001 -- a baseline that cannot run is reported as such, never fatal
        return {"error": repr(e)[:200]}
    return {"value": 2 * n / us, "unit": "Mtuples/s", "cores": threads, "kind": "reference",
            "sample": f"mc PRO (partition R and S + build R, probe disabled in the fork), |R|=|S|=2^{log2n} of its own "
                      f"pk/fk generator, best of 2: {us / 1e3:.1f} ms; compare other_workloads.prj_local_shuffle_1024"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n_ranks, argv):
    """--gpus N without a launcher: start N fresh children of this very script, one rank per GPU, and relay rank 0's
    JSON line. The parent has made no HIP call and never will (torch.cuda.device_count() reads the device list without
    initialising the runtime on this image); every child is a new process that initialises its own GPU -- no exec of a
    process that has touched the device. Returns the exit code."""
    import subprocess
    stub = os.environ.get("HJ_BENCH_TEST_ENGINE")
    if not stub:
        import torch
        have = torch.cuda.device_count()
        if have < n_ranks:
            print(f"bench.py --gpus {n_ranks}: only {have} HIP device(s) visible (there is no CPU fallback)", file=sys.stderr)
            return 3
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HJ_BENCH_LAUNCHER="self-spawned")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is the one JSON line; read it while waiting so a large line cannot fill the pipe
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n_ranks))
    # ranks that wait for each other forever (a collective one of them never joins) must not hold the caller forever
    deadline = time.time() + float(os.environ.get("HJ_BENCH_RANKS_TIMEOUT_S", "2400"))
    while alive:
        if time.time() > deadline:
            print(f"bench.py --gpus {n_ranks}: ranks {sorted(alive)} still running after the time limit; stopping them", file=sys.stderr)
            for o in alive:
                procs[o].terminate()
            for o in alive:
                try:
                    procs[o].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[o].kill()
            rc = 5
            break
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py --gpus {n_ranks}: rank {r} exited with {code}; stopping the others", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()                         # exactly the PIDs started above
        time.sleep(0.05)
    reader.join(timeout=10)
    text = (out[0] if out else "") or ""
    lines = [ln for ln in text.splitlines() if ln.strip().startswith("{")]
    if rc == 0 and not lines:
        print(f"bench.py --gpus {n_ranks}: rank 0 printed no JSON line", file=sys.stderr)
        rc = 4
    if rc == 0:
        print(lines[-1], flush=True)
    return rc


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))      # before anything touches the GPU
    if env_world is not None and int(env_world) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} disagrees with WORLD_SIZE={env_world} (refusing to report a "
                         f"{a.gpus}-GPU line from {env_world} rank(s))")
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST HOOK (tests/test_bench_launcher.py): "module:Class" of a stand-in compute engine. With it the ranks talk gloo
    # on the CPU, so the launcher, the rendezvous, the pre-flight exchange and the JSON contract can be driven where
    # there is no GPU. The product engine is the only one this file knows; nothing is ever measured with the stand-in
    # (the line says "engine": "TEST STAND-IN ...").
    stub = os.environ.get("HJ_BENCH_TEST_ENGINE") if world > 1 else None
    if stub is None:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} needs HIP device {local_rank}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
    import htm_hashjoin_amd as hj

    n = 1 << a.log2n
    if world > 1 or os.environ.get("HJ_BENCH_FORCE_SHARDED") == "1":   # the env var rehearses the N>1 code path on one GPU
        import torch.distributed as dist_mod
        from htm_hashjoin_amd import sharded
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        engine = device = None
        if stub is not None:
            import importlib
            mod, cls = stub.split(":")
            engine, device = getattr(importlib.import_module(mod), cls)(), "cpu"
        # RCCL prints a version banner on STDOUT when the first communicator comes up; the contract is one JSON
        # line there, so fd 1 points at stderr until the line is ready
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist_mod.init_process_group("gloo" if stub is not None else "nccl", rank=rank, world_size=world)
            line = sharded.bench_sharded(a, torch, dist_mod, hj, rank, world, local_rank, engine=engine, device=device)
            line["rccl_world"] = dist_mod.get_world_size()
            line["backend"] = dist_mod.get_backend()
            line["launcher"] = os.environ.get("HJ_BENCH_LAUNCHER", "external (torch.distributed.run)" if world > 1 else "none")
            if stub is not None:
                line["engine"] = f"TEST STAND-IN {stub} over gloo: not a measurement"
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist_mod.destroy_process_group()
        return

    stream = torch.cuda.current_stream().cuda_stream
    ctx = hj.HashJoinContext(local_rank, stream=stream)
    host_R = [] if not a.no_cpu_baseline else None
    main_leg, S_dev = oa_leg(torch, hj, ctx, n, a.dist, a.shuffle_range, a.steps, a.warmup, variant=a.build_variant, keep=host_R)

    # roofline of the dominant kernel (the build): algorithmic bytes = 16 B per R tuple
    # (8 read + 8 slot write), duration = HIP-event time of that launch
    ku = main_leg["kernel_us"]
    v2 = main_leg["buildVariant"] >= 2          # an LDS build (2: workgroup window, 3: wavefront rings, 4: rings + compact table): its kernel timed alone
    # Dominant kernel = the build. Variant 2: k_build_own (phase A) timed alone by its own HIP events
    # (hj_result.buildPhaseA_us); build_us additionally covers k_clear_unowned + k_build_deferred.
    # Algorithmic bytes per launch: build = R read 8 + slot write 8 = 16 B per R tuple;
    # probe = S read 8 + home-slot read 8 = 16 B per S tuple; table clear = 16 B per R tuple (2|R| slots).
    dom_name = {4: "k_build_wave<compact>", 3: "k_build_wave", 2: "k_build_own"}.get(main_leg["buildVariant"], "k_build_atomic_min")
    dom_us = ku["buildPhaseA_us"] if v2 else ku["build_us"]
    alg = 16.0 * n
    achieved = alg / (dom_us * 1e-6) / 1e9
    # HBM bytes per launch from rocprofv3 PMC passes of this very command -- only while the kernel sources are the ones
    # the passes were taken on
    traffic, traffic_note = committed_traffic(dom_name, a.log2n == 30 and a.dist == "uniform" and a.shuffle_range == 16)
    # what the kernel moves by design: R read once (8 B) + every reachable slot written once -- 8 B per slot, or 4 B with
    # the compact table (the index words that order the inserts stay in LDS); SURVEY.md 8d's algorithmic 16 B per tuple is
    # what `achieved` is computed from either way
    moved_model = (12.0 if main_leg["buildVariant"] == 4 else 16.0) * n
    # Whole step in SURVEY.md 8d's accounting: 16 B per R tuple (read 8 + slot written 8) + 16 B per S tuple (read 8 +
    # home slot 8) = 32 B per tuple pair. The table clear is NOT a separate 16 B: the LDS builds write every reachable
    # slot exactly once, empties included (PMC round 1: build group 17.3 GB at 2^30, not 34 GB).
    step_bytes = 32.0 * n
    roofline = {"bound": "hbm", "kernel": dom_name,
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": alg, "bytes_moved_by_design_per_launch": moved_model, "launch_us": dom_us,
                "other_kernels": {
                    "build_group_us (seam/bounds pre-pass + LDS build kernel + edge/unowned clear + deferred phase)" if v2 else "k_fill_empty_us":
                        ku["build_us"] if v2 else ku["clear_us"],
                    "k_probe_us": ku["probe_us"], "k_probe_GBps": 16.0 * n / (ku["probe_us"] * 1e-6) / 1e9,
                    "k_sample_locality_us (the pre-round decides on the device: no host read-back)": ku["clear_us"] if v2 else None},
                "whole_step": {"algorithmic_bytes": step_bytes, "bytes_per_tuple_pair": 32,
                               "GBps": step_bytes / (main_leg["ms_per_step"] * 1e-3) / 1e9,
                               "frac": step_bytes / (main_leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}}

    extra = {}
    if not a.no_extra:
        k2 = max(2, a.steps // 2)
        extra["oa_local_shuffle_1024"], _ = oa_leg(torch, hj, ctx, n, "local_shuffle", 1024, k2, 1, S_dev,
                                                   variant=a.build_variant)
        # unique keys within the reference's default shuffle window: the compact rings without retry rounds
        extra["oa_local_shuffle_16"], _ = oa_leg(torch, hj, ctx, n, "local_shuffle", 16, k2, 1, S_dev,
                                                 variant=a.build_variant)
        ctx2 = hj.HashJoinContext(local_rank, stream=stream)
        extra["prj_local_shuffle_1024"] = prj_leg(torch, hj, ctx2, n, "local_shuffle", 1024, k2, 1, S_dev)
        ctx2.close()
        if a.log2n > 27:      # BASELINE configs[1] at its own size: |R| = |S| = 2^27, uniform (the reference's experiment size)
            ctx3 = hj.HashJoinContext(local_rank, stream=stream)
            extra["oa_uniform_2p27"], _ = oa_leg(torch, hj, ctx3, 1 << 27, a.dist, a.shuffle_range, 2 * a.steps, 2,
                                                 variant=a.build_variant)
            ctx3.close()
    ctx.close()
    del S_dev
    torch.cuda.empty_cache()

    cpu = cpu_prj = None
    if not a.no_cpu_baseline:
        cpu = cpu_baseline(hj, a.log2n, a.dist, a.shuffle_range, R=host_R[0] if host_R else None)
        del host_R
        cpu_prj = cpu_baseline_prj(min(a.cpu_sample_log2n, a.log2n))

    line = {
        "metric": "Mtuples/sec build+probe, |R|=|S|=1B uint32, uniform vs local_shuffle",
        "value": main_leg["mtuples_per_s"], "unit": "Mtuples/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 tuples (u32 key), integer", "data": "synthetic (DataGen restatement, srand(0) glibc stream)",
        "config": {"workload": f"open-addressing build+probe (atomic), |R|=|S|=2^{a.log2n}, dataDistr={a.dist} "
                               f"W={a.shuffle_range}, S=sorted, probeLength=4, tableSize=2|R|; "
                               "step = build (locality pre-round; every reachable slot written once, empties included: "
                               "no separate table clear) + probe, inputs resident in HBM",
                   "algo": "atomic", "rSize": n, "sSize": n, "dataDistr": a.dist, "shuffleRange": a.shuffle_range},
        "result": {k: main_leg[k] for k in ("conflicts", "totalMatches", "inputSum", "buildVariant",
                                            "buildDeferred", "compactFallback", "checks")},
        "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_prj": cpu_prj, "other_workloads": extra,
    }
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
