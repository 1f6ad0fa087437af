#!/usr/bin/env python3
"""bench.py -- tracer-DOF-steps/s of the prim_run tracer path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one prim_step of the reference's time loop (prescribed-field refresh + Prim_Advec_Tracers_remap_rk2: three
RK-SSP euler_steps with limiter 8, hyperviscosity and DSS + time average) plus vertical_remap on every rsplit-th step.
Workload at every N: ne120 / 72 levels / qsize=35 DCMIP 1-1 (BASELINE.json configs[2], configs[3]): the elements are
sharded over the N GPUs along the reference's space-filling curve (strong scaling), all state resident in HBM, fp64,
synthetic (analytic DCMIP) data; the DSS halo is exchanged inside the library with RCCL send/recv.
Without WORLD_SIZE in the environment and N > 1 this script starts the N rank processes itself (fresh children through
torch.distributed.run; this parent never touches a GPU) and exits with their status.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_DOF_STEP = 400.0 / 3.0   # 16 2/3 mandatory fp64 field passes per tracer step
# what the pipeline that is actually built must move, in tracer-field passes per step: stage 1: 2 (Qdp(n0) -> T),
# stage 2: 2 (T -> B), stage 3a: 1 1/4 (k_lap1 reads B and stores only the quarter of its Laplacian that other patches and ranks read),
# stage 3b: 2 (B -> C; k_advance<2,3> forms the Laplacian of its own slots itself and reads only the halo ring of T), final DSS + time
# average: 3 (C, Qdp(n0) -> Qdp(np1)); on every rsplit-th step the remap does that last part on read (3 passes, no k_dss_patch<1>).
# 10 1/4 passes = 82 B per DOF-step (round 3: 12 2/3 = 101.3; with TSE_REMAP_FUSED=0: + 2/3).
REMAP_FUSED = os.environ.get("TSE_REMAP_FUSED", "1") != "0" and os.environ.get("TSE_DSS_ON_READ", "1") != "0"
PIPELINE_BYTES_PER_DOF_STEP = (10.25 if REMAP_FUSED else 10.25 + 2.0 / 3.0) * 8.0
BASELINE_PUBLISHED = 3.92e9            # ne120/72L/q35, 960 Edison cores (README:174), other hardware
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per DOF of ONE launch of each kernel (DESIGN.md "kernels"): tracer fields read + written, 8 B each.
# Default path (DSS on read): advance0 = k_advance<0> (Qdp -> T), advance1 = k_advance<1,1> (T -> B), lap = k_lap1<1>
# (B -> Laplacian in T), advance2 = k_advance<2,3> (B, T -> C), dss = k_dss_patch<1> (C, Qdp(n0) -> Qdp(np1)): 12 field passes per
# tracer step + 2/3 for the remap = 101 B per DOF-step.
KERNEL_BYTES_PER_DOF = {"advance0": 16.0, "advance1": 16.0, "lap": 10.0, "advance2": 16.0, "dss": 24.0, "remap": 24.0 if REMAP_FUSED else 16.0}
# what round 3's accounting charged the same kernels (k_lap1 stored its whole Laplacian, k_advance<2,3> read it back as a third field):
# kept so that roofline.frac can be compared across rounds (roofline.frac_round3_accounting)
KERNEL_BYTES_PER_DOF_R3 = {"lap": 16.0, "advance2": 24.0}
KERNEL_NAMES = {"advance0": "k_advance<0,0>", "advance1": "k_advance<1,1>", "advance2": "k_advance<2,3>", "lap": "k_lap1<1>",
                "dss": "k_dss_patch<1>", "remap": "k_remap<1,0,1>" if REMAP_FUSED else "k_remap<1,0,0>"}
if os.environ.get("TSE_DSS_ON_READ", "1") == "0":   # one DSS pass per stage: 4 dss launches (3 x 16 + 24), lap = k_lap1<0>
    KERNEL_BYTES_PER_DOF.update({"dss": (16.0 + 16.0 + 16.0 + 24.0) / 4.0})
    KERNEL_NAMES.update({"advance1": "k_advance<1,0>", "advance2": "k_advance<2,0>", "lap": "k_lap1<0>", "dss": "k_dss_patch<0|1>"})


def kernel_source_hash():
    """identifies the kernel sources a PMC traffic profile was taken with (tools/pmc_traffic.py stores the same hash)"""
    from transport_se_amd import _lib
    return _lib.source_hash()


def _kernel_entry(kernels, name):
    """entry of a kernel-keyed profile dict by short name ("k_advance<2,3>"), whatever block-shape parameter follows"""
    if name in kernels:
        return kernels[name]
    stem = name[:-1]   # "k_advance<2,3" matches "k_advance<2,3,32>"
    for k, v in kernels.items():
        if k.startswith(stem + ",") or k.startswith(stem + ">"):
            return v
    raise KeyError(name)


def measured_traffic(ne, qsize, n_gpus, group, launches_per_step):
    """(HBM bytes per launch of the dominant kernel, HBM bytes per tracer step of all kernels, source file) from the newest
    committed PMC passes under profiles/ that were collected with THIS build's kernel sources and workload (collected as
    MI355X_MICROARCH.md prescribes: separate --pmc runs, FETCH_SIZE doubled on gfx950); (None, None, reason) otherwise --
    counters cannot be read in the timed run itself."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            t = json.load(open(path))
            c = t["config"]
            if (c["ne"], c["qsize"], c["n_gpus"]) != (ne, qsize, n_gpus) or os.environ.get("TSE_DSS_ON_READ", "1") == "0":
                continue
            if t.get("kernel_source_hash") != kernel_source_hash():
                continue
            per_step = 0.0
            for g, n in launches_per_step.items():
                if g in KERNEL_NAMES and n:
                    per_step += _kernel_entry(t["kernels"], KERNEL_NAMES[g])["hbm_bytes_per_launch"] * n
            for k, v in t["kernels"].items():   # the small kernels: one launch per step each
                if k.startswith("k_nbr_minmax") or k.startswith("k_dcmip_step"):
                    per_step += v["hbm_bytes_per_launch"]
            return _kernel_entry(t["kernels"], KERNEL_NAMES[group])["hbm_bytes_per_launch"], per_step, os.path.relpath(path, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None, "no PMC profile of this build and workload under profiles/"


def measured_valu(ne, qsize, n_gpus, group):
    """vector-issue figures of the dominant kernel from the newest committed SQ-counter record of THIS build's kernel sources
    (tools/profile_round.sh: separate rocprofv3 --pmc passes): VALU instructions per wave, the share of the wave cycles with a VALU
    instruction in flight, waves per SIMD the kernel holds, and their product = the share of a SIMD's issue slots in use"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters*.json")), reverse=True):
        try:
            t = json.load(open(path))
            c = t["config"]
            if (c["ne"], c["qsize"], c["n_gpus"]) != (ne, qsize, n_gpus) or t.get("kernel_source_hash") != kernel_source_hash():
                continue
            e = _kernel_entry(t["kernels"], KERNEL_NAMES[group])
            wps = {"advance2": 2, "remap": 2}.get(group, 3)   # waves per SIMD the kernel's registers / LDS allow (tools/kres.sh)
            return {"insts_per_wave": round(e["valu_insts_per_wave"]), "active_frac_of_wave_cycles": e["active_inst_valu_frac_of_wave_cycles"],
                    "waves_per_simd": wps, "issue_slot_utilisation": e["active_inst_valu_frac_of_wave_cycles"] * wps,
                    "source": os.path.relpath(path, ROOT), "measured_in_this_run": False}
        except Exception:  # noqa: BLE001
            continue
    return None


def l2_record(ne):
    """the second half of the metric ("+ DCMIP1-1 L2 vs ref"): the 12-day DCMIP 1-1 error norm at this resolution from the newest
    committed record under profiles/ that was produced with THIS build's kernel sources (tests/test_gpu_dcmip_norms.py writes
    it in every -m gpu session; 4-6 GPU-minutes at ne120, so it cannot run inside the timed bench)."""
    import glob
    ref = {120: 0.782613, 30: 0.789052, 8: 0.865526}.get(ne)   # /root/reference/README:152,128,95
    last = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*l2_dcmip11_ne%d*.json" % ne)), reverse=True):
        try:
            r = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        last = last or (r.get("kernel_source_hash"), os.path.relpath(path, ROOT))
        if r.get("kernel_source_hash") == kernel_source_hash():
            return {"value": r["L2"], "ref": ref, "rel_err": abs(r["L2"] - ref) / ref if ref else None, "L1": r.get("L1"), "Linf": r.get("Linf"),
                    "config": r.get("config"), "source": os.path.relpath(path, ROOT)}
    return {"value": None, "ref": ref, "rel_err": None, "config": None, "source": None,
            "reason": "no record of this build's kernel sources under profiles/" + (" (newest: %s, hash %s)" % (last[1], last[0]) if last else "")}


def host_cores():
    """(usable cores for this process, logical CPUs of the box, sockets)"""
    total = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total
    try:   # cgroup v2 quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    sockets = 1
    try:
        ids = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                ids.add(line.split(":")[1].strip())
        sockets = max(1, len(ids))
    except Exception:  # noqa: BLE001
        pass
    return usable, total, sockets


def cpu_baseline(qsize):
    """the reference itself (oracle/_ref/ref_harness: unmodified reference modules, one MPI rank per usable host core) on a
    bounded sample of the same workload -- the ne30 mesh with the bench's qsize (1.7 GB per tracer field at qsize 35, far
    beyond the caches, so the rate is the bandwidth-bound one), 6 tracer steps + 2 remaps, timing the same prim_run
    region; falls back to the C restatement (oracle/, OpenMP) if the binary is absent."""
    usable, total, sockets = host_cores()
    cores = int(os.environ.get("TSE_CPU_RANKS", usable))
    where = "%d ranks on %d usable of %d logical CPUs, %d socket(s)" % (cores, usable, total, sockets)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    vdir = os.path.join(ROOT, "transport_se_amd", "data", "vcoord")
    ne, nsteps = int(os.environ.get("TSE_CPU_NE", "30")), 6
    if os.path.exists(harness) and os.path.exists("/opt/conda/bin/mpiexec"):
        try:
            import tempfile
            out = tempfile.mkdtemp(prefix="tse_cpu_")
            stdin = "%d %d %d %r %r 1 -1\n'%s'\n'%s'\n" % (ne, qsize, nsteps, 300.0, 1e15, out, vdir)
            env = dict(os.environ); env.pop("HIP_VISIBLE_DEVICES", None)
            # the reference keeps Qtens_biharmonic(np,np,nlev,qsize,nets:nete) on the stack (prim_advection_mod.F90:708)
            cmd = "ulimit -s unlimited 2>/dev/null || ulimit -s $(ulimit -H -s); exec /opt/conda/bin/mpiexec -n %d %s" % (cores, harness)
            res = subprocess.run(["bash", "-c", cmd], input=stdin.encode(),
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
            for line in res.stdout.decode().splitlines():
                if "tracer-DOF-steps/s" in line:
                    v = float(line.split("=")[1])
                    return dict(value=v, unit="tracer-DOF-steps/s", cores=cores, kind="reference",
                                sample="ne%d/72L/qsize=%d DCMIP1-1, %d tracer steps + 2 remaps (prim_run region), reference Fortran/MPI: %s"
                                       % (ne, qsize, nsteps, where),
                                why_sample="the reference at ne120/qsize=35 keeps elem(:) + its stack scratch + edge buffers (Qdp alone 55.7 GB, "
                                           "Qtens_biharmonic 27.9 GB per rank set, edge buffers sized 3*qsize*nlev per column): about 0.45 TB of host RAM "
                                           "and minutes per step on the %d cores a 1-GPU box grants; ne30/qsize=35 (1.7 GB per field, far beyond the caches) "
                                           "runs the same bandwidth-bound code in about 20 s" % cores)
            print("cpu_baseline: reference harness printed no rate:\n" + res.stdout.decode()[-800:], file=sys.stderr)
        except Exception as ex:  # noqa: BLE001
            print("cpu_baseline: reference harness failed (%s); using the C port" % ex, file=sys.stderr)
    # the same workload key whatever the kind: the SAME sample (ne30 mesh, the bench's qsize, 6 tracer steps + 2 remaps) through the C
    # restatement when the reference binary is not on this box
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    o = po.Oracle(ne, qsize, nu_q=1e15, threads=cores)
    o.dcmip_init(1)
    t0 = time.time(); done, _ = o.prim_run(1, 300.0, nsteps // 3); dt = time.time() - t0
    v = o.nelem * 16 * 72 * qsize * done / dt
    o.close()
    return dict(value=v, unit="tracer-DOF-steps/s", cores=cores, kind="port",
                sample="ne%d/72L/qsize=%d DCMIP1-1, %d tracer steps + %d remaps (prim_run region), C restatement (oracle/) with %d OpenMP threads: %s"
                       % (ne, qsize, done, done // 3, cores, where),
                why_sample="oracle/_ref/ref_harness (the reference itself) is not on this box; same ne30 sample as for kind=reference")


def self_launch(a):
    """start the N rank processes (fresh children; nothing in this process has touched a GPU) and relay their output"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ne", type=int, default=int(os.environ.get("TSE_BENCH_NE", "120")))
    ap.add_argument("--qsize", type=int, default=int(os.environ.get("TSE_BENCH_QSIZE", "35")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start N>1 ranks with `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)" % (a.gpus, world))
    # a rank stuck in communicator set-up or in a halo exchange whose peer died would otherwise block the job for ever
    limit = float(os.environ.get("TSE_BENCH_WATCHDOG_S", "1800"))
    watchdog = None
    if limit > 0:
        import threading

        def give_up():
            print("bench.py: rank %d has no result after %g s (TSE_BENCH_WATCHDOG_S): giving up" % (rank, limit), file=sys.stderr, flush=True)
            os._exit(4)
        watchdog = threading.Timer(limit, give_up); watchdog.daemon = True; watchdog.start()
    import torch
    import torch.distributed as dist
    # TSE_EXCHANGE: "rccl" (default: in-library RCCL send/recv), "torch" (torch.distributed P2P in the exchange callback),
    # "staged" (rehearsal: host-staged slots over gloo; the ranks may then share one GPU)
    exchange = os.environ.get("TSE_EXCHANGE", "rccl")
    ndev = torch.cuda.device_count()
    if world > 1 and exchange != "staged" and ndev < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (TSE_EXCHANGE=staged rehearses several ranks on one GPU)" % (world, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node; the container hostname may not resolve
        if exchange == "torch":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")   # control plane only: communicator id, barrier, max-over-ranks of the time
    from transport_se_amd.driver import PrimRun
    run = PrimRun(a.ne, a.qsize, test_case=1, rank=rank, world=world, device=local, dist_mod=dist, torch_mod=torch, exchange=exchange)
    comm_rank, comm_world = run.hip.comm_info()
    if world > 1:
        # one line per rank on stderr BEFORE the first halo exchange: who talks to whom with what -- so that a run that hangs or dies in
        # its first exchange (the one code path no 1-GPU box can execute) can be read off the driver's tail
        nb_, ni_ = run.hip.boundary_layout(); pb_, pi_ = run.hip.patch_layout()
        nlyr = (a.qsize + 1) * 72
        slots = ["%d:%dcol/%.1fMB" % (int(p), int(l), int(l) * nlyr * 8 / 1e6) for (p, _, l) in run.hip.schedule["send"]]
        v = run.hip.comm_version()
        print("bench.py: rank %d/%d device %d: %d elements (%d on the rank boundary), %d+%d patches (first launch + interior), exchange=%s comm=(%d,%d) "
              "send slots peer:columns/main message [%s], bounds message %d entries; RCCL runtime %s built %s (%s)"
              % (rank, world, local, int(run.mine.size), nb_, pb_, pi_, run.exchange_kind, comm_rank, comm_world, " ".join(slots),
                 int(sum(run.hip.minmax_send_len)), v["runtime"], v["built"], v["path"]), file=sys.stderr, flush=True)

    def barrier():
        run.hip.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        run.hip.synchronize(); torch.cuda.synchronize()

    # (the timed K steps contain floor((W + K) / rsplit) - floor(W / rsplit) remaps: K/rsplit when W is a multiple of rsplit, one
    # more otherwise -- the count is in the JSON line as remaps_in_timed_region)
    run.run(a.warmup)
    barrier()
    # per-kernel HIP-event timing inside the timed region (2 events per kernel group and launch, resolved after the region): its cost is
    # measured, not assumed -- TSE_BENCH_KERNEL_TIMING=0 runs the same region without it (profiles/r04_timing_on_off.txt: below 0.1 %)
    kernel_timing = os.environ.get("TSE_BENCH_KERNEL_TIMING", "1") != "0"
    run.hip.timing(kernel_timing)
    t0 = time.perf_counter()
    run.run(a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    np1_final = 2 if ((a.warmup + a.steps) % 2 == 1) else 1     # TimeLevel_Qdp: the last step wrote 3 - n0
    checksum = run.state_checksum(np1_final, torch)
    if world > 1:
        dev = "cuda:%d" % local if exchange == "torch" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cs = torch.tensor([checksum], dtype=torch.int64, device=dev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        checksum = int(cs.item())
    if watchdog:
        watchdog.cancel()
    ktimes = {k: run.hip.kernel_time(k) for k in ("advance0", "advance1", "advance2", "lap", "dss", "minmax", "remap", "level", "dcmip", "avg")}
    run.hip.timing(False)
    if rank == 0:
        nelem_total = 6 * a.ne * a.ne
        dof_steps = float(nelem_total) * 16 * 72 * a.qsize * a.steps
        value = dof_steps / elapsed
        dom = max(KERNEL_BYTES_PER_DOF, key=lambda k: ktimes[k][0])   # the kernel the step spends most time in
        ms, n = ktimes[dom]
        dof_local = float(run.mine.size) * 16 * 72 * a.qsize
        ach = (KERNEL_BYTES_PER_DOF[dom] * dof_local / 1e9) / (ms / max(n, 1) / 1e3) if ms > 0 else 0.0
        launches_per_step = {k: ktimes[k][1] / a.steps for k in KERNEL_NAMES}
        traffic, traffic_step, traffic_src = measured_traffic(a.ne, a.qsize, world, dom, launches_per_step)
        nb, ni = run.hip.boundary_layout()
        rccl = run.hip.comm_version()
        pipe_bytes_step = PIPELINE_BYTES_PER_DOF_STEP * dof_local   # per GPU and tracer step
        out = {
            "metric": "advected tracer-DOF/sec (ne%d, 72L, qsize=%d) + DCMIP1-1 L2 vs ref" % (a.ne, a.qsize),
            "value": value, "unit": "tracer-DOF-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong",
            # the reference's own published rate for this metric and config (960 Edison cores, README:174)
            "vs_baseline": value / BASELINE_PUBLISHED if (a.ne, a.qsize) == (120, 35) else None,
            "remaps_in_timed_region": int(ktimes["remap"][1]),
            "dtype": "f64", "data": "synthetic (analytic DCMIP 1-1 fields, no files)",
            "config": {"workload": "ne%d DCMIP1-1 prim_run, NP=4, 72L, qsize=%d, rsplit=3, limiter8, nu_q=%g, tstep=%g; "
                                   "%d elements sharded over %d GPU(s) along the space-filling curve"
                                   % (a.ne, a.qsize, run.nu_q, run.tstep, nelem_total, world),
                       "ne": a.ne, "nlev": 72, "qsize": a.qsize, "elements_per_gpu": int(run.mine.size),
                       "rank_boundary_elements": nb, "interior_elements": ni},
            # what the communicator itself reports (ncclCommCount / ncclCommUserRank); 1 / "none" on a single GPU
            "world_size": comm_world if run.exchange_kind == "rccl" else world, "exchange": run.exchange_kind,
            **({"exchange_note": run.exchange_note} if run.exchange_note else {}),
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES.get(dom, dom), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # achieved / avg_ms: HIP events of THIS run; traffic*: PMC counters cannot be read inside a timed run -- they come
                         # from the committed profile of the same kernel sources (hash-checked), collected as the guide prescribes
                         "traffic_measured_in_this_run": False, "avg_ms": ms / max(n, 1), "launches": n, "kernel_timing_in_timed_region": kernel_timing,
                         "alg_bytes_per_launch": KERNEL_BYTES_PER_DOF[dom] * dof_local,
                         # the same kernel time priced with round 3's bytes for that kernel (k_advance<2,3> then read the first Laplacian as
                         # a third field; now it computes it: fewer bytes, more VALU work, a shorter step)
                         "frac_round3_accounting": (KERNEL_BYTES_PER_DOF_R3.get(dom, KERNEL_BYTES_PER_DOF[dom]) * dof_local / 1e9) / (ms / max(n, 1) / 1e3) / HBM_PEAK_GBS if ms > 0 else 0.0,
                         "valu": measured_valu(a.ne, a.qsize, world, dom),
                         "dominant_kernel_note": ("k_advance<2,3> forms the first Laplacian of its own slots itself since round 4: it is bound by fp64 VALU issue "
                                                  "(SQ counters in profiles/), not by HBM; whole_step_frac is the figure that tracks the step") if dom == "advance2" else None,
                         # whole step against SURVEY 8(d)'s 133.3 B per DOF-step (kept for continuity between rounds) ...
                         "whole_step_frac": value * ALG_BYTES_PER_DOF_STEP / (world * HBM_PEAK_GBS * 1e9),
                         # ... and against the bytes THIS pipeline must move (12 2/3 field passes = 101.3 B per DOF-step)
                         "pipeline_alg_bytes_per_dof_step": PIPELINE_BYTES_PER_DOF_STEP,
                         "whole_step_frac_pipeline": value * PIPELINE_BYTES_PER_DOF_STEP / (world * HBM_PEAK_GBS * 1e9),
                         # counter traffic of all kernels of one tracer step (same PMC profile; remap weighted by its launch rate)
                         "traffic_total_per_step": traffic_step,
                         "traffic_ratio_vs_pipeline": traffic_step / pipe_bytes_step if traffic_step else None},
            "l2_dcmip11": dict(l2_record(a.ne), measured_in_this_run=False),   # (a 4-minute run: tests/test_gpu_dcmip_norms.py writes the record)
            "rccl": rccl,
            # the chunks tried at init for T, Qdp1, Qdp2, B, C, their streaming-write rates and which try each field kept
            "placement": (lambda p: dict(p, write_GBs=[round(x) for x in p["write_GBs"]]))(run.hip.placement()),
            "kernel_ms_per_step": {k: v[0] / a.steps for k, v in ktimes.items()},
            "kernel_source_hash": kernel_source_hash(),
            # wrap-around int64 sum of the bit patterns of the final Qdp over all ranks: equal for every --gpus N
            # (same steps/warmup) iff the result is bit-for-bit independent of the partition
            "state_checksum": checksum & 0xFFFFFFFFFFFFFFFF,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.qsize)
        print(json.dumps(out), flush=True)
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
