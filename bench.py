#!/usr/bin/env python3
"""bench.py -- tracer-DOF-steps/s of the prim_run tracer path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one prim_step of the reference's time loop (prescribed-field refresh + Prim_Advec_Tracers_remap_rk2: three
RK-SSP euler_steps with limiter 8, hyperviscosity and DSS + time average) plus vertical_remap on every rsplit-th step.
Workload at every N: ne120 / 72 levels / qsize=35 DCMIP 1-1 (BASELINE.json configs[2], configs[3]): the elements are
sharded over the N GPUs (strong scaling), all state resident in HBM, fp64, synthetic (analytic DCMIP) data.
Prints ONE JSON line on rank 0.
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
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per DOF of ONE launch of each kernel (DESIGN.md "kernels"): tracer fields read + written, 8 B each.
# Default path (DSS on read): advance0 = k_advance<0> (Qdp -> T), advance1 = k_advance<1,1> (T -> B), lap = k_lap1<1>
# (B -> Qdp(np1), Laplacian), advance2 = k_advance<2,2> (Qdp(np1), Laplacian -> B), dss = k_dss_t2<1> (B, Qdp(n0) -> Qdp(np1)).
KERNEL_BYTES_PER_DOF = {"advance0": 16.0, "advance1": 16.0, "lap": 24.0, "advance2": 24.0, "dss": 24.0, "remap": 16.0}
if os.environ.get("TSE_DSS_ON_READ", "1") == "0":   # one DSS pass per stage: 4 dss launches (3 x 16 + 24), lap = k_lap1<0>
    KERNEL_BYTES_PER_DOF.update({"lap": 16.0, "dss": (16.0 + 16.0 + 16.0 + 24.0) / 4.0})
KERNEL_NAMES = {"advance0": "k_advance<0,0>", "advance1": "k_advance<1,1>", "advance2": "k_advance<2,2>", "lap": "k_lap1<1>",
                "dss": "k_dss_t2<1>", "remap": "k_remap<1>"}


def measured_traffic(ne, qsize, n_gpus, group):
    """HBM bytes per launch of the dominant kernel group from the committed PMC passes (profiles/), collected exactly as
    MI355X_MICROARCH.md prescribes (separate --pmc runs, FETCH_SIZE doubled on gfx950); None when the workload differs"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc_traffic_ne120_q35.json")))
        c = t["config"]
        if (c["ne"], c["qsize"], c["n_gpus"]) == (ne, qsize, n_gpus) and os.environ.get("TSE_DSS_ON_READ", "1") != "0":
            return t["kernels"][KERNEL_NAMES[group]]["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    return None


def cpu_baseline(qsize, seconds_hint=20.0):
    """the reference itself (oracle/_ref/ref_harness: unmodified reference modules, MPI ranks on the host cores) on a
    bounded sample of the same workload; falls back to the C restatement (oracle/, OpenMP) if the binary is absent."""
    import multiprocessing
    cores = max(1, min(multiprocessing.cpu_count(), 16))
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    vdir = os.path.join(ROOT, "tests", "golden", "vcoord")
    ne, nsteps = 16, 6
    if os.path.exists(harness) and os.path.exists("/opt/conda/bin/mpiexec"):
        try:
            import tempfile
            out = tempfile.mkdtemp(prefix="tse_cpu_")
            stdin = "%d %d %d %r %r 1 -1\n'%s'\n'%s'\n" % (ne, qsize, nsteps, 300.0, 7e15, out, vdir)
            env = dict(os.environ); env.pop("HIP_VISIBLE_DEVICES", None)
            # the reference keeps Qtens_biharmonic(np,np,nlev,qsize,nets:nete) on the stack (prim_advection_mod.F90:708)
            cmd = "ulimit -s unlimited 2>/dev/null || ulimit -s $(ulimit -H -s); exec /opt/conda/bin/mpiexec -n %d %s" % (cores, harness)
            res = subprocess.run(["bash", "-c", cmd], input=stdin.encode(),
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
            for line in res.stdout.decode().splitlines():
                if "tracer-DOF-steps/s" in line:
                    v = float(line.split("=")[1])
                    return dict(value=v, unit="tracer-DOF-steps/s", cores=cores, kind="reference",
                                sample="ne%d/72L/qsize=%d DCMIP1-1, %d tracer steps + 2 remaps, reference Fortran/MPI on %d ranks"
                                       % (ne, qsize, nsteps, cores))
        except Exception as ex:  # noqa: BLE001
            print("cpu_baseline: reference harness failed (%s); using the C port" % ex, file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    ne = 8
    o = po.Oracle(ne, qsize, nu_q=6e16, threads=cores)
    o.dcmip_init(1)
    t0 = time.time(); done, _ = o.prim_run(1, 400.0, 1); dt = time.time() - t0
    v = o.nelem * 16 * 72 * qsize * done / dt
    o.close()
    return dict(value=v, unit="tracer-DOF-steps/s", cores=cores, kind="port",
                sample="ne%d/72L/qsize=%d DCMIP1-1, %d tracer steps + 1 remap, C restatement with %d OpenMP threads" % (ne, qsize, done, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ne", type=int, default=int(os.environ.get("TSE_BENCH_NE", "120")))
    ap.add_argument("--qsize", type=int, default=int(os.environ.get("TSE_BENCH_QSIZE", "35")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    backend = os.environ.get("TSE_DIST_BACKEND", "nccl")   # "gloo": rehearsal with host-staged slots (ranks may share a GPU)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from transport_se_amd.driver import PrimRun
    run = PrimRun(a.ne, a.qsize, test_case=1, rank=rank, world=world, device=local, dist_mod=dist, torch_mod=torch,
                  stage_through_host=(backend != "nccl"))

    def barrier():
        run.hip.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        run.hip.synchronize(); torch.cuda.synchronize()

    # warm-up leaves the time loop at a multiple of rsplit so the timed K steps contain K/rsplit remaps
    run.run(a.warmup)
    barrier()
    run.hip.timing(True)
    t0 = time.perf_counter()
    run.run(a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    np1_final = 2 if ((a.warmup + a.steps) % 2 == 1) else 1     # TimeLevel_Qdp: the last step wrote 3 - n0
    checksum = run.state_checksum(np1_final, torch)
    if world > 1:
        dev = "cuda:%d" % local if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cs = torch.tensor([checksum], dtype=torch.int64, device=dev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        checksum = int(cs.item())
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
        out = {
            "metric": "advected tracer-DOF/sec (ne120, 72L, qsize=35) + DCMIP1-1 L2 vs ref",
            "value": value, "unit": "tracer-DOF-steps/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (analytic DCMIP 1-1 fields, no files)",
            "config": {"workload": "ne%d DCMIP1-1 prim_run, NP=4, 72L, qsize=%d, rsplit=3, limiter8, nu_q=%g, tstep=%g; "
                                   "%d elements sharded over %d GPU(s)" % (a.ne, a.qsize, run.nu_q, run.tstep, nelem_total, a.gpus),
                       "ne": a.ne, "nlev": 72, "qsize": a.qsize, "elements_per_gpu": int(run.mine.size)},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES.get(dom, dom), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic(a.ne, a.qsize, a.gpus, dom), "avg_ms": ms / max(n, 1), "launches": n,
                         "alg_bytes_per_launch": KERNEL_BYTES_PER_DOF[dom] * dof_local,
                         "whole_step_frac": value * ALG_BYTES_PER_DOF_STEP / (a.gpus * HBM_PEAK_GBS * 1e9)},
            "kernel_ms_per_step": {k: v[0] / a.steps for k, v in ktimes.items()},
            # wrap-around int64 sum of the bit patterns of the final Qdp over all ranks: equal for every --gpus N
            # (same steps/warmup) iff the result is bit-for-bit independent of the partition
            "state_checksum": checksum & 0xFFFFFFFFFFFFFFFF,
        }
        if a.gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.qsize)
        print(json.dumps(out))
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
