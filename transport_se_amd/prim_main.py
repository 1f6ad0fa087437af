"""prim_main -- namelist-compatible front end, one process per GPU.

    python -m transport_se_amd.prim_main < dcmip1-1.nl                 # one GPU
    python -m transport_se_amd.prim_main --gpus 8 < dcmip1-1.nl        # starts 8 ranks itself (one per GPU)
    bin/preqx < dcmip1-1.nl                                            # the name the reference's run scripts call ($EXE)

Reads the reference's Fortran namelists from stdin the way prim_main/readnl do (src/prim_main.F90:47,
src/share/namelist_mod.F90:159-351: rank 0 reads, everybody gets a copy), runs the device-resident prim_run loop and leaves
a `HommeTime_stats` file with the four timer names the reference's perf scripts grep (test/run_ne120_perf.sh:140-144:
prim_run, prim_advance_exp, prim_advec_tracers_remap_rk2, vertical_remap; wallmax = maximum over the ranks).  Only the ~12
ctl_nl/vert_nl keys the tracer path reads are honoured; output (analysis_nl), restart and threading keys are accepted and
ignored.  At the end it prints the error-norm line the reference's NCL script prints and the tracer-mass lines; both are
computed reproducibly (diagnostics.py), so the same run prints the same digits on 1, 2, 4 or 8 GPUs.
Under an MPI-style launcher (mpiexec/srun: PMI_RANK, OMPI_COMM_WORLD_RANK or SLURM_PROCID in the environment) the ranks
are taken from it; the control plane is torch.distributed/gloo on MASTER_ADDR:MASTER_PORT (default 127.0.0.1:29531 when every
rank runs on this node; several nodes must set MASTER_ADDR; one GPU per LOCAL rank is required).  No run on more than one
physical GPU has been recorded yet: the multi-rank evidence is staged ranks sharing a GPU and loopback rehearsals.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

from . import diagnostics as dg

DEFAULTS = dict(ne=0, qsize=4, ndays=0, nmax=12, tstep=0.0, nu=0.0, nu_q=-1.0, rsplit=0, qsplit=1, limiter_option=0,
                hypervis_order=0, test_case="", statefreq=1, vfile_mid="", vfile_int="", disable_diagnostics=False,
                # control_mod defaults (control_mod.F90:48,61-66,134-140) of the keys the tracer path depends on
                vert_remap_q_alg=0, hypervis_subcycle_q=1, hypervis_power=0.0, hypervis_scaling=0.0)


def parse_namelists(text):
    """minimal Fortran namelist reader: {group: {key: value}}; comments start with '!'"""
    groups = {}
    cur = None
    for raw in text.splitlines():
        line = raw.split("!")[0].strip()
        if not line:
            continue
        if line.startswith("&"):
            cur = groups.setdefault(line[1:].strip().lower(), {})
            continue
        if line.startswith("/"):
            cur = None
            continue
        if cur is None:
            continue
        for key, val in re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*=\s*([^=]+?)(?=\s+[A-Za-z_][A-Za-z0-9_]*\s*=|$)", line):
            cur[key.lower()] = _value(val.strip().rstrip(","))
    return groups


def _value(v):
    if v[:1] in "'\"":
        return v.strip("'\"")
    lv = v.lower()
    if lv in (".true.", "t", ".t."):
        return True
    if lv in (".false.", "f", ".f."):
        return False
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(lv.replace("d", "e"))
    except ValueError:
        return v.split(",")[0].strip().strip("'\"")


def settings(groups):
    s = dict(DEFAULTS)
    for g in ("ctl_nl", "vert_nl"):
        for k, v in groups.get(g, {}).items():
            if k in s:
                s[k] = v
    # the reference's namelist templates carry placeholders (ne = NE, tstep = TIME_STEP: test/dcmip1-1/dcmip1-1.nl:7,18) that its run
    # scripts substitute with sed before they feed the file to $EXE (test/run_ne8_tests.sh:109-115); a template fed as it is must end
    # with a message, not with a Python traceback
    for k, typ in (("ne", int), ("qsize", int), ("ndays", int), ("nmax", int), ("rsplit", int), ("qsplit", int), ("limiter_option", int),
                   ("statefreq", int), ("vert_remap_q_alg", int), ("hypervis_subcycle_q", int), ("tstep", float), ("nu", float), ("nu_q", float),
                   ("hypervis_power", float), ("hypervis_scaling", float)):
        v = s[k]
        if isinstance(v, bool) or not isinstance(v, (int, float)) or (typ is int and int(v) != v):
            raise SystemExit("prim_main: ctl_nl key %s = %r is not %s (namelist templates such as test/dcmip1-1/dcmip1-1.nl carry the placeholders "
                             "NE and TIME_STEP: substitute them first, as test/run_ne8_tests.sh:109-115 does)" % (k, v, "an integer" if typ is int else "a number"))
    if s["ne"] <= 0 or s["qsize"] <= 0 or s["tstep"] <= 0:
        raise SystemExit("prim_main: ne = %r, qsize = %r, tstep = %r (all three must be positive)" % (s["ne"], s["qsize"], s["tstep"]))
    if s["ndays"] and s["tstep"]:
        s["nmax"] = int(s["ndays"] * 86400 / s["tstep"])       # namelist_mod.F90:347-351
    if s["nu_q"] < 0:
        s["nu_q"] = s["nu"]                                    # :700
    if s["limiter_option"] != 8:
        raise SystemExit("prim_main: only limiter_option = 8 is supported (the only limiter the reference wires)")
    if s["rsplit"] <= 0 or s["qsplit"] != 1:
        raise SystemExit("prim_main: needs rsplit > 0 (vertically lagrangian) and qsplit = 1")
    # what the device path does not implement is refused, never silently replaced (the Fortran seam aborts on the same keys)
    if s["vert_remap_q_alg"] not in (0, 1, 2):
        raise SystemExit("prim_main: vert_remap_q_alg = %r (0|1: PPM with mirrored ghost cells, 2: piecewise-constant boundary cells)" % (s["vert_remap_q_alg"],))
    if s["hypervis_subcycle_q"] != 1:
        raise SystemExit("prim_main: hypervis_subcycle_q = %r is not supported (1 only)" % (s["hypervis_subcycle_q"],))
    if s["hypervis_power"] != 0 or s["hypervis_scaling"] != 0:
        raise SystemExit("prim_main: hypervis_power / hypervis_scaling must be 0 (constant-coefficient hyperviscosity only)")
    tc = str(s["test_case"]).lower()
    if not tc.startswith("dcmip1-"):
        raise SystemExit("prim_main: test_case must be dcmip1-1 or dcmip1-2")
    s["test"] = 1 if tc.startswith("dcmip1-1") else 2
    return s


def _fortran_e(x, w, d):
    """Fortran Ew.d: 0.dddE+ee (the mantissa is below 1, unlike C's %E)"""
    if x == 0 or not np.isfinite(x):
        return ("%*.*E" % (w, d, x)) if not np.isfinite(x) else "%*s" % (w, "0." + "0" * d + "E+00")
    e = int(np.floor(np.log10(abs(x)))) + 1
    m = x / 10.0 ** e
    if abs(round(m, d)) >= 1.0:
        m /= 10.0; e += 1
    return "%*s" % (w, ("%.*f" % (d, m)).replace("0.", "0.", 1) + "E%+03d" % e)


def write_hommetime(path, timers, nranks=1):
    """GPTL-like summary (utils/timing/gptl.c prints name, processes, threads, count, walltotal, wallmax, wallmin)"""
    with open(path, "w") as f:
        f.write("name                                     processes  threads      count    walltotal      wallmax      wallmin\n")
        for name, (count, tot, mx, mn) in timers.items():
            f.write("%-40s %9d %8d %10d %12.6f %12.6f %12.6f\n" % (name, nranks, 1, count * nranks, tot, mx, mn))


def _rank_world():
    env = os.environ
    for r, w in (("RANK", "WORLD_SIZE"), ("PMI_RANK", "PMI_SIZE"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"), ("SLURM_PROCID", "SLURM_NTASKS")):
        if r in env and w in env:
            return int(env[r]), int(env[w]), int(env.get("LOCAL_RANK", env.get("OMPI_COMM_WORLD_LOCAL_RANK", env.get("SLURM_LOCALID", env[r]))))
    return 0, 1, 0


def _local_world(world):
    """ranks on this node (LOCAL_WORLD_SIZE and its MPI / SLURM spellings); all of them when the launcher does not say"""
    env = os.environ
    for k in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE"):
        if k in env:
            try:
                return max(1, min(world, int(str(env[k]).split("(")[0])))
            except ValueError:
                pass
    return world


WATCHDOG_CHUNK = 360   # tracer steps a rank runs at most between two host synchronisations while the watchdog is armed


class Watchdog:
    """re-armed on progress: kick() after every milestone; a rank that misses one for `limit` seconds ends with exit code 4"""

    def __init__(self, limit, rank):
        self.limit, self.rank, self.timer, self.what = limit, rank, None, "set-up"
        self.kick("set-up")

    def _give_up(self):
        print("prim_main: rank %d made no progress for %g s (TSE_WATCHDOG_S) in: %s -- giving up" % (self.rank, self.limit, self.what),
              file=sys.stderr, flush=True)
        os._exit(4)

    def kick(self, what):
        """the previous milestone was reached; `what` is what the rank does next"""
        self.stop()
        self.what = what
        if self.limit > 0:
            import threading
            self.timer = threading.Timer(self.limit, self._give_up); self.timer.daemon = True; self.timer.start()

    def stop(self):
        if self.timer is not None:
            self.timer.cancel(); self.timer = None


def main(argv=None):
    ap = argparse.ArgumentParser(prog="prim_main")
    ap.add_argument("--gpus", type=int, default=0, help="start this many ranks (one per GPU) and feed them the namelist from stdin")
    ap.add_argument("--namelist", default=None, help="read the namelists from this file instead of stdin")
    a = ap.parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        text = open(a.namelist).read() if a.namelist else sys.stdin.read()
        with tempfile.NamedTemporaryFile("w", suffix=".nl", delete=False) as f:
            f.write(text)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
                              "--master-port", str(port), "-m", "transport_se_amd.prim_main", "--namelist", f.name])
        os.unlink(f.name)
        return rc
    rank, world, local = _rank_world()
    local_world = _local_world(world)
    dist = torch = None
    exchange = os.environ.get("TSE_EXCHANGE", "rccl")
    # a rank stuck in communicator set-up or in a halo exchange whose peer died must not block the job for ever (the reference's
    # abortmp = MPI_Abort ends every rank): a PROGRESS watchdog -- exit non-zero when TSE_WATCHDOG_S seconds (default 1800; 0 disables)
    # pass without this rank reaching its next milestone (set-up done, a chunk of at most WATCHDOG_CHUNK tracer steps done, the final
    # barrier passed); it stays armed until the process group is gone
    watchdog = Watchdog(float(os.environ.get("TSE_WATCHDOG_S", "1800")) if world > 1 else 0.0, rank)
    if world > 1:
        import torch
        import torch.distributed as dist
        if local_world == world:   # every rank on this node: the loopback defaults are safe (the container hostname may not resolve)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        elif "MASTER_ADDR" not in os.environ:
            raise SystemExit("prim_main: %d ranks on several nodes (%d on this one): set MASTER_ADDR / MASTER_PORT for the control plane" % (world, local_world))
        ndev = torch.cuda.device_count()
        if exchange != "staged" and ndev < local_world:   # one GPU per LOCAL rank (ranks on other nodes have their own)
            raise SystemExit("prim_main: %d ranks on this node but %d GPU(s) visible (TSE_EXCHANGE=staged rehearses several ranks on one GPU)"
                             % (local_world, ndev))
        local = local % max(ndev, 1)
        if exchange == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 reads the namelists (stdin or file), everybody gets the text (namelist_mod.F90:318-319 + the MPI_Bcasts after it)
    box = [None]
    if rank == 0:
        box[0] = open(a.namelist).read() if a.namelist else sys.stdin.read()
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    s = settings(parse_namelists(box[0]))

    from . import cube_mesh as cm
    from .driver import PrimRun
    from .hip_mod import TseError
    if os.environ.get("TSE_TEST_FAIL_REMAP_RANK") == str(rank):   # tests: this rank alone meets a negative layer thickness
        os.environ["TSE_TEST_FAIL_REMAP"] = "1"

    def abort_all(msg):
        """abortmp (parallel_mod.F90:274-287): message, then every rank must end.  The failing rank exits non-zero at once -- it
        must not enter another collective -- and the launcher (torch.distributed.run, mpiexec, srun) takes the other ranks down;
        where a launcher does not, their watchdog (TSE_WATCHDOG_S) does."""
        print(" %d ABORTING WITH ERROR: %s" % (rank, msg), flush=True)
        print("prim_main: rank %d: %s" % (rank, msg), file=sys.stderr, flush=True)
        os._exit(2)
    run = PrimRun(s["ne"], s["qsize"], test_case=s["test"], nu_q=s["nu_q"], tstep=s["tstep"], rsplit=s["rsplit"], rank=rank, world=world,
                  device=local, dist_mod=dist, torch_mod=torch, exchange=exchange, vert_remap_q_alg=s["vert_remap_q_alg"])
    hip, gid, nelem = run.hip, run.mine, run.nelem
    say = print if rank == 0 else (lambda *x, **k: None)

    def gsum(partials):
        return dg.global_sum(partials, gid, nelem, dist, rank, world)

    def gather(x):
        return dg.gather_by_gid(x, gid, nelem, dist, rank, world)

    tr = 0 if s["test"] == 1 else 1
    q0 = gather(run.fetch_qdp(1)[:, tr].copy())              # the norm script's tracer only (Q for 1-1, Q2 for 1-2)
    m0 = gsum(hip.element_mass(1))
    nsteps = (s["nmax"] // s["rsplit"]) * s["rsplit"]
    say(" nmax = %d tracer steps, tstep = %g, ne = %d, qsize = %d, nu_q = %g, %d rank(s), exchange = %s"
        % (nsteps, s["tstep"], s["ne"], s["qsize"], s["nu_q"], world, run.exchange_kind))
    if world > 1:
        v = hip.comm_version()
        say(" RCCL runtime %s (%s), library built with the headers of %s" % (v["runtime"], v["path"], v["built"]))
    hip.timing(True)
    hip.synchronize()
    if world > 1:
        dist.barrier()
    watchdog.kick("the first tracer steps")
    t0 = time.perf_counter()
    np1 = 2
    diag = (not s["disable_diagnostics"]) and s["statefreq"] > 0
    n = 0
    # prim_run_subcycle computes diagnostics in the remap cycle that ENDS at a multiple of statefreq (prim_driver_mod.F90:765-768):
    # prim_diag_scalars before the cycle's first step (n = 1) and after its remap (n = 2), then prim_printstate.  So the run stops once
    # at the start of such a cycle (for the "before" integrals) and once at its end.
    scale = 1.0 / dg.G                      # prim_state_mod.F90:176-177 (ps0 = 1e5 Pa: no mb factor)
    before = None                           # (nstep, Qmass(:,1), Qvar(:,1)) of the diagnostics cycle under way

    def qdiag(tl):
        """(Qmass, Qvar, min Q, max Q) per tracer of time level tl on rank 0: exact sums of the element shares / extrema over all ranks"""
        m, v, lo, hi = hip.element_qdiag(tl)
        m, v, lo, hi = gsum(m), gsum(v), gather(lo), gather(hi)
        return (m * scale / (4.0 * np.pi), v * scale / (4.0 * np.pi), lo.min(0), hi.max(0)) if rank == 0 else (None,) * 4
    while n < nsteps:
        # run up to the next diagnostics point (a multiple of rsplit and of statefreq) without host synchronisation
        stop = nsteps
        if diag:
            f = s["statefreq"] * s["rsplit"] // np.gcd(s["statefreq"], s["rsplit"])
            end = (run.nstep // f + 1) * f                       # end of the next diagnostics cycle
            if end - s["rsplit"] > run.nstep:
                stop = min(nsteps, end - s["rsplit"])            # ... stop at its start first
            else:
                stop = min(nsteps, end)
                if end <= nsteps:
                    before = (run.nstep,) + qdiag(np1 if run.nstep else 1)[:2]   # Qdp(n0) of the cycle's first step = what the last step wrote
        if watchdog.limit > 0:   # bounded chunks, so that "no progress" can be told from "a long run" (whole rsplit cycles)
            stop = min(stop, n + max(s["rsplit"], WATCHDOG_CHUNK // s["rsplit"] * s["rsplit"]))
        try:
            np1 = run.run(stop - n)
        except TseError as ex:   # e.g. "negative layer thickness" (prim_advection_mod.F90:1323): the reference aborts the whole job there
            abort_all("%s (tracer step %s)" % (ex, getattr(ex, "nstep", "?")))
        n = stop
        watchdog.kick("tracer steps %d.." % n)
        if diag and run.nstep % s["statefreq"] == 0 and run.nstep % s["rsplit"] == 0:
            m2, v2, qlo, qhi = qdiag(np1)   # (formed on the device: no copy of the tracer field to the host)
            if rank == 0:
                # prim_printstate (prim_state_mod.F90:341-347,376-385): format 100 = (A10,3(E23.15)); qvsum = global_integral(sum_k Q*dp)*scale
                print(" nstep= %d" % run.nstep)
                for t in range(s["qsize"]):
                    print("%-10s%s" % ("qv= ", "".join(_fortran_e(x, 23, 15) for x in (qlo[t], qhi[t], m2[t]))))
                if before is not None and before[0] == run.nstep - s["rsplit"]:
                    for t in range(s["qsize"]):   # '(a,i1,a,E22.14,a,2E15.7)': i1 overflows to '*' beyond tracer 9, as in the reference
                        print("Q%s,Q diss, dQ^2/dt:%s kg/m^2%s%s" % (str(t + 1) if t < 9 else "*", _fortran_e(m2[t], 22, 14),
                                                                    _fortran_e((m2[t] - before[1][t]) / s["tstep"], 15, 7),
                                                                    _fortran_e((v2[t] - before[2][t]) / s["tstep"], 15, 7)))
            before = None
    hip.synchronize()
    wall = time.perf_counter() - t0
    groups = {k: hip.kernel_time(k)[0] / 1e3 for k in ("advance", "dss", "lap", "minmax", "level", "remap", "dcmip")}
    mine = np.array([wall, groups["dcmip"], sum(groups[k] for k in ("advance", "dss", "lap", "minmax", "level")), groups["remap"]])
    if world > 1:
        ranks_t = [None] * world if rank == 0 else None
        dist.gather_object(mine, ranks_t, dst=0)
        stack = np.stack(ranks_t) if rank == 0 else None
    else:
        stack = mine[None]
    if rank == 0:
        names = ("prim_run", "prim_advance_exp", "prim_advec_tracers_remap_rk2", "vertical_remap")
        counts = (nsteps // s["rsplit"], nsteps, nsteps, nsteps // s["rsplit"])
        write_hommetime("HommeTime_stats", {nm: (cnt, float(stack[:, i].sum()), float(stack[:, i].max()), float(stack[:, i].min()))
                                            for i, (nm, cnt) in enumerate(zip(names, counts))}, world)
    # error norms on the unique-column grid (dcmip1-*_error_norm_ng.ncl:39-77) and tracer mass ("Q, Q diss")
    q1 = gather(run.fetch_qdp(np1)[:, tr].copy())
    ps_v = gather(hip.fetch("ps_v", (gid.size, 4, 4)))
    m1 = gsum(hip.element_mass(np1))
    if rank == 0:
        geo = cm.geometry(s["ne"])
        dp0 = dg.hybrid_dp(run.hv.hyai, run.hv.hybi, np.full((nelem, 4, 4), dg.P0))
        nrm = dg.dcmip_norms(s["ne"], geo["lat"], geo["lon"], q0 / dp0, q1 / dg.hybrid_dp(run.hv.hyai, run.hv.hybi, ps_v),
                             dg.level_heights(run.hv.hyam, run.hv.hybm))
        print("DCMIP 1-%d: L1=%8.6f L2=%8.6f Linf=%8.6f q_max=%8.6f q_min=%14.6e" % (s["test"], nrm["L1"], nrm["L2"], nrm["Linf"], nrm["q_max"], nrm["q_min"]))
        for t in range(s["qsize"]):
            print("Q%d mass: %22.14E -> %22.14E (relative change %10.3e)" % (t + 1, m0[t], m1[t], (m1[t] - m0[t]) / max(abs(m0[t]), 1e-300)))
        print("prim_run wall %.3f s: %.4e tracer-DOF-steps/s on %d rank(s)" % (float(stack[:, 0].max()), nelem * 16 * 72 * s["qsize"] * nsteps / float(stack[:, 0].max()), world))
    run.close()
    if world > 1:
        watchdog.kick("the final barrier")   # (still armed: a peer that died leaves this rank in the barrier)
        dist.barrier()
        dist.destroy_process_group()
    watchdog.stop()
    return 0


if __name__ == "__main__":
    sys.exit(main())
