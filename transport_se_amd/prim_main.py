"""prim_main -- namelist-compatible front end.

    python -m transport_se_amd.prim_main < dcmip1-1.nl

Reads the reference's Fortran namelists from stdin the way prim_main/readnl do (src/prim_main.F90:47,
src/share/namelist_mod.F90:159-351), runs the device-resident prim_run loop and leaves a `HommeTime_stats` file with the
four timer names the reference's perf scripts grep (test/run_ne120_perf.sh:140-144: prim_run, prim_advance_exp,
prim_advec_tracers_remap_rk2, vertical_remap).  Only the ~12 ctl_nl/vert_nl keys the tracer path reads are honoured;
output (analysis_nl), restart and threading keys are accepted and ignored.  At the end it prints the error-norm line the
reference's NCL script prints.
"""
import re
import sys
import time

import numpy as np

from . import diagnostics as dg

DEFAULTS = dict(ne=0, qsize=4, ndays=0, nmax=12, tstep=0.0, nu=0.0, nu_q=-1.0, rsplit=0, qsplit=1, limiter_option=0,
                hypervis_order=0, test_case="", statefreq=1, vfile_mid="", vfile_int="", disable_diagnostics=False)


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
    if s["ndays"] and s["tstep"]:
        s["nmax"] = int(s["ndays"] * 86400 / s["tstep"])       # namelist_mod.F90:347-351
    if s["nu_q"] < 0:
        s["nu_q"] = s["nu"]                                    # :700
    if s["limiter_option"] != 8:
        raise SystemExit("prim_main: only limiter_option = 8 is supported (the only limiter the reference wires)")
    if s["rsplit"] <= 0 or s["qsplit"] != 1:
        raise SystemExit("prim_main: needs rsplit > 0 (vertically lagrangian) and qsplit = 1")
    tc = str(s["test_case"]).lower()
    if not tc.startswith("dcmip1-"):
        raise SystemExit("prim_main: test_case must be dcmip1-1 or dcmip1-2")
    s["test"] = 1 if tc.startswith("dcmip1-1") else 2
    return s


def write_hommetime(path, timers, nranks=1):
    """GPTL-like summary (utils/timing/gptl.c prints name, processes, threads, count, walltotal, wallmax, wallmin)"""
    with open(path, "w") as f:
        f.write("name                                     processes  threads      count    walltotal      wallmax      wallmin\n")
        for name, (count, sec) in timers.items():
            f.write("%-40s %9d %8d %10d %12.6f %12.6f %12.6f\n" % (name, nranks, 1, count, sec, sec, sec))


def main(argv=None):
    from .driver import PrimRun
    s = settings(parse_namelists(sys.stdin.read()))
    run = PrimRun(s["ne"], s["qsize"], test_case=s["test"], nu_q=s["nu_q"], tstep=s["tstep"], rsplit=s["rsplit"])
    hip = run.hip
    from . import cube_mesh as cm
    geo = cm.geometry(s["ne"])
    q0 = run.fetch_qdp(1).copy()
    dp0 = dg.hybrid_dp(run.hv.hyai, run.hv.hybi, np.full((run.nelem, 4, 4), dg.P0))
    nsteps = (s["nmax"] // s["rsplit"]) * s["rsplit"]
    print(" nmax = %d tracer steps, tstep = %g, ne = %d, qsize = %d, nu_q = %g" % (nsteps, s["tstep"], s["ne"], s["qsize"], s["nu_q"]))
    hip.timing(True)
    t0 = time.perf_counter()
    np1 = 2
    for n in range(nsteps):
        np1 = run.step()
        if (not s["disable_diagnostics"]) and s["statefreq"] > 0 and run.nstep % s["statefreq"] == 0 and run.nstep % s["rsplit"] == 0:
            ps_v = hip.fetch("ps_v", (run.nelem, 4, 4))
            for t, (mn, mx, sm) in enumerate(dg.printstate_lines(run.fetch_qdp(np1), dg.hybrid_dp(run.hv.hyai, run.hv.hybi, ps_v))):
                print("nstep=%d qv(%d)= %23.15E %23.15E %23.15E" % (run.nstep, t + 1, mn, mx, sm))
    hip.synchronize()
    wall = time.perf_counter() - t0
    groups = {k: hip.kernel_time(k) for k in ("advance", "dss", "lap", "minmax", "level", "remap", "dcmip")}
    adv = sum(groups[k][0] for k in ("advance", "dss", "lap", "minmax", "level")) / 1e3
    write_hommetime("HommeTime_stats", {
        "prim_run": (nsteps // s["rsplit"], wall),
        "prim_advance_exp": (nsteps, groups["dcmip"][0] / 1e3),
        "prim_advec_tracers_remap_rk2": (nsteps, adv),
        "vertical_remap": (nsteps // s["rsplit"], groups["remap"][0] / 1e3)})
    q1 = run.fetch_qdp(np1)
    ps_v = hip.fetch("ps_v", (run.nelem, 4, 4))
    tr = 0 if s["test"] == 1 else 1
    nrm = dg.dcmip_norms(s["ne"], run.lat, run.lon, (q0 / dp0[:, None])[:, tr], (q1 / dg.hybrid_dp(run.hv.hyai, run.hv.hybi, ps_v)[:, None])[:, tr],
                         dg.level_heights(run.hv.hyam, run.hv.hybm))
    print("DCMIP 1-%d: L1=%8.6f L2=%8.6f Linf=%8.6f q_max=%8.6f q_min=%14.6e" % (s["test"], nrm["L1"], nrm["L2"], nrm["Linf"], nrm["q_max"], nrm["q_min"]))
    m0, m1 = dg.tracer_mass(geo["spheremp"], q0), dg.tracer_mass(geo["spheremp"], q1)
    for t in range(s["qsize"]):
        print("Q%d mass: %22.14E -> %22.14E (relative change %10.3e)" % (t + 1, m0[t], m1[t], (m1[t] - m0[t]) / max(abs(m0[t]), 1e-300)))
    print("prim_run wall %.3f s: %.4e tracer-DOF-steps/s" % (wall, run.nelem * 16 * 72 * s["qsize"] * nsteps / wall))
    run.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
