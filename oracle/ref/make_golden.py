#!/usr/bin/env python3
"""make_golden.py -- TEST INFRASTRUCTURE: generate tests/golden/*.npz from the reference itself.

Runs oracle/_ref/ref_harness (the unmodified reference hot path built by oracle/ref/Makefile, driven by
this repo's ref_harness.F90) in the build container and stores its outputs as small numpy fixtures.
Only data is written (inputs and expected outputs); no reference source travels.

    python oracle/ref/make_golden.py            # ne2 fixtures (seconds)
    python oracle/ref/make_golden.py --long     # + ne8 12-day DCMIP 1-1 / 1-day 1-2 norms (minutes, 6 ranks)
    python oracle/ref/make_golden.py --sfc      # only the space-filling-curve partition fixture (ref_sfc_partition.npz)
    python oracle/ref/make_golden.py --alg2     # only the vert_remap_q_alg=2 fixture (ref_ne2_alg2.npz)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
import norms  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
VCOORD = "/root/reference/test/vcoord"
GOLD = os.path.join(ROOT, "tests", "golden")
MPIEXEC = "/opt/conda/bin/mpiexec"


def run(ne, qsize, nsteps, tstep, nu_q, test, dumpfreq, nranks=1, outdir=None, env=None):
    """env: namelist overrides the harness takes from the environment (TSE_NL_VERT_REMAP_Q_ALG, ...)"""
    outdir = outdir or tempfile.mkdtemp(prefix="tse_ref_")
    os.makedirs(outdir, exist_ok=True)
    stdin = "%d %d %d %r %r %d %d\n'%s'\n'%s'\n" % (ne, qsize, nsteps, tstep, nu_q, test, dumpfreq, outdir, VCOORD)
    res = subprocess.run([MPIEXEC, "-n", str(nranks), HARNESS], input=stdin.encode(), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, check=True, env=dict(os.environ, **(env or {})))
    return outdir, res.stdout.decode()


def make_alg2():
    """vert_remap_q_alg = 2 (control_mod.F90:61-66; prim_advection_mod.F90:230-250,283-341): the reference's own single-call
    remap_Q_ppm outputs and a 6-step DCMIP 1-1 run (two remaps) with the piecewise-constant boundary cells."""
    cfg = dict(ne=2, qsize=5, nsteps=6, tstep=1800.0, nu_q=1e19, test=1)
    outdir, _ = run(dumpfreq=0, env={"TSE_NL_VERT_REMAP_Q_ALG": "2"}, **cfg)
    _, remaps = po.read_ops(os.path.join(outdir, "ops_000000_r0000.bin"))
    s6 = po.read_state(os.path.join(outdir, "state_000006_r0000.bin"))
    np.savez_compressed(
        os.path.join(GOLD, "ref_ne2_alg2.npz"), config=json.dumps(dict(cfg, vert_remap_q_alg=2)),
        remap_dp1=np.array([r["dp1"] for r in remaps]), remap_dp2=np.array([r["dp2"] for r in remaps]),
        remap_Qin=np.array([r["Qin"] for r in remaps]), remap_Qout=np.array([r["Qout"] for r in remaps]),
        qdp_step6=s6["qdp"], dp3d_step6=s6["dp3d"], ps_v_step6=s6["ps_v"])
    ref0 = np.load(os.path.join(GOLD, "ref_ne2_dcmip11.npz"))
    print("alg 2 vs alg 0 after 6 steps: max |diff| / max = %.3e" % (np.abs(s6["qdp"] - ref0["qdp_step6"]).max() / np.abs(s6["qdp"]).max()))


def make_ne2():
    cfg = dict(ne=2, qsize=5, nsteps=6, tstep=1800.0, nu_q=1e19, test=1)
    outdir, log = run(dumpfreq=1, **cfg)
    st = po.read_static(os.path.join(outdir, "static_000000_r0000.bin"))
    ops, remaps = po.read_ops(os.path.join(outdir, "ops_000000_r0000.bin"))
    np.savez_compressed(os.path.join(GOLD, "ref_ne2_static.npz"), **{k: v for k, v in st.items()})
    np.savez_compressed(
        os.path.join(GOLD, "ref_ops.npz"),
        ie=np.array([o["ie"] for o in ops]), s=np.array([o["s"] for o in ops]), v=np.array([o["v"] for o in ops]),
        div=np.array([o["div"] for o in ops]), grad=np.array([o["grad"] for o in ops]),
        div_wk=np.array([o["div_wk"] for o in ops]), lap_wk=np.array([o["lap_wk"] for o in ops]),
        remap_dp1=np.array([r["dp1"] for r in remaps]), remap_dp2=np.array([r["dp2"] for r in remaps]),
        remap_Qin=np.array([r["Qin"] for r in remaps]), remap_Qout=np.array([r["Qout"] for r in remaps]))
    s1 = po.read_state(os.path.join(outdir, "state_000001_r0000.bin"))
    s3 = po.read_state(os.path.join(outdir, "state_000003_r0000.bin"))
    s6 = po.read_state(os.path.join(outdir, "state_000006_r0000.bin"))
    sub = np.array([0, 3, 5, 10, 17, 23])  # elements whose step-1 derived fields are kept
    np.savez_compressed(
        os.path.join(GOLD, "ref_ne2_dcmip11.npz"), config=json.dumps(cfg),
        qdp_step1=s1["qdp"], qdp_step3=s3["qdp"], qdp_step6=s6["qdp"],
        sub=sub, vn0_step1=s1["vn0"][sub], dp_step1=s1["dp"][sub], divdp_step1=s1["divdp"][sub],
        divdp_proj_step1=s1["divdp_proj"][sub], eta_dot_dpdn_step1=s1["eta_dot_dpdn"][sub],
        dp3d_step3=s3["dp3d"], ps_v_step3=s3["ps_v"])
    # DCMIP 1-2
    cfg2 = dict(ne=2, qsize=3, nsteps=3, tstep=600.0, nu_q=1e19, test=2)
    outdir2, _ = run(dumpfreq=0, **cfg2)
    t3 = po.read_state(os.path.join(outdir2, "state_000003_r0000.bin"))
    np.savez_compressed(os.path.join(GOLD, "ref_ne2_dcmip12.npz"), config=json.dumps(cfg2), qdp_step3=t3["qdp"])
    # the reference on 2 ranks: its own bit-for-bit check + a multi-rank schedule (putmap/getmap with
    # neighbour-rank slots) for the host-side descriptor conversion tests
    outdir3, _ = run(dumpfreq=0, nranks=2, **cfg)
    sts = [po.read_static(os.path.join(outdir3, "static_000000_r%04d.bin" % r)) for r in range(2)]
    g6 = gather_state_with_gid(outdir3, "000006", 2, sts)
    bfb = bool(np.array_equal(g6["qdp"], s6["qdp"]))
    print("reference 1 rank vs 2 ranks bit-for-bit:", bfb)
    np.savez_compressed(
        os.path.join(GOLD, "ref_ne2_2rank_sched.npz"), bfb_1v2=bfb,
        **{"r%d_%s" % (r, k): sts[r][k] for r in range(2) for k in ("gid", "putmap", "getmap", "reverse", "send_cycles", "move_cycle")})


def gather_state_with_gid(outdir, tag, nranks, statics):
    parts = [po.read_state(os.path.join(outdir, "state_%s_r%04d.bin" % (tag, r))) for r in range(nranks)]
    gid = np.concatenate([s["gid"] for s in statics]) - 1
    out = {}
    for k, v in parts[0].items():
        if isinstance(v, np.ndarray):
            cat = np.concatenate([p[k] for p in parts])
            full = np.empty_like(cat)
            full[gid] = cat
            out[k] = full
    return out


def make_long(nranks=6):
    """ne8 DCMIP 1-1 (12 d) and 1-2 (1 d) as test/run_ne8_tests.sh runs them: norms + mass of the reference."""
    res = {}
    for name, test, nsteps, q in (("dcmip1-1", 1, 2592, 4), ("dcmip1-2", 2, 216, 4)):
        outdir, log = run(ne=8, qsize=q, nsteps=nsteps, tstep=400.0, nu_q=6e16, test=test, dumpfreq=0, nranks=nranks)
        sts = [po.read_static(os.path.join(outdir, "static_000000_r%04d.bin" % r)) for r in range(nranks)]
        s0 = gather_state_with_gid(outdir, "000000", nranks, sts)
        s1 = gather_state_with_gid(outdir, "%06d" % nsteps, nranks, sts)
        tr = 0 if test == 1 else 1
        gid = np.concatenate([s["gid"] for s in sts]) - 1
        lat = np.empty((gid.size, 4, 4)); lon = np.empty((gid.size, 4, 4))
        lat[gid] = np.concatenate([s["lat"] for s in sts]); lon[gid] = np.concatenate([s["lon"] for s in sts])
        spheremp = np.empty((gid.size, 4, 4)); spheremp[gid] = np.concatenate([s["spheremp"] for s in sts])
        res[name] = norms.dcmip_norms_from_qdp(8, lat, lon, s0["qdp"][:, tr], s1["qdp"][:, tr], s1["ps_v"], *po.read_vcoord())
        res[name]["mass0"] = [float(x) for x in norms.tracer_mass(spheremp, s0["qdp"])]
        res[name]["mass1"] = [float(x) for x in norms.tracer_mass(spheremp, s1["qdp"])]
        res[name]["config"] = dict(ne=8, qsize=q, nsteps=nsteps, tstep=400.0, nu_q=6e16, test=test, tracer=tr + 1)
        print(name, res[name])
    json.dump(res, open(os.path.join(GOLD, "ref_ne8_norms.json"), "w"), indent=1)


SFC_CASES = ((2, 2), (4, 3), (5, 3), (6, 4), (7, 5), (8, 8), (9, 4), (10, 7), (15, 8), (30, 8))


def make_sfc():
    """owner rank of every element as the reference's genspacepart assigns it (read back from each rank's GlobalId list),
    for face sizes that exercise the Hilbert (2), Peano (3), Cinco (5) levels, mixed factorisations, uneven chunk sizes and
    the non-factorable fallback (ne=7)."""
    out = {}
    for ne, nranks in SFC_CASES:
        outdir, _ = run(ne=ne, qsize=1, nsteps=0, tstep=300.0, nu_q=1e15, test=1, dumpfreq=0, nranks=nranks)
        owner = -np.ones(6 * ne * ne, dtype=np.int16)
        for r in range(nranks):
            st = po.read_static(os.path.join(outdir, "static_000000_r%04d.bin" % r))
            assert np.all(np.diff(st["gid"]) > 0)      # local order = ascending global number (metagraph_mod.F90:317-323)
            owner[st["gid"] - 1] = r
        assert owner.min() >= 0
        out["owner_ne%d_np%d" % (ne, nranks)] = owner
        print("sfc ne=%d nranks=%d: chunk sizes %s" % (ne, nranks, np.bincount(owner).tolist()))
        import shutil
        shutil.rmtree(outdir, ignore_errors=True)
    np.savez_compressed(os.path.join(GOLD, "ref_sfc_partition.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sfc", action="store_true")
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--only-long", action="store_true")
    ap.add_argument("--alg2", action="store_true", help="only the vert_remap_q_alg=2 fixture (ref_ne2_alg2.npz)")
    a = ap.parse_args()
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", HERE])
    if a.sfc:
        make_sfc()
        sys.exit(0)
    if a.alg2:
        make_alg2()
        sys.exit(0)
    if not a.only_long:
        make_ne2()
    if a.long or a.only_long:
        make_long()
