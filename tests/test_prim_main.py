"""namelist front end and product-side diagnostics (8f-3)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import norms
from transport_se_amd import diagnostics as dg
from transport_se_amd import prim_main as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NL = """
&ctl_nl
  NThreads = 1
  partmethod = 4
  topology = "cube"
  test_case = "dcmip1-1"
  ne = 8
  qsize = 4
  nmax = 6             ! six tracer steps
  statefreq = 3
  tstep = 400
  qsplit = 1, rsplit = 3
  nu_q = 6e16
  limiter_option = 8
  hypervis_order = 2
/
&vert_nl
  vform = "ccm"
  vfile_mid = "vcoord/acme-72m.ascii"
  vfile_int = "vcoord/acme-72i.ascii"
/
&analysis_nl
  output_frequency = 6
/
"""


def test_namelist_parsing():
    s = pm.settings(pm.parse_namelists(NL))
    assert (s["ne"], s["qsize"], s["nmax"], s["rsplit"], s["test"]) == (8, 4, 6, 3, 1)
    assert s["tstep"] == 400 and s["nu_q"] == 6e16 and s["limiter_option"] == 8
    s2 = pm.settings(pm.parse_namelists(NL.replace("nmax = 6", "ndays = 12")))
    assert s2["nmax"] == 12 * 86400 // 400          # namelist_mod.F90:347-351
    with pytest.raises(SystemExit):
        pm.settings(pm.parse_namelists(NL.replace("limiter_option = 8", "limiter_option = 4")))


def test_product_diagnostics_agree_with_the_checker():
    rng = np.random.default_rng(5)
    from transport_se_amd import cube_mesh as cm
    g = cm.geometry(2)
    q0 = rng.uniform(size=(24, 72, 4, 4)); q1 = q0 + 0.1 * rng.uniform(size=q0.shape)
    zm = np.linspace(12000, 100, 72)
    assert dg.dcmip_norms(2, g["lat"], g["lon"], q0, q1, zm) == norms.dcmip_norms(2, g["lat"], g["lon"], q0, q1, zm)


@pytest.mark.gpu
def test_prim_main_runs_a_reference_style_namelist(tmp_path):
    res = subprocess.run([sys.executable, "-m", "transport_se_amd.prim_main"], input=NL.encode(), cwd=str(tmp_path),
                         env=dict(os.environ, PYTHONPATH=ROOT), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-2000:]
    assert "DCMIP 1-1: L1=" in out and "qv(1)=" in out
    stats = open(os.path.join(str(tmp_path), "HommeTime_stats")).read()
    for name in ("prim_run", "prim_advance_exp", "prim_advec_tracers", "vertical_remap"):   # run_ne120_perf.sh:140-144
        assert name in stats
    # mass conserved (the "Q,Q diss" check)
    for line in out.splitlines():
        if line.startswith("Q") and "relative change" in line:
            assert abs(float(line.split("relative change")[1].strip(" )"))) < 1e-12
