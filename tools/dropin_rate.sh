#!/bin/bash
# drop-in prim_run rate: the reference's own hooks (per-stage) and the one-call whole-step hook through the Fortran seam, host elem(:) as
# the source of truth (copy_qdp_h2d/d2h around every rsplit cycle).  usage: tools/dropin_rate.sh <ne> <qsize> <nsteps>
R=${GRAFT_REPO_ROOT:-/root/repo}
ne=${1:-30}; q=${2:-35}; n=${3:-6}
out=$(mktemp -d)
for ws in 0 1; do
  echo "== TSE_HARNESS_WHOLE_STEP=$ws ne=$ne qsize=$q steps=$n"
  printf "%d %d %d 300.0 1e15 1 -1\n'%s'\n'%s'\n" $ne $q $n $out $R/transport_se_amd/data/vcoord | \
    (ulimit -s unlimited 2>/dev/null; TSE_HARNESS_WHOLE_STEP=$ws /opt/conda/bin/mpiexec -n 1 $R/tests/fortran_dropin/_build/hip_harness) 2>&1 | grep -i "tracer steps\|DOF-steps\|error\|abort" 
done
rm -rf $out
