"""transport_se_amd -- MI355X-native spectral-element tracer advection behind the reference's
prim_advec_tracers_remap()/euler_step()/vertical_remap() call surface.

The compute path is libtransport_se_hip.so (hand-written HIP for gfx950, C ABI in include/transport_se_hip.h);
this package is the Python mirror of the Fortran host seam (hip_mod <-> cuda_mod) plus the mesh/driver
pieces SURVEY.md 8(f) lists as "next".  There is no CPU fallback: importing works anywhere, running needs the GPU.
"""
from . import _lib  # noqa: F401
from .hip_mod import HipMod, NLEV, NLEVP, NP  # noqa: F401
