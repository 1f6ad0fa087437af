"""prim_run driver: the reference's time loop around the tracer path, device-resident.

Mirrors prim_run_subcycle / prim_step (reference src/share/prim_driver_mod.F90:701-943) and TimeLevel_Qdp
(time_mod.F90:85-109): rsplit tracer steps, each preceded by the prescribed-field refresh of prim_advance_exp
(prim_advance_mod.F90:110-149), then vertical_remap.  One process per GPU; for world_size > 1 the elements are
cut into contiguous chunks of the reference's space-filling curve and the DSS halo is exchanged INSIDE the library
with RCCL send/recv over xGMI (tse_comm_init; the communicator id travels over torch.distributed, which is only the
control plane here).  `exchange="torch"` / `"staged"` select the callback form of the seam instead (torch.distributed
P2P ops on the device buffers, or host-staged slots for rehearsals where several ranks share one GPU).
"""
import numpy as np

from . import cube_mesh as cm
from .spacecurve import sfc_partition
from .hip_mod import HipMod
from .hybvcoord import HvCoord

NU_Q = {8: 6e16, 30: 1e15, 120: 1e13}     # test/run_ne8_tests.sh:27, run_ne30_tests.sh, run_ne120_perf.sh:29
TSTEP = {8: 400.0, 30: 300.0, 120: 75.0}  # test/run_ne8_tests.sh:25, run_ne120_perf.sh:28


def partition(ne, nranks):
    """owner rank of every element (index = global element number - 1): contiguous chunks of the reference's
    space-filling curve, the first mod(nelem,npart) ranks take one extra (spacecurve_mod.F90:1218-1273,
    cube_mod.F90:1574-1633).  Local element order on a rank = ascending global number (metagraph_mod.F90:317-323)."""
    return sfc_partition(ne, nranks)


class HaloExchange:
    """bndry_exchangeV (bndry_mod.F90:74-124) on device-resident packed slots: one isend + one irecv per
    neighbour rank, message = nlyr * lengthP doubles, slot s at column offset sum(lengthP[:s])."""

    def __init__(self, sched, device, dist_mod, torch_mod, stage_through_host=False):
        """device: "cuda:N" (buffers are HIP device pointers) or "cpu" (host pointers; gloo tests).
        stage_through_host: copy the device slots to the host and exchange them with a CPU backend (gloo) -- lets
        several ranks share one GPU for rehearsals; the production path sends the device buffers with RCCL."""
        self.sched, self.dist, self.torch, self.device = sched, dist_mod, torch_mod, device
        self.stage = stage_through_host
        self.len_s = {0: [int(s[2]) for s in sched["send"]]}; self.len_r = {0: [int(s[2]) for s in sched["recv"]]}
        self._plans = {}

    def set_minmax_layout(self, send_len, recv_len):
        """per-slot entry counts of the compact neighbour min/max exchange (kind 1), from tse_halo_minmax_layout"""
        self.len_s[1] = [int(x) for x in send_len]; self.len_r[1] = [int(x) for x in recv_len]

    def _wrap(self, ptr, count):
        if str(self.device) == "cpu":
            import ctypes
            arr = np.ctypeslib.as_array(ctypes.cast(int(ptr), ctypes.POINTER(ctypes.c_double)), shape=(int(count),))
            return self.torch.from_numpy(arr)
        iface = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        holder = type("DevArr", (), {"__cuda_array_interface__": iface})()
        return self.torch.as_tensor(holder, device=self.device)

    def _plan(self, sbuf, rbuf, nlyr, kind):
        """tensor views of the slots and the P2POp list of one (buffers, nlyr, kind) exchange, built once: the library's
        halo buffers never move, and six exchanges per tracer step make the per-call Python work visible at 8 GPUs"""
        key = (int(sbuf), int(rbuf), int(nlyr), int(kind))
        plan = self._plans.get(key)
        if plan is None:
            dist = self.dist
            ls, lr = self.len_s[kind], self.len_r[kind]
            off_s = np.concatenate([[0], np.cumsum(ls)]).astype(int); off_r = np.concatenate([[0], np.cumsum(lr)]).astype(int)
            ns, nr = int(off_s[-1]), int(off_r[-1])
            st = self._wrap(sbuf, ns * nlyr); rt_dev = self._wrap(rbuf, nr * nlyr)
            staged = self.stage and str(self.device) != "cpu"
            rt = self.torch.empty(nr * nlyr, dtype=self.torch.float64) if staged else rt_dev
            st_host = self.torch.empty(ns * nlyr, dtype=self.torch.float64) if staged else None
            src = st_host if staged else st
            ops = []
            for i, (peer, _, _) in enumerate(self.sched["recv"]):
                if lr[i]:
                    ops.append(dist.P2POp(dist.irecv, rt[off_r[i] * nlyr:(off_r[i] + lr[i]) * nlyr], int(peer)))
            for i, (peer, _, _) in enumerate(self.sched["send"]):
                if ls[i]:
                    ops.append(dist.P2POp(dist.isend, src[off_s[i] * nlyr:(off_s[i] + ls[i]) * nlyr], int(peer)))
            plan = self._plans[key] = (ops, st, st_host, rt, rt_dev if staged else None)
        return plan

    def __call__(self, sbuf, rbuf, nlyr, kind=0):
        ops, st, st_host, rt, rt_dev = self._plan(sbuf, rbuf, nlyr, kind)
        if st_host is not None:
            st_host.copy_(st)
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        if rt_dev is not None:
            rt_dev.copy_(rt)
        if str(self.device) != "cpu":
            self.torch.cuda.current_stream(self.device).synchronize()
        return 0


def check_schedules_match(sched, minmax_len, rank, dist_mod):
    """every rank's slot list must mirror its peers' (same peer set, same entry counts for both exchange kinds): a
    mismatch would leave the grouped send/recv of some rank waiting forever, so it is caught here, on the control plane."""
    mine = dict(send=[(int(p), int(l), int(m)) for (p, _, l), m in zip(sched["send"], minmax_len[0])],
                recv=[(int(p), int(l), int(m)) for (p, _, l), m in zip(sched["recv"], minmax_len[1])])
    world = dist_mod.get_world_size()
    everyone = [None] * world
    dist_mod.all_gather_object(everyone, mine)
    for peer, ln, mm in mine["send"]:
        theirs = [x for x in everyone[peer]["recv"] if x[0] == rank]
        if theirs != [(rank, ln, mm)]:
            raise RuntimeError("halo schedule mismatch: rank %d sends (%d cols, %d min/max entries) to rank %d, which expects %s"
                               % (rank, ln, mm, peer, theirs))
    for peer, ln, mm in mine["recv"]:
        theirs = [x for x in everyone[peer]["send"] if x[0] == rank]
        if theirs != [(rank, ln, mm)]:
            raise RuntimeError("halo schedule mismatch: rank %d expects (%d cols, %d min/max entries) from rank %d, which sends %s"
                               % (rank, ln, mm, peer, theirs))


class PrimRun:
    def __init__(self, ne, qsize, test_case=1, nu_q=None, tstep=None, rsplit=3, rank=0, world=1, device=0,
                 dist_mod=None, torch_mod=None, exchange="rccl", vert_remap_q_alg=0):
        """exchange (world > 1): "rccl" = in-library RCCL send/recv (production), "torch" = torch.distributed P2P ops in the
        exchange callback, "staged" = callback with host-staged slots over a CPU backend (ranks may share a GPU)."""
        self.ne, self.qsize, self.rsplit, self.test_case = ne, qsize, rsplit, test_case
        self.nu_q = NU_Q.get(ne, 1e15 * (30.0 / ne) ** 3.2) if nu_q is None else nu_q
        self.tstep = TSTEP.get(ne, 300.0 * 30.0 / ne) if tstep is None else tstep
        self.hv = HvCoord()
        topo = cm.topology(ne)
        geo = cm.geometry(ne, topo)
        self.nelem = 6 * ne * ne
        owner = partition(ne, world)
        desc = cm.edge_descriptors(topo, owner, rank)
        mine = desc["elems"]
        self.mine = mine
        self.elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine],
                         spheremp=geo["spheremp"][mine], rspheremp=geo["rspheremp"][mine],
                         putmapP=desc["putmapP"], getmapP=desc["getmapP"], reverse=desc["reverse"])
        self.lat, self.lon = geo["lat"][mine], geo["lon"][mine]
        callback = None
        if world > 1:
            if exchange not in ("rccl", "torch", "staged"):
                raise ValueError("exchange=%r" % (exchange,))
            # with "rccl" the callback stays dormant: it is the transport of last resort when the communicator cannot be
            # built on some rank (see below), and is never entered while the library holds a communicator
            callback = HaloExchange(desc, "cuda:%d" % device, dist_mod, torch_mod, stage_through_host=(exchange != "torch"))
        self._exchange = callback
        self.exchange_kind = exchange if world > 1 else "none"
        self.exchange_note = None
        self.hip_device = device
        self.rank, self.world = rank, world
        self.hip = HipMod(self.elem, cm.dvv(), (self.hv.hyai, self.hv.hybi, self.hv.ps0), qsize, self.nu_q,
                          rsplit=rsplit, device=device, schedule=dict(send=desc["send"], recv=desc["recv"]), exchange=callback,
                          vert_remap_q_alg=vert_remap_q_alg)
        if world > 1:
            check_schedules_match(desc, (self.hip.minmax_send_len, self.hip.minmax_recv_len), rank, dist_mod)
            if exchange == "rccl":
                # ncclCommInitRank is a blocking collective: a rank that cannot take part must say so BEFORE anyone enters it,
                # or its peers wait in the bootstrap for ever.  So: local pre-check on every rank, agreement over the control
                # plane, and only then the collective (whose own failures are symmetric: every rank gets an error back).
                err = None
                try:
                    self.hip.comm_precheck(rank, world)
                except RuntimeError as ex:
                    err = str(ex)
                errs = [None] * world
                dist_mod.all_gather_object(errs, err)
                bad = [(r, e) for r, e in enumerate(errs) if e]
                if not bad:
                    box = [HipMod.comm_unique_id() if rank == 0 else None]
                    dist_mod.broadcast_object_list(box, src=0)
                    try:
                        self.hip.comm_init(box[0], rank, world)
                    except RuntimeError as ex:
                        err = str(ex)
                    dist_mod.all_gather_object(errs, err)
                    bad = [(r, e) for r, e in enumerate(errs) if e]
                if bad:
                    # every rank leaves RCCL together and says so: host-staged slots over the control-plane backend
                    self.hip.comm_abort()
                    self.exchange_kind = "staged"
                    self.exchange_note = "RCCL communicator could not be initialised on rank(s) %s (%s): halo staged through the host" \
                                         % (",".join(str(r) for r, _ in bad), bad[0][1])
                    if rank == 0:
                        import sys
                        print("WARNING: " + self.exchange_note, file=sys.stderr, flush=True)
        self.hip.dcmip_init(test_case, self.lat, self.lon, self.hv.hyam, self.hv.hybm)
        self.hip.dcmip_set_initial()
        self.nstep = 0

    def step(self):
        """one prim_step (+ vertical_remap when it closes an rsplit cycle)"""
        hip = self.hip
        hip.dcmip_step_inputs(self.nstep, self.tstep)
        n0 = 1 if self.nstep % 2 == 0 else 2
        hip.advec_tracers_remap_rk2(self.tstep, n0, 3 - n0)
        self.nstep += 1
        if self.nstep % self.rsplit == 0:
            hip.vertical_remap(self.tstep * self.rsplit, 3 - n0)
        return 3 - n0

    def run(self, nsteps):
        """nsteps prim_steps; whole rsplit cycles go through the device-resident loop (no host synchronisation inside)"""
        np1 = 2
        lead = (self.rsplit - self.nstep % self.rsplit) % self.rsplit     # steps that complete the cycle under way
        if lead > nsteps:                                                 # ... which this call does not reach: step by step
            lead = 0
            while nsteps > 0:
                np1 = self.step(); nsteps -= 1
        ncyc = (1 if lead else 0) + (nsteps - lead) // self.rsplit       # (the device-resident loop completes a cycle under way first)
        if ncyc:
            before = self.nstep
            self.nstep = self.hip.prim_run_subcycle(self.tstep, ncyc, self.nstep)
            nsteps -= self.nstep - before
            np1 = 2 if (self.nstep - 1) % 2 == 0 else 1        # the last step wrote 3 - n0, n0 = 1 + mod(nstep, 2)
        for _ in range(nsteps):
            np1 = self.step()
        return np1

    def state_checksum(self, tl, torch_mod):
        """order-independent checksum of Qdp(:,:,:,:,tl) on this rank: the wrap-around int64 sum of the bit patterns
        (summing it over ranks gives a number that is identical for every partition iff the fields are bit-for-bit equal)"""
        ptr, nbytes = self.hip.device_ptr("qdp%d" % tl)
        n = self.mine.size * self.qsize * 72 * 16
        self.hip.synchronize()
        iface = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        holder = type("DevArr", (), {"__cuda_array_interface__": iface})()
        t = torch_mod.as_tensor(holder, device="cuda:%d" % self.hip_device)
        return int(t.sum().item())

    def fetch_qdp(self, tl):
        n, q = self.mine.size, self.qsize
        return self.hip.fetch("qdp%d" % tl, (n, q, 72, 4, 4))   # (one time level: the two are separate allocations)

    def close(self):
        self.hip.close()
