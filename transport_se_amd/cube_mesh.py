"""cube_mesh -- cubed-sphere mesh, metric terms and edge descriptors for the tracer engine.

Produces, for a uniform ne x ne x 6 equi-angular cubed sphere with NP=4 elements, the element_t fields the hot
path reads -- spherep lat/lon, D, Dinv, metdet, rmetdet, mp, spheremp, rspheremp -- with the same formulas and
the same operation order as the reference's prim_init1 chain:
    set_corner_coordinates (cube_mod.F90:1280-1312), coordinates_atomic/ref2sphere (:136-181, :2483-2511),
    metric_atomic + dmap_equiangular + vmap (:241-486, :582-743), mass_matrix (mass_matrix_mod.F90:27-137) and the
    global area correction alpha (prim_driver_mod.F90:265-283),
and the neighbour topology/orientation the reference derives in CubeTopology (cube_mod.F90:1432-2162) with the
reverse rule of cube_mod.F90:2371-2381, expressed as reference-style putmapP/getmapP/reverse descriptors
(schedule_mod.F90:905,929).  Vectorised numpy; the few quantities the reference evaluates in quad precision
(GLL points/weights, Dvv) are evaluated with 50-digit decimals and rounded once.
"""
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

NP = 4
DD_PI = 3.141592653589793238462643383279
DIST_THRESHOLD = 1.0e-9
W, E, S, N, SW, SE, NW, NE = range(8)  # control_mod.F90:173-181 (0-based)

getcontext().prec = 50


def gll_decimal():
    s = (Decimal(1) / Decimal(5)).sqrt()
    return [Decimal(-1), -s, s, Decimal(1)], [Decimal(1) / 6, Decimal(5) / 6, Decimal(5) / 6, Decimal(1) / 6]


def gll():
    x, w = gll_decimal()
    return np.array([float(v) for v in x]), np.array([float(v) for v in w])


def dvv():
    """Dvv[l][i] = Dvv(i,l) of derivative_mod.F90:451-486 (C order == Fortran memory)"""
    x, _ = gll_decimal()
    leg = [(5 * v ** 3 - 3 * v) / 2 for v in x]
    d = np.zeros((4, 4))
    for j in range(4):
        for i in range(4):
            if i != j:
                d[i, j] = float((Decimal(1) / (x[i] - x[j])) * leg[i] / leg[j])  # Fortran dvv(j,i) -> memory [i][j]
    d[3, 3] = 3.0
    d[0, 0] = -3.0
    return d


def _vertex_keys(ne, face, A, B):
    x = np.where(face == 1, ne, np.where(face == 2, -A, np.where(face == 3, -ne, np.where(face == 4, A, np.where(face == 5, B, -B)))))
    y = np.where(face == 1, A, np.where(face == 2, ne, np.where(face == 3, -A, np.where(face == 4, -ne, A))))
    z = np.where(face <= 4, B, np.where(face == 5, -ne, ne))
    m = 2 * ne + 1
    return ((x + ne).astype(np.int64) * m + (y + ne)) * m + (z + ne)


_EDGE_C0 = np.array([0, 1, 0, 3])  # W,E,S,N start corner (corners SW,SE,NE,NW = 0..3)
_EDGE_C1 = np.array([3, 2, 1, 2])
_CDIR_CORNER = np.array([0, 1, 3, 2])  # SW,SE,NW,NE -> corner index


def topology(ne):
    """neighbour element / neighbour direction / reversed flag per (element, direction); -1 where a corner
    neighbour does not exist (the 3 elements around each of the 8 cube vertices)."""
    nelem = 6 * ne * ne
    e = np.arange(nelem)
    face = e // (ne * ne) + 1
    je = (e % (ne * ne)) // ne
    ie = e % ne
    A0, B0 = -ne + 2 * ie, -ne + 2 * je
    vk = np.stack([_vertex_keys(ne, face, A0, B0), _vertex_keys(ne, face, A0 + 2, B0),
                   _vertex_keys(ne, face, A0 + 2, B0 + 2), _vertex_keys(ne, face, A0, B0 + 2)], 1)  # [e][corner]
    nbr_elem = -np.ones((nelem, 8), dtype=np.int64); nbr_dir = -np.ones((nelem, 8), dtype=np.int64)
    nbr_rev = np.zeros((nelem, 8), dtype=np.int64)
    # edges: match unordered vertex pairs
    k0 = vk[:, _EDGE_C0]; k1 = vk[:, _EDGE_C1]                       # [e][d]
    lo = np.minimum(k0, k1).reshape(-1); hi = np.maximum(k0, k1).reshape(-1)
    order = np.lexsort((hi, lo))
    lo_s, hi_s = lo[order], hi[order]
    assert np.all(lo_s[0::2] == lo_s[1::2]) and np.all(hi_s[0::2] == hi_s[1::2]), "every edge must be shared by exactly 2 elements"
    a, b = order[0::2], order[1::2]
    for x, y in ((a, b), (b, a)):
        ex, dx, ey, dy = x // 4, x % 4, y // 4, y % 4
        nbr_elem[ex, dx] = ey; nbr_dir[ex, dx] = dy
        nbr_rev[ex, dx] = (k0.reshape(-1)[x] != k0.reshape(-1)[y]).astype(np.int64)  # start vertices differ -> reversed
    # corners: the element around the vertex that is neither me nor one of my two edge neighbours there
    flat = vk.reshape(-1)
    order = np.argsort(flat, kind="stable")
    ks = flat[order]
    start = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    count = np.diff(np.r_[start, ks.size])
    grp_of = np.repeat(np.arange(start.size), count)
    grp_index = np.empty(flat.size, dtype=np.int64); grp_index[order] = grp_of
    # table group -> up to 4 (element, corner)
    members = -np.ones((start.size, 4), dtype=np.int64)
    pos = np.arange(ks.size) - np.repeat(start, count)
    members[grp_of, pos] = order
    corner_edges = {0: (W, S), 1: (E, S), 3: (W, N), 2: (E, N)}  # corner index -> the two edges meeting there
    for cd in range(4):
        c = _CDIR_CORNER[cd]
        g = grp_index[np.arange(nelem) * 4 + c]
        mem = members[g]                                           # [e][4] flat ids (e*4+corner) or -1
        me = mem // 4
        ea, eb = corner_edges[int(c)]
        excl = (me == np.arange(nelem)[:, None]) | (me == nbr_elem[:, ea][:, None]) | (me == nbr_elem[:, eb][:, None]) | (mem < 0)
        full = (count[g] == 4)
        cand = np.where(excl, -1, mem)
        pick = cand.max(1)
        ok = full & (pick >= 0)
        nbr_elem[ok, 4 + cd] = pick[ok] // 4
        ncorner = pick[ok] % 4
        inv = np.empty(4, dtype=np.int64); inv[_CDIR_CORNER] = np.arange(4)
        nbr_dir[ok, 4 + cd] = 4 + inv[ncorner]
    return dict(face=face, ie=ie, je=je, nbr_elem=nbr_elem, nbr_dir=nbr_dir, nbr_rev=nbr_rev)


def edge_point(d, k):
    return [k * 4, k * 4 + 3, k, 12 + k][d]


CORNER_POINT = {SW: 0, SE: 3, NW: 12, NE: 15}


def dss_sum(field, topo):
    """edgeVpack + edgeVunpack of a [nelem][16] field in the reference's order (all S, E, N, W, then SW, SE, NE, NW)"""
    out = field.copy()
    ne_, nd, rev = topo["nbr_elem"], topo["nbr_dir"], topo["nbr_rev"]
    ks = np.arange(4)
    for d in (S, E, N, W):
        for k in range(4):
            kk = np.where(rev[:, d] == 1, 3 - k, k)
            src_pt = np.choose(nd[:, d], [kk * 4, kk * 4 + 3, kk, 12 + kk])
            out[:, edge_point(d, k)] = out[:, edge_point(d, k)] + field[ne_[:, d], src_pt]
    for d in (SW, SE, NE, NW):
        ok = ne_[:, d] >= 0
        src_pt = np.array([0, 3, 12, 15])[nd[ok, d] - 4]
        out[ok, CORNER_POINT[d]] = out[ok, CORNER_POINT[d]] + field[ne_[ok, d], src_pt]
    return out


def _vmap(x1, x2, face):
    t1, t2, c1, c2 = np.tan(x1), np.tan(x2), np.cos(x1), np.cos(x2)
    r = np.sqrt(1.0 + t1 * t1 + t2 * t2)
    eq = face <= 4
    pd = np.sqrt(t1 * t1 + t2 * t2)
    pdz = np.where(pd <= DIST_THRESHOLD, 1.0, pd)
    sgn = np.where(face == 6, -1.0, 1.0)
    D11 = np.where(eq, 1.0 / (r * c1), sgn * t2 / (pdz * c1 * c1 * r))
    D12 = np.where(eq, 0.0, -sgn * t1 / (pdz * c2 * c2 * r))
    D21 = np.where(eq, -t1 * t2 / (c1 * r * r), sgn * t1 / (pdz * c1 * c1 * r * r))
    D22 = np.where(eq, 1.0 / (r * r * c1 * c2 * c2), sgn * t2 / (pdz * c2 * c2 * r * r))
    pole = (~eq) & (pd <= DIST_THRESHOLD)
    D11 = np.where(pole, 1.0, D11); D12 = np.where(pole, 0.0, D12); D21 = np.where(pole, 0.0, D21); D22 = np.where(pole, 1.0, D22)
    return D11, D12, D21, D22


def geometry(ne, topo=None):
    """dict of [nelem][4 j][4 i]... arrays: lat lon D Dinv metdet rmetdet mp spheremp rspheremp + alpha"""
    topo = topo or topology(ne)
    face = topo["face"][:, None, None]
    xq, wq = gll_decimal()
    pts = np.array([float(v) for v in xq])
    pd = np.array([float((1 - v) / 2) for v in xq]); qd = np.array([float((1 + v) / 2) for v in xq])
    xstart, xend = -0.25 * DD_PI, 0.25 * DD_PI
    dx = (xend - xstart) / ne
    sx = (xstart + topo["ie"] * dx)[:, None, None]; sy = (xstart + topo["je"] * dx)[:, None, None]
    cx = [sx, sx + dx, sx + dx, sx]; cy = [sy, sy, sy + dx, sy + dx]
    pi_, pj_ = pd[None, None, :], pd[None, :, None]; qi_, qj_ = qd[None, None, :], qd[None, :, None]
    x = pi_ * pj_ * cx[0] + qi_ * pj_ * cx[1] + qi_ * qj_ * cx[2] + pi_ * qj_ * cx[3]
    y = pi_ * pj_ * cy[0] + qi_ * pj_ * cy[1] + qi_ * qj_ * cy[2] + pi_ * qj_ * cy[3]
    X, Y = np.tan(x), np.tan(y)
    r = np.sqrt(1.0 + X * X + Y * Y)
    lat = np.where(face <= 4, np.arcsin(Y / r), np.where(face == 5, np.arcsin(-1.0 / r), np.arcsin(1.0 / r)))
    far = (np.abs(Y) > DIST_THRESHOLD) | (np.abs(X) > DIST_THRESHOLD)
    lon = np.where(face == 1, np.arctan2(X, 1.0), np.where(face == 2, np.arctan2(1.0, -X), np.where(
        face == 3, np.arctan2(-X, -1.0), np.where(face == 4, np.arctan2(-1.0, X), np.where(
            face == 5, np.where(far, np.arctan2(X, Y), 0.0), np.where(far, np.arctan2(X, -Y), 0.0))))))
    lon = np.where(lon < 0.0, lon + 2.0 * DD_PI, lon)
    # elem_jacobians from the corner cartp (exactly the element corners)
    k = [(cx[c], cy[c]) for c in range(4)]
    u21 = (-k[0][0] + k[1][0] + k[2][0] - k[3][0]) / 4.0; u22 = (-k[0][1] + k[1][1] + k[2][1] - k[3][1]) / 4.0
    u31 = (-k[0][0] - k[1][0] + k[2][0] + k[3][0]) / 4.0; u32 = (-k[0][1] - k[1][1] + k[2][1] + k[3][1]) / 4.0
    u41 = (k[0][0] - k[1][0] + k[2][0] - k[3][0]) / 4.0; u42 = (k[0][1] - k[1][1] + k[2][1] - k[3][1]) / 4.0
    a, b = pts[None, None, :], pts[None, :, None]
    J11 = u21 + u41 * b; J12 = u31 + u41 * a; J21 = u22 + u42 * b; J22 = u32 + u42 * a
    p2i, p2j, q2i, q2j = (1 - a) / 2, (1 - b) / 2, (1 + a) / 2, (1 + b) / 2
    x1 = p2i * p2j * cx[0] + q2i * p2j * cx[1] + q2i * q2j * cx[2] + p2i * q2j * cx[3]
    x2 = p2i * p2j * cy[0] + q2i * p2j * cy[1] + q2i * q2j * cy[2] + p2i * q2j * cy[3]
    T11, T12, T21, T22 = _vmap(x1, x2, np.broadcast_to(face, x1.shape))
    D11 = T11 * J11 + T12 * J21; D12 = T11 * J12 + T12 * J22; D21 = T21 * J11 + T22 * J21; D22 = T21 * J12 + T22 * J22
    det = D11 * D22 - D12 * D21
    I11, I12, I21, I22 = D22 / det, -D12 / det, -D21 / det, D11 / det
    md0, rmd0 = np.abs(det), 1.0 / np.abs(det)
    wf = [Fraction(1, 6), Fraction(5, 6), Fraction(5, 6), Fraction(1, 6)]
    mp = np.array([[float(wf[i] * wf[j]) for i in range(4)] for j in range(4)])[None]
    # area correction: alpha = 4 pi / sum(mp*metdet) with an (effectively) exact global sum
    per_elem = np.zeros(md0.shape[0])
    prod = (mp * md0).reshape(md0.shape[0], 16)
    for p in range(16):
        per_elem = per_elem + prod[:, p]
    import math
    alpha = 4 * DD_PI / math.fsum(per_elem.tolist())
    sa = np.sqrt(alpha)
    n = md0.shape[0]
    D = np.empty((n, 4, 4, 2, 2)); Dinv = np.empty((n, 4, 4, 2, 2))     # [...][b][a] == Fortran (a,b,i,j) memory
    D[..., 0, 0], D[..., 0, 1], D[..., 1, 0], D[..., 1, 1] = D11 * sa, D21 * sa, D12 * sa, D22 * sa
    Dinv[..., 0, 0], Dinv[..., 0, 1], Dinv[..., 1, 0], Dinv[..., 1, 1] = I11 / sa, I21 / sa, I12 / sa, I22 / sa
    metdet = md0 * alpha; rmetdet = rmd0 / alpha
    mp = np.broadcast_to(mp, metdet.shape).copy()
    spheremp = mp * metdet
    rspheremp = 1.0 / dss_sum(spheremp.reshape(n, 16), topo).reshape(n, 4, 4)
    return dict(lat=lat, lon=lon, D=D, Dinv=Dinv, metdet=metdet, rmetdet=rmetdet, mp=mp, spheremp=spheremp,
                rspheremp=rspheremp, alpha=alpha)


_OFF = np.array([0, 4, 8, 12, 16, 17, 18, 19])


def edge_descriptors(topo, owner_rank=None, rank=0):
    """Reference-style descriptors (putmapP/getmapP/reverse, 0-based columns, -1 = no neighbour) for the elements
    of `rank` (all elements when owner_rank is None), plus the neighbour-rank message slots.

    Column numbering (ours; only the contract matters -- the sender writes where the receiver reads,
    schedule_mod.F90:905,929): rank-internal edges first (receiver-owned column 20*local_e + slot), then one
    contiguous slot per neighbour rank in ascending rank order; inside a slot, columns are ordered by the
    directed-edge key (min global element, its direction) so both ranks enumerate them identically."""
    ne_, nd, rev = topo["nbr_elem"], topo["nbr_dir"], topo["nbr_rev"]
    nelem = ne_.shape[0]
    if owner_rank is None:
        owner_rank = np.zeros(nelem, dtype=np.int64)
    mine = np.flatnonzero(owner_rank == rank)
    g2l = -np.ones(nelem, dtype=np.int64); g2l[mine] = np.arange(mine.size)
    n = mine.size
    put = -np.ones((n, 8), dtype=np.int32); get = -np.ones((n, 8), dtype=np.int32); rv = np.zeros((n, 8), dtype=np.int32)
    width = np.array([4, 4, 4, 4, 1, 1, 1, 1])
    # internal: receiver (le,d) owns column 20*le+off[d]; the sender (nbr, nd) puts there
    remote = []  # (peer, key_elem, key_dir, local_e, d) for pairs crossing a rank boundary
    for le, ge in enumerate(mine):
        for d in range(8):
            nb = ne_[ge, d]
            if nb < 0:
                continue
            if d < 4:
                rv[le, d] = rev[ge, d]
            if owner_rank[nb] == rank:
                col = 20 * le + _OFF[d]
                get[le, d] = col
                put[g2l[nb], nd[ge, d]] = col
            else:
                # undirected pair key: the (element, direction) with the smaller global element id
                key = (int(ge), int(d)) if ge < nb else (int(nb), int(nd[ge, d]))
                remote.append((int(owner_rank[nb]), key[0], key[1], le, d))
    base = 20 * n
    sched = []
    for peer in sorted(set(r[0] for r in remote)):
        items = sorted([r for r in remote if r[0] == peer], key=lambda r: (r[1], r[2]))
        ptr = base
        for (_, _, _, le, d) in items:
            put[le, d] = base; get[le, d] = base
            base += int(width[d])
        sched.append((peer, ptr + 1, base - ptr))   # ptrP is 1-based as the reference stores it
    return dict(putmapP=put, getmapP=get, reverse=rv, send=sched, recv=list(sched), nbuf=base, elems=mine)
