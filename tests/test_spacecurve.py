"""SFC partition: transport_se_amd.spacecurve against the reference's own per-rank element lists.

tests/golden/ref_sfc_partition.npz holds, for ten (ne, nranks) cases, the owner rank of every element as the reference's
genspacepart assigned it (oracle/ref/make_golden.py --sfc: the unmodified spacecurve_mod/cube_mod run under mpiexec, each
rank's GlobalId list read back).  tests/golden/ref_ne2_2rank_sched.npz holds the 2-rank gid lists of the ne2 run."""
import numpy as np
import pytest

from transport_se_amd import spacecurve as sc
from transport_se_amd.driver import partition


def _cases(gold):
    z = gold("ref_sfc_partition.npz")
    for key in z.files:
        ne, nranks = (int(x[2:]) for x in key.split("_")[1:])
        yield ne, nranks, z[key].astype(np.int64)


def test_partition_equals_reference_genspacepart(gold):
    seen = 0
    for ne, nranks, owner_ref in _cases(gold):
        owner = sc.sfc_partition(ne, nranks)
        assert np.array_equal(owner, owner_ref), "ne=%d nranks=%d" % (ne, nranks)
        assert np.array_equal(partition(ne, nranks), owner_ref)     # what the driver uses
        seen += 1
    assert seen == 10


def test_ne2_two_rank_ownership_fixture(gold):
    z = gold("ref_ne2_2rank_sched.npz")
    owner = partition(2, 2)
    for r in range(2):
        assert np.array_equal(np.flatnonzero(owner == r) + 1, z["r%d_gid" % r])


@pytest.mark.parametrize("ne", [2, 3, 5, 6, 8, 12, 30, 120])
def test_curve_is_a_continuous_walk(ne):
    """every cell visited once and consecutive cells are edge neighbours (what makes SFC chunks compact)"""
    m = sc.face_curve(ne)
    assert sorted(m.reshape(-1).tolist()) == list(range(ne * ne))
    pos = np.empty((ne * ne, 2), dtype=np.int64)
    ii, jj = np.meshgrid(np.arange(ne), np.arange(ne), indexing="ij")
    pos[m.reshape(-1)] = np.stack([ii.reshape(-1), jj.reshape(-1)], 1)
    assert np.all(np.abs(np.diff(pos, axis=0)).sum(1) == 1)


def test_chunk_sizes_and_neighbour_ranks_at_the_headline_config():
    """ne120 on 8 ranks: 10 800 elements each, and compact chunks: few neighbour ranks, short boundary"""
    from transport_se_amd import cube_mesh as cm
    owner = sc.sfc_partition(120, 8)
    assert np.bincount(owner).tolist() == [10800] * 8
    topo = cm.topology(120)
    nb = topo["nbr_elem"]
    for r in range(8):
        mine = owner == r
        peers = owner[nb[mine][nb[mine] >= 0]]
        assert len(set(peers.tolist()) - {r}) <= 5
        boundary = np.any((owner[np.where(nb >= 0, nb, 0)] != r) & (nb >= 0), axis=1) & mine
        assert boundary.sum() < 0.09 * mine.sum()      # a 104 x 104 patch has ~4 % of its elements on the rim
