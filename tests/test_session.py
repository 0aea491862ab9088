"""Several devices behind the Step seam (origin_amd/session.py): one process, one context and one
thread per device, the reference's call pattern unchanged (origin.py:193-208: one session
object, ``orig.stepNN_...()`` called one after the other).

CPU: the thread group's collectives and the column-list exchange between threads.
GPU (-m gpu): ``SimpleOrig(devices=[0, 0])`` -- two contexts on ONE card, strips through the host
group -- against the one-context chain, on a regular and on an irregular (reference-made) area map;
steps 6 and 7's reductions over the pieces; a twin with ``devices=[0, 1]`` over RCCL that skips
itself on boxes with one GPU.
"""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _threads(world, fn):
    out, errs = [None] * world, []

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return out


def test_thread_group_collectives_and_column_exchange():
    from origin_amd import multigpu
    from origin_amd.session import ThreadGroup, _Shared
    world = 3
    sh = _Shared(world)
    groups = [ThreadGroup(sh, r) for r in range(world)]

    def body(r):
        g = groups[r]
        assert g.broadcast(b"id-from-1" if r == 1 else b"", src=1) == b"id-from-1"
        s = g.allreduce(np.array([r + 1.0, 10.0 * r]), "sum")
        m = g.allreduce(np.array([float(r)]), "max")
        g.barrier()
        return s, m
    for s, m in _threads(world, body):
        assert np.array_equal(s, [6.0, 30.0]) and m[0] == 2.0

    # the halo of an OwnerTiling between threads: every rank rebuilds what it needs
    G = np.load(os.path.join(ROOT, "tests", "golden", "g10_areas.npz"))
    amap = G["many_areamap"].astype(int)
    Ny, Nx = amap.shape
    tl = multigpu.OwnerTiling.from_areamap(amap, world, halo=4)
    field = np.arange(2 * Ny * Nx, dtype=np.float64).reshape(2, Ny, Nx)

    class C:   # the part of TileComm exchange_halo_host uses
        def __init__(self, g):
            self.group = g

    def halo(r):
        t = tl.tile(r)
        tile = np.where(tl.owned_tile(r)[None], field[:, t.y0:t.y1, t.x0:t.x1], -1.0)
        return multigpu.exchange_halo_host(C(groups[r]), tl, r, tile)
    for r, ext in enumerate(_threads(world, halo)):
        (ey0, ey1, ex0, ex1), _ = tl.extended(r)
        need = tl.needed(r)[ey0:ey1, ex0:ex1] | tl.owned_ext(r)
        assert np.array_equal(ext[:, need], field[:, ey0:ey1, ex0:ex1][:, need])


def test_thread_group_failure_breaks_the_barrier_instead_of_hanging():
    from origin_amd.session import ThreadGroup, _Shared
    sh = _Shared(2)
    groups = [ThreadGroup(sh, r) for r in range(2)]
    seen = []

    def body(r):
        if r == 0:
            groups[0].abort()       # what DeviceGroup does for a rank that raised
            return
        try:
            groups[1].allreduce(np.zeros(1))
        except threading.BrokenBarrierError:
            seen.append("broken")
    _threads(2, body)
    assert seen == ["broken"]


# ----------------------------------------------------------------------------------- GPU
def _chain(orig, areamap, purity=None):
    orig.step01_preprocessing()
    orig.step02_areas.set_areamap(areamap)
    orig.step03_compute_PCA_threshold()
    orig.step04_compute_greedy_PCA()
    orig.step05_compute_TGLR()
    if purity is not None:
        orig.step06_compute_purity_threshold(purity=purity)
    return orig


def _compare(one, two, mask):
    # DCT + standardisation: the per-channel mean is summed over two ranks instead of one
    # (float64: the last bit of a float32 cube_std may move)
    assert np.max(np.abs(two.cube_std._data - one.cube_std._data)) <= 1e-6
    assert np.max(np.abs(two.cont_dct._data - one.cont_dct._data)) <= 1e-6
    assert np.allclose(np.asarray(two.ima_std), np.asarray(one.ima_std), atol=1e-6)
    assert np.array_equal(np.asarray(two.segmap_merged), np.asarray(one.segmap_merged))
    assert np.allclose(two.thresO2, one.thresO2, rtol=1e-9)
    # greedy PCA: the same areas, whole, on one rank each
    assert np.array_equal(np.asarray(two.mapO2), np.asarray(one.mapO2))
    assert np.max(np.abs(two.cube_faint._data - one.cube_faint._data)) <= 1e-6
    # GLR: to rounding (another tile geometry: other waves hold a given spaxel)
    for name in ("cube_correl", "cube_correl_min"):
        assert np.max(np.abs(getattr(two, name)._data - getattr(one, name)._data)) <= 1e-4, name
    assert np.mean(two.cube_profile._data != one.cube_profile._data) <= 1e-4
    assert np.max(np.abs(np.asarray(two.maxmap) - np.asarray(one.maxmap))) <= 1e-4
    assert np.max(np.abs(np.asarray(two.minmap) - np.asarray(one.minmap))) <= 1e-4
    for name in ("cube_local_max", "cube_local_min", "cube_std_local_max", "cube_std_local_min"):
        a, b = getattr(two, name)._data, getattr(one, name)._data
        assert np.mean((a != 0) != (b != 0)) <= 1e-4, name
        both = (a != 0) & (b != 0)
        assert np.max(np.abs(a - b)[both]) <= 1e-4, name
        assert not np.any(a[mask] != 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["grid", "areas"])
def test_two_contexts_on_one_card_behind_the_step_seam(kind):
    """``SimpleOrig(devices=[0, 0])``: the user's calls are those of the one-device session; the
    hot steps run on two contexts of the same card (threads of this process, strips through the
    host group).  Against the one-context chain: PCA outputs identical (mapO2) / to the last bit
    of the all-reduced mean (cubes), GLR to rounding.  ``areas``: the irregular, reference-made
    area map of golden G10 (whole areas per rank, boxes, column lists)."""
    from _mp_tiled_worker import areas_field, field
    from origin_amd.session import TiledCube
    from origin_amd.steps import SimpleOrig
    if kind == "areas":
        f, raw, var, mask = areas_field()
    else:
        os.environ.pop("TILED_FIELD", None)
        f, raw, var, mask = field()
    psf = f.PSF.astype(float)
    one = _chain(SimpleOrig(raw, var, mask, psf, f.profiles), f.areamap, purity=0.8)
    two = SimpleOrig(raw, var, mask, psf, f.profiles, devices=[0, 0])
    _chain(two, f.areamap, purity=0.8)
    assert isinstance(two._hip_cache["cube_faint"], TiledCube)
    assert two._hip_session.world == 2 and two._hip_session.group.backend == "host"
    _compare(one, two, mask)
    # step 6: the purity curves are sums of per-piece counts
    assert np.isclose(two.param["threshold"], one.param["threshold"], rtol=1e-5, equal_nan=True)
    assert np.isclose(two.param["threshold_std"], one.param["threshold_std"], rtol=1e-5,
                      equal_nan=True)
    assert np.allclose(np.asarray(two.Pval["Det_M"]), np.asarray(one.Pval["Det_M"]), atol=2)
    # step 7's thresholding over the pieces: np.where's order, the same detections
    from origin_amd import detection
    lmax = one.cube_local_max._data
    smax = one.cube_std_local_max._data
    t_cor = float(np.sort(lmax[lmax > 0])[-200]) + 1e-3
    t_std = float(np.sort(smax[smax > 0])[-200]) + 1e-3
    c1 = detection.from_session(one, threshold=t_cor, threshold_std=t_std)[0]
    c2 = detection.from_session(two, threshold=t_cor, threshold_std=t_std)[0]
    a = set(zip(c1["z0"].tolist(), c1["y0"].tolist(), c1["x0"].tolist(), c1["comp"].tolist()))
    b = set(zip(c2["z0"].tolist(), c2["y0"].tolist(), c2["x0"].tolist(), c2["comp"].tolist()))
    assert len(a ^ b) <= max(2, len(a) // 100)      # (values within rounding of the threshold)
    n2 = int(np.sum(c2["comp"] == 0))
    lin = (c2["z0"][:n2] * raw.shape[1] + c2["y0"][:n2]) * raw.shape[2] + c2["x0"][:n2]
    assert np.all(np.diff(lin) > 0)                 # C order
    two._hip_session.close()


@pytest.mark.gpu
def test_three_contexts_on_one_card_irregular_areas():
    """Three ranks (an odd split: 1 + 2 in the bisection of the areas) on the irregular area map:
    same checks as with two."""
    from _mp_tiled_worker import areas_field
    from origin_amd.steps import SimpleOrig
    f, raw, var, mask = areas_field()
    psf = f.PSF.astype(float)
    one = _chain(SimpleOrig(raw, var, mask, psf, f.profiles), f.areamap)
    three = _chain(SimpleOrig(raw, var, mask, psf, f.profiles, devices=[0, 0, 0]), f.areamap)
    sess = three._hip_session
    assert sess.world == 3 and sorted(sess.p2.balance()["areas_per_rank"]) == [1, 2, 2]
    _compare(one, three, mask)
    sess.close()


@pytest.mark.gpu
def test_tiled_steps_take_cubes_from_elsewhere():
    """A session reloaded from its files (steps.py:342-352) hands step 4 a cube_std and step 5 a
    cube_faint that no rank holds: the tiled steps distribute the host cubes over the ranks' boxes
    and give what the resident chain gives."""
    from _mp_tiled_worker import field
    from origin_amd.steps import LazyCube, SimpleOrig
    os.environ.pop("TILED_FIELD", None)
    f, raw, var, mask = field()
    psf = f.PSF.astype(float)
    a = _chain(SimpleOrig(raw, var, mask, psf, f.profiles, devices=[0, 0]), f.areamap)
    b = SimpleOrig(raw, var, mask, psf, f.profiles, devices=[0, 0])
    b.step01_preprocessing()
    b.step02_areas.set_areamap(f.areamap)
    b.step03_compute_PCA_threshold()
    # cube_std as a host array only (what a DataObj holds after a reload), nothing cached
    host_std = b.cube_std._data.astype(np.float32)
    b._hip_cache.pop("cube_std")
    b.steps["preprocessing"].cube_std = LazyCube(host=host_std.astype(np.float64))
    for st in b._hip_session.rk:
        st.pop("cube_std", None)
    b.step04_compute_greedy_PCA()
    assert np.array_equal(np.asarray(b.mapO2), np.asarray(a.mapO2))
    assert np.max(np.abs(b.cube_faint._data - a.cube_faint._data)) <= 1e-6
    host_faint = b.cube_faint._data.copy()
    b._hip_cache.pop("cube_faint")
    b.steps["compute_greedy_PCA"].cube_faint = LazyCube(host=host_faint)
    b.step05_compute_TGLR()
    assert np.max(np.abs(b.cube_correl._data - a.cube_correl._data)) <= 1e-4
    assert np.max(np.abs(np.asarray(b.maxmap) - np.asarray(a.maxmap))) <= 1e-4
    a._hip_session.close()
    b._hip_session.close()


@pytest.mark.gpu
def test_a_failing_rank_fails_the_step_and_leaves_the_group_usable():
    from _mp_tiled_worker import field
    from origin_amd.session import DeviceGroup
    os.environ.pop("TILED_FIELD", None)
    g = DeviceGroup([0, 0])

    def bad(r):
        if r == 1:
            raise ValueError("rank 1 gives up")
        g.comms[0].group.barrier()
    with pytest.raises(ValueError, match="rank 1 gives up"):
        g.run(bad)
    tot = g.run(lambda r: g.comms[r].allreduce_sum(np.array([1.0 + r])))
    assert tot[0][0] == 3.0 and tot[1][0] == 3.0
    g.close()


@pytest.mark.gpu
def test_two_devices_behind_the_step_seam_over_rccl():
    """The same session on two GPUs: cubes travel over RCCL, one communicator per context, both in
    this process.  Skips itself on a box with one device."""
    from origin_amd.device import device_count
    if device_count() < 2:
        pytest.skip("needs two GPUs")
    from _mp_tiled_worker import areas_field
    from origin_amd.steps import SimpleOrig
    f, raw, var, mask = areas_field()
    psf = f.PSF.astype(float)
    one = _chain(SimpleOrig(raw, var, mask, psf, f.profiles), f.areamap)
    two = _chain(SimpleOrig(raw, var, mask, psf, f.profiles, devices=[0, 1]), f.areamap)
    assert two._hip_session.group.backend == "rccl"
    assert all(c.device_p2p for c in two._hip_session.group.comms)
    _compare(one, two, mask)
    two._hip_session.close()
