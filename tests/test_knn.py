"""SURVEY.md 8f row f-2 -- K nearest template vertices + the SMPL neighbour-blended LBS quantities.
CPU: the numpy oracle against vectors produced by the reference's own statements (hugs_wo_trimlp.py:47-119, compiled
from /root/reference by tests/golden/make_golden.py; the pytorch3d search itself is absent from the reference, so the
search is pinned only by its published contract -- brute force, checked here independently in float64).
GPU: the HIP kernels (through the C ABI and the drop-in Python functions) against the oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_substeps.npz"))
FLOAT_TOL = 2e-6   # relative, on fp32 sums of <= 6 products (order of summation differs between torch, numpy and the kernel)


def body(n, m, J, seed):
    r = np.random.default_rng(seed)
    templ = (r.standard_normal((m, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)
    joints = templ[r.choice(m, J, replace=False)]
    logits = -np.linalg.norm(templ[:, None, :] - joints[None], axis=-1) / 0.01
    w = np.exp(logits - logits.max(1, keepdims=True))
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    pts = (templ[r.integers(0, m, n)] + 0.02 * r.standard_normal((n, 3))).astype(np.float32)
    return templ, w, pts


def test_oracle_search_is_the_brute_force_k_smallest():
    templ, _, pts = body(300, 500, 24, 1)
    pts[:7] = templ[:7]                                   # exact hits
    templ[100] = templ[3]                                 # a duplicated template vertex: a genuine tie
    d, i = ko.knn_points(pts, templ, 6)
    d64 = ((pts[:, None, :].astype(np.float64) - templ[None].astype(np.float64)) ** 2).sum(-1)
    ref = np.sort(d64, axis=1)[:, :6]
    np.testing.assert_allclose(d, ref, rtol=1e-5, atol=1e-12)
    assert (np.diff(d, axis=1) >= 0).all() and d[3, 0] == 0 and d[3, 1] == 0
    assert i[3, 0] == 3 and i[3, 1] == 100                # tie -> lower index first
    assert np.array_equal(np.take_along_axis(((pts[:, None] - templ[None]) ** 2).astype(np.float32).sum(-1) * 0 +
                                             ((pts[:, None, 0] - templ[None, :, 0]) ** 2 + (pts[:, None, 1] - templ[None, :, 1]) ** 2 +
                                              (pts[:, None, 2] - templ[None, :, 2]) ** 2), i, axis=1), d)


def test_oracle_matches_reference_statements():
    d, i = ko.knn_points(G["knn_points"], G["knn_template"], 6)
    assert np.array_equal(d, G["knn_search_dists"]) and np.array_equal(i, G["knn_search_idx"])
    xd, w = ko.smpl_lbsweight_top_k(G["knn_lbs_weights"], G["knn_points"], G["knn_template"], K=6)
    np.testing.assert_allclose(xd, G["knn_lbsweight_dist"], rtol=FLOAT_TOL, atol=1e-9)
    np.testing.assert_allclose(w, G["knn_lbsweight_weights"], rtol=FLOAT_TOL, atol=1e-8)
    xd2, T, info = ko.smpl_lbsmap_top_k(G["knn_lbs_weights"], G["knn_verts_transform"], G["knn_points"], G["knn_template"],
                                        K=6, addition_info=G["knn_addition_info"])
    np.testing.assert_allclose(xd2, G["knn_lbsmap_dist"], rtol=FLOAT_TOL, atol=1e-9)
    np.testing.assert_allclose(T, G["knn_lbsmap_transform"], rtol=FLOAT_TOL, atol=1e-7)
    np.testing.assert_allclose(info, G["knn_lbsmap_info"], rtol=FLOAT_TOL, atol=1e-7)
    # the fixture exercises the confidence gate both ways, and the blended weights still sum to one
    nb = G["knn_lbs_weights"][i]
    conf = np.exp(-np.abs(nb - nb[:, :1]).sum(-1) / 0.02) > 0.9
    assert conf[:, 0].all() and 0.05 < conf[:, 1:].mean() < 0.95
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_hip_matches_golden_vectors(device):
    from hugs_amd.knn import knn_points, smpl_lbsmap_top_k, smpl_lbsweight_top_k
    t = lambda k: torch.from_numpy(G[k].copy()).to(device)
    res = knn_points(t("knn_points")[None], t("knn_template")[None], K=6)
    assert res.dists.shape == (1, 250, 6) and res.idx.dtype == torch.int64 and res.knn is None
    assert np.array_equal(res.dists[0].cpu().numpy(), G["knn_search_dists"])
    assert np.array_equal(res.idx[0].cpu().numpy(), G["knn_search_idx"])
    xd, w = smpl_lbsweight_top_k(t("knn_lbs_weights"), t("knn_points")[None], t("knn_template")[None])
    np.testing.assert_allclose(xd[0].cpu().numpy(), G["knn_lbsweight_dist"], rtol=FLOAT_TOL, atol=1e-9)
    np.testing.assert_allclose(w[0].cpu().numpy(), G["knn_lbsweight_weights"], rtol=FLOAT_TOL, atol=1e-8)
    vT = t("knn_verts_transform")[None].requires_grad_(True)
    xd2, T, info = smpl_lbsmap_top_k(t("knn_lbs_weights"), vT, t("knn_points")[None], t("knn_template")[None], K=6,
                                     addition_info=t("knn_addition_info")[None])
    np.testing.assert_allclose(xd2[0].detach().cpu().numpy(), G["knn_lbsmap_dist"], rtol=FLOAT_TOL, atol=1e-9)
    np.testing.assert_allclose(T[0].detach().cpu().numpy(), G["knn_lbsmap_transform"], rtol=FLOAT_TOL, atol=1e-7)
    np.testing.assert_allclose(info[0].detach().cpu().numpy(), G["knn_lbsmap_info"], rtol=FLOAT_TOL, atol=1e-7)
    # gradients reach verts_transform (and addition_info) as upstream: d/dverts_transform[v] = sum over the (point, neighbour)
    # pairs that picked vertex v of wgt * dL/dxyz_transform[point] -- the fused backward's scatter-add against numpy's
    aI = t("knn_addition_info")[None].requires_grad_(True)
    vT2 = t("knn_verts_transform")[None].requires_grad_(True)
    _, T2, info2 = smpl_lbsmap_top_k(t("knn_lbs_weights"), vT2, t("knn_points")[None], t("knn_template")[None], K=6, addition_info=aI)
    r = np.random.default_rng(0)
    gT, gI = r.standard_normal(T2.shape[1:]).astype(np.float32), r.standard_normal(info2.shape[1:]).astype(np.float32)
    ((T2[0] * torch.from_numpy(gT).to(device)).sum() + (info2[0] * torch.from_numpy(gI).to(device)).sum()).backward()
    _, wgt = ko._blend_weights(G["knn_lbs_weights"], G["knn_search_dists"], G["knn_search_idx"])
    want_T = np.zeros(G["knn_verts_transform"].shape, np.float64)
    want_I = np.zeros(G["knn_addition_info"].shape, np.float64)
    np.add.at(want_T, G["knn_search_idx"], wgt[:, :, None, None].astype(np.float64) * gT[:, None].astype(np.float64))
    np.add.at(want_I, G["knn_search_idx"], wgt[:, :, None].astype(np.float64) * gI[:, None].astype(np.float64))
    np.testing.assert_allclose(vT2.grad[0].cpu().numpy(), want_T, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(aI.grad[0].cpu().numpy(), want_I, rtol=1e-5, atol=1e-6)
    T.sum().backward()
    assert vT.grad is not None and float(vT.grad.abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,K", [(1, 8, 1), (63, 6, 6), (1000, 6890, 6), (4097, 1023, 8), (20011, 6890, 6), (300, 501, 3)])
def test_hip_search_is_bit_exact_against_the_oracle(n, m, K, device):
    from hugs_amd.knn import knn_points
    templ, _, pts = body(n, m, min(24, m), seed=n + m)
    if m > 200:
        templ[m // 2] = templ[5]                          # tie
        pts[0] = templ[5]
    res = knn_points(torch.from_numpy(pts)[None].to(device), torch.from_numpy(templ)[None].to(device), K=K, return_nn=True)
    d, i = ko.knn_points(pts, templ, K)
    assert np.array_equal(res.idx[0].cpu().numpy(), i)
    assert np.array_equal(res.dists[0].cpu().numpy().view(np.uint32), d.view(np.uint32))
    assert np.array_equal(res.knn[0].cpu().numpy(), templ[i])


@pytest.mark.gpu
def test_hip_lbsweight_full_size_against_the_oracle(device):
    """SMPL-sized template (6 890 vertices, 24 joints), 50k Gaussians: the shape of hugs_trimlp.py:480-484."""
    from hugs_amd.knn import smpl_lbsweight_top_k
    templ, w, pts = body(50_000, 6890, 24, seed=5)
    xd, out = smpl_lbsweight_top_k(torch.from_numpy(w).to(device), torch.from_numpy(pts)[None].to(device),
                                   torch.from_numpy(templ)[None].to(device))
    rd, rw = ko.smpl_lbsweight_top_k(w, pts, templ, K=6)
    # a neighbour whose confidence exp(.) sits within rounding of the 0.9 threshold may flip: allow a handful of rows
    bad = (np.abs(out[0].cpu().numpy() - rw) > FLOAT_TOL * np.maximum(np.abs(rw), 1e-3)).any(-1)
    assert bad.sum() <= 3, f"{bad.sum()} of {len(bad)} rows differ"
    good = ~bad
    np.testing.assert_allclose(xd[0].cpu().numpy()[good], rd[good], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(out[0].sum(-1).cpu().numpy(), 1.0, atol=1e-5)   # hugs_trimlp.py:486-489's sanity check


@pytest.mark.gpu
def test_hip_search_takes_batches_and_offset_views(device):
    """B = 2 with the SMPL template: element 1 starts 6890*12 bytes in (8 mod 16); likewise a sliced cloud handed to
    distCUDA2.  pytorch3d / simple_knn have no alignment rule, so neither has the library (float alignment only)."""
    from hugs_amd.knn import distCUDA2, knn_points, smpl_lbsweight_top_k
    sets = [body(500, 6890, 24, seed=s) for s in (21, 22)]
    templ = np.stack([s[0] for s in sets]); pts = np.stack([s[2] for s in sets]); w = sets[0][1]
    res = knn_points(torch.from_numpy(pts).to(device), torch.from_numpy(templ).to(device), K=6)
    xd, out = smpl_lbsweight_top_k(torch.from_numpy(w).to(device), torch.from_numpy(pts).to(device), torch.from_numpy(templ).to(device))
    for b in range(2):
        d, i = ko.knn_points(pts[b], templ[b], 6)
        assert np.array_equal(res.idx[b].cpu().numpy(), i)
        assert np.array_equal(res.dists[b].cpu().numpy().view(np.uint32), d.view(np.uint32))
        rd, rw = ko.smpl_lbsweight_top_k(w, pts[b], templ[b], K=6)
        np.testing.assert_allclose(out[b].cpu().numpy(), rw, rtol=1e-4, atol=1e-6)
    cloud = np.random.default_rng(3).standard_normal((1003, 3)).astype(np.float32)
    for off in (1, 2, 3):   # storage offsets of 12, 24, 36 bytes
        got = distCUDA2(torch.from_numpy(cloud).to(device)[off:])
        assert np.array_equal(got.cpu().numpy().view(np.uint32), ko.dist_cuda2(cloud[off:]).view(np.uint32))


@pytest.mark.gpu
def test_hip_knn_errors(device):
    from hugs_amd.knn import knn_points
    p = torch.zeros(1, 4, 3, device=device)
    with pytest.raises(RuntimeError):
        knn_points(p, torch.zeros(1, 3, 3, device=device), K=6)        # fewer template points than K
    with pytest.raises(RuntimeError):
        knn_points(p, torch.zeros(1, 30, 3, device=device), K=9)       # K beyond the compiled range
    with pytest.raises(RuntimeError):
        knn_points(torch.zeros(1, 4, 3), torch.zeros(1, 30, 3), K=2)   # CPU tensors: no fallback
    assert knn_points(torch.zeros(1, 0, 3, device=device), torch.zeros(1, 30, 3, device=device), K=2).idx.shape == (1, 0, 2)


def test_oracle_dist_cuda2_small_known_answer():
    """unit grid: every interior point of a 4x4x4 lattice has 6 neighbours at distance 1 -> mean of the 3 nearest = 1"""
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    d = ko.dist_cuda2(g)
    assert np.all(d == 1.0)
    g2 = np.concatenate([g, g[:1]])                        # a duplicated point: its twin is at distance 0
    d2 = ko.dist_cuda2(g2)
    assert d2[0] == d2[-1] == np.float32(2.0 / 3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 65, 1000, 30011])
def test_hip_dist_cuda2_is_bit_exact_against_the_oracle(n, device):
    """row f-4: the initial-scale statistic of scene.py:181, including the reference's use of it"""
    from hugs_amd.knn import distCUDA2
    r = np.random.default_rng(n)
    pts = r.standard_normal((n, 3)).astype(np.float32)
    if n > 100:
        pts[7] = pts[3]                                    # duplicate
    got = distCUDA2(torch.from_numpy(pts).to(device))
    ref = ko.dist_cuda2(pts)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    scales = torch.log(torch.sqrt(torch.clamp_min(got, 0.0000001)))[..., None].repeat(1, 3)   # scene.py:181-182
    assert torch.isfinite(scales).all()
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(3, 3, device=device))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["gaussian", "clusters_and_outliers", "plane", "duplicates"])
def test_hip_dist_cuda2_grid_search_is_bit_exact(shape, device):
    """From 32 768 points on distCUDA2 searches a uniform grid (O(n)) instead of scanning the cloud (O(n^2)): same three
    nearest distances, same fp32 expression, same bits -- on a Gaussian cloud, on tight clusters with far outliers (the
    outliers fall back to scanning the cloud), on a planar cloud (a degenerate grid axis) and with many exact duplicates."""
    from hugs_amd.knn import distCUDA2
    import diff_gaussian_rasterization as dgr
    n = 33_000
    assert dgr._load().hgs_dist_cuda2_workspace(n) > 0 and dgr._load().hgs_dist_cuda2_workspace(30_000) == 0
    r = np.random.default_rng(len(shape))
    if shape == "gaussian":
        pts = r.standard_normal((n, 3))
    elif shape == "clusters_and_outliers":
        centres = r.uniform(-5, 5, (40, 3))
        pts = centres[r.integers(0, 40, n)] + 0.01 * r.standard_normal((n, 3))
        pts[:25] = r.uniform(-300, 300, (25, 3))
    elif shape == "plane":
        pts = np.concatenate([r.uniform(-1, 1, (n, 2)), np.full((n, 1), 0.25)], 1)
    else:
        pts = r.standard_normal((n // 3, 3))
        pts = np.concatenate([pts, pts, pts[: n - 2 * (n // 3)]], 0)     # every point two or three times
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).to(device)).cpu().numpy()
    ref = ko.dist_cuda2(pts)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"{(got != ref).sum()} of {n} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("K,layout", [(1, "blob"), (6, "blob"), (8, "blob"), (6, "surface"), (6, "uniform_box"), (3, "line")])
def test_grid_search_equals_the_scan(K, layout, device):
    """hgs_knn_points_ws (template on a uniform grid, shells of cells around every query) against hgs_knn_points (the scan of
    the whole template): the same K neighbours in the same order, bit for bit -- for queries on the body, far outside its
    bounding box (the fall-back to the scan), exactly on template vertices, and a template with every vertex duplicated
    (ties: the lower index first) and a degenerate flat part.  Layouts: a normal cloud with sparse tails; a surface of
    human size (waves of one or two groups); queries spread evenly over the box (many groups per wave: the open list); every
    vertex on one line (a grid of one row)."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    lib = dgr._load()
    lib.hgs_knn_workspace.restype = C.c_size_t
    lib.hgs_knn_workspace.argtypes = [C.c_int32, C.c_int32]
    args = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.hgs_knn_points.argtypes = args + [C.c_void_p]
    lib.hgs_knn_points_ws.argtypes = args + [C.c_void_p, C.c_void_p]
    r = np.random.default_rng(K)
    templ, _, pts = body(30_000, 3445, 24, seed=40 + K)
    if layout == "surface":
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from bench_knn import body_surface
        templ, pts = body_surface(3445, r), body_surface(30_000, r, noise=0.004)
    elif layout == "uniform_box":
        pts = r.uniform(templ.min(0), templ.max(0), (30_000, 3)).astype(np.float32)
    elif layout == "line":
        templ[:, 1:] = 0.25
    templ = np.concatenate([templ, templ], 0)                      # 6 890 vertices, every one twice
    templ[100:400, 1] = 0.125                                       # a flat patch
    pts[:2000] = templ[r.integers(0, templ.shape[0], 2000)]          # queries exactly on vertices
    pts[2000:2300] = r.uniform(-40, 40, (300, 3))                   # far away
    pts[2300:2310] = np.array([np.inf, -np.inf, np.nan, 1e30, -1e30, 0, 0, 0, 0, 0], np.float32)[:, None]
    n, m = pts.shape[0], templ.shape[0]
    tp, tt = torch.from_numpy(pts.astype(np.float32)).to(device), torch.from_numpy(templ.astype(np.float32)).to(device)
    out = {}
    for name in ("scan", "grid"):
        d = torch.empty(n, K, dtype=torch.float32, device=device)
        i = torch.empty(n, K, dtype=torch.int64, device=device)
        if name == "scan":
            rc = lib.hgs_knn_points(n, tp.data_ptr(), m, tt.data_ptr(), K, d.data_ptr(), i.data_ptr(), None)
        else:
            nbytes = lib.hgs_knn_workspace(n, m)
            assert nbytes > 0
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            rc = lib.hgs_knn_points_ws(n, tp.data_ptr(), m, tt.data_ptr(), K, d.data_ptr(), i.data_ptr(), ws.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        out[name] = (d.cpu().numpy(), i.cpu().numpy())
    ok = np.isfinite(pts).all(1)                                    # (non-finite queries: whatever the scan leaves, unspecified)
    assert np.array_equal(out["scan"][1][ok], out["grid"][1][ok])
    assert np.array_equal(out["scan"][0][ok].view(np.uint32), out["grid"][0][ok].view(np.uint32))
    assert lib.hgs_knn_workspace(1000, 100) == 0


@pytest.mark.gpu
def test_grids_survive_non_finite_coordinates(device):
    """A template vertex / a cloud point at infinity (or NaN) leaves the bounding box without a cell size: the grid falls back to
    one cell and every finite query still gets the scan's answer."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    from hugs_amd.knn import distCUDA2
    lib = dgr._load()
    lib.hgs_knn_workspace.restype = C.c_size_t
    lib.hgs_knn_workspace.argtypes = [C.c_int32, C.c_int32]
    args = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.hgs_knn_points.argtypes = args + [C.c_void_p]
    lib.hgs_knn_points_ws.argtypes = args + [C.c_void_p, C.c_void_p]
    templ, _, pts = body(8000, 1500, 24, seed=77)
    for bad in (np.inf, -np.inf, np.nan):
        t = templ.copy()
        t[17, 1] = bad
        n, m, K = pts.shape[0], t.shape[0], 4
        tp, tt = torch.from_numpy(pts).to(device), torch.from_numpy(t).to(device)
        out = []
        for use_ws in (False, True):
            d = torch.empty(n, K, dtype=torch.float32, device=device)
            i = torch.empty(n, K, dtype=torch.int64, device=device)
            ws = torch.empty(lib.hgs_knn_workspace(n, m), dtype=torch.uint8, device=device)
            rc = (lib.hgs_knn_points_ws(n, tp.data_ptr(), m, tt.data_ptr(), K, d.data_ptr(), i.data_ptr(), ws.data_ptr(), None) if use_ws
                  else lib.hgs_knn_points(n, tp.data_ptr(), m, tt.data_ptr(), K, d.data_ptr(), i.data_ptr(), None))
            assert rc == 0
            torch.cuda.synchronize()
            out.append((d.cpu().numpy(), i.cpu().numpy()))
        assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
        assert (out[1][1] != 17).all()                                # the vertex at infinity is nobody's neighbour
    cloud = np.random.default_rng(5).standard_normal((40_000, 3)).astype(np.float32)
    want = distCUDA2(torch.from_numpy(cloud).to(device)).cpu().numpy()
    cloud2 = cloud.copy()
    cloud2[123] = np.inf                                              # one point at infinity: the grid search degenerates, the others keep their answer
    got = distCUDA2(torch.from_numpy(cloud2).to(device)).cpu().numpy()
    keep = np.ones(len(cloud), bool)
    keep[123] = False
    d123 = ((cloud - cloud[123]) ** 2).sum(1)
    unaffected = keep & (d123 > np.sort(d123)[40])                    # points that did not count #123 among their three nearest
    assert np.array_equal(got[unaffected], want[unaffected])
