"""Analytic known-answer tests for the parts of the oracle whose arithmetic lives in absent
third-party packages (torchdiffeq, faiss): PARITY UNPINNED there, so these anchor it."""
import numpy as np
import torch

from oracle import ode, nets, knn


def test_grid_constructor():
    assert len(ode.grid_dts(0.1)) == 10
    assert len(ode.grid_dts(0.25)) == 4
    g = ode.fixed_grid(0.3)
    assert len(g) == 5 and float(g[-1]) == 1.0
    np.testing.assert_allclose(float(ode.grid_dts(0.3)[-1]), 0.1, rtol=1e-6)
    # dt is computed in fp32 from the fp32 grid, like the solver does
    d = ode.grid_dts(0.1)
    assert d.dtype == torch.float32 and abs(float(d.sum()) - 1.0) < 1e-6


def test_euler_identity_closed_form():
    torch.manual_seed(0)
    W = torch.randn(8, 8, dtype=torch.float64) * 0.3
    b = torch.randn(8, dtype=torch.float64)
    y0 = torch.randn(3, 8, dtype=torch.float64)
    y = ode.odeint_fixed(lambda v: ode.fc(v, W, b, "id"), y0, "euler", 0.25, dt_dtype=torch.float64)
    ref = ode.euler_linear_closed_form(y0, W, b, 0.25, 4)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-12)


def test_rk4_and_midpoint_linear():
    torch.manual_seed(1)
    A = torch.randn(6, 6, dtype=torch.float64) * 0.4
    y0 = torch.randn(2, 6, dtype=torch.float64)
    y = ode.odeint_fixed(lambda v: v @ A.T, y0, "rk4", 0.25, dt_dtype=torch.float64)
    np.testing.assert_allclose(y.numpy(), ode.rk4_linear_closed_form(y0, A, 0.25, 4).numpy(), rtol=1e-12)
    # midpoint on a linear field: y1 = (I + hA + (hA)^2/2) y0
    h = 0.5
    ym = ode.odeint_fixed(lambda v: v @ A.T, y0, "midpoint", h, dt_dtype=torch.float64)
    P = torch.eye(6, dtype=torch.float64) + h * A + (h * A) @ (h * A) / 2
    np.testing.assert_allclose(ym.numpy(), (y0 @ P.T @ P.T).numpy(), rtol=1e-12)


def test_rk4_is_three_eighths_rule_not_classic():
    # 1-D nonlinear field crossing relu's kink mid-step distinguishes the two tableaus
    w, b = torch.tensor([[-3.0]], dtype=torch.float64), torch.tensor([1.0], dtype=torch.float64)
    f = lambda v: torch.relu(v @ w.T + b)
    y0 = torch.tensor([[0.2]], dtype=torch.float64)
    h = 1.0
    k1 = f(y0); k2 = f(y0 + h * k1 / 3); k3 = f(y0 + h * (k2 - k1 / 3)); k4 = f(y0 + h * (k1 - k2 + k3))
    three8 = y0 + h * (k1 + 3 * (k2 + k3) + k4) / 8
    c1 = f(y0); c2 = f(y0 + h * c1 / 2); c3 = f(y0 + h * c2 / 2); c4 = f(y0 + h * c3)
    classic = y0 + h * (c1 + 2 * c2 + 2 * c3 + c4) / 6
    got = ode.odeint_fixed(f, y0, "rk4", 1.0, dt_dtype=torch.float64)
    assert abs(float(three8 - classic)) > 1e-3
    np.testing.assert_allclose(got.numpy(), three8.numpy(), rtol=1e-14)


def test_gem_limits():
    x = torch.rand(2, 4, 5, 6) + 0.1
    np.testing.assert_allclose(nets.gem(x, torch.tensor([1.0])).flatten(1).numpy(),
                               x.mean((2, 3)).numpy(), rtol=1e-5)
    big = nets.gem(x.double(), torch.tensor([200.0], dtype=torch.float64)).flatten(1)
    np.testing.assert_allclose(big.numpy(), x.double().amax((2, 3)).numpy(), rtol=3e-2)
    # clamp: everything below eps behaves as eps
    z = -torch.ones(1, 1, 3, 3)
    np.testing.assert_allclose(float(nets.gem(z, torch.tensor([3.0]))), 1e-6, rtol=1e-4)


def test_netvlad_single_cluster():
    torch.manual_seed(2)
    x = torch.randn(2, 8, 3, 3)
    c = torch.randn(1, 8)
    w = torch.randn(1, 8, 1, 1)
    y = nets.netvlad(x, w, c)
    xn = torch.nn.functional.normalize(x, dim=1).reshape(2, 8, -1)
    v = (xn - c.view(1, 8, 1)).sum(-1)
    np.testing.assert_allclose(y.numpy(), torch.nn.functional.normalize(v, dim=1).numpy(), rtol=1e-5, atol=1e-7)


def test_knn_planted_duplicates_and_padding():
    rng = np.random.default_rng(0)
    db = rng.standard_normal((50, 32)).astype(np.float32)
    db[7] = db[3]                     # exact duplicate rows: tie -> lower index first
    q = db[[3, 20]] + 1e-3
    D, I, D64 = knn.knn_l2_fp64(q, db, 5)
    assert list(I[0, :2]) == [3, 7] and I[1, 0] == 20
    assert np.all(np.diff(D64, axis=1) >= 0)
    # brute-force check of every entry
    full = ((q[:, None, :].astype(np.float64) - db[None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_allclose(D64, np.sort(full, axis=1)[:, :5], rtol=1e-12)
    # k > ntotal pads with (FLT_MAX, -1), faiss-style
    D2, I2, _ = knn.knn_l2_fp64(q, db[:3], 5)
    assert np.all(I2[:, 3:] == -1) and np.all(D2[:, 3:] == knn.FLT_MAX)
    # fp32 faiss-like path agrees where gaps are not tiny
    Df, If = knn.knn_l2_faisslike_fp32(q, db, 5)
    ok = knn.unambiguous_mask(D64, 1e-5)
    assert np.array_equal(If[ok], I[ok])


def test_resnet_macs_match_survey():
    from oracle import resnet
    assert abs(resnet.gmacs("resnet18", 3, 224, 224) / 1e9 - 1.403) < 0.01
    assert abs(resnet.gmacs("resnet18", 3, 224, 1344) / 1e9 - 8.415) < 0.03
    assert abs(resnet.gmacs("resnet50", 3, 224, 224) / 1e9 - 3.278) < 0.03


# ---------------------------------------------------------------------------- triplet mining
def test_mining_known_answer():
    """Hand-made case for the restated per-query mining loop (datasets_ws_nuscenes.py:1241-1258,
    1394-1404): database rows sit at x = 0,1,2,...; queries at fractional positions."""
    from oracle import mining
    db = np.zeros((12, 32), dtype=np.float32)
    db[:, 0] = np.arange(12)
    db[7] = db[6]                                      # duplicate row: the earlier candidate wins
    q = np.zeros((2, 32), dtype=np.float32)
    q[0, 0], q[1, 0] = 2.2, 6.0
    hard = {0: np.array([5, 3, 1]), 1: np.array([7, 6, 9])}
    soft = {0: np.array([1, 2, 3, 5]), 1: np.array([6])}
    sampled = np.array([9, 0, 2, 4, 6, 7, 8, 10, 11, 3])
    t = mining.compute_triplets_partial(q, db, [0, 1], hard, soft, sampled, negs_num_per_query=3)
    # query 0: best positive among (5,3,1) is 3 (|2.2-3| = 0.8 < 1.2); negatives from sorted
    # sample minus soft = (0,4,6,7,8,9,10,11): nearest 4 (1.8), 0 (2.2), then 6 and 7 tie -> 6 first
    assert t[0].tolist() == [0, 3, 4, 0, 6]
    # query 1 sits exactly on rows 6 and 7 (duplicates): positives listed (7,6,9) -> 7 comes first;
    # negatives: sample minus {6} = (0,2,3,4,7,8,9,10,11) -> 7 (0), 8 (4), 4 (4): 4 precedes 8 in
    # candidate order, so the tie goes to 4
    assert t[1].tolist() == [1, 7, 7, 4, 8]


# ---------------------------------------------------------------------------- sparse-voxel branch
def _dense(sp, extent, c):
    import torch
    g = torch.zeros((sp.nbatch, c, extent, extent, extent), dtype=sp.feats.dtype)
    for i, (b, x, y, z) in enumerate(sp.coords):
        g[b, :, x // sp.stride, y // sp.stride, z // sp.stride] = sp.feats[i]
    return g


def test_sparse_conv_equals_masked_dense_conv3d():
    """The dictionary-based sparse convolution of the oracle against torch's dense conv3d evaluated at
    the occupied sites: kernel 5 / 3 (stride 1, centred) and kernel 2 / stride 2, with the kernel index
    -> offset convention (first spatial axis fastest) made explicit."""
    import torch
    import torch.nn.functional as F
    from oracle import sparse
    torch.manual_seed(0)
    E = 12
    coords, feats = sparse.synth_cloud(2, 60, extent=E, seed=1)
    x = sparse.from_coords(feats.double(), coords)
    for ksize in (5, 3):
        cin = x.feats.shape[1]
        kern = torch.randn(ksize ** 3, cin, 6, dtype=torch.float64)
        y = sparse.conv(x, kern, ksize)
        # dense weight [co][ci][dx][dy][dz] = kern[dx + k*dy + k*k*dz][ci][co]
        wd = kern.view(ksize, ksize, ksize, cin, 6).permute(4, 3, 2, 1, 0).contiguous()
        yd = F.conv3d(_dense(x, E, cin), wd, padding=ksize // 2)
        for i, (b, cx, cy, cz) in enumerate(y.coords):
            assert torch.allclose(y.feats[i], yd[b, :, cx, cy, cz], atol=1e-12)
        x = sparse.SpT(y.coords, torch.tanh(y.feats), 1, 2)
    kern = torch.randn(8, 6, 4, dtype=torch.float64)
    y = sparse.conv(x, kern, 2, stride=2)
    wd = kern.view(2, 2, 2, 6, 4).permute(4, 3, 2, 1, 0).contiguous()
    yd = F.conv3d(_dense(x, E, 6), wd, stride=2)
    assert y.stride == 2 and len(y.coords) == len({(b, cx // 2, cy // 2, cz // 2) for b, cx, cy, cz in x.coords})
    for i, (b, cx, cy, cz) in enumerate(y.coords):
        assert torch.allclose(y.feats[i], yd[b, :, cx // 2, cy // 2, cz // 2], atol=1e-12)
    # second level: a stride-2 tensor convolved with kernel 3 reaches +-2 in original units
    k3 = torch.randn(27, 4, 3, dtype=torch.float64)
    z = sparse.conv(y, k3, 3)
    zd = F.conv3d(_dense(y, E // 2, 4), k3.view(3, 3, 3, 4, 3).permute(4, 3, 2, 1, 0).contiguous(), padding=1)
    for i, (b, cx, cy, cz) in enumerate(z.coords):
        assert torch.allclose(z.feats[i], zd[b, :, cx // 2, cy // 2, cz // 2], atol=1e-12)


def test_sparse_transposed_conv_equals_dense_conv_transpose3d():
    """oracle.sparse.conv_transpose (kernel 2 / stride 2 onto the finer level's existing coordinates) against torch's dense
    conv_transpose3d of the coarse grid, read at the fine sites."""
    import torch
    import torch.nn.functional as F
    from oracle import sparse
    torch.manual_seed(1)
    E = 12
    coords, feats = sparse.synth_cloud(2, 70, extent=E, seed=3)
    fine = sparse.from_coords(feats.double(), coords)
    fine = sparse.SpT(fine.coords, torch.randn(len(fine.coords), 5, dtype=torch.float64), 1, 2)
    coarse = sparse.conv(fine, torch.randn(8, 5, 6, dtype=torch.float64), 2, stride=2)
    kern = torch.randn(8, 6, 4, dtype=torch.float64)
    up = sparse.conv_transpose(coarse, kern, fine)
    assert up.coords == fine.coords and up.stride == 1
    # dense weight [ci][co][dx][dy][dz] = kern[dx + 2*dy + 4*dz][ci][co]
    wd = kern.view(2, 2, 2, 6, 4).permute(3, 4, 2, 1, 0).contiguous()
    ud = F.conv_transpose3d(_dense(coarse, E // 2, 6), wd, stride=2)
    for i, (b, cx, cy, cz) in enumerate(up.coords):
        assert torch.allclose(up.feats[i], ud[b, :, cx, cy, cz], atol=1e-12)
    # one level up (stride 4 -> 2)
    c4 = sparse.conv(coarse, torch.randn(8, 6, 6, dtype=torch.float64), 2, stride=2)
    up2 = sparse.conv_transpose(c4, kern, coarse)
    u2 = F.conv_transpose3d(_dense(c4, E // 4, 6), wd, stride=2)
    for i, (b, cx, cy, cz) in enumerate(up2.coords):
        assert torch.allclose(up2.feats[i], u2[b, :, cx // 2, cy // 2, cz // 2], atol=1e-12)


def test_sparse_tensor_merges_duplicates_and_floors():
    import torch
    from oracle import sparse
    c = torch.tensor([[0, 1.2, 2.9, 0.1], [0, 1.7, 2.1, 0.9], [0, -0.5, 0.0, 0.0], [1, 1.0, 2.0, 0.0]])
    f = torch.tensor([[1.0], [3.0], [5.0], [7.0]])
    x = sparse.from_coords(f, c)
    assert x.coords == [(0, -1, 0, 0), (0, 1, 2, 0), (1, 1, 2, 0)]
    assert x.feats.view(-1).tolist() == [5.0, 2.0, 7.0]
