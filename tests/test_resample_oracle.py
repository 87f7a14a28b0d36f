"""F3 (affine Lanczos-3 resample) - CPU tests of the oracle's definition.  There is no reference arithmetic
for this step (the reference shells out to SWarp, scripts/resample_all.sh:330-342): parity with SWarp is
unpinned; the oracle is checked against closed-form properties and an independent float64 evaluation."""
import numpy as np

from oracle import apref


def _lanczos(d):
    r = np.sinc(d) * np.sinc(d / 3.0)
    r[np.abs(d) >= 3] = 0
    return r


def test_table_rows():
    lut = apref.lanczos3_table(256)
    assert lut.shape == (257, 6) and lut.dtype == np.float32
    assert np.array_equal(lut[0], np.array([0, 0, 1, 0, 0, 0], np.float32))        # zero offset: the pixel itself
    assert np.array_equal(lut[256], np.array([0, 0, 0, 1, 0, 0], np.float32))      # offset 1: the next pixel
    np.testing.assert_allclose(lut.sum(1), 1.0, atol=3e-7)
    np.testing.assert_allclose(lut[128], lut[128][::-1], atol=1e-7)                 # half-pixel offset is symmetric


def test_identity_and_integer_shift_are_exact():
    rng = np.random.default_rng(0)
    img = rng.normal(100, 10, (40, 50)).astype(np.float32)
    o, w = apref.resample_affine(img, [[1, 0, 0, 0, 1, 0]])
    assert w[0].sum() == (40 - 5) * (50 - 5)                   # 2 pixels on the low side, 3 on the high side missing
    assert np.array_equal(o[0][w[0] == 1], img[w[0] == 1]) and np.isnan(o[0][w[0] == 0]).all()
    assert w[0, 2, 2] == 1 and w[0, 1, 2] == 0 and w[0, 36, 46] == 1 and w[0, 37, 46] == 0
    o, w = apref.resample_affine(img, [[1, 0, 3, 0, 1, 2]], fscale=[0.5])
    yy, xx = np.nonzero(w[0])
    assert np.array_equal(o[0][yy, xx], img[yy + 2, xx + 3] * np.float32(0.5))


def test_matches_direct_float64_lanczos():
    yy, xx = np.mgrid[0:80, 0:90]
    img = (100 + 20 * np.sin(xx / 7.0) + 15 * np.cos(yy / 5.0) + 0.1 * xx).astype(np.float32)
    A = np.array([[np.cos(0.003), -np.sin(0.003), 1.37, np.sin(0.003), np.cos(0.003), 0.61]])
    o, w = apref.resample_affine(img, A, n_phases=4096)
    xin = A[0, 0] * xx + A[0, 1] * yy + A[0, 2]
    yin = A[0, 3] * xx + A[0, 4] * yy + A[0, 5]
    ref = np.full(img.shape, np.nan)
    for y, x in zip(*np.nonzero(w[0])):
        kx = np.arange(np.floor(xin[y, x]) - 2, np.floor(xin[y, x]) + 4).astype(int)
        ky = np.arange(np.floor(yin[y, x]) - 2, np.floor(yin[y, x]) + 4).astype(int)
        wx, wy = _lanczos(kx - xin[y, x]), _lanczos(ky - yin[y, x])
        ref[y, x] = (wy / wy.sum()) @ img[np.ix_(ky, kx)].astype(np.float64) @ (wx / wx.sum())
    assert w[0].mean() > 0.85
    assert np.nanmax(np.abs(o[0] - ref)) < 1.5e-3             # phase quantisation 1/4096 px x gradient ~3 / px + f32 rounding
    # constant images stay constant to float32 rounding of the weights
    o, w = apref.resample_affine(np.full((30, 30), 1234.5, np.float32), [[1.001, 0.002, 0.3, -0.002, 0.999, 0.7]])
    assert np.abs(o[0][w[0] == 1] - 1234.5).max() < 1e-3


def test_mask_and_nonfinite_poison_their_6x6_neighbourhood():
    img = np.ones((32, 32), np.float32)
    img[10, 20] = np.inf
    mask = np.zeros((32, 32), np.uint8)
    mask[20, 8] = 2
    o, w = apref.resample_affine(img, [[1, 0, 0.25, 0, 1, 0.5]], mask=mask)
    bad = np.zeros((32, 32), bool)
    bad[10 - 3:10 + 3, 20 - 3:20 + 3] = True                  # output (x, y) uses input columns x-2 .. x+3, rows y-2 .. y+3
    bad[20 - 3:20 + 3, 8 - 3:8 + 3] = True
    inside = np.zeros((32, 32), bool)
    inside[2:29, 2:29] = True                                 # 2 <= xin = x + 0.25 < w_in - 3 = 29
    assert np.array_equal(w[0] == 1, inside & ~bad)
    assert np.array_equal(np.isnan(o[0]), w[0] == 0)


def test_frames_use_their_own_transform():
    rng = np.random.default_rng(1)
    cube = rng.normal(50, 5, (3, 24, 28)).astype(np.float32)
    A = [[1, 0, 0, 0, 1, 0], [1, 0, 1, 0, 1, 0], [1, 0, 0.5, 0, 1, -0.25]]
    o, w = apref.resample_affine(cube, A, out_shape=(20, 22))
    assert o.shape == (3, 20, 22)
    for k in range(3):
        ok, wk = apref.resample_affine(cube[k], [A[k]], out_shape=(20, 22))
        assert np.array_equal(o[k], ok[0], equal_nan=True) and np.array_equal(w[k], wk[0])


def test_per_tile_affines_reduce_to_per_frame():
    rng = np.random.default_rng(2)
    cube = rng.normal(50, 5, (2, 40, 150)).astype(np.float32)
    A = np.array([[1.0, 0.001, 0.4, -0.001, 1.0, 0.3], [0.999, 0.0, -1.2, 0.0, 1.001, 0.8]])
    tiles = np.broadcast_to(A[:, None, None, :], (2, 3, 3, 6)).copy()
    o1, w1 = apref.resample_affine(cube, A)
    o2, w2 = apref.resample_affine(cube, tiles)
    assert np.array_equal(o1, o2, equal_nan=True) and np.array_equal(w1, w2)
    # a different transform in one tile changes that tile only
    tiles[0, 1, 2] = [1.0, 0.0, 2.0, 0.0, 1.0, 0.0]
    o3, _ = apref.resample_affine(cube, tiles)
    same = np.ones((40, 150), bool)
    same[16:32, 128:150] = False
    assert np.array_equal(o3[0][same], o1[0][same], equal_nan=True)
    yy, xx = np.mgrid[16:32, 128:145]
    assert np.array_equal(o3[0][yy, xx], cube[0][yy, xx + 2])
