"""Image pre-processing ahead of load_current_img: map construction and the remap kernel.
The kernel is bit-exact against the numpy restatement oracle/undistort_oracle.py; parity with cv::remap itself cannot be pinned here (OpenCV is
absent from the image): the deviation class is stated in include/hnet.h, and an end-to-end distort -> undistort property test bounds it."""
import numpy as np
import pytest

from oracle import undistort_oracle as uo

# uzhfpv.launch:75-82 (sensor_config 1): 640 x 480 fisheye
K_UZH = (275.46015578667294, 274.9948095922592, 315.958384100568, 242.7123497822731)
D_UZH = (-6.545154718304953e-06, -0.010379525898159981, 0.014935312423953146, -0.005639061406567785)


def _raw(seed, rows=480, cols=640):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols]
    img = 128 + 60 * np.sin(x / 23.0 + seed) * np.cos(y / 17.0) + 40 * np.sin((x + 2 * y) / 41.0) + rng.integers(-12, 13, (rows, cols))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_map_construction_properties():
    mx, my = uo.build_maps(K_UZH, D_UZH, fisheye=True)
    # the centre of the virtual camera looks along the optical axis: it samples the raw principal point
    c = (111, 159), (112, 160)
    assert abs(0.5 * (mx[c[0]] + mx[c[1]]) - K_UZH[2]) < 1e-3 and abs(0.5 * (my[c[0]] + my[c[1]]) - K_UZH[3]) < 1e-3
    # equidistant model at the horizontal edge: x = +-1 (45 deg) -> theta = pi/4
    th = np.pi / 4
    thd = th * (1 + D_UZH[0] * th ** 2 + D_UZH[1] * th ** 4 + D_UZH[2] * th ** 6 + D_UZH[3] * th ** 8)
    row = np.float64(mx[111]) * 0.5 + np.float64(mx[112]) * 0.5          # y ~ 0
    assert abs(row[319] - (K_UZH[2] + K_UZH[0] * thd)) < 0.05 and abs(row[0] - (K_UZH[2] - K_UZH[0] * thd)) < 0.05
    # no distortion and K equal to the virtual camera: the identity resampling
    ident = uo.build_maps((uo.F_VIRTUAL, uo.F_VIRTUAL, uo.CX, uo.CY), (0, 0, 0, 0), fisheye=False)
    v, u = np.mgrid[0:224, 0:320]
    assert np.abs(ident[0] - u).max() < 1e-4 and np.abs(ident[1] - v).max() < 1e-4
    img = _raw(1, 224, 320)
    assert np.array_equal(uo.remap(img, *ident), img)
    # a half-pixel shift averages neighbours (round half up in the integer blend), zeros enter at the border
    sh = uo.remap(img, ident[0] + np.float32(0.5), ident[1])
    want = ((img[:, :-1].astype(np.int64) + img[:, 1:] + 1) >> 1).astype(np.uint8)
    assert np.array_equal(sh[:, :-1], want) and np.array_equal(sh[:, -1], ((img[:, -1].astype(np.int64) * 512 + 512) >> 10).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("fisheye", [True, False])
def test_undistort_kernel_matches_oracle(blob, fisheye):
    from cuahn_vio_amd.homography_net import HnetEngine
    eng = HnetEngine(blob, variant="prior1", mc_samples=4, dropout_p=0.0, max_batch=1)
    d = D_UZH if fisheye else (-0.28, 0.07, 1e-3, -5e-4)
    eng.set_camera(K_UZH, d, 480, 640, fisheye=fisheye)
    mx, my = eng.get_undistort_maps()                     # built on the host in C++ (libm) ...
    rx, ry = uo.build_maps(K_UZH, d, fisheye=fisheye)     # ... against numpy: float32 roundings of doubles that agree to ~1e-13
    assert np.abs(mx - rx).max() < 1e-4 and np.abs(my - ry).max() < 1e-4
    assert (mx != rx).mean() < 1e-3 and (my != ry).mean() < 1e-3
    for seed in (3, 4):
        raw = _raw(seed)
        assert np.array_equal(eng.op_undistort(raw), uo.remap(raw, mx, my))
    # maps that leave the image, NaN maps: zeros, never a fault
    bad_x = mx.copy(); bad_x[:40] += 5000.0; bad_x[40:60] = np.nan
    eng.set_undistort_maps(bad_x, my, 480, 640)
    raw = _raw(5)
    out = eng.op_undistort(raw)
    assert np.array_equal(out, uo.remap(raw, bad_x, my)) and (out[:60] == 0).all()
    # raw frames through the pre-processing == the remapped frames through load_current_img
    eng.set_camera(K_UZH, d, 480, 640, fisheye=fisheye)
    raws = [_raw(10 + i) for i in range(2)]
    e2 = HnetEngine(blob, variant="prior1", mc_samples=4, dropout_p=0.0, max_batch=1)
    prior = np.zeros(8)
    for i, r in enumerate(raws):
        eng.push_raw_image(r, float(i))
        std = uo.remap(r, mx, my)
        e2._L.hnet_push_image(e2.handle, std.ctypes.data, 224, 320, 320, float(i))
    import ctypes as C
    outs = []
    for e in (eng, e2):
        mean, cov = np.zeros(8, np.float32), np.zeros((8, 8), np.float32)
        pr = (C.c_double * 8)(*prior)
        rc = e._L.hnet_infer(e.handle, pr, 0, mean.ctypes.data_as(C.POINTER(C.c_float)), cov.ctypes.data_as(C.POINTER(C.c_float)), None)
        assert rc == 0
        outs.append((mean, cov))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception):
        eng.push_raw_image(_raw(1, 100, 100), 9.0)          # wrong raw size
    eng.close(); e2.close()


# ---- an end-to-end property that does not use the restated map formulas -------------------------------------------
def _scene(x, y):
    """an analytic grey-level pattern on the z = 1 plane of the virtual (undistorted) camera: a soft checkerboard, 8-bit range"""
    return 128.0 + 55.0 * np.sin(9.0 * x) * np.sin(9.0 * y) + 40.0 * np.sin(2.3 * x - 1.1 * y + 0.4) + 15.0 * np.cos(5.0 * x + 3.0 * y)


def _fisheye_photo_of_scene(k, d, rows=480, cols=640):
    """what a fisheye camera (k, d: the equidistant model, forward direction only) records of _scene: every RAW pixel is back-projected by
    inverting theta_d = theta (1 + d1 theta^2 + ...) with Newton's method - the opposite direction of the undistortion map, written
    independently of include/hnet.h / oracle/undistort_oracle.py - and the scene is evaluated at that ray.  Quantised to 8 bits."""
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
    xd, yd = (u - k[2]) / k[0], (v - k[3]) / k[1]
    thd = np.sqrt(xd * xd + yd * yd)
    th = thd.copy()
    for _ in range(20):
        t2 = th * th
        f = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3])))) - thd
        fp = 1 + t2 * (3 * d[0] + t2 * (5 * d[1] + t2 * (7 * d[2] + t2 * 9 * d[3])))
        th = th - f / fp
    r = np.tan(np.clip(th, 0, 1.55))
    sc = np.where(thd > 1e-12, r / np.maximum(thd, 1e-12), 1.0)
    return np.clip(np.floor(_scene(xd * sc, yd * sc) + 0.5), 0, 255).astype(np.uint8)


def _ideal_undistorted():
    f = (320 - 1.0) / 2.0 / np.tan(np.pi / 4)                 # the 90-degree virtual camera of CamBase.h:166-169
    v, u = np.mgrid[0:224, 0:320].astype(np.float64)
    return _scene((u - (320 - 1.0) / 2.0) / f, (v - (224 - 1.0) / 2.0) / f)


def test_undistorting_a_fisheye_photo_recovers_the_scene_cpu():
    """numpy restatement: remap(fisheye photo) == the scene as the virtual camera sees it, to <= 1 grey level RMS (8-bit quantisation of the
    photo 0.29 + of the output 0.29 + the 1/32-px position table + bilinear interpolation of a smooth pattern)"""
    raw = _fisheye_photo_of_scene(K_UZH, D_UZH)
    mx, my = uo.build_maps(K_UZH, D_UZH, fisheye=True)
    got = uo.remap(raw, mx, my).astype(np.float64)
    err = got - _ideal_undistorted()
    assert np.sqrt((err ** 2).mean()) < 1.0 and np.abs(err).max() < 4.0


@pytest.mark.gpu
def test_undistorting_a_fisheye_photo_recovers_the_scene_gpu(blob):
    """the same through hnet_set_camera + undistort_kernel: the GPU pre-processing inverts an independently simulated fisheye camera"""
    from cuahn_vio_amd.homography_net import HnetEngine
    eng = HnetEngine(blob, variant="prior1", mc_samples=4, dropout_p=0.0, max_batch=1)
    eng.set_camera(K_UZH, D_UZH, 480, 640, fisheye=True)
    got = eng.op_undistort(_fisheye_photo_of_scene(K_UZH, D_UZH)).astype(np.float64)
    eng.close()
    err = got - _ideal_undistorted()
    print(f"undistort(fisheye photo) vs the analytic scene: RMS {np.sqrt((err ** 2).mean()):.3f}, max {np.abs(err).max():.2f} grey levels")
    assert np.sqrt((err ** 2).mean()) < 1.0 and np.abs(err).max() < 4.0
