"""The header-only C++ adapter include/HomographyNet.h (class surface of the reference pytorch::HomographyNet):
compiled here against tiny cv::Mat / Eigen::Matrix stand-ins (tests/cpp/shims.h; neither library is in the
image), run on the GPU box and compared with the batch entry point."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "adapter_smoke.bin")


def _build(src="adapter_smoke.cpp", out=None, hip_headers=False):
    from cuahn_vio_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    rocm_lib = "/opt/rocm/lib"
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include")] + \
          (["-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-Wno-unused-result"] if hip_headers else []) + [      # (a host program that calls the HIP runtime API itself)
           os.path.join(ROOT, "tests", "cpp", src), "-o", out or BIN,
           "-L", os.path.join(ROOT, "cuahn_vio_amd"), "-lhnet_hip", f"-Wl,-rpath,{os.path.join(ROOT, 'cuahn_vio_amd')}",
           "-L", rocm_lib, "-lamdhip64", f"-Wl,-rpath,{rocm_lib}"]
    subprocess.run(cmd, check=True)


def test_adapter_compiles_and_links():
    _build()
    assert os.path.exists(BIN)
    out = subprocess.run(["nm", "-D", "--undefined-only", BIN], capture_output=True, text=True, check=True).stdout
    used = {l.split()[-1] for l in out.splitlines() if "hnet_" in l}
    # the adapter reaches the kernels only through the C ABI
    assert {"hnet_create", "hnet_push_image", "hnet_infer", "hnet_destroy", "hnet_latest_time"} <= used
    assert not any("torch" in l or "c10" in l for l in out.splitlines())


@pytest.mark.gpu
def test_adapter_runs_like_the_reference_call_sites(blob, tmp_path):
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HnetEngine
    _build()
    wpath = tmp_path / "traced_model_3_blocks_using_prior_showError.hnw"   # the launch default's name (uzhfpv.launch:58)
    wpath.write_bytes(blob)
    frames = np.stack([synth.make_pair(80 + i)[0] for i in range(3)])
    fpath = tmp_path / "frames.u8"
    frames.tofile(fpath)
    env = dict(os.environ, HNET_MC_SEED="1234", HNET_DROPOUT_P="0.05")
    r = subprocess.run([BIN, str(wpath), str(fpath), "3", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "HNet cannot inference! Only has one image!" in r.stdout          # first frame (HomographyNet.cpp:155-158)
    res = [l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert [int(x[1]) for x in res] == [1, 2]
    eng = HnetEngine(blob, variant="prior3", mc_samples=16, dropout_p=0.05, mc_seed=1234, max_batch=1)
    prior = np.array([[1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75]], np.float32)
    for x in res:
        k = int(x[1])
        vals = np.array([float(v) for v in x[2:]], np.float64)
        m, c = eng.infer_batch(frames[k - 1][None], frames[k][None], prior, pair_seq0=k - 1)
        assert np.array_equal(vals[:8].astype(np.float32), m[0])
        assert np.array_equal(vals[8:].astype(np.float32).reshape(8, 8), c[0])
    eng.close()


@pytest.mark.gpu
def test_adapter_routes_iterations_to_the_iterative_model(blob, tmp_path):
    """network_model_iterative_path (HomographyNet.cpp:20-24,104-124,209-219): a prior-3 main model and a prior-1 ITERATIVE model (another variant,
    as the reference allows: it is a separate traced file); iteration 0 of every frame must be the prior-3 engine's answer, iteration 1 the prior-1
    engine's, bit for bit, with one shared MC-dropout sequence count"""
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HnetEngine
    _build()
    wmain = tmp_path / "traced_model_3_blocks_using_prior.hnw"
    witer = tmp_path / "traced_model_1_blocks_using_prior.hnw"
    wmain.write_bytes(blob)
    witer.write_bytes(blob)
    frames = np.stack([synth.make_pair(90 + i)[0] for i in range(4)])
    fpath = tmp_path / "frames.u8"
    frames.tofile(fpath)
    env = dict(os.environ, HNET_MC_SEED="4321", HNET_DROPOUT_P="0.05", HNET_BLOCKS_TO_RUN="3", HNET_ITER_BLOCKS_TO_RUN="1")
    r = subprocess.run([BIN, str(wmain), str(fpath), "4", "1", str(witer), "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "IEKF! Load the Network for Iteration!" in r.stdout
    res = [l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert [(int(x[1]), int(x[2])) for x in res] == [(k, it) for k in (1, 2, 3) for it in (0, 1)]
    engs = {0: HnetEngine(blob, variant="prior3", mc_samples=16, dropout_p=0.05, mc_seed=4321, max_batch=1),
            1: HnetEngine(blob, variant="prior1", mc_samples=16, dropout_p=0.05, mc_seed=4321, max_batch=1)}
    prior = np.array([[1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75]], np.float32)
    for seq, x in enumerate(res):                    # one sequence count for both models
        k, it = int(x[1]), int(x[2])
        vals = np.array([float(v) for v in x[3:]], np.float64)
        m, c = engs[it].infer_batch(frames[k - 1][None], frames[k][None], prior, pair_seq0=seq)
        assert np.array_equal(vals[:8].astype(np.float32), m[0]), (k, it)
        assert np.array_equal(vals[8:].astype(np.float32).reshape(8, 8), c[0]), (k, it)
    m3, _ = engs[0].infer_batch(frames[0][None], frames[1][None], prior, pair_seq0=1)
    m1, _ = engs[1].infer_batch(frames[0][None], frames[1][None], prior, pair_seq0=1)
    assert not np.array_equal(m3, m1)                # the two variants really differ
    for e in engs.values():
        e.close()


@pytest.mark.gpu
def test_adapter_takes_the_variant_from_the_weight_files(state, tmp_path):
    """VERDICT r4 missing 3: the reference freezes blocks_to_run, N, p and the error-map twin into the traced .pt it loads (trace_model.py:16,36-46;
    HomographyNet.cpp:81-124).  Two HNETW001 files written with their variant record (python -m cuahn_vio_amd.weights --variant ...): a prior-3 / N = 16
    main model and a prior-1 / N = 8 / p = 0.1 iterative model, and an environment WITHOUT any HNET_* variable: every call must be the answer of the engine
    configured explicitly that way, bit for bit."""
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import HnetEngine
    _build()
    wmain = tmp_path / "main.hnw"
    witer = tmp_path / "iter.hnw"
    weights.save_blob(str(wmain), state, dict(variant="prior3", mc_samples=16, dropout_p=0.05))
    weights.save_blob(str(witer), state, dict(variant="prior1", mc_samples=8, dropout_p=0.1))
    frames = np.stack([synth.make_pair(95 + i)[0] for i in range(3)])
    fpath = tmp_path / "frames.u8"
    frames.tofile(fpath)
    env = {k: v for k, v in os.environ.items() if not k.startswith("HNET_")}
    r = subprocess.run([BIN, str(wmain), str(fpath), "3", "1", str(witer), "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "main model: EKF prior, blocks_to_run 3, MC-dropout N = 16, p = 0.05" in r.stdout
    assert "iterative model: EKF prior, blocks_to_run 1, MC-dropout N = 8, p = 0.1" in r.stdout
    res = [l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert [(int(x[1]), int(x[2])) for x in res] == [(k, it) for k in (1, 2) for it in (0, 1)]
    blob = weights.pack_state_dict(state)
    engs = {0: HnetEngine(blob, variant="prior3", mc_samples=16, dropout_p=0.05, mc_seed=0, max_batch=1),
            1: HnetEngine(blob, variant="prior1", mc_samples=8, dropout_p=0.1, mc_seed=0, max_batch=1)}
    prior = np.array([[1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75]], np.float32)
    for seq, x in enumerate(res):
        k, it = int(x[1]), int(x[2])
        vals = np.array([float(v) for v in x[3:]], np.float64)
        m, c = engs[it].infer_batch(frames[k - 1][None], frames[k][None], prior, pair_seq0=seq)
        assert np.array_equal(vals[:8].astype(np.float32), m[0]), (k, it)
        assert np.array_equal(vals[8:].astype(np.float32).reshape(8, 8), c[0]), (k, it)
    for e in engs.values():
        e.close()
    # the Python mirror reads the same record
    e = HnetEngine(str(witer), variant=None, mc_samples=None, dropout_p=None, emit_error_map=None, max_batch=1)
    u = e.config()
    assert (e.variant, u.mc_samples, u.emit_error_map) == ("prior1", 8, 0) and abs(u.dropout_p - 0.1) < 1e-7
    e.close()


IEKF_BIN = os.path.join(ROOT, "tests", "cpp", "iekf_demo.bin")


def test_iekf_demo_compiles():
    _build("iekf_demo.cpp", IEKF_BIN)
    assert os.path.exists(IEKF_BIN)


@pytest.mark.gpu
def test_adapter_feeds_the_iterated_ekf_update(blob, tmp_path):
    """drop-in end to end: C++ adapter + include/hnet_ekf.h run the reference's per-frame loop
    (prior from the state, network_inference, UpdaterHNet::update, twice per frame, offsets reset); the same loop in
    Python (batch entry point + numpy restatement of the update) must land on the same filter state"""
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HnetEngine
    from oracle import ekf_oracle
    _build("iekf_demo.cpp", IEKF_BIN)
    wpath = tmp_path / "traced_model_3_blocks_using_prior.hnw"
    wpath.write_bytes(blob)
    n_frames = 14                                   # the filter only uses the network from the 11th image on (VioManager.cpp:257)
    frames = np.stack([synth.make_pair(60 + i)[0] for i in range(n_frames)])
    fpath = tmp_path / "frames.u8"
    frames.tofile(fpath)
    env = dict(os.environ, HNET_MC_SEED="99", HNET_DROPOUT_P="0.05")
    csv_path = tmp_path / "traj_timing.txt"
    r = subprocess.run([IEKF_BIN, str(wpath), str(fpath), str(n_frames), "2", str(csv_path)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    # the timing file in the reference's format (VioManager.cpp:98,304-311), parsed the way ov_eval does
    from cuahn_vio_amd import timing_csv
    names, trows = timing_csv.parse(str(csv_path))
    assert [x.strip() for x in names] == list(timing_csv.COLUMNS) and len(trows) == n_frames - 1
    assert all(len(t) == 6 and t[5] >= t[3] > 0.0 for t in trows)           # total >= network inference > 0 ms
    rows = [l.split() for l in r.stdout.splitlines() if l.startswith("STATE")]
    assert [int(x[1]) for x in rows] == list(range(1, n_frames))
    assert [int(x[2]) for x in rows] == [2 if k + 1 > 10 else 0 for k in range(1, n_frames)]     # img_counter = k + 1 > 10

    eng = HnetEngine(blob, variant="prior3", mc_samples=16, dropout_p=0.05, mc_seed=99, max_batch=1)
    c_R_i = np.array([[-0.027256691772188965, -0.9996260641688061, 0.0021919370477445077],
                      [-0.7139206120417471, 0.017931469899155242, -0.6999970157716363],
                      [0.6996959571525168, -0.020644471939022302, -0.714142404092339]])
    t_i2c = np.array([0.0070507, 0.0240435, 0.0057731])
    w_hat, a_hat = np.array([0.02, -0.03, 0.05]), np.array([0.1, -0.05, 9.81])
    Q = ekf_oracle.noise_q(0.00559017, 0.01118034, 8.94427e-04, 0.04472136)
    st = dict(p=np.array([0.0, 0.0, 1.5]), q=np.array([1.0, 0, 0, 0]), v=np.array([0.4, -0.2, 0.0]), ba=np.zeros(3), bg=np.zeros(3),
              offset=np.zeros((4, 3)), cov=np.diag([1e-4] * 15 + [0.0] * 12))
    seq = 0
    for x in rows:
        k = int(x[1])
        for _ in range(16):                          # Propagator::propagate_with_imu's loop: Jacobians at the old state, mean, covariance
            F, Fw = ekf_oracle.jacobians(st, c_R_i, t_i2c, 0.002, w_hat)
            cov = ekf_oracle.propagate_cov(st["cov"], F, Fw, Q)
            st = ekf_oracle.propagate_mean(st, c_R_i, t_i2c, 0.002, w_hat, a_hat)
            st["cov"] = cov
        for it in range(2):                          # the network runs in every iteration (the mask sequence number advances), the update is gated
            prop = st["offset"][:, :2].reshape(8).copy()
            m, c = eng.infer_batch(frames[k - 1][None], frames[k][None], (prop * ekf_oracle.F_PIX).astype(np.float32)[None], pair_seq0=seq)
            seq += 1
            if k + 1 > 10:
                st = ekf_oracle.update(st, m[0].astype(np.float64), c[0].astype(np.float64), prop, 10.0, it == 0)
        st = ekf_oracle.reset_4pt_offset(st)
        got = np.array([float(v) for v in x[3:]])
        want = np.concatenate([st["p"], st["q"], st["v"], np.diag(st["cov"])[:15], [st["cov"][0, 7], st["cov"][4, 13]]])
        assert np.abs(got - want).max() < 1e-9, (k, np.abs(got - want).max())
    assert np.abs(st["p"]).max() > 0 and (np.diag(st["cov"])[:15] > 0).all()
    eng.close()


LAT_BIN = os.path.join(ROOT, "tests", "cpp", "adapter_latency.bin")


@pytest.mark.gpu
def test_adapter_latency_from_cpp(blob, tmp_path):
    """per-frame wall clock of the drop-in class measured in C++ (no Python in the loop); prints the figure DESIGN.md quotes"""
    _build("adapter_latency.cpp", LAT_BIN)
    wpath = tmp_path / "w.hnw"
    wpath.write_bytes(blob)
    env = dict(os.environ, HNET_MC_SAMPLES="32", HNET_BLOCKS_TO_RUN="3")
    for use_prior in ("0", "1"):
        r = subprocess.run([LAT_BIN, str(wpath), "220", use_prior], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        line = [l for l in r.stderr.splitlines() if l.startswith("LATENCY")][0]
        print(("full" if use_prior == "0" else "prior-3"), line)
        assert float(line.split()[2]) < 2.0


@pytest.mark.gpu
def test_cpp_group_example_runs(blob, tmp_path):
    """tests/cpp/group_example.cpp: a C++ caller of hnet_group (round 6) - twelve independent steps round-robin on three contexts, a consumer stream joined behind
    them, every step bit-identical to a single context's; with --time the pairs/s of the same loop (what bench.py --contexts measures through the Python mirror)"""
    exe = os.path.join(ROOT, "tests", "cpp", "group_example.bin")
    _build("group_example.cpp", exe, hip_headers=True)
    w = tmp_path / "w.hnw"
    w.write_bytes(blob)
    r = subprocess.run([exe, str(w), "3", "32", "--time"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GROUP_OK members=3 batch=32" in r.stdout and "GROUP_TIME" in r.stdout
    print(r.stdout[-300:])
