"""CPU: include/hnet_ekf.h (the filter update that consumes the network output) against the numpy
restatement oracle/ekf_oracle.py of cuahn::UpdaterHNet::update."""
import os
import subprocess

import numpy as np

from oracle import ekf_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "ekf_check.bin")
NSTATE = 3 + 4 + 3 + 3 + 3 + 12 + 729


def _flat(s):
    return np.concatenate([s["p"], s["q"], s["v"], s["ba"], s["bg"], s["offset"].reshape(-1), s["cov"].reshape(-1)])


def _cases(n, rng):
    out = []
    for i in range(n):
        a = rng.standard_normal((27, 27))
        cov = a @ a.T * 1e-3 + np.eye(27) * 1e-4
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        st = dict(p=rng.standard_normal(3), q=q, v=rng.standard_normal(3), ba=rng.standard_normal(3) * 0.1, bg=rng.standard_normal(3) * 0.01,
                  offset=rng.standard_normal((4, 3)) * 0.02, cov=cov)
        b = rng.standard_normal((8, 2))
        ncov = np.zeros((8, 8))
        for c in range(4):                                   # block diagonal like the network's output
            m = rng.standard_normal((2, 2))
            ncov[2 * c:2 * c + 2, 2 * c:2 * c + 2] = m @ m.T + np.eye(2) * 0.5
        mean = rng.standard_normal(8) * 5.0
        prop = st["offset"][:, :2].reshape(8).copy()
        out.append((st, mean, ncov, prop, 10.0 if i % 3 else 1.0, i % 2 == 0))
    # a zero innovation / zero rotation update (the reference divides 0/0 there)
    st = dict(out[0][0])
    st["cov"] = np.eye(27) * 1e-3
    out.append((st, out[0][3] * ekf_oracle.F_PIX, out[0][2], out[0][3], 10.0, True))
    return out


def test_ekf_update_matches_numpy_restatement(tmp_path):
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "ekf_check.cpp"), "-o", BIN], check=True)
    cases = _cases(12, np.random.default_rng(7))
    blob = [np.array([float(len(cases))])]
    for st, mean, ncov, prop, k, upd in cases:
        blob += [_flat(st), mean, ncov.reshape(-1), prop, np.array([k, 1.0 if upd else 0.0])]
    fin, fout = tmp_path / "in.f64", tmp_path / "out.f64"
    np.concatenate(blob).astype("<f8").tofile(fin)
    subprocess.run([BIN, str(fin), str(fout)], check=True, timeout=60)
    got = np.fromfile(fout, "<f8").reshape(len(cases), 1 + 2 * NSTATE)
    for (st, mean, ncov, prop, k, upd), g in zip(cases, got):
        assert g[0] == 1.0
        ref = ekf_oracle.update(st, mean, ncov, prop, k, upd)
        r1 = _flat(ref)
        assert np.abs(g[1:1 + NSTATE] - r1).max() < 1e-11 * max(1.0, np.abs(r1).max())
        r2 = _flat(ekf_oracle.reset_4pt_offset(ref))
        assert np.abs(g[1 + NSTATE:] - r2).max() < 1e-11 * max(1.0, np.abs(r2).max())
        # sanity of the restated update itself: the posterior offsets move towards the measurement
        if upd:
            before = np.abs(mean / ekf_oracle.F_PIX - prop)
            after = np.abs(mean / ekf_oracle.F_PIX - ref["offset"][:, :2].reshape(8))
            assert after.sum() < before.sum() + 1e-12
