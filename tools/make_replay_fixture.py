#!/usr/bin/env python3
"""Build-container tool: down-samples one UZH-FPV ground-truth trajectory the reference ships
(/root/reference/cuahn_ros/ov_data/uzh_fpv/<seq>_snapdragon_with_gt.txt: `t tx ty tz qx qy qz qw`, ~500 Hz, body pose in
the world) to the camera rate (30 Hz) and writes the KB-scale pose fixture tests/golden/replay_<seq>.npz that
cuahn_vio_amd/replay.py renders frame pairs from (BASELINE.json config 5; SURVEY.md §8d "commit only a down-sampled pose
fixture").  The fixture holds DATA only: times, positions, quaternions, and the camera constants of the launch file
(cuahn/launch/uzhfpv.launch:75-90, sensor_config 1).  The real image bags are not available (uzhfpv.launch:9 points to the
author's disk).

  python tools/make_replay_fixture.py [indoor_forward_7] [--frames 640]
"""
import argparse
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cuahn_ros/ov_data/uzh_fpv"

# cuahn/launch/uzhfpv.launch:81-90 (indoor, 45 degree downward facing camera, sensor_config 1)
CAM0_WH = (640, 480)                                                                                  # :75
CAM0_K = (275.46015578667294, 274.9948095922592, 315.958384100568, 242.7123497822731)                 # :82
CAM0_D = (-6.545154718304953e-06, -0.010379525898159981, 0.014935312423953146, -0.005639061406567785)  # :83
T_I_TO_C = ((-0.027256691772188965, -0.9996260641688061, 0.0021919370477445077, 0.02422852666805565),  # :84-91
            (-0.7139206120417471, 0.017931469899155242, -0.6999970157716363, 0.008974432843748055),
            (0.6996959571525168, -0.020644471939022302, -0.714142404092339, -0.000638971731537894),
            (0.0, 0.0, 0.0, 1.0))
INIT_HEIGHT = 0.1                                                                                     # :66 init_height


def slerp(q0, q1, a):
    d = float(np.dot(q0, q1))
    if d < 0:
        q1, d = -q1, -d
    if d > 0.9995:
        q = q0 + a * (q1 - q0)
    else:
        th = np.arccos(d)
        q = (np.sin((1 - a) * th) * q0 + np.sin(a * th) * q1) / np.sin(th)
    return q / np.linalg.norm(q)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seq", nargs="?", default="indoor_forward_7")
    ap.add_argument("--frames", type=int, default=640)
    ap.add_argument("--rate", type=float, default=30.0)
    ap.add_argument("--skip", type=float, default=8.0, help="seconds skipped at the start (the drone sits on the ground)")
    a = ap.parse_args()
    src = os.path.join(REF, a.seq + "_snapdragon_with_gt.txt")
    g = np.loadtxt(src)
    t = g[:, 0] - g[0, 0]
    ts = a.skip + np.arange(a.frames) / a.rate
    assert ts[-1] < t[-1], "sequence too short"
    idx = np.searchsorted(t, ts, side="right") - 1
    al = (ts - t[idx]) / (t[idx + 1] - t[idx])
    p = g[idx, 1:4] * (1 - al[:, None]) + g[idx + 1, 1:4] * al[:, None]
    q = np.stack([slerp(g[i, 4:8], g[i + 1, 4:8], x) for i, x in zip(idx, al)])      # (qx, qy, qz, qw), Hamilton, body -> world
    out = os.path.join(ROOT, "tests", "golden", f"replay_{a.seq}.npz")
    np.savez_compressed(out, name=a.seq, source=os.path.basename(src), rate_hz=a.rate, t=ts.astype(np.float64), p=p.astype(np.float64),
                        q_xyzw=q.astype(np.float64), floor_z=np.float64(g[:, 3].min() - INIT_HEIGHT), cam0_wh=np.array(CAM0_WH),
                        cam0_k=np.array(CAM0_K), cam0_d=np.array(CAM0_D), T_ItoC=np.array(T_I_TO_C))
    print(out, os.path.getsize(out), "bytes;", a.frames, "poses at", a.rate, "Hz; height above the floor",
          float((p[:, 2] - (g[:, 3].min() - INIT_HEIGHT)).min()), "..", float((p[:, 2] - (g[:, 3].min() - INIT_HEIGHT)).max()), "m")


if __name__ == "__main__":
    main()
