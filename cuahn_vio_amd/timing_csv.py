"""The reference's per-frame timing file — Python mirror of include/hnet_timing_csv.h.

Format: cuahn/src/core/VioManager.cpp:98 (header) and :304-311 (row: time stamp with 15 decimals, five millisecond
figures with 5 decimals, comma separated); read by ov_eval/src/utils/Loader.cpp:236-300."""
from __future__ import annotations

import os

HEADER = "# timestamp, loading image, state propagation, network inference, EKF update, total time"
COLUMNS = ("loading image", "state propagation", "network inference", "EKF update", "total time")


class TimingCsv:
    def __init__(self, path: str):
        if os.path.exists(path):                 # VioManager.cpp:87-90: an old file is deleted
            os.remove(path)
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)        # :92-93
        self._f = open(path, "a")
        self._f.write(HEADER + "\n")

    def append(self, timestamp_in_imu: float, load_img_ms: float, prop_ms: float, nn_ms: float, update_ms: float, total_ms: float) -> None:
        self._f.write(f"{timestamp_in_imu:.15f},{load_img_ms:.5f},{prop_ms:.5f},{nn_ms:.5f},{update_ms:.5f},{total_ms:.5f}\n")
        self._f.flush()

    def close(self) -> None:
        self._f.close()


def parse(path: str):
    """(names, rows) the way ov_eval's Loader::load_timing_flamegraph reads the file: category names from the '#' line
    (first field skipped), one list of floats per data line"""
    names, rows = [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("#"):
                names = [x for x in line.split(",") if x][1:]
                continue
            if line:
                rows.append([float(x) for x in line.split(",") if x])
    return names, rows
