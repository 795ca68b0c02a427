"""the timing CSV writer against a file the REFERENCE itself recorded — tests/golden/traj_timing_head.txt is
the first 25 lines of cuahn_ros/ov_data/uzh_fpv/traj_timing.txt (written by VioManager.cpp:98,304-311).  Parsing it and
re-emitting every row through the C++ writer (include/hnet_timing_csv.h) and its Python mirror must reproduce the bytes."""
import os
import subprocess

from cuahn_vio_amd import timing_csv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "traj_timing_head.txt")


def test_python_writer_reproduces_the_reference_file(tmp_path):
    names, rows = timing_csv.parse(GOLD)
    assert [n.strip() for n in names] == list(timing_csv.COLUMNS) and len(rows) == 24 and all(len(r) == 6 for r in rows)
    out = tmp_path / "sub" / "timing.txt"
    w = timing_csv.TimingCsv(str(out))
    for r in rows:
        w.append(*r)
    w.close()
    assert out.read_bytes() == open(GOLD, "rb").read()
    # an existing file is replaced, not appended to (VioManager.cpp:87-90)
    w = timing_csv.TimingCsv(str(out))
    w.close()
    assert out.read_text() == timing_csv.HEADER + "\n"


def test_cpp_writer_reproduces_the_reference_file(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "timing_csv_check.bin")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "timing_csv_check.cpp"), "-o", exe], check=True)
    _names, rows = timing_csv.parse(GOLD)
    out = tmp_path / "timing.txt"
    out.write_text("stale contents\n")
    text = "".join(" ".join(repr(x) for x in r) + "\n" for r in rows)
    subprocess.run([exe, str(out)], input=text, text=True, check=True, timeout=30)
    assert out.read_bytes() == open(GOLD, "rb").read()
    # a fresh output directory is created like the reference does (boost::filesystem::create_directories, VioManager.cpp:92-93)
    deep = tmp_path / "new" / "dir" / "timing.txt"
    subprocess.run([exe, str(deep)], input=text, text=True, check=True, timeout=30)
    assert deep.read_bytes() == open(GOLD, "rb").read()
