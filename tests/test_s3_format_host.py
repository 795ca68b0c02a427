"""csrc/s3_format.h on the host: hnet_create packs the fp16-plane weights with these functions (HNET_PREC_F16X2).  The conversions must agree
bit for bit with the compiler's _Float16 conversions (what the device code uses), the split formats must keep their stated error bounds."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_fp16_conversions_and_split_bounds(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("hipcc")
    if not clang:
        pytest.skip("no HIP compiler on this machine (the header includes hip_runtime.h)")
    out = str(tmp_path / "s3_format_check.bin")
    subprocess.run([clang, "-O2", "-x", "hip", "--offload-host-only", "-I" + ROOT, "-I/opt/rocm/include", "-w",
                    os.path.join(ROOT, "tests", "cpp", "s3_format_check.cpp"), "-o", out], check=True, timeout=300)
    r = subprocess.run([out], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


def test_block41_tap_table_and_bordered_layout_compile_time_checks(tmp_path):
    """tests/cpp/b41_tap_check.cpp: static_asserts over csrc/kernels.h (the tap table of the fused block-4 kernel's phase 2, the bordered block_4_1 layout)"""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("hipcc")
    if not clang:
        pytest.skip("no HIP compiler on this machine (the header includes hip_runtime.h)")
    out = str(tmp_path / "b41_tap_check.bin")
    subprocess.run([clang, "-O1", "-std=c++17", "-x", "hip", "--offload-host-only", "-I" + ROOT, "-I/opt/rocm/include", "-w",
                    os.path.join(ROOT, "tests", "cpp", "b41_tap_check.cpp"), "-o", out], check=True, timeout=300)
    assert subprocess.run([out], timeout=60).returncode == 0
