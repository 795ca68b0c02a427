"""CPU: tests/cpp/rccl_gather_example.cpp — the C++ caller of the multi-GPU split — compiles and links against librccl + the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "rccl_gather_example.bin")


def build():
    lib_dir = os.path.join(ROOT, "cuahn_vio_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "rccl_gather_example.cpp"), "-o", EXE, "-L", lib_dir, "-lhnet_hip", f"-Wl,-rpath,{lib_dir}",
           "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return EXE


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/lib/librccl.so") and os.path.exists(os.path.join(ROOT, "cuahn_vio_amd", "libhnet_hip.so"))),
                    reason="needs the ROCm toolchain's librccl and the built libhnet_hip.so (python -c 'import __graft_entry__ as g; g.build()')")
def test_rccl_gather_example_compiles_and_links():
    exe = build()
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    used = {l.split()[-1].split("@")[0] for l in out.splitlines()}
    assert {"ncclAllGather", "ncclCommInitAll", "ncclGroupStart", "hnet_infer_batch_packed_device", "hnet_create"} <= used
