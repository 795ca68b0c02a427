#!/usr/bin/env python3
"""VGPR / spill / scratch figures of the kernels of one gfx950 code object (llvm-readelf --notes), filtered by a substring of the demangled name.
   python tools/kregs.py /tmp/x.co lean_kernel
   (the code object of an object file: /opt/rocm/lib/llvm/bin/llvm-objdump --offloading file.o writes it next to the file)"""
import re, subprocess, sys
t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", sys.argv[1]], capture_output=True, text=True).stdout
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for e in re.split(r"\n\s+- \.agpr_count", t)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", e) or [None, "?"])[1]
    d = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if pat in d:
        print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}  {d[:150]}")
