"""cross-process determinism of the C++ drop-in: tests/cpp/adapter_smoke.bin and iekf_demo.bin run N times, every RESULT / STATE line must repeat bit for bit and the
timing rows must be consistent (total >= network inference > 0).   python tools/adapter_determinism_stress.py [N=100]   (150 runs: 0 differences, round 5)"""
import os, subprocess, sys, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuahn_vio_amd import synth, weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = tempfile.mkdtemp()
state = weights.synthetic_state(0)
weights.save_blob(d + "/main.hnw", state, dict(variant="prior3", mc_samples=16, dropout_p=0.05))
weights.save_blob(d + "/iter.hnw", state, dict(variant="prior1", mc_samples=8, dropout_p=0.1))
open(d + "/traced_model_3_blocks_using_prior.hnw", "wb").write(weights.pack_state_dict(state))
np.stack([synth.make_pair(95 + i)[0] for i in range(3)]).tofile(d + "/frames.u8")
np.stack([synth.make_pair(60 + i)[0] for i in range(14)]).tofile(d + "/frames14.u8")
env = {k: v for k, v in os.environ.items() if not k.startswith("HNET_")}
env2 = dict(os.environ, HNET_MC_SEED="99", HNET_DROPOUT_P="0.05")
ref = ref2 = None
bad = 0
for i in range(n):
    r = subprocess.run([ROOT + "/tests/cpp/adapter_smoke.bin", d + "/main.hnw", d + "/frames.u8", "3", "1", d + "/iter.hnw", "2"], capture_output=True, text=True, env=env)
    out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if ref is None: ref = out
    if out != ref or r.returncode: bad += 1; print("adapter_smoke run", i, "DIFFERS rc", r.returncode, flush=True)
    r = subprocess.run([ROOT + "/tests/cpp/iekf_demo.bin", d + "/traced_model_3_blocks_using_prior.hnw", d + "/frames14.u8", "14", "2", d + "/t.csv"], capture_output=True, text=True, env=env2)
    out = [l for l in r.stdout.splitlines() if l.startswith("STATE")]
    if ref2 is None: ref2 = out
    if out != ref2 or r.returncode: bad += 1; print("iekf_demo run", i, "DIFFERS rc", r.returncode, flush=True)
    rows = [l.split(",") for l in open(d + "/t.csv") if not l.startswith("#")]
    if not all(float(t[5]) >= float(t[3]) > 0 for t in rows): bad += 1; print("timing rows inconsistent in run", i, [t for t in rows if not float(t[5]) >= float(t[3]) > 0], flush=True)
print("runs", n, "bad", bad, "lines", len(ref), len(ref2))
