"""The contig of stress_variants seed SEED, index K, through the host model of the wave decomposition (CPU): python tools/r06_repro.py SEED K"""
import sys, os, numpy as np, ctypes, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
import importlib.util
from tests.util import synthetic_contig
from tests import test_dpw_model as T
from pyrodigal_amd import benchdata
from oracle import oracle as orc
spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
models = [b for _, b in benchdata.load_model_set()]
seed = int(sys.argv[1]); want = int(sys.argv[2])
rng = np.random.default_rng(seed)
seqs = []
for k in range(300):
    L = int(rng.choice([300, 900, 2500, 7000, 20000, 60000], p=[0.1, 0.15, 0.25, 0.25, 0.2, 0.05]))
    gc = float(rng.uniform(0.22, 0.78))
    s = bytearray(mm.planted_genome(L, gc, seed * 1000 + k) if rng.random() < 0.7 else synthetic_contig(L, gc, seed * 1000 + k))
    if rng.random() < 0.2 and L > 1000:
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, L - 200)); n = int(rng.choice([1, 3, 49, 50, 200]))
            s[at:at + n] = b"N" * n
    seqs.append(bytes(s))
mi = int(rng.integers(0, 16))
print("model", mi, "contig", want, "len", len(seqs[want]))
open("gpurun_in_seq.bin", "wb").write(seqs[want]) if len(sys.argv) > 3 else None
subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", T.LIB, T.SRC], check=True)
Lb = ctypes.CDLL(T.LIB); vp = ctypes.c_void_p
Lb.dpw_model_run.restype = ctypes.c_int
Lb.dpw_model_run.argtypes = [ctypes.c_int] + [vp] * 9 + [ctypes.c_double] + [vp] * 5
nn, st = T.check(Lb, seqs[want], orc.Training(models[mi]), closed=True, is_meta=False, mask=True)
print("model agrees with the oracle on", nn, "nodes", st)
