"""How often does the device's double sin / cos differ from glibc's on the float-valued arguments MicroFacet.cpp:220-223 produces? (dev tool)"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnxraytracer_amd as gx
gx.init(0)
libm = C.CDLL("libm.so.6")
rng = np.random.default_rng(1)
u = rng.random(2_000_000, dtype=np.float32)
phi = (6.28318530718 * u.astype(np.float64)).astype(np.float32)
for fn in ("sin", "cos", "sqrt", "tan"):
    f = getattr(libm, fn); f.restype = C.c_double; f.argtypes = [C.c_double]
    x = phi if fn != "sqrt" else (u / (1 - u)).astype(np.float32)
    x = x[:300000]
    ref = np.array([f(float(v)) for v in x], np.float64)
    dev = gx.eval_libm_f64(fn, x)
    ne = dev.view(np.uint64) != ref.view(np.uint64)
    print(fn, "mismatching doubles:", int(ne.sum()), "of", len(x), " as float32:", int((dev.astype(np.float32) != ref.astype(np.float32)).sum()))
