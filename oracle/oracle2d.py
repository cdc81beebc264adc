"""ctypes bindings of the 2D CPU oracle (oracle/fluid2d_oracle.c) and, where it has been built, of
the reference's own 2D solver (oracle/_ref/libref2d.so, built from /root/reference by `make -C oracle ref`).

TEST INFRASTRUCTURE: only tests/, scripts/make_golden_2d.py and the benchmark's cpu_baseline leg may
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DENSITY, PRESSURE, DIVERGENCE, VORTICITY, U, V, FX, FY = range(8)
FIELD_NAMES = ["density", "pressure", "divergence", "vorticity", "u", "v", "fx", "fy"]
# stage numbers shared by ref2d_stage() and o2_stage()
(APPLY_SOURCE, ADD_BUOYANCY, ADD_WIND, ADD_FORCE, SOLVE_PRESSURE, ADVECT_VALUES, SET_NEUMANN, SET_ZERO,
 COMPUTE_DIVERGENCE, COPY_BORDER, SOLVE_POISSON, CORRECT_VELOCITY, COMPUTE_VORTICITY, RESET_FORCE,
 SOLVE_PRESSURE_RB, SOLVE_POISSON_RB) = range(16)

REF_PATH = os.path.join(_HERE, "_ref", "libref2d.so")


def reference_available():
    return os.path.exists(REF_PATH)


class _Sim2D:
    prefix = None
    lib = None

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def __init__(self, rx, ry):
        self.rx, self.ry = rx, ry
        f = self._f("create"); f.restype = C.c_void_p; f.argtypes = [C.c_int, C.c_int]
        self.h = C.c_void_p(f(rx, ry))
        for name, res, args in (("destroy", None, [C.c_void_p]), ("size", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
                                ("get", None, [C.c_void_p, C.c_int, C.c_void_p]), ("set", None, [C.c_void_p, C.c_int, C.c_void_p]),
                                ("params", None, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
                                ("dt", C.c_double, [C.c_void_p]), ("time", C.c_double, [C.c_void_p]),
                                ("step", C.c_ulong, [C.c_void_p]), ("reset", None, [C.c_void_p]),
                                ("stage", None, [C.c_void_p, C.c_int])):
            fn = self._f(name); fn.restype = res; fn.argtypes = args

    def __del__(self):
        if getattr(self, "h", None):
            self._f("destroy")(self.h)
            self.h = None

    def shape(self, which):
        return (self._f("size")(self.h, which, 0), self._f("size")(self.h, which, 1))

    def get(self, which):
        """Field as an array indexed [x, y] (the reference's Array2T layout, y fastest)."""
        a = np.empty(self.shape(which), dtype=np.float64)
        self._get(which, a)
        return a

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.shape(which)
        self._set(which, a)

    def params(self, dt=0.0, acc=0.0, iters=0, wind=False, maccormack=True):
        self._f("params")(self.h, dt, acc, iters, int(wind), int(maccormack))

    def stage(self, op):
        self._f("stage")(self.h, op)

    def reset(self):
        self._f("reset")(self.h)

    @property
    def dt(self): return self._f("dt")(self.h)
    @property
    def time(self): return self._f("time")(self.h)
    @property
    def step(self): return int(self._f("step")(self.h))

    def state(self):
        return {FIELD_NAMES[k]: self.get(k) for k in range(6)}


class Reference2D(_Sim2D):
    """The reference's own FluidSim (3_fluid/FluidSim.{h,cpp}), compiled from its sources."""
    prefix = "ref2d_"

    def __init__(self, rx=128, ry=192):
        if Reference2D.lib is None:
            Reference2D.lib = C.CDLL(REF_PATH)
        super().__init__(rx, ry)
        self.lib.ref2d_advance.argtypes = [C.c_void_p]

    def _get(self, which, a): self.lib.ref2d_get(self.h, which, a.ctypes.data_as(C.c_void_p))
    def _set(self, which, a): self.lib.ref2d_set(self.h, which, a.ctypes.data_as(C.c_void_p))

    def advance(self):
        self.lib.ref2d_advance(self.h)


class Oracle2D(_Sim2D):
    """The plain-C restatement (oracle/fluid2d_oracle.c)."""
    prefix = "o2_"

    def __init__(self, rx=128, ry=192):
        if Oracle2D.lib is None:
            path = os.path.join(_HERE, "libfluid2d_oracle.so")
            if not os.path.exists(path):
                subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
            Oracle2D.lib = C.CDLL(path)
            Oracle2D.lib.o2_field.restype = C.c_void_p
            Oracle2D.lib.o2_field.argtypes = [C.c_void_p, C.c_int]
            Oracle2D.lib.o2_advance.argtypes = [C.c_void_p, C.c_int]
            Oracle2D.lib.o2_last_iters.restype = C.c_int
            Oracle2D.lib.o2_last_iters.argtypes = [C.c_void_p]
            Oracle2D.lib.o2_last_residual.restype = C.c_double
            Oracle2D.lib.o2_last_residual.argtypes = [C.c_void_p]
            Oracle2D.lib.o2_set_wind_calls.argtypes = [C.c_void_p, C.c_double]
        super().__init__(rx, ry)

    def _get(self, which, a): self.lib.o2_get(self.h, which, a.ctypes.data_as(C.c_void_p))
    def _set(self, which, a): self.lib.o2_set(self.h, which, a.ctypes.data_as(C.c_void_p))

    def advance(self, red_black=False):
        self.lib.o2_advance(self.h, int(red_black))

    def set_wind_calls(self, r):
        """addWind's call counter (a process-wide static in the reference, FluidSim.h:123): for checks that run
        several reference instances in one process."""
        self.lib.o2_set_wind_calls(self.h, float(r))

    @property
    def last_iters(self): return self.lib.o2_last_iters(self.h)
    @property
    def last_residual(self): return self.lib.o2_last_residual(self.h)
