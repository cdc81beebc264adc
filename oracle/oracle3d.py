"""ctypes binding of the CPU oracle (oracle/fluid3d_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py may import this module.  The shipped library never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

FLUID, OBJECT, STATIC = 0, 1, 2
U, V, W, D, SU, SV, SW, SD = range(8)


def build():
    """Compile the oracle libraries (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


_NATIVE_PATH = None


def build_native():
    """The fp32 oracle once more with -O3 -march=native (same source, still no fused multiply-add), for timing it as
    the CPU baseline on the machine it is built on: compiled into a temporary directory at run time, never shipped
    (-march=native code may not run on another machine).  Returns False if that build is not possible here."""
    global _NATIVE_PATH
    if _NATIVE_PATH:
        return True
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="fsamd_oracle_"), "libfluid3d_oracle_f32_native.so")
    cmd = [os.environ.get("CC", "gcc"), "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-shared", "-fopenmp",
           "-DREAL=float", "-o", out, os.path.join(_HERE, "fluid3d_oracle.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _NATIVE_PATH = out
    return True


def _lib(dtype, native=False):
    name = {np.float32: "f32", np.float64: "f64"}[dtype] + ("_native" if native else "")
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(_HERE, f"libfluid3d_oracle_{name}.so")
    if native:
        if dtype is not np.float32 or not build_native():
            raise RuntimeError("no native build of the oracle")
        path = _NATIVE_PATH
    elif not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    real = C.c_float if dtype is np.float32 else C.c_double
    P = C.c_void_p
    lib.o3_create.restype = P
    lib.o3_threads.argtypes = [C.c_int]
    lib.o3_threads.restype = C.c_int
    lib.o3_create.argtypes = [C.c_int] * 3
    lib.o3_destroy.argtypes = [P]
    lib.o3_field.restype = P
    lib.o3_field.argtypes = [P, C.c_int]
    lib.o3_cells.restype = P
    lib.o3_cells.argtypes = [P]
    lib.o3_sphere_emitter.argtypes = [P] + [C.c_float] * 5
    lib.o3_sphere_influence.argtypes = [P, C.c_float]
    lib.o3_fill_sphere.argtypes = [P] + [C.c_float] * 4 + [C.c_int]
    lib.o3_fill_sphere_influence.argtypes = [P, C.c_float, C.c_int]
    lib.o3_solid_sphere.argtypes = [P] + [C.c_float] * 4
    lib.o3_apply_source.argtypes = [P] + [C.c_double] * 6
    lib.o3_bottom_influence.argtypes = [P]
    lib.o3_clear.argtypes = [P]
    lib.o3_gs_lexi.restype = real
    lib.o3_gs_lexi.argtypes = [P]
    lib.o3_semi_lagrange.argtypes = [P, real, P, P, P, P]
    lib.o3_update.restype = C.c_int
    lib.o3_update.argtypes = [P, real, real, C.c_int, C.POINTER(real)]
    lib.o3_rb_halfsweep.argtypes = [P, C.c_int, real]
    lib.o3_rb_halfsweep_range.argtypes = [P, C.c_int, real, C.c_int, C.c_int, C.c_int]
    lib.o3_rb_sweep.restype = C.c_double
    lib.o3_rb_sweep.argtypes = [P, real]
    lib.o3_div_sumsq.restype = C.c_double
    lib.o3_div_sumsq.argtypes = [P, C.POINTER(C.c_double)]
    lib.o3_residual_l2.restype = C.c_double
    lib.o3_residual_l2.argtypes = [P]
    lib.o3_advect.argtypes = [P, real] + [P] * 11 + [C.c_int]
    lib.o3_reflect.argtypes = [C.c_int64, P, P, P]
    lib.o3_maccormack.argtypes = [P, real] + [P] * 15
    lib.o3_mg_vcycle.argtypes = [P, C.c_int, C.c_int, C.c_int, real, C.c_int]
    lib.o3_vorticity.argtypes = [P, P]
    lib.o3_vorticity.restype = C.c_double
    lib.o3_diagnostics.argtypes = [P, C.POINTER(C.c_double)]
    lib.o3_downsample_density.argtypes = [P, C.c_int, P]
    lib.o3_buoyancy.argtypes = [P, real]
    lib.o3_project_rb.restype = C.c_int
    lib.o3_project_rb.argtypes = [P, real, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib._real = real
    _LIBS[name] = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _View(np.ndarray):
    """ndarray view onto the oracle's C arrays that keeps the owning object alive."""
    _owner = None


def _view(arr, owner):
    v = arr.view(_View)
    v._owner = owner
    return v


class Oracle3D:
    """The reference's 3D FluidSim (advection-reflection/main.cpp:21-442), restated on the CPU."""

    def __init__(self, nx, ny, nz, dtype=np.float32, native=False):
        self.dtype = dtype
        self.lib = _lib(dtype, native)
        self.nx, self.ny, self.nz = nx, ny, nz
        self.h = self.lib.o3_create(nx, ny, nz)
        self.n = nx * ny * nz

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.o3_destroy(self.h)
            self.h = None

    # --- state views (zero-copy numpy views onto the C arrays, shape (nz, ny, nx)) ---
    def field(self, which):
        p = self.lib.o3_field(self.h, which)
        ct = C.c_float if self.dtype is np.float32 else C.c_double
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(self.n,))
        return _view(arr.reshape(self.nz, self.ny, self.nx), self)

    @property
    def u(self): return self.field(U)
    @property
    def v(self): return self.field(V)
    @property
    def w(self): return self.field(W)
    @property
    def d(self): return self.field(D)

    @property
    def cells(self):
        p = self.lib.o3_cells(self.h)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n,))
        return _view(arr.reshape(self.nz, self.ny, self.nx), self)

    # --- reference surface ---
    def sphere_influence(self, radius):
        self.lib.o3_sphere_influence(self.h, radius)

    def sphere_emitter(self, radius, cx, cy, cz, strength):
        self.lib.o3_sphere_emitter(self.h, radius, cx, cy, cz, strength)

    def fill_sphere_influence(self, radius, it):
        self.lib.o3_fill_sphere_influence(self.h, radius, it)

    def fill_sphere(self, radius, cx, cy, cz, it):
        self.lib.o3_fill_sphere(self.h, radius, cx, cy, cz, it)

    def solid_sphere(self, radius, cx, cy, cz):
        self.lib.o3_solid_sphere(self.h, radius, cx, cy, cz)

    def apply_source(self, xmin, xmax, ymin, ymax, zmin, zmax):
        self.lib.o3_apply_source(self.h, xmin, xmax, ymin, ymax, zmin, zmax)

    def bottom_influence(self):
        self.lib.o3_bottom_influence(self.h)

    def clear(self):
        self.lib.o3_clear(self.h)

    def gauss_seidel_projection(self):
        return float(self.lib.o3_gs_lexi(self.h))

    def semi_lagrange(self, dt):
        """Self-advection from a fresh snapshot, as update() does (main.cpp:122-129)."""
        for a, b in ((SU, U), (SV, V), (SW, W), (SD, D)):
            self.field(a)[...] = self.field(b)
        f = lambda k: self.lib.o3_field(self.h, k)
        self.lib.o3_semi_lagrange(self.h, dt, f(SU), f(SV), f(SW), f(SD))

    def update(self, dt, change_tol=5.0, max_iter=1000):
        last = self.lib._real(0)
        it = self.lib.o3_update(self.h, dt, change_tol, max_iter, C.byref(last))
        return it, float(last.value)

    # --- extensions ---
    def rb_halfsweep(self, colour, omega=1.0):
        self.lib.o3_rb_halfsweep(self.h, colour, omega)

    def rb_halfsweep_range(self, colour, zb, ze, gz0, omega=1.0):
        self.lib.o3_rb_halfsweep_range(self.h, colour, omega, zb, ze, gz0)

    def rb_sweep(self, omega=1.0):
        return float(self.lib.o3_rb_sweep(self.h, omega))

    def div_sumsq(self):
        mx = C.c_double(0)
        s = self.lib.o3_div_sumsq(self.h, C.byref(mx))
        return float(s), float(mx.value)

    def residual_l2(self):
        return float(self.lib.o3_residual_l2(self.h))

    def project_rb(self, omega=1.0, stop_mode=0, tol=0.0, max_iter=100, check_every=1):
        m = C.c_double(0)
        it = self.lib.o3_project_rb(self.h, omega, stop_mode, tol, max_iter, check_every, C.byref(m))
        return it, float(m.value)

    def advect(self, dt, q, vel, density_mode):
        """q = (qu,qv,qw,qd), vel = (U,V,W) numpy arrays; returns the four advected arrays."""
        q = [np.ascontiguousarray(a, dtype=self.dtype) for a in q]
        vel = [np.ascontiguousarray(a, dtype=self.dtype) for a in vel]
        out = [np.empty_like(q[0]) for _ in range(4)]
        self.lib.o3_advect(self.h, dt, *[_ptr(a) for a in q], *[_ptr(a) for a in vel],
                           *[_ptr(a) for a in out], density_mode)
        return out

    def maccormack(self, dt, q, vel, f):
        """MacCormack correction + limiter: q, f = (u,v,w,d) before and after the forward advection over dt
        along vel = (U,V,W); returns the four corrected arrays."""
        q = [np.ascontiguousarray(a, dtype=self.dtype) for a in q]
        vel = [np.ascontiguousarray(a, dtype=self.dtype) for a in vel]
        f = [np.ascontiguousarray(a, dtype=self.dtype) for a in f]
        out = [np.empty_like(q[0]) for _ in range(4)]
        self.lib.o3_maccormack(self.h, dt, *[_ptr(a) for a in q], *[_ptr(a) for a in vel], *[_ptr(a) for a in f],
                               *[_ptr(a) for a in out])
        return out

    def mg_vcycle(self, nu1=2, nu2=2, nu_bottom=20, alpha=2.0, nlev_max=0):
        """One multigrid V-cycle on the velocity field (see o3_mg_vcycle)."""
        self.lib.o3_mg_vcycle(self.h, nu1, nu2, nu_bottom, alpha, nlev_max)

    def vorticity(self):
        """(|curl u| field, sum of |curl u|^2)"""
        out = np.empty((self.nz, self.ny, self.nx), dtype=self.dtype)
        ens = self.lib.o3_vorticity(self.h, _ptr(out))
        return out, float(ens)

    def diagnostics(self):
        """(kinetic energy, max speed, density sum)"""
        d = (C.c_double * 3)()
        self.lib.o3_diagnostics(self.h, d)
        return d[0], d[1], d[2]

    def downsample_density(self, f):
        shape = tuple((n + f - 1) // f for n in (self.nz, self.ny, self.nx))
        out = np.empty(shape, dtype=self.dtype)
        self.lib.o3_downsample_density(self.h, f, _ptr(out))
        return out

    def reflect(self, proj, tilde):
        proj = np.ascontiguousarray(proj, dtype=self.dtype)
        tilde = np.ascontiguousarray(tilde, dtype=self.dtype)
        out = np.empty_like(proj)
        self.lib.o3_reflect(proj.size, _ptr(proj), _ptr(tilde), _ptr(out))
        return out

    def buoyancy(self, dt):
        self.lib.o3_buoyancy(self.h, dt)
