"""ctypes binding of libfluidsim_amd.so (include/fluidsim_amd.h).

The classes keep the method names of the reference's solvers so that code and
tests written against them read the same:

  FluidSim3D  <->  class FluidSim of advection-reflection/main.cpp:21-442
                   (sphere_influence, fill_sphere_influence, bottom_influence,
                   clear, get_density, update, gauss_seidel_projection,
                   semi_lagrange) plus the Simulation interface of
                   include/Simulation.h:11-78 (reset, advance, setTimestep,
                   getTime, getStep).

There is no CPU fallback: if the HIP library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfluidsim_amd.so")

FS_OK = 0
F32, F64 = 0, 1
CELL_FLUID, CELL_OBJECT, CELL_STATIC_FLUID = 0, 1, 2
FIELD_U, FIELD_V, FIELD_W, FIELD_DENSITY, FIELD_DIVERGENCE, FIELD_VORTICITY = range(6)
ORDER_RED_BLACK, ORDER_LEXICOGRAPHIC = 0, 1
STOP_FIXED, STOP_L1_CHANGE, STOP_L2_RESIDUAL = 0, 1, 2
CHECK_ADAPTIVE, CHECK_DEFERRED = 0, -1
STEP_REFERENCE, STEP_REFLECTION = 0, 1
SWEEP_FUSED, SWEEP_TWO_PASS, SWEEP_FUSED_LDS = 0, 1, 2
ADVECT_TILED, ADVECT_DIRECT = 0, 1
ADVECTION_SEMI_LAGRANGE, ADVECTION_MACCORMACK = 0, 1
SOLVER_SWEEPS, SOLVER_MULTIGRID = 0, 1
COMM_ID_BYTES = 128
NOALLOC_MAX_CELLS = 1 << 24

# every symbol include/fluidsim_amd.h declares
EXPORTS = [
    "fsamd_version", "fs3d_default_config", "fs3d_create", "fs3d_destroy", "fs3d_reset", "fs3d_set_config",
    "fs3d_get_config", "fs3d_last_error", "fs3d_set_field", "fs3d_get_field", "fs3d_get_field_plane",
    "fs3d_set_cell_types", "fs3d_get_cell_types", "fs3d_sphere_influence", "fs3d_sphere_emitter", "fs3d_solid_sphere",
    "fs3d_fill_sphere_influence", "fs3d_fill_sphere", "fs3d_bottom_influence", "fs3d_clear", "fs3d_apply_source",
    "fs3d_gauss_seidel_projection", "fs3d_sweeps", "fs3d_time_sweeps", "fs3d_time_sweep_pairs", "fs3d_time_field_copies", "fs3d_time_advections", "fs3d_project", "fs3d_residual", "fs3d_semi_lagrange", "fs3d_sweep_advect",
    "fs3d_update", "fs3d_advance", "fs3d_get_stats", "fs3d_synchronize", "fs3d_synchronize_all", "fs3d_diagnostics",
    "fs3d_get_density_downsampled", "fs3d_save", "fs3d_load", "fs3d_mg_vcycle", "fs3d_residual_noalloc",
    "fsamd_comm_unique_id", "fsamd_comm_selftest",
    "fs3d_create_slab", "fs3d_slab_range", "fs3d_comm_timing", "fs3d_get_comm_stats",
    "fs2d_create", "fs2d_destroy", "fs2d_last_error", "fs2d_reset", "fs2d_set_params", "fs2d_get_info",
    "fs2d_field_size", "fs2d_get_field", "fs2d_set_field", "fs2d_advance", "fs2d_stage",
]


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("dtype", C.c_int32), ("ordering", C.c_int32), ("stop_mode", C.c_int32),
        ("max_iters", C.c_int32), ("check_every", C.c_int32), ("step_mode", C.c_int32), ("local_slabs", C.c_int32),
        ("sweep_kernel", C.c_int32), ("advect_kernel", C.c_int32), ("advection", C.c_int32), ("solver", C.c_int32),
        ("tol", C.c_double), ("omega", C.c_double), ("buoyancy", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("step", C.c_int64), ("time", C.c_double), ("sweeps", C.c_int32), ("sweeps_step", C.c_int32),
        ("metric", C.c_double), ("ms_project", C.c_double), ("ms_advect", C.c_double), ("ms_step", C.c_double),
        ("evaluations_step", C.c_int32), ("redone", C.c_int32), ("fused", C.c_int64),
        ("sweeps_total", C.c_int64), ("evaluations_total", C.c_int64),
    ]


class CommStats(C.Structure):
    _fields_ = [("exchanges", C.c_int64), ("waits", C.c_int64), ("reductions", C.c_int64), ("bytes_sent_lo", C.c_int64),
                ("bytes_sent_hi", C.c_int64), ("exchange_ms", C.c_double), ("wait_ms", C.c_double),
                ("reduction_host_ms", C.c_double), ("poll_fallbacks", C.c_int64)]


class Diag(C.Structure):
    _fields_ = [("kinetic_energy", C.c_double), ("max_speed", C.c_double), ("density_sum", C.c_double),
                ("enstrophy", C.c_double)]


class FluidSimError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """Loads the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FluidSimError(
            f"{path} not found: build it with `make -C fluid-simulation_amd/csrc` "
            "(there is no CPU fallback for the product path)")
    lib = C.CDLL(path)
    P, I, D, F = C.c_void_p, C.c_int, C.c_double, C.c_float
    sig = {
        "fsamd_version": (C.c_char_p, []),
        "fs3d_default_config": (None, [C.POINTER(Config)]),
        "fs3d_create": (I, [I, I, I, C.POINTER(Config), C.POINTER(P)]),
        "fs3d_destroy": (None, [P]),
        "fs3d_reset": (I, [P]),
        "fs3d_set_config": (I, [P, C.POINTER(Config)]),
        "fs3d_get_config": (I, [P, C.POINTER(Config)]),
        "fs3d_last_error": (C.c_char_p, [P]),
        "fs3d_set_field": (I, [P, I, P, C.c_size_t]),
        "fs3d_get_field": (I, [P, I, P, C.c_size_t]),
        "fs3d_get_field_plane": (I, [P, I, I, P, C.c_size_t]),
        "fs3d_set_cell_types": (I, [P, P, C.c_size_t]),
        "fs3d_get_cell_types": (I, [P, P, C.c_size_t]),
        "fs3d_sphere_influence": (I, [P, F]),
        "fs3d_sphere_emitter": (I, [P, F, F, F, F, F]),
        "fs3d_solid_sphere": (I, [P, F, F, F, F]),
        "fs3d_fill_sphere_influence": (I, [P, F, I]),
        "fs3d_fill_sphere": (I, [P, F, F, F, F, I]),
        "fs3d_bottom_influence": (I, [P]),
        "fs3d_clear": (I, [P]),
        "fs3d_apply_source": (I, [P, D, D, D, D, D, D]),
        "fs3d_gauss_seidel_projection": (I, [P, C.POINTER(D)]),
        "fs3d_sweeps": (I, [P, I]),
        "fs3d_time_sweeps": (I, [P, I, C.POINTER(D)]),
        "fs3d_time_sweep_pairs": (I, [P, I, C.POINTER(D)]),
        "fs3d_time_field_copies": (I, [P, I, C.POINTER(D)]),
        "fs3d_time_advections": (I, [P, I, D, I, C.POINTER(D)]),
        "fs3d_project": (I, [P, C.POINTER(I), C.POINTER(D)]),
        "fs3d_residual": (I, [P, C.POINTER(D), C.POINTER(D)]),
        "fs3d_semi_lagrange": (I, [P, D]),
        "fs3d_sweep_advect": (I, [P, D, C.POINTER(D), C.POINTER(I)]),
        "fs3d_update": (I, [P, D]),
        "fs3d_advance": (I, [P, D]),
        "fs3d_get_stats": (I, [P, C.POINTER(Stats)]),
        "fs3d_synchronize": (I, [P]),
        "fs3d_synchronize_all": (I, [P]),
        "fs3d_diagnostics": (I, [P, C.POINTER(Diag)]),
        "fs3d_get_density_downsampled": (I, [P, I, P, C.c_size_t]),
        "fs3d_mg_vcycle": (I, [P]),
        "fs3d_save": (I, [P, C.c_char_p]),
        "fs3d_load": (I, [P, C.c_char_p]),
        "fs3d_residual_noalloc": (I, [I, I, I, I, I, P, P, P, P, P, P]),
        "fsamd_comm_unique_id": (I, [P]),
        "fsamd_comm_selftest": (I, [I]),
        "fs3d_create_slab": (I, [I, I, I, C.POINTER(Config), I, I, P, C.POINTER(P)]),
        "fs3d_slab_range": (I, [P, C.POINTER(I), C.POINTER(I), C.POINTER(I)]),
        "fs3d_comm_timing": (I, [P, I]),
        "fs3d_get_comm_stats": (I, [P, C.POINTER(CommStats), I]),
        "fs2d_create": (I, [I, I, I, C.POINTER(P)]),
        "fs2d_destroy": (None, [P]),
        "fs2d_last_error": (C.c_char_p, [P]),
        "fs2d_reset": (I, [P]),
        "fs2d_set_params": (I, [P, D, D, I, I, I, I]),
        "fs2d_get_info": (I, [P, C.POINTER(D), C.POINTER(D), C.POINTER(C.c_uint64), C.POINTER(I), C.POINTER(D)]),
        "fs2d_field_size": (I, [P, I, C.POINTER(I), C.POINTER(I)]),
        "fs2d_get_field": (I, [P, I, P, C.c_size_t]),
        "fs2d_set_field": (I, [P, I, P, C.c_size_t]),
        "fs2d_advance": (I, [P]),
        "fs2d_stage": (I, [P, I]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_config():
    cfg = Config()
    load_library().fs3d_default_config(C.byref(cfg))
    return cfg


def comm_unique_id():
    """Rank 0 calls this and sends the 128 bytes to the other ranks by any host channel."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = load_library().fsamd_comm_unique_id(buf)
    if rc != FS_OK:
        raise FluidSimError(f"fsamd_comm_unique_id failed ({rc})")
    return bytes(buf)


class FluidSim3D:
    """The reference's 3D FluidSim (advection-reflection/main.cpp:21-442) on one MI355X, or one
    z-slab of it when rank/nranks are given."""

    def __init__(self, nx, ny, nz, config=None, rank=0, nranks=1, comm_id=None, **overrides):
        self.lib = load_library()
        cfg = config if config is not None else default_config()
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dtype = np.float64 if cfg.dtype == F64 else np.float32
        h = C.c_void_p()
        if nranks > 1:
            idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
            rc = self.lib.fs3d_create_slab(nx, ny, nz, C.byref(cfg), rank, nranks, idbuf, C.byref(h))
        else:
            rc = self.lib.fs3d_create(nx, ny, nz, C.byref(cfg), C.byref(h))
        self.h = h
        self.m_dt = 0.0   # Simulation::m_dt (include/Simulation.h:75)
        if rc != FS_OK:
            msg = self.lib.fs3d_last_error(h).decode() if h else "invalid argument"
            if h:
                self.lib.fs3d_destroy(h)
            self.h = None
            raise FluidSimError(f"fs3d_create failed ({rc}): {msg}")
        z0, z1, halo = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.fs3d_slab_range(self.h, C.byref(z0), C.byref(z1), C.byref(halo)))
        self.z0, self.z1 = z0.value, z1.value
        self.shape = (self.z1 - self.z0, ny, nx)
        self.count = self.shape[0] * ny * nx

    # ---- plumbing ----
    def _chk(self, rc):
        if rc != FS_OK:
            raise FluidSimError(f"fluidsim_amd call failed ({rc}): {self.lib.fs3d_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.fs3d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def config(self):
        cfg = Config()
        self._chk(self.lib.fs3d_get_config(self.h, C.byref(cfg)))
        return cfg

    def set_config(self, **kw):
        cfg = self.config
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        self._chk(self.lib.fs3d_set_config(self.h, C.byref(cfg)))

    # ---- grid accessors ----
    def get_field(self, which):
        out = np.empty(self.shape, dtype=self.dtype)
        self._chk(self.lib.fs3d_get_field(self.h, which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_field(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        if a.size != self.count:
            raise ValueError("field size mismatch")
        self._chk(self.lib.fs3d_set_field(self.h, which, a.ctypes.data_as(C.c_void_p), a.size))

    def mg_vcycle(self):
        self._chk(self.lib.fs3d_mg_vcycle(self.h))

    def vorticity(self):
        """|curl u| with the stencil of computeVorticity (3_fluid/FluidSim.h:220-229)."""
        return self.get_field(FIELD_VORTICITY)

    def diagnostics(self):
        d = Diag()
        self._chk(self.lib.fs3d_diagnostics(self.h, C.byref(d)))
        return d

    def downsample_density(self, factor):
        """Density averaged over blocks of factor^3 cells (of the owned planes)."""
        shape = tuple((n + factor - 1) // factor for n in self.shape)
        out = np.empty(shape, dtype=self.dtype)
        self._chk(self.lib.fs3d_get_density_downsampled(self.h, factor, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def save(self, path):
        self._chk(self.lib.fs3d_save(self.h, os.fsencode(path)))

    def load(self, path):
        self._chk(self.lib.fs3d_load(self.h, os.fsencode(path)))

    def get_field_plane(self, which, z):
        out = np.empty((self.ny, self.nx), dtype=self.dtype)
        self._chk(self.lib.fs3d_get_field_plane(self.h, which, z, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    u = property(lambda self: self.get_field(FIELD_U))
    v = property(lambda self: self.get_field(FIELD_V))
    w = property(lambda self: self.get_field(FIELD_W))
    d = property(lambda self: self.get_field(FIELD_DENSITY))
    divergence = property(lambda self: self.get_field(FIELD_DIVERGENCE))

    def get_density(self):            # main.cpp:98-100
        return self.get_field(FIELD_DENSITY)

    @property
    def cells(self):
        out = np.empty(self.shape, dtype=np.uint8)
        self._chk(self.lib.fs3d_get_cell_types(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_cell_types(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint8)
        self._chk(self.lib.fs3d_set_cell_types(self.h, a.ctypes.data_as(C.c_void_p), a.size))

    # ---- scene (main.cpp:44-96) ----
    def sphere_influence(self, radius):
        self._chk(self.lib.fs3d_sphere_influence(self.h, radius))

    def sphere_emitter(self, radius, cx, cy, cz, strength):
        self._chk(self.lib.fs3d_sphere_emitter(self.h, radius, cx, cy, cz, strength))

    def solid_sphere(self, radius, cx, cy, cz):
        self._chk(self.lib.fs3d_solid_sphere(self.h, radius, cx, cy, cz))

    def fill_sphere_influence(self, radius, it):
        self._chk(self.lib.fs3d_fill_sphere_influence(self.h, radius, it))

    def fill_sphere(self, radius, cx, cy, cz, it):
        self._chk(self.lib.fs3d_fill_sphere(self.h, radius, cx, cy, cz, it))

    def bottom_influence(self):
        self._chk(self.lib.fs3d_bottom_influence(self.h))

    def clear(self):
        self._chk(self.lib.fs3d_clear(self.h))

    def apply_source(self, xmin, xmax, ymin, ymax, zmin, zmax):
        self._chk(self.lib.fs3d_apply_source(self.h, xmin, xmax, ymin, ymax, zmin, zmax))

    # ---- the step ----
    def gauss_seidel_projection(self):            # main.cpp:189-254
        c = C.c_double()
        self._chk(self.lib.fs3d_gauss_seidel_projection(self.h, C.byref(c)))
        return c.value

    def sweeps(self, n):
        self._chk(self.lib.fs3d_sweeps(self.h, n))

    def time_sweeps(self, n):
        ms = C.c_double()
        self._chk(self.lib.fs3d_time_sweeps(self.h, n, C.byref(ms)))
        return ms.value

    def time_sweep_pairs(self, npairs):
        ms = C.c_double()
        self._chk(self.lib.fs3d_time_sweep_pairs(self.h, npairs, C.byref(ms)))
        return ms.value

    def time_field_copies(self, n):
        ms = C.c_double()
        self._chk(self.lib.fs3d_time_field_copies(self.h, n, C.byref(ms)))
        return ms.value

    def time_advections(self, n, t_step, reflected=False):
        ms = C.c_double()
        self._chk(self.lib.fs3d_time_advections(self.h, n, t_step, int(reflected), C.byref(ms)))
        return ms.value

    def project(self):
        n, m = C.c_int(), C.c_double()
        self._chk(self.lib.fs3d_project(self.h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def residual(self):
        l2, mx = C.c_double(), C.c_double()
        self._chk(self.lib.fs3d_residual(self.h, C.byref(l2), C.byref(mx)))
        return l2.value, mx.value

    def semi_lagrange(self, t_step):              # main.cpp:256-291 on a fresh snapshot
        self._chk(self.lib.fs3d_semi_lagrange(self.h, t_step))

    def sweep_advect(self, t_step):
        """One red-black sweep and semi_lagrange(t_step) in one launch; returns (L2 residual of the swept field,
        served).  served False: the state is unchanged, use sweeps(1) + semi_lagrange(t_step)."""
        l2, ok = C.c_double(), C.c_int()
        self._chk(self.lib.fs3d_sweep_advect(self.h, t_step, C.byref(l2), C.byref(ok)))
        return l2.value, bool(ok.value)

    def update(self, t_step):                     # main.cpp:102-130
        self._chk(self.lib.fs3d_update(self.h, t_step))
        return self.stats().sweeps

    # ---- Simulation interface (include/Simulation.h:11-78) ----
    def reset(self):
        self._chk(self.lib.fs3d_reset(self.h))

    def setTimestep(self, dt):
        self.m_dt = float(dt)

    def advance(self, dt=None):
        self._chk(self.lib.fs3d_advance(self.h, self.m_dt if dt is None else dt))
        return False                              # fluids never report "done" (3_fluid/FluidSim.h:95)

    def getTime(self):
        return self.stats().time

    def getStep(self):
        return self.stats().step

    def stats(self):
        st = Stats()
        self._chk(self.lib.fs3d_get_stats(self.h, C.byref(st)))
        return st

    def synchronize(self):
        self._chk(self.lib.fs3d_synchronize(self.h))

    def synchronize_all(self):
        """Collective on a multi-rank handle: every rank reports a backtrace that left any rank's halo."""
        self._chk(self.lib.fs3d_synchronize_all(self.h))

    def comm_timing(self, on=True):
        self._chk(self.lib.fs3d_comm_timing(self.h, int(on)))

    def comm_stats(self, reset=False):
        st = CommStats()
        self._chk(self.lib.fs3d_get_comm_stats(self.h, C.byref(st), int(reset)))
        return st


class FluidSim2D:
    """The reference's 2D smoke solver (class FluidSim, 3_fluid/FluidSim.{h,cpp}) on the GPU, fp64.
    Fields are numpy arrays indexed [x, y] like the reference's Array2d(x, y)."""

    DENSITY, PRESSURE, DIVERGENCE, VORTICITY, U, V, FORCE_X, FORCE_Y = range(8)

    def __init__(self, res_x=128, res_y=192, device=-1):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.fs2d_create(res_x, res_y, device, C.byref(h))
        self.h = h
        if rc != FS_OK:
            msg = self.lib.fs2d_last_error(h).decode() if h else "invalid argument"
            if h:
                self.lib.fs2d_destroy(h)
            self.h = None
            raise FluidSimError(f"fs2d_create failed ({rc}): {msg}")
        self.res_x, self.res_y = res_x, res_y
        self._wind, self._mac, self._ordering = False, True, ORDER_LEXICOGRAPHIC

    def _chk(self, rc):
        if rc != FS_OK:
            raise FluidSimError(f"fluidsim_amd 2D call failed ({rc}): {self.lib.fs2d_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.fs2d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shape(self, which):
        a, b = C.c_int(), C.c_int()
        self._chk(self.lib.fs2d_field_size(self.h, which, C.byref(a), C.byref(b)))
        return (a.value, b.value)

    def get(self, which):
        out = np.empty(self.shape(which), dtype=np.float64)
        self._chk(self.lib.fs2d_get_field(self.h, which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if a.shape != self.shape(which):
            raise ValueError("field shape mismatch")
        self._chk(self.lib.fs2d_set_field(self.h, which, a.ctypes.data_as(C.c_void_p), a.size))

    def params(self, dt=0.0, acc=0.0, iters=0, wind=None, maccormack=None, ordering=None):
        self._wind = self._wind if wind is None else bool(wind)
        self._mac = self._mac if maccormack is None else bool(maccormack)
        self._ordering = self._ordering if ordering is None else ordering
        self._chk(self.lib.fs2d_set_params(self.h, dt, acc, iters, int(self._wind), int(self._mac), self._ordering))

    # the reference's setters (3_fluid/FluidSim.h:258-277, include/Simulation.h:58)
    def setTimestep(self, t): self.params(dt=t)
    def setAccuracy(self, acc): self.params(acc=acc)
    def setIteration(self, n): self.params(iters=n)
    def setWind(self, w): self.params(wind=w)
    def setMacCormack(self, m): self.params(maccormack=m)

    def info(self):
        dt, t, res = C.c_double(), C.c_double(), C.c_double()
        step, it = C.c_uint64(), C.c_int()
        self._chk(self.lib.fs2d_get_info(self.h, C.byref(dt), C.byref(t), C.byref(step), C.byref(it), C.byref(res)))
        return {"dt": dt.value, "time": t.value, "step": step.value, "last_iters": it.value, "last_residual": res.value}

    def getTimestep(self): return self.info()["dt"]
    def getTime(self): return self.info()["time"]
    def getStep(self): return self.info()["step"]

    def reset(self):
        self._chk(self.lib.fs2d_reset(self.h))

    def advance(self):
        self._chk(self.lib.fs2d_advance(self.h))
        return False

    def stage(self, op):
        self._chk(self.lib.fs2d_stage(self.h, op))
