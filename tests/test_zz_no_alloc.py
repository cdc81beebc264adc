"""fs3d_residual_noalloc (the entry point that allocates nothing) against fs3d_residual on a handle.

The entry point takes device arrays, so the test moves a simulation's fields into device buffers of its
own (plain hipMalloc / hipMemcpy through the HIP runtime the library has already loaded) and compares the
two doubles the entry point writes with what fs3d_residual returns for the same state: the kernels are the
same ones, so the values are equal bit for bit.  Two calls in a row share the fixed device array; the
second runs on a smaller grid, so anything the first call left behind would show.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = 1.0 / 30.0


class Hip:
    def __init__(self):
        self.rt = C.CDLL(self.runtime_path())
        self.rt.hipSetDevice.argtypes = [C.c_int]
        self.chk(self.rt.hipSetDevice(0))   # the uploads below go to the device the entry point is called for
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.bufs = []

    @staticmethod
    def runtime_path():
        """The HIP runtime the library itself has mapped into this process (whatever its soname)."""
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                if "libamdhip64.so" in path:
                    return path
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                C.CDLL(name)
                return name
            except OSError:
                pass
        raise OSError("no HIP runtime found")

    def chk(self, rc):
        assert rc == 0, f"HIP runtime call failed ({rc})"

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self.chk(self.rt.hipMalloc(C.byref(p), a.nbytes))
        self.bufs.append(p)
        self.chk(self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1))   # host to device
        return p

    def alloc(self, nbytes, fill=0xFF):
        p = C.c_void_p()
        self.chk(self.rt.hipMalloc(C.byref(p), nbytes))
        self.bufs.append(p)
        self.chk(self.rt.hipMemset(p, fill, nbytes))
        return p

    def download(self, p, dtype, count):
        out = np.empty(count, dtype=dtype)
        self.chk(self.rt.hipDeviceSynchronize())
        self.chk(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2))   # device to host
        return out

    def free_all(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


def state(fsamd, shape, dtype, steps, obstacle):
    nx, ny, nz = shape
    sim = fsamd.FluidSim3D(nx, ny, nz, device=0, dtype=dtype, stop_mode=fsamd.STOP_FIXED, max_iters=3)
    sim.sphere_influence(nx / 10.0)
    if obstacle:
        sim.solid_sphere(nx / 8.0, nx / 2.0, ny / 2.0, nz / 2.0)
    for it in range(steps):
        sim.fill_sphere_influence(nx / 10.0, it)
        sim.advance(DT)
    return sim


def noalloc(fsamd, hip, sim, dtype):
    lib = fsamd.load_library()
    u, v, w = (hip.upload(sim.get_field(k)) for k in range(3))
    cells = hip.upload(sim.cells)
    out = hip.alloc(16)
    rc = lib.fs3d_residual_noalloc(0, sim.nx, sim.ny, sim.nz, dtype, u, v, w, cells, out, None)
    assert rc == 0, lib.fs3d_last_error(None).decode()
    return hip.download(out, np.float64, 2)


@pytest.mark.parametrize("dtype", [0, 1])
def test_no_alloc_residual_equals_the_handle_entry_point_twice_in_a_row(fsamd, dtype):
    fsamd.load_library()
    hip = Hip()
    try:
        # first a grid with 16-byte rows (the wide path of the residual kernel) and an obstacle, then a smaller,
        # ragged one: the second call finds the first call's flags and partial sums in the shared array
        for shape, steps, obstacle in [((64, 48, 40), 3, True), ((30, 21, 17), 2, False)]:
            sim = state(fsamd, shape, dtype, steps, obstacle)
            l2, mx = sim.residual()
            got = noalloc(fsamd, hip, sim, dtype)
            assert l2 > 0 and mx > 0
            assert got[0] == l2 and got[1] == mx, (shape, got, (l2, mx))
            sim.close()
    finally:
        hip.free_all()


def test_no_alloc_residual_refuses_what_does_not_fit(fsamd):
    lib = fsamd.load_library()
    hip = Hip()
    try:
        p = hip.alloc(64)
        rc = lib.fs3d_residual_noalloc(0, 512, 512, 512, 0, p, p, p, p, p, None)   # 2^27 cells > 2^24
        assert rc == -1
        assert b"exceed the fixed scratch" in lib.fs3d_last_error(None)
        assert lib.fs3d_residual_noalloc(0, 2, 8, 8, 0, p, p, p, p, p, None) == -1
        assert lib.fs3d_residual_noalloc(0, 8, 8, 8, 0, None, p, p, p, p, None) == -1
    finally:
        hip.free_all()
