"""The C++ host side (include/fsamd/*.hpp): builds with plain g++ against the C ABI on any machine;
on the GPU it must reproduce the oracle when driven with the call order of the reference's main()."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_adapter_builds():
    subprocess.check_call(["make", "-s", "-C", CPP])
    assert os.path.exists(os.path.join(CPP, "adapter_main"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="needs the reference tree (build container only)")
def test_adapters_compile_against_the_reference_s_own_simulation_header():
    # FluidSim3DSimulation<Simulation> and FluidSim2DSimulation<Simulation> over include/Simulation.h:11-78 itself
    # (libigl viewer type, RigidObject list and all), driven through a Simulation* like Simulator::runSimThread
    subprocess.check_call(["make", "-s", "-C", CPP, "real_interface_check"])


@pytest.mark.gpu
def test_adapter_reproduces_oracle(tmp_path):
    from oracle.oracle3d import Oracle3D
    subprocess.check_call(["make", "-s", "-C", CPP])
    out = tmp_path / "fields.bin"
    n, steps = 32, 3
    res = subprocess.run([os.path.join(CPP, "adapter_main"), str(n), str(steps), str(out)],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "density mismatches 0" in res.stdout and "after reset: step 0 time 0.0" in res.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(4, n, n, n)
    o = Oracle3D(n, n, n)
    o.sphere_influence(n / 10.0)
    for it in range(steps):
        o.fill_sphere_influence(n / 10.0, it)
        o.project_rb(stop_mode=1, tol=5.0, max_iter=1000)
        o.semi_lagrange(np.float32(1.0 / 30.0))
    for k in range(4):
        assert np.array_equal(got[k], o.field(k)), f"field {k}"
    # the 2D solver through the Simulation-shaped adapter
    from oracle import oracle2d as o2
    ref = o2.Oracle2D(32, 48)
    for _ in range(3):
        ref.advance()
    line = [l for l in res.stdout.splitlines() if l.startswith("2D run:")][0].split()
    assert line[3] == "3" and float(line[7]) == pytest.approx(ref.dt, abs=1e-6)
    assert float(line[10]) == pytest.approx(ref.get(o2.DENSITY).sum(), abs=1e-8)
    assert float(line[12]) == pytest.approx(ref.get(o2.U)[16, 10], rel=1e-8, abs=1e-15)
