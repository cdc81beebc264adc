"""The synthetic scenes of the benchmark configurations, as sequences of calls that work on the GPU
solver (binding.FluidSim3D) and on the CPU oracle alike (same method names).

vortex_ring      the reference's own scene: sphere_influence(N/10) (advection-reflection/main.cpp:462)
ring_collision   two opposed emitters at y = N/10 (+20) and y = 9N/10 (-20)
smoke_plume      source box x,z in [0.45, 0.55], y in [0.10, 0.15] (3_fluid/FluidSim.h:73), buoyancy, and a
                 solid sphere of radius nx/8 in the middle of the grid as interior Object cells
"""


def vortex_ring(sim):
    sim.sphere_influence(sim.nx / 10.0)


def vortex_ring_per_step(sim, it):
    sim.fill_sphere_influence(sim.nx / 10.0, it)          # main.cpp:507


def ring_collision(sim):
    r = sim.nx / 10.0
    sim.sphere_emitter(r, sim.nx / 2.0, r, sim.nz / 2.0, 20.0)
    sim.sphere_emitter(r, sim.nx / 2.0, sim.ny - 1 - r, sim.nz / 2.0, -20.0)


def ring_collision_per_step(sim, it):
    r = sim.nx / 10.0
    sim.fill_sphere(r, sim.nx / 2.0, r, sim.nz / 2.0, it)
    sim.fill_sphere(r, sim.nx / 2.0, sim.ny - 1 - r, sim.nz / 2.0, it)


def smoke_plume(sim):
    sim.solid_sphere(sim.nx / 8.0, sim.nx / 2.0, sim.ny / 2.0, sim.nz / 2.0)
    sim.apply_source(0.45, 0.55, 0.10, 0.15, 0.45, 0.55)


def smoke_plume_per_step(sim, it):
    sim.apply_source(0.45, 0.55, 0.10, 0.15, 0.45, 0.55)  # FluidSim.h:73: re-applied every step


SCENES = {
    "vortex_ring": (vortex_ring, vortex_ring_per_step),
    "ring_collision": (ring_collision, ring_collision_per_step),
    "smoke_plume": (smoke_plume, smoke_plume_per_step),
}
