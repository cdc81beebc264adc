// Compile-only check of the adapters against the reference's OWN interface header (include/Simulation.h:11-78,
// which pulls in libigl's viewer and RigidObject): both adapter templates are instantiated over the reference's
// `Simulation`, and a function drives them through a `Simulation*` the way Simulator::runSimThread does
// (include/Simulator.h:172-231: advance(), then updateRenderGeometry(); reset() from the GUI thread, :74-86).
// Built only where the reference tree is mounted (tests/test_cpp_adapter.py); nothing is linked or run: the viewer
// needs GLFW/OpenGL, which this container does not have.
#include "Simulation.h"

#include <fsamd/FluidSim2D.hpp>
#include <fsamd/FluidSim3D.hpp>

using Gpu3D = fsamd::FluidSim3DSimulation<Simulation>;
using Gpu2D = fsamd::FluidSim2DSimulation<Simulation>;

static_assert(std::is_base_of<Simulation, Gpu3D>::value, "the 3D adapter is a Simulation");
static_assert(std::is_base_of<Simulation, Gpu2D>::value, "the 2D adapter is a Simulation");
static_assert(!std::is_abstract<Gpu3D>::value && !std::is_abstract<Gpu2D>::value, "every pure virtual is implemented");

// what Simulator does with the object it owns
double drive(Simulation* sim, igl::opengl::glfw::Viewer& viewer, int steps) {
    sim->reset();
    sim->setTimestep(1.0 / 30.0);
    for (int i = 0; i < steps; ++i) {
        if (sim->advance()) break;
        sim->updateRenderGeometry();
        sim->renderRenderGeometry(viewer);
    }
    // fluid simulations hold no rigid objects (3_fluid/FluidSim.h never touches m_objects): the list the
    // reference's GUI iterates over (include/Gui.cpp) is there and stays empty
    return sim->getTime() + (double)sim->getStep() + (double)sim->getObjects().size();
}

double both(igl::opengl::glfw::Viewer& viewer) {
    Gpu3D a(32, 32, 32);
    Gpu2D b(32, 48);
    return drive(&a, viewer, 2) + drive(&b, viewer, 2);
}
