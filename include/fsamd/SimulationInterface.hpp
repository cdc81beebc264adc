// Headless mirror of the reference's abstract simulation interface
// (include/Simulation.h:11-78): the same member functions with the same
// meaning, with the libigl viewer type only forward-declared so that the
// adapters build without libigl, GLFW or OpenGL.
//
// One member of the reference's class is left out on purpose: getObjects() / m_objects
// (Simulation.h:64, :73), a std::vector<RigidObject>.  RigidObject is a libigl/Eigen mesh
// type that cannot be declared without those libraries, and no fluid simulation of the
// reference ever fills the list (3_fluid/FluidSim.h never touches m_objects).  Over the
// reference's own base class the adapters inherit the (empty) list unchanged;
// tests/cpp/real_interface_check.cpp compiles exactly that.
//
// A caller that has the reference tree uses the reference's own `Simulation`
// as the adapter's base class instead (fsamd::FluidSim3DSimulation<Simulation>,
// the template takes the base as its parameter); this mirror is what the tests
// and headless drivers derive from.
#pragma once
#include <cstdint>

namespace igl { namespace opengl { namespace glfw { class Viewer; } } }

namespace fsamd {

class HeadlessSimulation {
public:
    HeadlessSimulation() = default;
    virtual ~HeadlessSimulation() = default;

    virtual void init() {}                         // Simulation.h:21

    void reset() {                                  // Simulation.h:26-30
        m_time = 0.0;
        m_step = 0;
        resetMembers();
    }

    virtual void updateRenderGeometry() = 0;        // Simulation.h:38
    virtual bool advance() = 0;                     // Simulation.h:47; true = simulation finished
    virtual void renderRenderGeometry(igl::opengl::glfw::Viewer& viewer) = 0;   // Simulation.h:56

    void setTimestep(double t) { m_dt = t; }        // Simulation.h:58
    double getTime() const { return m_time; }       // Simulation.h:60
    unsigned long getStep() const { return m_step; }   // Simulation.h:62

protected:
    virtual void resetMembers() = 0;                // Simulation.h:71
    double m_dt = 0.0;                              // Simulation.h:75-77
    double m_time = 0.0;
    unsigned long m_step = 0;
};

}  // namespace fsamd
