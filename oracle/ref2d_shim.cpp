// TEST INFRASTRUCTURE.  C entry points around the reference's own 2D solver
// (class FluidSim, 3_fluid/FluidSim.{h,cpp}) so that tests and the fixture
// generator can drive it headless.  This file contains no reference code: it
// includes the reference's headers from where they lie (see oracle/Makefile,
// target `ref`) and is linked with the reference's own FluidSim.cpp.
//
// The solver's grids and parameters are private and its resolution is fixed
// in init() (3_fluid/FluidSim.h:15-41), so the class is opened with the usual
// access-specifier trick; every system / Eigen / libigl header is included
// first so that the trick only reaches the reference's own classes.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <igl/colormap.h>
#include <igl/edges.h>
#include <igl/per_face_normals.h>
#include <igl/per_vertex_normals.h>
#include <igl/readOBJ.h>
#include <igl/readOFF.h>
#include <igl/opengl/glfw/Viewer.h>

#define private public
#define protected public
#include "FluidSim.h"
#undef private
#undef protected

namespace {
Array2d& field(FluidSim* s, int which) {
    switch (which) {
        case 0: return s->p_density->x();
        case 1: return s->p_pressure->x();
        case 2: return s->p_divergence->x();
        case 3: return s->p_vorticity->x();
        case 4: return s->p_velocity->x();
        case 5: return s->p_velocity->y();
        case 6: return s->p_force->x();
        default: return s->p_force->y();
    }
}
}  // namespace

extern "C" {

// A solver of res_x * res_y cells with every derived parameter computed as init() does
// (3_fluid/FluidSim.h:16-29).  res 128 x 192 is the reference's own configuration.
void* ref2d_create(int res_x, int res_y) {
    FluidSim* s = new FluidSim();   // 128 x 192
    if (res_x != s->m_res_x || res_y != s->m_res_y) {
        s->m_res_x = res_x;
        s->m_res_y = res_y;
        s->m_size_x = s->m_res_x;
        s->m_dx = s->m_size_x / s->m_res_x;
        s->m_idx = s->m_res_x / s->m_size_x;
        s->m_size_y = s->m_dx * s->m_res_y;
        s->m_dt = 0.005 * sqrt((s->m_res_x + s->m_res_y) * 0.5);
        s->p_density = new Grid2(res_x, res_y, s->m_dx);
        s->p_pressure = new Grid2(res_x, res_y, s->m_dx);
        s->p_divergence = new Grid2(res_x, res_y, s->m_dx);
        s->p_vorticity = new Grid2(res_x, res_y, s->m_dx);
        s->p_velocity = new MACGrid2(res_x, res_y, s->m_dx);
        s->p_force = new MACGrid2(res_x, res_y, s->m_dx);
        s->reset();
    }
    return s;
}
void ref2d_destroy(void* h) { delete static_cast<FluidSim*>(h); }

int ref2d_size(void* h, int which, int dim) { return field(static_cast<FluidSim*>(h), which).size(dim); }
void ref2d_get(void* h, int which, double* out) {
    Array2d& a = field(static_cast<FluidSim*>(h), which);
    for (int i = 0; i < a.size(0); ++i)
        for (int j = 0; j < a.size(1); ++j) out[(size_t)i * a.size(1) + j] = a(i, j);
}
void ref2d_set(void* h, int which, const double* in) {
    Array2d& a = field(static_cast<FluidSim*>(h), which);
    for (int i = 0; i < a.size(0); ++i)
        for (int j = 0; j < a.size(1); ++j) a(i, j) = in[(size_t)i * a.size(1) + j];
}
void ref2d_params(void* h, double dt, double acc, int iter, int wind, int maccormack) {
    FluidSim* s = static_cast<FluidSim*>(h);
    if (dt > 0) s->setTimestep(dt);
    if (acc > 0) s->setAccuracy(acc);
    if (iter > 0) s->setIteration(iter);
    s->setWind(wind != 0);
    s->setMacCormack(maccormack != 0);
}
double ref2d_dt(void* h) { return static_cast<FluidSim*>(h)->getTimestep(); }
double ref2d_time(void* h) { return static_cast<FluidSim*>(h)->getTime(); }
unsigned long ref2d_step(void* h) { return static_cast<FluidSim*>(h)->getStep(); }
void ref2d_reset(void* h) { static_cast<FluidSim*>(h)->reset(); }
int ref2d_advance(void* h) { return static_cast<FluidSim*>(h)->advance() ? 1 : 0; }

// single stages of advance() (3_fluid/FluidSim.h:71-96, :150-167)
void ref2d_stage(void* h, int op) {
    FluidSim* s = static_cast<FluidSim*>(h);
    switch (op) {
        case 0: s->p_density->applySource(0.45, 0.55, 0.1, 0.15); break;
        case 1: s->addBuoyancy(); break;
        case 2: s->addWind(); break;
        case 3: s->addForce(); break;
        case 4: s->solvePressure(); break;
        case 5: s->advectValues(); break;
        case 6: s->setNeumann(); break;
        case 7: s->setZero(); break;
        case 8: s->computeDivergence(); break;
        case 9: s->copyBorder(); break;
        case 10: s->solvePoisson(); break;
        case 11: s->correctVelocity(); break;
        case 12: s->computeVorticity(); break;
        case 13: s->p_force->reset(); break;
    }
}

}  // extern "C"
