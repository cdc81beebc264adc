// C++ host side of the 2D fp64 smoke solver over the C ABI (fs2d_* in fluidsim_amd.h): the public
// surface of the reference's class FluidSim (3_fluid/FluidSim.h:11-303) behind its Simulation
// interface (include/Simulation.h:11-78).  Base is the reference's `Simulation` in a build against the
// reference tree, fsamd::HeadlessSimulation otherwise.
//
// Grid accessors: the reference exposes Grid2::x() / MACGrid2::x(), y() as Array2d& (include/Grid2.h:
// 20-22, include/MACGrid2.h:19-20).  Here field(which) returns a host copy in the same layout
// (element (x, y) at [x*size1 + y]) and Array2 gives the (i, j) indexing of include/Array2T.h:59-70.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../fluidsim_amd.h"
#include "SimulationInterface.hpp"

namespace fsamd {

struct Array2 {                       // the read side of Array2T<double>
    int size0 = 0, size1 = 0;
    std::vector<double> data;
    double operator()(unsigned i, unsigned j) const { return data[(size_t)i * size1 + j]; }
    int size(int dimension) const { return dimension == 0 ? size0 : size1; }
};

template <class Base = HeadlessSimulation>
class FluidSim2DSimulation : public Base {
public:
    explicit FluidSim2DSimulation(int res_x = 128, int res_y = 192) : res_x_(res_x), res_y_(res_y) { init(); }
    ~FluidSim2DSimulation() override { if (sim_) fs2d_destroy(sim_); }

    void init() override {                                    // FluidSim.h:15-41
        if (!sim_) {
            const int rc = fs2d_create(res_x_, res_y_, -1, &sim_);
            if (rc != FS_OK) {
                std::string msg = sim_ ? fs2d_last_error(sim_) : "invalid argument";
                if (sim_) fs2d_destroy(sim_);
                sim_ = nullptr;
                throw std::runtime_error("fs2d_create: " + msg);
            }
        }
        double dt = 0;
        check(fs2d_get_info(sim_, &dt, nullptr, nullptr, nullptr, nullptr));
        this->m_dt = dt;
        this->reset();
    }
    bool advance() override {                                 // FluidSim.h:71-96
        push_params();
        check(fs2d_advance(sim_));
        this->m_time += this->m_dt;
        this->m_step++;
        return false;
    }
    void updateRenderGeometry() override { render_ = field(m_field); }   // FluidSim.h:52-69 (data side)
    void renderRenderGeometry(igl::opengl::glfw::Viewer&) override {}
    const Array2& renderField() const { return render_; }

    Array2 field(int which) {
        Array2 a;
        check(fs2d_field_size(sim_, which, &a.size0, &a.size1));
        a.data.resize((size_t)a.size0 * a.size1);
        check(fs2d_get_field(sim_, which, a.data.data(), a.data.size()));
        return a;
    }

    // FluidSim.h:258-277
    void selectField(int field) { m_field = field; }
    void setAccuracy(double acc) { m_acc = acc; }
    void setIteration(int iter) { m_iter = iter; }
    void setWind(bool w) { m_windOn = w; }
    void setMacCormack(bool m) { m_macOn = m; }
    void setPoissonOrdering(int ordering) { m_ordering = ordering; }   // FS_ORDER_LEXICOGRAPHIC (exact) / FS_ORDER_RED_BLACK
    int getField() const { return m_field; }
    int getResX() const { return res_x_; }
    int getResY() const { return res_y_; }
    double getAccuracy() const { return m_acc; }
    int getIteration() const { return m_iter; }
    bool getWind() const { return m_windOn; }
    bool getMacCormack() const { return m_macOn; }
    double getTimestep() const { return this->m_dt; }

protected:
    void resetMembers() override { check(fs2d_reset(sim_)); }   // FluidSim.h:43-50

private:
    void push_params() {
        check(fs2d_set_params(sim_, this->m_dt, m_acc, m_iter, m_windOn ? 1 : 0, m_macOn ? 1 : 0, m_ordering));
    }
    void check(int rc) const {
        if (rc != FS_OK) throw std::runtime_error(std::string("fluidsim_amd 2D: ") + fs2d_last_error(sim_));
    }
    int res_x_, res_y_;
    fs2d_sim* sim_ = nullptr;
    int m_field = 0, m_iter = 1000, m_ordering = FS_ORDER_LEXICOGRAPHIC;
    double m_acc = 1e-5;
    bool m_windOn = false, m_macOn = true;
    Array2 render_;
};

}  // namespace fsamd
