// C++ host side of the 3D fluid step over the C ABI (fluidsim_amd.h).
//
//  fsamd::FluidSim3D            the public surface of the reference's 3D solver, class FluidSim of
//                               advection-reflection/main.cpp:21-442, method for method:
//                               sphere_influence, fill_sphere_influence, bottom_influence, clear,
//                               get_density, update, gauss_seidel_projection, semi_lagrange, and the
//                               slice read of draw_velocity_field (main.cpp:135-152).
//  fsamd::FluidSim3DSimulation  the same solver behind the reference's Simulation interface
//                               (include/Simulation.h:11-78): reset / advance / setTimestep / getTime /
//                               getStep.  Base is the reference's `Simulation` when the caller builds
//                               against the reference tree, fsamd::HeadlessSimulation otherwise.
//
// Errors: the reference's methods cannot fail; these throw std::runtime_error carrying
// fs3d_last_error() when the device does.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fluidsim_amd.h"
#include "SimulationInterface.hpp"

namespace fsamd {

class FluidSim3D {
public:
    FluidSim3D(int res_x, int res_y, int res_z, const fs3d_config* cfg = nullptr) : nx_(res_x), ny_(res_y), nz_(res_z) {
        const int rc = fs3d_create(res_x, res_y, res_z, cfg, &sim_);
        if (rc != FS_OK) {
            std::string msg = sim_ ? fs3d_last_error(sim_) : "invalid argument";
            if (sim_) fs3d_destroy(sim_);
            sim_ = nullptr;
            throw std::runtime_error("fs3d_create: " + msg);
        }
        density_.resize(cells());
    }
    ~FluidSim3D() { if (sim_) fs3d_destroy(sim_); }
    FluidSim3D(const FluidSim3D&) = delete;
    FluidSim3D& operator=(const FluidSim3D&) = delete;

    size_t cells() const { return (size_t)nx_ * ny_ * nz_; }
    int res_x() const { return nx_; }
    int res_y() const { return ny_; }
    int res_z() const { return nz_; }
    fs3d_sim* handle() { return sim_; }

    // scene, main.cpp:44-96
    void sphere_influence(float radius) { check(fs3d_sphere_influence(sim_, radius)); }
    void fill_sphere_influence(float radius, int iter) { check(fs3d_fill_sphere_influence(sim_, radius, iter)); }
    void bottom_influence() { check(fs3d_bottom_influence(sim_)); }
    void clear() { check(fs3d_clear(sim_)); }

    // main.cpp:98-100: x-fastest array of the whole grid, valid until the next call on this object;
    // this is what Renderer::bind_fluid_shader_data uploads with glTexImage3D (Renderer.h:205-241)
    const std::vector<float>& get_density() {
        check(fs3d_get_field(sim_, FS_FIELD_DENSITY, density_.data(), density_.size()));
        return density_;
    }

    // main.cpp:102-130
    void update(float t_step) { check(fs3d_update(sim_, t_step)); }

    // main.cpp:189-254; returns the L1 change of the sweep
    float gauss_seidel_projection() {
        double change = 0;
        check(fs3d_gauss_seidel_projection(sim_, &change));
        return (float)change;
    }

    // main.cpp:256-291, applied to a snapshot of the current state as update() does
    void semi_lagrange(float t_step) { check(fs3d_semi_lagrange(sim_, t_step)); }

    // The data draw_velocity_field(..., depth) reads (main.cpp:135-152): cell-centred velocity of the
    // interior cells of plane `depth`, (u[i]+u[x+1])/2 etc., and the largest norm on the plane.
    // Output arrays hold nx*ny entries (shell entries are zero).
    float velocity_slice(int depth, std::vector<float>& uc, std::vector<float>& vc, std::vector<float>& wc) {
        const size_t plane = (size_t)nx_ * ny_;
        std::vector<float> u(plane), v(plane), w(plane), w_up(plane);
        check(fs3d_get_field_plane(sim_, FS_FIELD_U, depth, u.data(), plane));
        check(fs3d_get_field_plane(sim_, FS_FIELD_V, depth, v.data(), plane));
        check(fs3d_get_field_plane(sim_, FS_FIELD_W, depth, w.data(), plane));
        check(fs3d_get_field_plane(sim_, FS_FIELD_W, depth + 1, w_up.data(), plane));
        uc.assign(plane, 0.f); vc.assign(plane, 0.f); wc.assign(plane, 0.f);
        float max_norm = 0.00001f;   // main.cpp:133
        for (int y = 1; y < ny_ - 1; ++y)
            for (int x = 1; x < nx_ - 1; ++x) {
                const size_t i = (size_t)y * nx_ + x;
                uc[i] = (u[i] + u[i + 1]) / 2.0f;
                vc[i] = (v[i] + v[i + nx_]) / 2.0f;
                wc[i] = (w[i] + w_up[i]) / 2.0f;
                max_norm = std::max(max_norm, std::sqrt(uc[i] * uc[i] + vc[i] * vc[i] + wc[i] * wc[i]));
            }
        return max_norm;
    }

    void get_field(int which, std::vector<float>& out) {
        out.resize(cells());
        check(fs3d_get_field(sim_, which, out.data(), out.size()));
    }
    fs3d_stats stats() { fs3d_stats st; check(fs3d_get_stats(sim_, &st)); return st; }
    fs3d_diag diagnostics() { fs3d_diag d; check(fs3d_diagnostics(sim_, &d)); return d; }
    // the density averaged over factor^3 blocks, for a renderer that cannot take the whole grid per frame
    void get_density_downsampled(int factor, std::vector<float>& out) {
        auto blocks = [factor](int n) { return (size_t)((n + factor - 1) / factor); };
        out.resize(blocks(nx_) * blocks(ny_) * blocks(nz_));
        check(fs3d_get_density_downsampled(sim_, factor, out.data(), out.size()));
    }
    void save(const std::string& path) { check(fs3d_save(sim_, path.c_str())); }
    void load(const std::string& path) { check(fs3d_load(sim_, path.c_str())); }
    void reset() { check(fs3d_reset(sim_)); }
    void advance(double dt) { check(fs3d_advance(sim_, dt)); }

private:
    void check(int rc) const {
        if (rc != FS_OK) throw std::runtime_error(std::string("fluidsim_amd: ") + fs3d_last_error(sim_));
    }
    int nx_, ny_, nz_;
    fs3d_sim* sim_ = nullptr;
    std::vector<float> density_;
};

// The vortex-ring scene of the reference's main() (main.cpp:462, :506-509) behind the Simulation
// interface.  advance() never reports "finished" (3_fluid/FluidSim.h:95).
template <class Base = HeadlessSimulation>
class FluidSim3DSimulation : public Base {
public:
    FluidSim3DSimulation(int res_x, int res_y, int res_z, const fs3d_config* cfg = nullptr)
        : sim_(res_x, res_y, res_z, cfg) {
        init();
    }
    void init() override {
        this->m_dt = 1.0 / 30.0;   // the cap of main.cpp:509
        this->reset();
    }
    bool advance() override {
        sim_.fill_sphere_influence(sim_.res_x() / 10.0f, (int)this->m_step);   // main.cpp:507
        sim_.advance(this->m_dt);
        this->m_time += this->m_dt;
        this->m_step++;
        return false;
    }
    void updateRenderGeometry() override { density_ = &sim_.get_density(); }
    void renderRenderGeometry(igl::opengl::glfw::Viewer&) override {}   // drawing is the renderer's job
    const std::vector<float>* renderDensity() const { return density_; }
    FluidSim3D& solver() { return sim_; }

protected:
    void resetMembers() override {
        sim_.reset();
        sim_.sphere_influence(sim_.res_x() / 10.0f);   // main.cpp:462
        density_ = nullptr;
    }

private:
    FluidSim3D sim_;
    const std::vector<float>* density_ = nullptr;
};

}  // namespace fsamd
