// Headless driver with the call order of the reference's main() (advection-reflection/main.cpp:
// 458-462, :506-509): FluidSim sim; sim.sphere_influence(RES_X/10); per frame
// fill_sphere_influence(RES_X/10, iter); update(1/30).  Writes u,v,w,density after the last step.
#include <cstdio>
#include <cstdlib>
#include <fsamd/FluidSim2D.hpp>
#include <fsamd/FluidSim3D.hpp>

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    const int steps = argc > 2 ? atoi(argv[2]) : 3;
    const char* out = argc > 3 ? argv[3] : "adapter_out.bin";
    try {
        fsamd::FluidSim3D sim(n, n, n);
        sim.sphere_influence(n / 10.0f);
        for (int iter = 0; iter < steps; ++iter) {
            sim.fill_sphere_influence(n / 10.0f, iter);
            sim.update(1.0f / 30.0f);
            printf("step %d sweeps %d change %.6g\n", iter, sim.stats().sweeps, sim.stats().metric);
        }
        FILE* f = fopen(out, "wb");
        std::vector<float> a;
        for (int k = 0; k < 3; ++k) { sim.get_field(k, a); fwrite(a.data(), 4, a.size(), f); }
        const std::vector<float>& d = sim.get_density();
        fwrite(d.data(), 4, d.size(), f);
        fclose(f);

        // the same scene through the Simulation interface
        fsamd::FluidSim3DSimulation<> s2(n, n, n);
        for (int iter = 0; iter < steps; ++iter) s2.advance();
        s2.updateRenderGeometry();
        const std::vector<float>& d2 = *s2.renderDensity();
        size_t diff = 0;
        for (size_t i = 0; i < d.size(); ++i) diff += d[i] != d2[i];
        std::vector<float> uc, vc, wc;
        const float mx = s2.solver().velocity_slice(n / 2, uc, vc, wc);
        printf("simulation-interface run: step %lu time %.6f density mismatches %zu slice max norm %.4f\n",
               s2.getStep(), s2.getTime(), diff, mx);
        s2.reset();
        printf("after reset: step %lu time %.1f\n", s2.getStep(), s2.getTime());
        // the 2D solver behind the same interface (3_fluid/main.cpp drives it through Simulator)
        fsamd::FluidSim2DSimulation<> s3(32, 48);
        for (int k = 0; k < 3; ++k) s3.advance();
        s3.selectField(0);
        s3.updateRenderGeometry();
        double mass = 0;
        for (double v : s3.renderField().data) mass += v;
        printf("2D run: step %lu time %.6f dt %.6f density sum %.9f u(16,10) %.9e\n", s3.getStep(), s3.getTime(),
               s3.getTimestep(), mass, s3.field(FS2D_U)(16, 10));
        return diff == 0 ? 0 : 2;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
