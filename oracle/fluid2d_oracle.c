/*
 * TEST INFRASTRUCTURE -- CPU oracle for the reference's 2D smoke solver.
 *
 * Plain-C restatement of class FluidSim in 3_fluid/FluidSim.h:11-303 and
 * 3_fluid/FluidSim.cpp:1-389 (fp64, like include/Array2T.h:99).  It is pinned
 * bit for bit against the reference itself: oracle/_ref/libref2d.so is the
 * reference's own FluidSim.cpp compiled in this container, and
 * tests/test_oracle2d.py compares every stage and whole advance() calls with
 * it, and with the golden vectors generated from it (tests/golden/ref2d_*.npz,
 * scripts/make_golden_2d.py) on machines where the reference is absent.
 *
 * Arrays use the reference's Array2T layout (include/Array2T.h:59-63):
 * element (i, j) at data[i*size1 + j] with i = x, so y is the fast axis.
 * Sizes: cell fields rx*ry, u (rx+1)*ry, v rx*(ry+1) (include/MACGrid2.h:14-15).
 *
 * The only extension is the red-black ordering of the pressure solve
 * (o2_solve_poisson_rb): same update, same residual, other visiting order.
 * Build with -O2 -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct O2 {
    int rx, ry;
    double dx, idx, dt, acc;
    int iter, wind_on, mac_on;
    double wind_r;            /* the static call counter of addWind (FluidSim.h:123-124) */
    double time;
    unsigned long step;
    int last_iters;
    double last_residual;
    double *d, *p, *div, *vort, *u, *v, *fx, *fy;
} O2;

#define C_(a, i, j) (a)[(size_t)(i) * o->ry + (j)]         /* cell fields and u: size1 = ry */
#define V_(a, i, j) (a)[(size_t)(i) * (o->ry + 1) + (j)]   /* v-like fields: size1 = ry + 1 */

static size_t n_c(const O2 *o) { return (size_t)o->rx * o->ry; }
static size_t n_u(const O2 *o) { return (size_t)(o->rx + 1) * o->ry; }
static size_t n_v(const O2 *o) { return (size_t)o->rx * (o->ry + 1); }

void o2_apply_source(O2 *o, double xmin, double xmax, double ymin, double ymax);

/* init() (FluidSim.h:15-41) with free resolution, then reset() */
O2 *o2_create(int rx, int ry) {
    O2 *o = (O2 *)calloc(1, sizeof(O2));
    o->rx = rx; o->ry = ry;
    const double size_x = rx;
    o->dx = size_x / rx;
    o->idx = rx / size_x;
    o->dt = 0.005 * sqrt((rx + ry) * 0.5);
    o->acc = 1e-5; o->iter = 1000; o->wind_on = 0; o->mac_on = 1;
    o->d = (double *)calloc(n_c(o), 8); o->p = (double *)calloc(n_c(o), 8);
    o->div = (double *)calloc(n_c(o), 8); o->vort = (double *)calloc(n_c(o), 8);
    o->u = (double *)calloc(n_u(o), 8); o->v = (double *)calloc(n_v(o), 8);
    o->fx = (double *)calloc(n_u(o), 8); o->fy = (double *)calloc(n_v(o), 8);
    o2_apply_source(o, 0.45, 0.55, 0.1, 0.15);   /* resetMembers, FluidSim.h:43-50 */
    return o;
}

void o2_destroy(O2 *o) {
    if (!o) return;
    free(o->d); free(o->p); free(o->div); free(o->vort); free(o->u); free(o->v); free(o->fx); free(o->fy);
    free(o);
}

double *o2_field(O2 *o, int which) {
    switch (which) {
        case 0: return o->d; case 1: return o->p; case 2: return o->div; case 3: return o->vort;
        case 4: return o->u; case 5: return o->v; case 6: return o->fx; default: return o->fy;
    }
}
int o2_size(O2 *o, int which, int dim) {
    if (which == 4 || which == 6) return dim == 0 ? o->rx + 1 : o->ry;
    if (which == 5 || which == 7) return dim == 0 ? o->rx : o->ry + 1;
    return dim == 0 ? o->rx : o->ry;
}
void o2_params(O2 *o, double dt, double acc, int iter, int wind, int mac) {
    if (dt > 0) o->dt = dt;
    if (acc > 0) o->acc = acc;
    if (iter > 0) o->iter = iter;
    o->wind_on = wind != 0; o->mac_on = mac != 0;
}
double o2_dt(O2 *o) { return o->dt; }
double o2_time(O2 *o) { return o->time; }
unsigned long o2_step(O2 *o) { return o->step; }
int o2_last_iters(O2 *o) { return o->last_iters; }
double o2_last_residual(O2 *o) { return o->last_residual; }

/* Simulation::reset + resetMembers (include/Simulation.h:26-30, FluidSim.h:43-50) */
void o2_reset(O2 *o) {
    o->time = 0; o->step = 0;
    memset(o->d, 0, n_c(o) * 8); memset(o->p, 0, n_c(o) * 8); memset(o->div, 0, n_c(o) * 8);
    memset(o->u, 0, n_u(o) * 8); memset(o->v, 0, n_v(o) * 8);
    memset(o->fx, 0, n_u(o) * 8); memset(o->fy, 0, n_v(o) * 8);
    o2_apply_source(o, 0.45, 0.55, 0.1, 0.15);
}

/* Grid2::applySource (include/Grid2.h:55-61) */
void o2_apply_source(O2 *o, double xmin, double xmax, double ymin, double ymax) {
    for (int y = (int)(ymin * o->ry); y < (int)(ymax * o->ry); y++)
        for (int x = (int)(xmin * o->rx); x < (int)(xmax * o->rx); x++) C_(o->d, x, y) = 1.0;
}

/* FluidSim.h:109-118 */
void o2_add_buoyancy(O2 *o) {
    const double scaling = 64.0 / o->rx;
    for (int i = 0; i < o->rx; ++i)
        for (int j = 1; j < o->ry + 1 - 1; ++j)
            V_(o->fy, i, j) += 0.1 * (C_(o->d, i, j - 1) + C_(o->d, i, j)) / 2.0 * scaling;
}

/* FluidSim.h:123-124: the reference's call counter is a function-level static, so one per process, shared by every
   FluidSim in it and never reset. The oracle keeps one per simulation (the same thing for a process that holds one
   FluidSim); a checker that drives several reference instances in one process carries the count over with this. */
void o2_set_wind_calls(O2 *o, double r) { o->wind_r = r; }

/* FluidSim.h:120-134 */
void o2_add_wind(O2 *o) {
    const double scaling = 64.0 / o->rx;
    o->wind_r += 1;
    const double fx = 2e-2 * cos(5e-2 * o->wind_r) * cos(3e-2 * o->wind_r) * scaling;
    for (size_t k = 0; k < n_u(o); ++k) o->fx[k] += fx;
}

/* FluidSim.h:136-148 */
void o2_add_force(O2 *o) {
    for (size_t k = 0; k < n_u(o); ++k) o->u[k] += o->dt * o->fx[k];
    for (size_t k = 0; k < n_v(o); ++k) o->v[k] += o->dt * o->fy[k];
}

/* FluidSim.h:169-187 */
void o2_set_neumann(O2 *o) {
    int sx = o->rx + 1, sy = o->ry;
    for (int y = 0; y < sy; ++y) {
        C_(o->u, 0, y) = C_(o->u, 2, y);
        C_(o->u, sx - 1, y) = C_(o->u, sx - 3, y);
    }
    sx = o->rx; sy = o->ry + 1;
    for (int x = 0; x < sx; ++x) {
        V_(o->v, x, 0) = V_(o->v, x, 2);
        V_(o->v, x, sy - 1) = V_(o->v, x, sy - 3);
    }
}

/* FluidSim.h:189-207 */
void o2_set_zero(O2 *o) {
    int sx = o->rx + 1, sy = o->ry;
    for (int x = 0; x < sx; ++x) {
        C_(o->u, x, 0) = 0;
        C_(o->u, x, sy - 1) = 0;
    }
    sx = o->rx; sy = o->ry + 1;
    for (int y = 0; y < sy; ++y) {
        V_(o->v, 0, y) = 0;
        V_(o->v, sx - 1, y) = 0;
    }
}

/* FluidSim.h:209-218 */
void o2_compute_divergence(O2 *o) {
    for (int y = 1; y < o->ry - 1; ++y)
        for (int x = 1; x < o->rx - 1; ++x) {
            const double dudx = (C_(o->u, x + 1, y) - C_(o->u, x, y)) * o->idx;
            const double dvdy = (V_(o->v, x, y + 1) - V_(o->v, x, y)) * o->idx;
            C_(o->div, x, y) = dudx + dvdy;
        }
}

/* FluidSim.h:220-229 */
void o2_compute_vorticity(O2 *o) {
    for (int y = 2; y < o->ry - 2; ++y)
        for (int x = 2; x < o->rx - 2; ++x) {
            const double dudy = (C_(o->u, x, y + 1) - C_(o->u, x, y - 1)) * o->idx * 0.5;
            const double dvdx = (V_(o->v, x + 1, y) - V_(o->v, x - 1, y)) * o->idx * 0.5;
            C_(o->vort, x, y) = dvdx - dudy;
        }
}

/* FluidSim.h:231-243 */
void o2_copy_border(O2 *o) {
    const int sx = o->rx, sy = o->ry;
    for (int y = 0; y < sy; ++y) {
        C_(o->p, 0, y) = C_(o->p, 1, y);
        C_(o->p, sx - 1, y) = C_(o->p, sx - 2, y);
    }
    for (int x = 0; x < sx; ++x) {
        C_(o->p, x, 0) = C_(o->p, x, 1);
        C_(o->p, x, sy - 1) = C_(o->p, x, sy - 2);
    }
}

static inline void poisson_visit(O2 *o, int x, int y, double dx2, double rho) {
    const double b = -C_(o->div, x, y) / o->dt * rho;
    C_(o->p, x, y) = (dx2 * b + C_(o->p, x - 1, y) + C_(o->p, x + 1, y) + C_(o->p, x, y - 1) + C_(o->p, x, y + 1)) / 4.0;
}

static double poisson_residual(O2 *o, double dx2, double rho) {
    double residual = 0;
    for (int y = 1; y < o->ry - 1; ++y)
        for (int x = 1; x < o->rx - 1; ++x) {
            const double b = -C_(o->div, x, y) / o->dt * rho;
            const double r = b - (4 * C_(o->p, x, y) - C_(o->p, x - 1, y) - C_(o->p, x + 1, y) - C_(o->p, x, y - 1) -
                                  C_(o->p, x, y + 1)) / dx2;
            residual += r * r;
        }
    residual = sqrt(residual);
    residual /= (o->rx - 2) * (o->ry - 2);
    return residual;
}

/* FluidSim.cpp:6-48 */
void o2_solve_poisson(O2 *o) {
    const double dx2 = o->dx * o->dx, rho = 1;
    double residual = o->acc + 1;
    int it;
    for (it = 0; residual > o->acc && it < o->iter; ++it) {
        for (int y = 1; y < o->ry - 1; ++y)
            for (int x = 1; x < o->rx - 1; ++x) poisson_visit(o, x, y, dx2, rho);
        residual = poisson_residual(o, dx2, rho);
    }
    o->last_iters = it; o->last_residual = residual;
}

/* extension: the same loop with the cells visited colour by colour ((x+y)&1 = 0 first) */
void o2_solve_poisson_rb(O2 *o) {
    const double dx2 = o->dx * o->dx, rho = 1;
    double residual = o->acc + 1;
    int it;
    for (it = 0; residual > o->acc && it < o->iter; ++it) {
        for (int colour = 0; colour < 2; ++colour)
            for (int y = 1; y < o->ry - 1; ++y)
                for (int x = 1; x < o->rx - 1; ++x)
                    if (((x + y) & 1) == colour) poisson_visit(o, x, y, dx2, rho);
        residual = poisson_residual(o, dx2, rho);
    }
    o->last_iters = it; o->last_residual = residual;
}

/* FluidSim.cpp:50-66 */
void o2_correct_velocity(O2 *o) {
    for (int y = 1; y < o->ry - 1; ++y)
        for (int x = 1; x < o->rx; ++x) C_(o->u, x, y) -= o->dt * (C_(o->p, x, y) - C_(o->p, x - 1, y)) * o->idx;
    for (int y = 1; y < o->ry; ++y)
        for (int x = 1; x < o->rx - 1; ++x) V_(o->v, x, y) -= o->dt * (C_(o->p, x, y) - C_(o->p, x, y - 1)) * o->idx;
}

/* FluidSim.h:150-167; rb selects the ordering of the Poisson sweeps */
void o2_solve_pressure(O2 *o, int rb) {
    o2_set_neumann(o);
    o2_set_zero(o);
    o2_compute_divergence(o);
    o2_copy_border(o);
    if (rb) o2_solve_poisson_rb(o); else o2_solve_poisson(o);
    o2_correct_velocity(o);
    o2_compute_vorticity(o);
    o2_compute_divergence(o);
}

static double clampd(double a, double lo, double hi) {   /* the if-chains of FluidSim.cpp:104-107 etc. */
    if (a < lo) a = lo;
    if (a > hi) a = hi;
    return a;
}

/* FluidSim.cpp:87-129: density from o->d into o->d, carried by (u, v); the temporary starts at zero */
static void advect_density_sl(O2 *o, const double *u, const double *v) {
    double *tmp = (double *)calloc(n_c(o), 8);
    for (int y = 1; y < o->ry - 1; ++y)
        for (int x = 1; x < o->rx - 1; ++x) {
            const double lxv = (C_(u, x, y) + C_(u, x + 1, y)) / 2;
            const double lyv = (V_(v, x, y) + V_(v, x, y + 1)) / 2;
            double lx = x - o->dt * o->idx * lxv;
            double ly = y - o->dt * o->idx * lyv;
            if (lx < 1) lx = 1;
            if (ly < 1) ly = 1;
            if (lx > o->rx - 2) lx = o->rx - 2;
            if (ly > o->ry - 2) ly = o->ry - 2;
            const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
            const double xw = lx - xl, yw = ly - yl;
            C_(tmp, x, y) = xw * yw * C_(o->d, xh, yh) + (1 - xw) * yw * C_(o->d, xl, yh) + xw * (1 - yw) * C_(o->d, xh, yl) +
                            (1 - xw) * (1 - yw) * C_(o->d, xl, yl);
        }
    memcpy(o->d, tmp, n_c(o) * 8);
    free(tmp);
}

/* FluidSim.cpp:131-215 */
static void advect_velocity_sl(O2 *o, const double *u, const double *v) {
    double *ut = (double *)calloc(n_u(o), 8), *vt = (double *)calloc(n_v(o), 8);
    for (int y = 1; y < o->ry - 1; ++y)
        for (int x = 1; x < o->rx; ++x) {
            const double lxv = C_(u, x, y);
            const double lyv = (V_(v, x, y) + V_(v, x - 1, y) + V_(v, x - 1, y + 1) + V_(v, x, y + 1)) / 4;
            double lx = x - o->dt * o->idx * lxv;
            double ly = y - o->dt * o->idx * lyv;
            if (lx < 1.5) lx = 1.5;
            if (ly < 1.5) ly = 1.5;
            if (lx > o->rx - 1.5) lx = o->rx - 1.5;
            if (ly > o->ry - 2.5) ly = o->ry - 2.5;
            const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
            const double xw = lx - xl, yw = ly - yl;
            C_(ut, x, y) = xw * yw * C_(o->u, xh, yh) + (1 - xw) * yw * C_(o->u, xl, yh) + xw * (1 - yw) * C_(o->u, xh, yl) +
                           (1 - xw) * (1 - yw) * C_(o->u, xl, yl);
        }
    for (int y = 1; y < o->ry; ++y)
        for (int x = 1; x < o->rx - 1; ++x) {
            const double lxv = (C_(u, x, y) + C_(u, x + 1, y) + C_(u, x + 1, y - 1) + C_(u, x, y - 1)) / 4;
            const double lyv = V_(v, x, y);
            double lx = x - o->dt * o->idx * lxv;
            double ly = y - o->dt * o->idx * lyv;
            if (lx < 1.5) lx = 1.5;
            if (ly < 1.5) ly = 1.5;
            if (lx > o->rx - 2.5) lx = o->rx - 2.5;
            if (ly > o->ry - 1.5) ly = o->ry - 1.5;
            const double xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
            const double xw = lx - xl, yw = ly - yl;
            V_(vt, x, y) = xw * yw * V_(o->v, (int)xh, (int)yh) + (1 - xw) * yw * V_(o->v, (int)xl, (int)yh) +
                           xw * (1 - yw) * V_(o->v, (int)xh, (int)yl) + (1 - xw) * (1 - yw) * V_(o->v, (int)xl, (int)yl);
        }
    memcpy(o->u, ut, n_u(o) * 8);
    memcpy(o->v, vt, n_v(o) * 8);
    free(ut); free(vt);
}

static double min4(double a, double b, double c, double d) {
    double m = 1e10;
    m = a < m ? a : m; m = b < m ? b : m; m = c < m ? c : m; m = d < m ? d : m;
    return m;
}
static double max4(double a, double b, double c, double d) {
    double m = -1e10;
    m = a > m ? a : m; m = b > m ? b : m; m = c > m ? c : m; m = d > m ? d : m;
    return m;
}

/* FluidSim.cpp:68-85, :217-389 */
void o2_advect_values(O2 *o) {
    double *d0 = (double *)malloc(n_c(o) * 8), *u0 = (double *)malloc(n_u(o) * 8), *v0 = (double *)malloc(n_v(o) * 8);
    memcpy(d0, o->d, n_c(o) * 8); memcpy(u0, o->u, n_u(o) * 8); memcpy(v0, o->v, n_v(o) * 8);
    advect_density_sl(o, u0, v0);
    advect_velocity_sl(o, u0, v0);
    if (o->mac_on) {
        double *df = (double *)malloc(n_c(o) * 8), *uf = (double *)malloc(n_u(o) * 8), *vf = (double *)malloc(n_v(o) * 8);
        memcpy(df, o->d, n_c(o) * 8); memcpy(uf, o->u, n_u(o) * 8); memcpy(vf, o->v, n_v(o) * 8);
        /* MacCormackUpdate: backward pass with dt -> -dt on the forward result */
        o->dt *= -1;
        advect_density_sl(o, u0, v0);
        advect_velocity_sl(o, u0, v0);
        double *db = o->d, *ub = o->u, *vb = o->v;   /* backward results live in the state arrays now */
        o->dt *= -1;
        double *dt_ = (double *)malloc(n_c(o) * 8), *ut = (double *)malloc(n_u(o) * 8), *vt = (double *)malloc(n_v(o) * 8);
        memcpy(dt_, df, n_c(o) * 8); memcpy(ut, uf, n_u(o) * 8); memcpy(vt, vf, n_v(o) * 8);
        for (int y = 1; y < o->ry - 1; ++y)
            for (int x = 1; x < o->rx - 1; ++x) C_(dt_, x, y) += 0.5 * (C_(d0, x, y) - C_(db, x, y));
        for (int y = 1; y < o->ry - 1; ++y)
            for (int x = 1; x < o->rx; ++x) C_(ut, x, y) += 0.5 * (C_(u0, x, y) - C_(ub, x, y));
        for (int y = 1; y < o->ry; ++y)
            for (int x = 1; x < o->rx - 1; ++x) V_(vt, x, y) += 0.5 * (V_(v0, x, y) - V_(vb, x, y));
        /* MacCormackClamp: revert to the forward value outside the min/max of the four source corners */
        for (int y = 1; y < o->ry - 1; ++y)
            for (int x = 1; x < o->rx - 1; ++x) {
                const double lxv = (C_(u0, x, y) + C_(u0, x + 1, y)) / 2, lyv = (V_(v0, x, y) + V_(v0, x, y + 1)) / 2;
                double lx = x - o->dt * o->idx * lxv, ly = y - o->dt * o->idx * lyv;
                if (lx < 1) lx = 1;
                if (ly < 1) ly = 1;
                if (lx > o->rx - 2) lx = o->rx - 2;
                if (ly > o->ry - 2) ly = o->ry - 2;
                const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
                const double mn = min4(C_(d0, xl, yl), C_(d0, xl, yh), C_(d0, xh, yl), C_(d0, xh, yh));
                const double mx = max4(C_(d0, xl, yl), C_(d0, xl, yh), C_(d0, xh, yl), C_(d0, xh, yh));
                if (C_(dt_, x, y) < mn || C_(dt_, x, y) > mx) C_(dt_, x, y) = C_(df, x, y);
            }
        for (int y = 1; y < o->ry - 1; ++y)
            for (int x = 1; x < o->rx; ++x) {
                const double lxv = C_(u0, x, y);
                const double lyv = (V_(v0, x, y) + V_(v0, x - 1, y) + V_(v0, x - 1, y + 1) + V_(v0, x, y + 1)) / 4;
                double lx = x - o->dt * o->idx * lxv, ly = y - o->dt * o->idx * lyv;
                lx = clampd(lx, 1.5, o->rx - 1.5);   /* same two-sided if-chain, low test first */
                if (ly < 1.5) ly = 1.5;
                if (ly > o->ry - 2.5) ly = o->ry - 2.5;
                const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
                const double mn = min4(C_(u0, xl, yl), C_(u0, xl, yh), C_(u0, xh, yl), C_(u0, xh, yh));
                const double mx = max4(C_(u0, xl, yl), C_(u0, xl, yh), C_(u0, xh, yl), C_(u0, xh, yh));
                if (C_(ut, x, y) < mn || C_(ut, x, y) > mx) C_(ut, x, y) = C_(uf, x, y);
            }
        for (int y = 1; y < o->ry; ++y)
            for (int x = 1; x < o->rx - 1; ++x) {
                const double lxv = (C_(u0, x, y) + C_(u0, x + 1, y) + C_(u0, x + 1, y - 1) + C_(u0, x, y - 1)) / 4;
                const double lyv = V_(v0, x, y);
                double lx = x - o->dt * o->idx * lxv, ly = y - o->dt * o->idx * lyv;
                if (lx < 1.5) lx = 1.5;
                if (ly < 1.5) ly = 1.5;
                if (lx > o->rx - 2.5) lx = o->rx - 2.5;
                if (ly > o->ry - 1.5) ly = o->ry - 1.5;
                const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
                const double mn = min4(V_(v0, xl, yl), V_(v0, xl, yh), V_(v0, xh, yl), V_(v0, xh, yh));
                const double mx = max4(V_(v0, xl, yl), V_(v0, xl, yh), V_(v0, xh, yl), V_(v0, xh, yh));
                if (V_(vt, x, y) < mn || V_(vt, x, y) > mx) V_(vt, x, y) = V_(vf, x, y);
            }
        memcpy(o->d, dt_, n_c(o) * 8); memcpy(o->u, ut, n_u(o) * 8); memcpy(o->v, vt, n_v(o) * 8);
        free(df); free(uf); free(vf); free(dt_); free(ut); free(vt);
    }
    free(d0); free(u0); free(v0);
}

/* advance() (FluidSim.h:71-96); rb selects the ordering of the Poisson sweeps */
void o2_advance(O2 *o, int rb) {
    o2_apply_source(o, 0.45, 0.55, 0.1, 0.15);
    o2_add_buoyancy(o);
    if (o->wind_on) o2_add_wind(o);
    o2_add_force(o);
    o2_solve_pressure(o, rb);
    o2_advect_values(o);
    memset(o->fx, 0, n_u(o) * 8); memset(o->fy, 0, n_v(o) * 8);
    o->time += o->dt;
    o->step++;
}

/* single stages, numbered like ref2d_stage() of oracle/ref2d_shim.cpp */
void o2_stage(O2 *o, int op) {
    switch (op) {
        case 0: o2_apply_source(o, 0.45, 0.55, 0.1, 0.15); break;
        case 1: o2_add_buoyancy(o); break;
        case 2: o2_add_wind(o); break;
        case 3: o2_add_force(o); break;
        case 4: o2_solve_pressure(o, 0); break;
        case 5: o2_advect_values(o); break;
        case 6: o2_set_neumann(o); break;
        case 7: o2_set_zero(o); break;
        case 8: o2_compute_divergence(o); break;
        case 9: o2_copy_border(o); break;
        case 10: o2_solve_poisson(o); break;
        case 11: o2_correct_velocity(o); break;
        case 12: o2_compute_vorticity(o); break;
        case 13: memset(o->fx, 0, n_u(o) * 8); memset(o->fy, 0, n_v(o) * 8); break;
        case 14: o2_solve_pressure(o, 1); break;
        case 15: o2_solve_poisson_rb(o); break;
    }
}

void o2_get(O2 *o, int which, double *out) {
    memcpy(out, o2_field(o, which), (size_t)o2_size(o, which, 0) * o2_size(o, which, 1) * 8);
}
void o2_set(O2 *o, int which, const double *in) {
    memcpy(o2_field(o, which), in, (size_t)o2_size(o, which, 0) * o2_size(o, which, 1) * 8);
}
