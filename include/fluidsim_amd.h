/*
 * fluidsim_amd.h -- C ABI of the MI355X (gfx950) fluid step.
 *
 * The reference has no FFI boundary: its 3D solver is the C++ class FluidSim
 * in advection-reflection/main.cpp:21-442 and its 2D solver the class FluidSim
 * in 3_fluid/FluidSim.h:11-303 behind the abstract Simulation interface
 * (include/Simulation.h:11-78).  Each entry point below names the reference
 * member function it replaces.  The C++ adapters that give those classes'
 * method names back to a C++ caller live in include/fsamd/.
 *
 * Conventions
 *  - every function returns FS_OK (0) or a negative FS_E* code; the text of the
 *    last failure on a handle is returned by fs3d_last_error();
 *  - fields are dense arrays of nx*ny*nz scalars, x fastest:
 *    i = (z*ny + y)*nx + x (main.cpp:319-325).  u[i], v[i], w[i] are the
 *    velocities on the low-x / low-y / low-z face of cell i, in cells per
 *    second (main.cpp:137-139, :217-219);
 *  - host pointers only; the library owns all device memory and its stream;
 *  - a handle may be used from different host threads over its life, but
 *    calls on one handle must not overlap (the discipline of the reference's
 *    Simulator thread, include/Simulator.h:172-231).  Every entry point
 *    selects the handle's device itself.
 */
#ifndef FLUIDSIM_AMD_H
#define FLUIDSIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FS_OK = 0,
    FS_EINVAL = -1,   /* bad argument                                    */
    FS_EHIP = -2,     /* a HIP runtime call failed                       */
    FS_ENOMEM = -3,   /* device or host allocation failed                */
    FS_ECOMM = -4,    /* an RCCL call failed or the communicator is bad  */
    FS_ESTATE = -5    /* call not valid in the handle's current state    */
};

/* arithmetic type of the fields */
enum { FS_F32 = 0, FS_F64 = 1 };

/* cell types, same values as enum CellType (main.cpp:294-298) */
enum { FS_CELL_FLUID = 0, FS_CELL_OBJECT = 1, FS_CELL_STATIC_FLUID = 2 };

/* field selectors for fs3d_set_field / fs3d_get_field */
enum { FS_FIELD_U = 0, FS_FIELD_V = 1, FS_FIELD_W = 2, FS_FIELD_DENSITY = 3, FS_FIELD_DIVERGENCE = 4,
       FS_FIELD_VORTICITY = 5 /* |curl u|, the 3D form of computeVorticity, 3_fluid/FluidSim.h:220-229 */ };

/* visiting order of the projection sweep */
enum {
    FS_ORDER_RED_BLACK = 0,     /* two half-sweeps, colour (x+y+z)&1 = 0 then 1          */
    FS_ORDER_LEXICOGRAPHIC = 1  /* the reference's order (main.cpp:235), run as
                                   diagonals of 16^3-cell blocks, one launch each;
                                   bit-identical to the serial sweep, an order of
                                   magnitude slower than red-black: for parity runs */
};

/* stopping rule of the projection loop */
enum {
    FS_STOP_FIXED = 0,       /* exactly max_iters sweeps                                         */
    FS_STOP_L1_CHANGE = 1,   /* until the L1 change of one sweep <= tol (main.cpp:105-111)       */
    FS_STOP_L2_RESIDUAL = 2  /* until sqrt(sum r^2)/#interior <= tol with r = -div/dt, the residual of
                                3_fluid/FluidSim.cpp:17-43 at dx = rho = 1; dt is the step length given
                                to fs3d_advance / fs3d_update, and 1 for a bare fs3d_project          */
};

/* special values of fs3d_config.check_every */
enum {
    FS_CHECK_ADAPTIVE = 0,   /* first evaluation one sweep before the count the same projection of the
                                previous step needed, then after every sweep; a projection that runs
                                past that count thins its evaluations out (by the excess, and with
                                omega == 1 by the decay rate two failed evaluations show), so a capped
                                projection may overshoot its stopping sweep by a few sweeps           */
    FS_CHECK_DEFERRED = -1   /* as FS_CHECK_ADAPTIVE, but inside fs3d_advance the last evaluation of a
                                projection is left to the advection that follows (which reads the
                                projected field anyway and adds up its L2 residual on the way): the
                                sweep the previous step's count expects to be the last runs unverified
                                (a projection that needed one sweep runs that sweep and nothing else).
                                If the advection finds the residual above tol, the advected field is
                                dropped, the projection goes on with evaluations of its own and the
                                advection is repeated (fs3d_stats.redone counts these).  Every
                                projection still ends on a verified residual <= tol.  Needs
                                FS_STOP_L2_RESIDUAL, red-black order, FS_ADVECT_TILED and semi-Lagrange
                                (under FS_SOLVER_MULTIGRID only projections of one sweep defer);
                                otherwise it acts as FS_CHECK_ADAPTIVE                              */
};

/* implementation of one red-black sweep (same result bit for bit) */
enum {
    FS_SWEEP_FUSED = 0,      /* both colours in one pass over HBM, out of place: the register-resident
                                kernel when nx is even, else the LDS-window kernel; on a single
                                domain with an even nx, two consecutive sweeps that no evaluation
                                of the stopping rule separates run as ONE pass over HBM            */
    FS_SWEEP_TWO_PASS = 1,   /* one in-place launch per colour                                     */
    FS_SWEEP_FUSED_LDS = 2   /* both colours in one pass, always the LDS-window kernel             */
};

/* implementation of the semi-Lagrangian advection (same result bit for bit) */
enum {
    FS_ADVECT_TILED = 0,     /* fields staged through LDS plane by plane; samples whose backtrace
                                leaves the staged window (|v| dt >= 1 cell) use direct loads       */
    FS_ADVECT_DIRECT = 1     /* every tap a global load                                            */
};

/* advection scheme */
enum {
    FS_ADVECTION_SEMI_LAGRANGE = 0, /* main.cpp:256-291                                          */
    FS_ADVECTION_MACCORMACK = 1     /* semi-Lagrange forward + the backward pass, correction and
                                       limiter of 3_fluid/FluidSim.cpp:217-389 carried over to 3D
                                       (not in the 3D reference); the density is then carried by the
                                       plain face average of the carrying velocity                  */
};

/* how a projection reduces the divergence between two evaluations of the stopping rule */
enum {
    FS_SOLVER_SWEEPS = 0,    /* Gauss-Seidel sweeps only: the reference (main.cpp:105-111)          */
    FS_SOLVER_MULTIGRID = 1  /* V-cycles: 2 red-black sweeps, a coarse-grid correction of the same
                                gradient form (levels of 2x2x2 blocks, red-black Gauss-Seidel on
                                each, piecewise-constant transfer), 2 sweeps; the stopping rule is
                                evaluated after every cycle and a cycle counts as 4 sweeps.  Same
                                fixed point as the sweeps; red-black ordering.  With z-slabs every
                                slab must start on an even plane; each rank then holds and solves
                                the coarse levels of the whole grid (the fine level stays
                                distributed), with the single-domain result bit for bit        */
};

/* step variants of fs3d_advance */
enum {
    FS_STEP_REFERENCE = 0,   /* project, then self-advect: FluidSim::update (main.cpp:102-130)   */
    FS_STEP_REFLECTION = 1   /* advection-reflection (Zehnder et al. 2018, Alg. 1); absent from
                                the reference's code although its README names it               */
};

typedef struct fs3d_config {
    int32_t device;       /* HIP device ordinal; -1 = the calling thread's current device        */
    int32_t dtype;        /* FS_F32 (the reference's precision) or FS_F64                         */
    int32_t ordering;     /* FS_ORDER_*                                                           */
    int32_t stop_mode;    /* FS_STOP_*                                                            */
    int32_t max_iters;    /* sweep cap; the reference uses 1000 (main.cpp:108)                    */
    int32_t check_every;  /* evaluate the stopping metric every n-th sweep; or FS_CHECK_ADAPTIVE (0),
                             FS_CHECK_DEFERRED (-1)                                                  */
    int32_t step_mode;    /* FS_STEP_*                                                            */
    int32_t local_slabs;  /* >= 1: cut this handle's planes into that many z-slabs on its own GPU
                             (exercises the multi-GPU slab path inside one process)            */
    int32_t sweep_kernel; /* FS_SWEEP_*                                                           */
    int32_t advect_kernel;/* FS_ADVECT_*                                                          */
    int32_t advection;    /* FS_ADVECTION_*, the scheme of fs3d_advance (fs3d_update and
                             fs3d_semi_lagrange stay the reference's semi-Lagrange)                */
    int32_t solver;       /* FS_SOLVER_*                                                          */
    double tol;           /* the reference uses 5.0 with FS_STOP_L1_CHANGE (main.cpp:108)         */
    double omega;         /* over-relaxation factor of the sweeps; 1.0 = the reference's plain
                             Gauss-Seidel (multigrid cycles always smooth with 1.0)               */
    double buoyancy;      /* 0 = off; else the 0.1 of 3_fluid/FluidSim.h:115 scaled by this value */
} fs3d_config;

typedef struct fs3d_stats {
    int64_t step;            /* steps performed since reset (Simulation::getStep)                */
    double time;             /* simulated time since reset (Simulation::getTime)                 */
    int32_t sweeps;          /* sweeps of the last projection                                    */
    int32_t sweeps_step;     /* sweeps of all projections of the last step                       */
    double metric;           /* last evaluated stopping metric (change or residual)              */
    double ms_project;       /* device time of the last step up to the end of its last projection,
                                milliseconds (in the reflection step the first half advection lies
                                between the two projections and is included)                      */
    double ms_advect;        /* device time of the last step's last advection (+ reflection)      */
    double ms_step;          /* device time of the whole last step                               */
    int32_t evaluations_step;/* passes over the velocity that evaluated the stopping metric in the last
                                step (an evaluation folded into an advection is not a pass)         */
    int32_t redone;          /* advections repeated since reset because a deferred evaluation found
                                the tolerance not met (FS_CHECK_DEFERRED)                           */
    int64_t fused;           /* projections of one sweep since reset that ran inside the launch of the
                                advection after them (see fs3d_sweep_advect)                        */
    int64_t sweeps_total;    /* sweeps of all projections of fs3d_advance since reset               */
    int64_t evaluations_total; /* the same for evaluations_step: a caller that times many steps reads
                                the two totals once at the end instead of asking after every step   */
} fs3d_stats;

typedef struct fs3d_sim fs3d_sim;

/* library-wide ---------------------------------------------------------- */
const char *fsamd_version(void);
/* Fills cfg with the reference's behaviour: f32, red-black order, L1-change
 * stop at 5.0 capped at 1000 sweeps checked every sweep, omega 1, reference step. */
void fs3d_default_config(fs3d_config *cfg);

/* lifetime -------------------------------------------------------------- */
/* FluidSim::FluidSim (main.cpp:23-42): zero fields, one-cell Object shell.  */
int fs3d_create(int nx, int ny, int nz, const fs3d_config *cfg, fs3d_sim **out);
void fs3d_destroy(fs3d_sim *sim);
/* Simulation::reset (include/Simulation.h:26-30): zero the fields, rebuild the
 * shell mask, forget emitters, time = 0, step = 0.                         */
int fs3d_reset(fs3d_sim *sim);
int fs3d_set_config(fs3d_sim *sim, const fs3d_config *cfg); /* device and dtype must not change */
int fs3d_get_config(const fs3d_sim *sim, fs3d_config *cfg);
const char *fs3d_last_error(const fs3d_sim *sim);

/* state access (the reference hands out m_density by reference, main.cpp:98-100,
 * and reads m_vu/m_vv/m_vw slices in draw_velocity_field, :132-187) -------- */
int fs3d_set_field(fs3d_sim *sim, int which, const void *host, size_t count);
int fs3d_get_field(fs3d_sim *sim, int which, void *host, size_t count);
/* One z-plane of a field (count = nx*ny), the access pattern of
 * draw_velocity_field(..., depth) (main.cpp:135-152).                       */
int fs3d_get_field_plane(fs3d_sim *sim, int which, int z, void *host, size_t count);
/* m_s (main.cpp:311).  The outermost shell is always forced to Object.     */
int fs3d_set_cell_types(fs3d_sim *sim, const uint8_t *host, size_t count);
int fs3d_get_cell_types(fs3d_sim *sim, uint8_t *host, size_t count);

/* scene (main.cpp:44-96) ------------------------------------------------- */
/* sphere_influence(radius): emitter centred (nx/2, radius, nz/2), jet 20 along +y. */
int fs3d_sphere_influence(fs3d_sim *sim, float radius);
/* the same emitter with free centre and signed jet strength along y         */
int fs3d_sphere_emitter(fs3d_sim *sim, float radius, float cx, float cy, float cz, float strength);
/* Interior obstacle: the cells inside the sphere (same distance test as main.cpp:50-51) become Object.
 * The reference's mask supports interior Object cells (main.cpp:199, :210-215) but never creates any. */
int fs3d_solid_sphere(fs3d_sim *sim, float radius, float cx, float cy, float cz);
/* fill_sphere_influence(radius, iter): re-stamps the density inside the sphere
 * centred (nx/2, radius, nz/2) when iter % 100 == 0 (main.cpp:60-74).       */
int fs3d_fill_sphere_influence(fs3d_sim *sim, float radius, int iter);
/* the same re-stamp for an emitter with free centre                         */
int fs3d_fill_sphere(fs3d_sim *sim, float radius, float cx, float cy, float cz, int iter);
int fs3d_bottom_influence(fs3d_sim *sim); /* main.cpp:76-87 */
int fs3d_clear(fs3d_sim *sim);            /* main.cpp:89-96 */
/* Smoke source box in fractions of the resolution, density = 1 inside
 * (Grid2::applySource, include/Grid2.h:55-61, extended by a z range).       */
int fs3d_apply_source(fs3d_sim *sim, double xmin, double xmax, double ymin, double ymax, double zmin,
                      double zmax);

/* the step ---------------------------------------------------------------- */
/* gauss_seidel_projection() (main.cpp:189-254): one sweep in the configured
 * order; *change receives the L1 norm of the sweep's net velocity update.   */
int fs3d_gauss_seidel_projection(fs3d_sim *sim, double *change);
/* n sweeps without evaluating any metric (benchmark / smoother use).        */
int fs3d_sweeps(fs3d_sim *sim, int n);
/* Benchmark support: n sweeps back to back on the handle's stream, bracketed by
 * two events on that stream; *ms receives the device time between them (the
 * reference times whole frames with std::chrono, main.cpp:502, :541-543).    */
int fs3d_time_sweeps(fs3d_sim *sim, int n, double *ms);
/* The same for passes of two sweeps each (what fs3d_project and fs3d_sweeps use between evaluations of the
 * stopping rule on a single domain with FS_SWEEP_FUSED and an even nx: one launch applies two full red-black
 * sweeps in one pass over HBM, bit for bit two launches of one sweep); FS_ESTATE where that kernel does not apply. */
int fs3d_time_sweep_pairs(fs3d_sim *sim, int npairs, double *ms);
/* The yardstick of those two: n times the three velocity fields streamed from the current generation of the arrays
 * to the other one (12 bytes read and 12 written per cell, what any pass of a sweep has to move at least), timed the
 * same way.  The state is not changed.  No counterpart in the reference.                                             */
int fs3d_time_field_copies(fs3d_sim *sim, int n, double *ms);
/* The same for the advection: n advections of length t_step back to back (reflected == 0: semi_lagrange of
 * the current state, main.cpp:256-291; else the second half advection of the advection-reflection step,
 * which carries the reflection of a copy of the current velocity), *ms = device time between the events.   */
int fs3d_time_advections(fs3d_sim *sim, int n, double t_step, int reflected, double *ms);
/* The projection loop of update() (main.cpp:105-111) under the configured
 * stopping rule; outputs may be NULL.                                       */
int fs3d_project(fs3d_sim *sim, int *sweeps, double *metric);
/* One V-cycle of FS_SOLVER_MULTIGRID on the current velocity (no counterpart in the
 * reference; what fs3d_project repeats under that solver).                   */
int fs3d_mg_vcycle(fs3d_sim *sim);
/* L2 residual of the current field and max |div| (3_fluid/FluidSim.cpp:25-43). */
int fs3d_residual(fs3d_sim *sim, double *l2, double *max_abs_div);
/* semi_lagrange(t_step, snapshot...) (main.cpp:256-291) applied to a snapshot
 * of the current state, as update() calls it (main.cpp:122-129).            */
int fs3d_semi_lagrange(fs3d_sim *sim, double t_step);
/* One sweep of gauss_seidel_projection() (main.cpp:189-254, red-black order) and semi_lagrange(t_step)
 * (main.cpp:256-291) of the swept field in ONE launch: the swept velocity lives in LDS only, so the launch
 * moves 33 instead of 58 bytes per cell.  On MI355X it is bound by instruction issue and takes longer than the two
 * launches (1.9 against 1.65 ms at 512^3), so fs3d_advance uses it only when the environment says FSAMD_FUSED_STEP=1
 * (then under FS_CHECK_DEFERRED wherever a projection needed a single sweep in the previous two steps); the result is
 * the same bit for bit either way.  *residual_l2 (may be NULL) receives the L2
 * residual of the swept field, as fs3d_residual would report it between the two.  *served = 0 means that some
 * cell's backtrace left the staged window (|v| t_step of a cell or more) or was not finite: the launch cannot
 * serve such a cell, the state is then left as it was before the call and the caller runs fs3d_sweeps(1) and
 * fs3d_semi_lagrange instead (same result, bit for bit, where both apply).  FS_ESTATE unless FS_F32, red-black
 * order, FS_SWEEP_FUSED, FS_ADVECT_TILED, a single domain.  No counterpart in the reference.               */
int fs3d_sweep_advect(fs3d_sim *sim, double t_step, double *residual_l2, int *served);
/* update(t_step) (main.cpp:102-130) = fs3d_project + fs3d_semi_lagrange.    */
int fs3d_update(fs3d_sim *sim, double t_step);
/* Simulation::advance (include/Simulation.h:47): one step of the configured
 * step_mode of length dt, then time += dt, step += 1.                       */
int fs3d_advance(fs3d_sim *sim, double dt);
int fs3d_get_stats(fs3d_sim *sim, fs3d_stats *out);
/* Blocks until all work queued on the handle has finished.  Local: a rank may call it alone.  On a slab
 * handle an advection whose backtrace needed a plane beyond the two halo planes (|v| dt > 2 cells at a slab
 * interface) is not waited for when it is queued: this call returns FS_ESTATE for it if this rank's own
 * kernels raised the flag or if the last reduction over the ranks reported it (until fs3d_reset or
 * fs3d_load).                                                                                            */
int fs3d_synchronize(fs3d_sim *sim);
/* The same, collective on a handle of fs3d_create_slab with several ranks (every rank must call it): the
 * out-of-halo flag is added up over the ranks, so every rank returns FS_ESTATE at the same call, whichever
 * rank's backtrace left its halo.  Any projection that evaluates its stopping metric reports the flag the
 * same way.  No counterpart in the reference (single process).                                          */
int fs3d_synchronize_all(fs3d_sim *sim);

/* Diagnostics of the current state (the reference prints none; they are what shows
 * what the reflection preserves).  Cell-centred velocities as in
 * draw_velocity_field (main.cpp:138-142), interior cells that are not Object;
 * enstrophy with the stencil of computeVorticity (3_fluid/FluidSim.h:220-229).    */
typedef struct fs3d_diag {
    double kinetic_energy;   /* sum (u^2 + v^2 + w^2) / 2                                       */
    double max_speed;        /* largest cell-centred velocity norm                              */
    double density_sum;      /* sum of the density                                              */
    double enstrophy;        /* sum |curl u|^2                                                  */
} fs3d_diag;
int fs3d_diagnostics(fs3d_sim *sim, fs3d_diag *out);
/* The density averaged over blocks of factor^3 cells, x fastest, ceil(n/factor)
 * blocks per axis: the per-frame fetch for a renderer (Renderer.h:205-241 uploads
 * the whole grid) when the full field is too large to move every frame.  On a slab
 * handle the owned planes, whose first plane must be a multiple of factor.        */
int fs3d_get_density_downsampled(fs3d_sim *sim, int factor, void *host, size_t count);
/* Restart files (the reference has none).  fs3d_save writes the grid size, element
 * type, step, time, the cell types and u, v, w, density of the whole grid to `path`
 * (every rank of a slab decomposition writes its planes into the same file);
 * fs3d_load reads them back into a handle of the same size and element type.  A run
 * continued from a file repeats the uninterrupted run bit for bit.  fs3d_save sets the
 * file to its exact size first (stale bytes of an older, larger file do not survive) and
 * fs3d_load refuses a file of any other size; with several ranks the caller puts a
 * barrier between fs3d_save and the first read of the file, since a rank returns when its
 * own planes are written.                                                              */
int fs3d_save(fs3d_sim *sim, const char *path);
int fs3d_load(fs3d_sim *sim, const char *path);

/* The residual of fs3d_residual without a handle and without any allocation.  It has no counterpart in the
 * reference; it is for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  The caller passes device arrays of a whole nx*ny*nz grid on GPU `device`
 * (-1 = the calling thread's current device): u, v, w of element type `dtype` (FS_F32 / FS_F64), the cell
 * types (one byte per cell, FS_CELL_*, shell included) and d_out, two doubles in device memory that receive
 * the L2 residual sqrt(sum div^2)/#interior and max |div| (what fs3d_residual returns).  All arrays x fastest
 * and 16-byte aligned.  The kernels are queued on `stream` (a hipStream_t, NULL = the null stream) and the
 * call returns without waiting.  What the kernels need beyond inputs and outputs (flag bytes, partial sums)
 * lives in a fixed device array of FS_NOALLOC_MAX_CELLS + 16 KiB bytes that every GPU the entry point is used
 * on holds until the process ends: grids of more cells are refused with FS_EINVAL (text by
 * fs3d_last_error(NULL), per host thread).  No hipMalloc, hipFree or other allocating call is made.  All
 * calls on one GPU share that array: calls that overlap on one GPU (from several streams or threads) race. */
#define FS_NOALLOC_MAX_CELLS (1u << 24)   /* 256^3 */
int fs3d_residual_noalloc(int device, int nx, int ny, int nz, int dtype, const void *d_u, const void *d_v,
                          const void *d_w, const uint8_t *d_cells, double *d_out, void *stream);

/* z-slab decomposition over the GPUs of one node (no counterpart in the
 * reference, which is single-process) ------------------------------------- */
#define FS_COMM_ID_BYTES 128
/* Rank 0 calls this and ships the bytes to all ranks by any host channel.   */
int fsamd_comm_unique_id(uint8_t id[FS_COMM_ID_BYTES]);
/* Loopback check of the RCCL binding on one GPU (one-rank communicator, all-reduce, grouped send and
 * receive to itself); FS_OK when every call succeeded and the data came back intact.                */
int fsamd_comm_selftest(int device);
/* Collective.  Creates this rank's slab of a global nx*ny*nz_global grid:
 * planes [z0, z1) plus halo planes, RCCL communicator over all ranks.       */
int fs3d_create_slab(int nx, int ny, int nz_global, const fs3d_config *cfg, int rank, int nranks,
                     const uint8_t id[FS_COMM_ID_BYTES], fs3d_sim **out);
/* Owned plane range of this handle in global coordinates.                   */
int fs3d_slab_range(const fs3d_sim *sim, int *z0, int *z1, int *halo);
/* What the slab decomposition costs this rank (no counterpart in the reference, which times whole frames,
 * main.cpp:502, :541-543): counts and bytes are always kept; the two device times need an event pair around
 * every exchange and every wait and are kept only while fs3d_comm_timing(sim, 1) is in force.             */
typedef struct fs3d_comm_stats {
    int64_t exchanges;         /* halo exchanges queued (one grouped send/receive per neighbour each)        */
    int64_t waits;             /* times the compute stream was made to wait for the exchange stream         */
    int64_t reductions;        /* all-reduced scalar records fetched to the host (stopping rule, diagnostics) */
    int64_t bytes_sent_lo;     /* bytes sent to the rank below / above (as many are received)                */
    int64_t bytes_sent_hi;
    double exchange_ms;        /* device time of the exchanges on the exchange stream                        */
    double wait_ms;            /* device time the compute stream spent waiting for them: exposed communication */
    double reduction_host_ms;  /* host wall time spent blocked on the reductions                             */
    int64_t poll_fallbacks;    /* reductions whose record did not show in the polled host word within 20 ms, so
                                  that the host fell back to a stream synchronisation (should stay 0)        */
} fs3d_comm_stats;
int fs3d_comm_timing(fs3d_sim *sim, int on);
/* reset != 0 zeroes the record after it is copied out */
int fs3d_get_comm_stats(fs3d_sim *sim, fs3d_comm_stats *out, int reset);

/* ------------------------------------------------------------------------
 * The 2D fp64 smoke solver: class FluidSim of 3_fluid/FluidSim.h:11-303 and
 * 3_fluid/FluidSim.cpp:1-389.  Fields use the reference's Array2T layout
 * (include/Array2T.h:59-63): element (x, y) at data[x*size1 + y], doubles.
 * ---------------------------------------------------------------------- */
enum {
    FS2D_DENSITY = 0, FS2D_PRESSURE = 1, FS2D_DIVERGENCE = 2, FS2D_VORTICITY = 3,   /* res_x * res_y          */
    FS2D_U = 4, FS2D_FORCE_X = 6,                                                   /* (res_x + 1) * res_y    */
    FS2D_V = 5, FS2D_FORCE_Y = 7                                                    /* res_x * (res_y + 1)    */
};

/* single stages of advance(), for stage-wise use and tests */
enum {
    FS2D_APPLY_SOURCE = 0,       /* Grid2::applySource(0.45, 0.55, 0.1, 0.15), include/Grid2.h:55-61 */
    FS2D_ADD_BUOYANCY = 1,       /* FluidSim.h:109-118 */
    FS2D_ADD_WIND = 2,           /* FluidSim.h:120-134 */
    FS2D_ADD_FORCE = 3,          /* FluidSim.h:136-148 */
    FS2D_SOLVE_PRESSURE = 4,     /* FluidSim.h:150-167, lexicographic Poisson sweeps */
    FS2D_ADVECT_VALUES = 5,      /* FluidSim.cpp:68-85 (semi-Lagrange, MacCormack update + clamp) */
    FS2D_SET_NEUMANN = 6,        /* FluidSim.h:169-187 */
    FS2D_SET_ZERO = 7,           /* FluidSim.h:189-207 */
    FS2D_COMPUTE_DIVERGENCE = 8, /* FluidSim.h:209-218 */
    FS2D_COPY_BORDER = 9,        /* FluidSim.h:231-243 */
    FS2D_SOLVE_POISSON = 10,     /* FluidSim.cpp:6-48 */
    FS2D_CORRECT_VELOCITY = 11,  /* FluidSim.cpp:50-66 */
    FS2D_COMPUTE_VORTICITY = 12, /* FluidSim.h:220-229 */
    FS2D_RESET_FORCE = 13,       /* FluidSim.h:89 */
    FS2D_SOLVE_PRESSURE_RB = 14, /* the same with red-black Poisson sweeps */
    FS2D_SOLVE_POISSON_RB = 15
};

typedef struct fs2d_sim fs2d_sim;

/* FluidSim::init (FluidSim.h:15-41) for a free resolution (the reference fixes 128 x 192): dx = 1,
 * dt = 0.005*sqrt((res_x+res_y)/2), accuracy 1e-5, 1000 iterations, MacCormack on, wind off; then reset. */
int fs2d_create(int res_x, int res_y, int device, fs2d_sim **out);
void fs2d_destroy(fs2d_sim *sim);
const char *fs2d_last_error(const fs2d_sim *sim);
/* Simulation::reset (include/Simulation.h:26-30) + FluidSim::resetMembers (FluidSim.h:43-50) */
int fs2d_reset(fs2d_sim *sim);
/* setTimestep / setAccuracy / setIteration / setWind / setMacCormack (Simulation.h:58, FluidSim.h:263-266);
 * dt, acc, iter <= 0 leave the value unchanged.  ordering: FS_ORDER_LEXICOGRAPHIC (the reference's Poisson
 * sweep order, bit-identical, default) or FS_ORDER_RED_BLACK (parallel, same tolerance). */
int fs2d_set_params(fs2d_sim *sim, double dt, double acc, int iter, int wind, int maccormack, int ordering);
/* getTimestep / getTime / getStep, plus the iterations and residual of the last Poisson solve */
int fs2d_get_info(fs2d_sim *sim, double *dt, double *time, uint64_t *step, int *last_iters, double *last_residual);
int fs2d_field_size(const fs2d_sim *sim, int which, int *size0, int *size1);
/* Grid2::x() / MACGrid2::x(), y() (include/Grid2.h:20-22, include/MACGrid2.h:19-20) as whole arrays */
int fs2d_get_field(fs2d_sim *sim, int which, double *host, size_t count);
int fs2d_set_field(fs2d_sim *sim, int which, const double *host, size_t count);
/* FluidSim::advance (FluidSim.h:71-96) */
int fs2d_advance(fs2d_sim *sim);
int fs2d_stage(fs2d_sim *sim, int stage);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDSIM_AMD_H */
