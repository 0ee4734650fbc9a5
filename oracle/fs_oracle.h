/*
 * fs_oracle.h -- interface of the CPU restatement in fs_oracle.c.
 * TEST INFRASTRUCTURE ONLY: used by tests/, the smoke check and bench.py's
 * cpu_baseline leg; never by the shipped library.
 */
#ifndef FS_ORACLE_H
#define FS_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FSO_SCHEME_RK4 = 0, FSO_SCHEME_MACCORMACK = 1 };
enum { FSO_PRESSURE_JACOBI = 0, FSO_PRESSURE_MGPCG = 1 };
enum { FSO_UX = 0, FSO_UY = 1, FSO_UZ = 2, FSO_RHO = 3, FSO_T = 4, FSO_P = 5, FSO_DIV = 6, FSO_R = 7,
       FSO_VORTX = 8, FSO_VORTY = 9, FSO_VORTZ = 10 };

typedef struct fso_params {
    int nx, ny, nz;
    float dt;
    int jacobi_iters;
    int scheme;
    float source_x, source_y, source_z, source_radius;
    float rho0, temperature0, temperature_env;
    int vort_confine;        /* Solver.cu:13 VORT_CONFINE, defined-behaviour variant (see fso_vort_size) */
    float curl_strength;     /* Solver.h:73 */
    int pressure_solver;     /* FSO_PRESSURE_*: Solver.cu:1178-1192; jacobi_iters is max_iter for both */
    int mg_level;            /* Solver.h:151; every dim must be a multiple of 2^(mg_level-1) */
    int init_smooth_steps;   /* Solver.h:153 */
    int bottom_smooth_steps; /* Solver.h:154 */
    float iter_precision;    /* Solver.h:69 */
} fso_params;

typedef struct fso_state fso_state;

/* 1: evaluate the x*y + z shaped expressions of the advection and the buoyancy with fused
   multiply-adds, as nvcc's default -fmad=true may (process-wide; default 0 = the uncontracted
   meaning every GPU kernel implements).  For sensitivity tests only. */
void fso_set_fmad(int on);
void fso_default_params(fso_params* p, int nx, int ny, int nz);
fso_state* fso_create(const fso_params* p);
void fso_destroy(fso_state* s);
void fso_step(fso_state* s);
void fso_stage_source(fso_state* s);
void fso_stage_force(fso_state* s);
void fso_stage_vorticity(fso_state* s);
void fso_stage_advect(fso_state* s);
void fso_stage_confine(fso_state* s);
void fso_stage_project(fso_state* s);
void fso_stage_gradient(fso_state* s);
float* fso_field(fso_state* s, int which);
size_t fso_field_size(const fso_state* s, int which);
double fso_residual(const fso_state* s);
int fso_frame(const fso_state* s);
int fso_last_cg_iters(const fso_state* s);
float fso_last_rtr(const fso_state* s);

/* kernel-level entry points (known-answer tests, per-kernel parity) */
float fso_trilerp(const float* field, float px, float py, float pz, float ox, float oy,
                  float oz, int dx, int dy, int dz, int clamp);
void fso_backtrace(const float* ux, const float* uy, const float* uz, float px, float py,
                   float pz, float dt, int nx, int ny, int nz, int scheme, float* out3);
void fso_grid_init(float* temperature, float temperature_env, int nx, int ny, int nz);
void fso_source(float source_x, float source_y, float source_z, float radius, float* rho,
                float* temperature, float rho0, float temperature0, int nx, int ny, int nz);
void fso_force_uy(float* uy, const float* temperature_field, float dt, float temperature_env,
                  int nx, int ny, int nz);
void fso_semilag(const float* field, float* new_field, const float* ux, const float* uy,
                 const float* uz, float dt, int nx, int ny, int nz, int dir, int scheme);
void fso_divergence(float* div_field, const float* ux, const float* uy, const float* uz,
                    int nx, int ny, int nz);
void fso_jacobi(const float* field, float* new_field, const float* div_field, float* r,
                int nx, int ny, int nz);
void fso_gradient(float* ux, float* uy, float* uz, const float* pressure, int nx, int ny, int nz);
double fso_sum_sq(const float* r, size_t n);
size_t fso_vort_size(int nx, int ny, int nz);
size_t fso_mg_space(int nx, int ny, int nz, int levels);
void fso_mg_smooth(const float* r, float* z, size_t offset, int phase, int dx, int dy, int dz);
void fso_mg_restrict(float* r, const float* z, size_t offset, int dx, int dy, int dz);
void fso_mg_prolongate(float* z, size_t offset, int dx, int dy, int dz);
void fso_mg_precondition(float* r, float* z, int nx, int ny, int nz, int levels, int init_smooth_steps,
                         int bottom_smooth_steps);
float fso_dot(const float* a, const float* b, size_t n);
void fso_apply_a(float* Ap, const float* p, int nx, int ny, int nz);
void fso_find_vort(float* vortx, float* vorty, float* vortz, const float* ux, const float* uy,
                   const float* uz, int nx, int ny, int nz);
void fso_vort_confine(float* u, const float* vortx, const float* vorty, const float* vortz,
                      int nx, int ny, int nz, int dir, float coeff);

#ifdef __cplusplus
}
#endif
#endif
