/*
 * fs_oracle.c -- CPU restatement of the CuSolver simulation step (Jacobi branch).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls
 * this file; it is the checker that the HIP path is compared against (tests/,
 * the smoke check and the cpu_baseline leg of bench.py).
 *
 * PARITY PINNING: the reference holds no golden vectors, known-answer tests or
 * fixtures for this path and its CUDA source cannot be compiled or run here, so
 * this restatement is "parity unpinned" by upstream data.  It is pinned only by
 * following the reference source expression by expression (citations below, all
 * relative to /root/reference), by analytic known-answer tests in tests/ and by a
 * second transcription of the same reference lines in numpy
 * (tests/test_oracle_crosscheck.py) that it has to match bit for bit.
 *
 * Arithmetic contract: every expression is evaluated in IEEE binary32 in the
 * association order the reference source spells, with NO fused multiply-add
 * (build with -ffp-contract=off -fno-fast-math).  nvcc's default -fmad=true would
 * contract some a*b+c in the shipped reference binary in a pattern the source does
 * not determine; the language-level (uncontracted) meaning is what is restated.
 *
 * Layout: every array is x-fastest, flat index x + y*dx + z*dx*dy with
 * (dx,dy,dz) the dims of that array (CuSolver/Source/Solver.cu:33,41).
 *   ux (nx+1,ny,nz)  uy (nx,ny+1,nz)  uz (nx,ny,nz+1)  others (nx,ny,nz)
 *   (Solver.cu:881-893)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>

#include "fs_oracle.h"

typedef struct { int x, y, z; } i3;
typedef struct { float x, y, z; } f3;

static inline i3 mk_i3(int a, int b, int c) { i3 r; r.x = a; r.y = b; r.z = c; return r; }
static inline f3 mk_f3(float a, float b, float c) { f3 r; r.x = a; r.y = b; r.z = c; return r; }

/* Math.cuh:21-31 */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Math.cuh:33-39 -- clamp every index into [0, dim-1] */
static inline i3 minmax3(i3 pos, i3 max_pos)
{
    pos.x = imax(0, imin(pos.x, max_pos.x - 1));
    pos.y = imax(0, imin(pos.y, max_pos.y - 1));
    pos.z = imax(0, imin(pos.z, max_pos.z - 1));
    return pos;
}

static inline size_t flat(i3 pos, i3 max_pos)
{
    return (size_t)pos.x + (size_t)pos.y * (size_t)max_pos.x +
           (size_t)pos.z * (size_t)max_pos.x * (size_t)max_pos.y;
}

/* Solver.cu:30-34 -- replicate-edge point sample */
static inline float sample(const float* field, i3 pos, i3 max_pos)
{
    pos = minmax3(pos, max_pos);
    return field[flat(pos, max_pos)];
}

/* Solver.cu:37-42 -- zero-outside point sample */
static inline float cg_sample(const float* field, i3 pos, i3 max_pos)
{
    if (pos.x < 0 || pos.x >= max_pos.x || pos.y < 0 || pos.y >= max_pos.y ||
        pos.z < 0 || pos.z >= max_pos.z)
        return 0;
    return field[flat(pos, max_pos)];
}

/* Contraction sensitivity switch (fso_set_fmad).  The reference is built by nvcc, whose default
   -fmad=true fuses a multiply into a following add or subtract wherever it likes; which of them it
   fuses is not determined by the source, so the oracle's contract is the uncontracted meaning
   (g_fmad = 0).  With g_fmad = 1 the expressions of the advection and the buoyancy that have the
   shape x*y + z or z - x*y are evaluated with one fused multiply-add the way LLVM-based compilers
   contract them (the left product of a sum of two products is the fused one).  It is what the
   library's FS_FLAG_FMAD mode is checked against, and what measures how far contraction can move
   the fields. */
static int g_fmad = 0;
void fso_set_fmad(int on) { g_fmad = on != 0; }

/* Math.cuh:57-60 */
static inline float lerpf(float a, float b, float s)
{
    if (g_fmad) return fmaf(a, 1 - s, b * s);
    return a * (1 - s) + b * s;
}

/* pos - c * u of the RK stages (Solver.cu:145, 150, 155) and a + 2 * u of :160 */
static inline f3 f3_step(f3 pos, float c, f3 u)
{
    if (g_fmad) { pos.x = fmaf(-c, u.x, pos.x); pos.y = fmaf(-c, u.y, pos.y); pos.z = fmaf(-c, u.z, pos.z); return pos; }
    pos.x -= c * u.x; pos.y -= c * u.y; pos.z -= c * u.z;
    return pos;
}
static inline f3 f3_add2(f3 a, f3 u)
{
    if (g_fmad) { a.x = fmaf(2.0f, u.x, a.x); a.y = fmaf(2.0f, u.y, a.y); a.z = fmaf(2.0f, u.z, a.z); return a; }
    a.x += 2.0f * u.x; a.y += 2.0f * u.y; a.z += 2.0f * u.z;
    return a;
}

/* Math.cuh:80-118 float3 operators, spelled out so the association is explicit */
static inline f3 f3_scale(float a, f3 b) { b.x *= a; b.y *= a; b.z *= a; return b; } /* :80-86  */
static inline f3 f3_sub(f3 a, f3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; } /* :96-102 */
static inline f3 f3_add(f3 a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; } /* :104-110*/
static inline f3 f3_div(f3 a, float b) { a.x /= b; a.y /= b; a.z /= b; return a; }    /* :112-118*/

/* Solver.cu:52-113 */
static float trilerp(const float* field, f3 pos, float fpos_x, float fpos_y, float fpos_z,
                     i3 max_pos, int b_minmax_sample)
{
    float x = pos.x - fpos_x;
    float y = pos.y - fpos_y;
    float z = pos.z - fpos_z;

    int ix = (int)x;             /* truncation toward zero, :58-60 */
    int iy = (int)y;
    int iz = (int)z;

    float fx = x - floorf(x);    /* :62-64 */
    float fy = y - floorf(y);
    float fz = z - floorf(z);

    /* :66-82 -- the comparison is float-vs-double-literal, i.e. done in double */
    if ((double)x < -1e-6) { ix -= 1; fx = x - (float)ix; }
    if ((double)y < -1e-6) { iy -= 1; fy = y - (float)iy; }
    if ((double)z < -1e-6) { iz -= 1; fz = z - (float)iz; }

    float a, b, c, d, e, f, g, h;
    if (b_minmax_sample) {       /* :86-96 */
        a = sample(field, mk_i3(ix, iy, iz), max_pos);
        b = sample(field, mk_i3(ix + 1, iy, iz), max_pos);
        c = sample(field, mk_i3(ix, iy + 1, iz), max_pos);
        d = sample(field, mk_i3(ix + 1, iy + 1, iz), max_pos);
        e = sample(field, mk_i3(ix, iy, iz + 1), max_pos);
        f = sample(field, mk_i3(ix + 1, iy, iz + 1), max_pos);
        g = sample(field, mk_i3(ix, iy + 1, iz + 1), max_pos);
        h = sample(field, mk_i3(ix + 1, iy + 1, iz + 1), max_pos);
    } else {                     /* :97-107 */
        a = cg_sample(field, mk_i3(ix, iy, iz), max_pos);
        b = cg_sample(field, mk_i3(ix + 1, iy, iz), max_pos);
        c = cg_sample(field, mk_i3(ix, iy + 1, iz), max_pos);
        d = cg_sample(field, mk_i3(ix + 1, iy + 1, iz), max_pos);
        e = cg_sample(field, mk_i3(ix, iy, iz + 1), max_pos);
        f = cg_sample(field, mk_i3(ix + 1, iy, iz + 1), max_pos);
        g = cg_sample(field, mk_i3(ix, iy + 1, iz + 1), max_pos);
        h = cg_sample(field, mk_i3(ix + 1, iy + 1, iz + 1), max_pos);
    }

    float lerp1 = lerpf(lerpf(a, b, fx), lerpf(c, d, fx), fy);   /* :109 */
    float lerp2 = lerpf(lerpf(e, f, fx), lerpf(g, h, fx), fy);   /* :110 */
    return lerpf(lerp1, lerp2, fz);                              /* :112 */
}

/* velocity at a point: the three staggered gathers every RK stage does (:142-144) */
static inline f3 velocity_at(const float* ux, const float* uy, const float* uz, f3 pos,
                             int nx, int ny, int nz)
{
    f3 u;
    u.x = trilerp(ux, pos, 0.f, 0.5f, 0.5f, mk_i3(nx + 1, ny, nz), 0);
    u.y = trilerp(uy, pos, 0.5f, 0.f, 0.5f, mk_i3(nx, ny + 1, nz), 0);
    u.z = trilerp(uz, pos, 0.5f, 0.5f, 0.f, mk_i3(nx, ny, nz + 1), 0);
    return u;
}

/* Solver.cu:139-161 */
static f3 RK4(const float* ux, const float* uy, const float* uz, f3 pos, float dt,
              int nx, int ny, int nz)
{
    f3 u1 = velocity_at(ux, uy, uz, pos, nx, ny, nz);
    /* (f3_step(pos, c, u) is f3_sub(pos, f3_scale(c, u)) unless the contraction switch is on) */
    f3 p1 = f3_step(pos, 0.5f * dt, u1);                          /* :145 */
    f3 u2 = velocity_at(ux, uy, uz, p1, nx, ny, nz);
    f3 p2 = f3_step(pos, 0.5f * dt, u2);                          /* :150 */
    f3 u3 = velocity_at(ux, uy, uz, p2, nx, ny, nz);
    f3 p3 = f3_step(pos, dt, u3);                                 /* :155 */
    f3 u4 = velocity_at(ux, uy, uz, p3, nx, ny, nz);
    /* :160  pos - dt * (u1 + 2.0*u2 + 2.0*u3 + u4) / 6.0 ; the double literals
       convert to float through the operator signatures (Math.cuh:80,112). */
    f3 sum = f3_add(f3_add2(f3_add2(u1, u2), u3), u4);   /* ((u1 + 2 u2) + 2 u3) + u4 */
    return f3_sub(pos, f3_div(f3_scale(dt, sum), 6.0f));
}

/* Solver.cu:163-169 */
static f3 MacCormack(const float* ux, const float* uy, const float* uz, f3 pos, float dt,
                     int nx, int ny, int nz)
{
    f3 coord_predict = RK4(ux, uy, uz, pos, dt, nx, ny, nz);
    f3 coord_reflect = RK4(ux, uy, uz, coord_predict, -dt, nx, ny, nz);
    return f3_add(coord_predict, f3_div(f3_sub(pos, coord_reflect), 2.f));
}

/* ------------------------------------------------------------------ kernels */

float fso_trilerp(const float* field, float px, float py, float pz, float ox, float oy,
                  float oz, int dx, int dy, int dz, int clamp)
{
    return trilerp(field, mk_f3(px, py, pz), ox, oy, oz, mk_i3(dx, dy, dz), clamp);
}

void fso_backtrace(const float* ux, const float* uy, const float* uz, float px, float py,
                   float pz, float dt, int nx, int ny, int nz, int scheme, float* out3)
{
    f3 c = scheme == FSO_SCHEME_MACCORMACK
               ? MacCormack(ux, uy, uz, mk_f3(px, py, pz), dt, nx, ny, nz)
               : RK4(ux, uy, uz, mk_f3(px, py, pz), dt, nx, ny, nz);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

/* Solver.cu:285-293 */
void fso_grid_init(float* temperature, float temperature_env, int nx, int ny, int nz)
{
    size_t n = (size_t)nx * ny * nz;
    for (size_t q = 0; q < n; ++q) temperature[q] = temperature_env;
}

/* Solver.cu:295-307; launch (ny,nz) x nx (:1028) */
void fso_source(float source_x, float source_y, float source_z, float radius, float* rho,
                float* temperature, float rho0, float temperature0, int nx, int ny, int nz)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                /* pow(v, 2) of exactly representable v: the square is exact */
                float ddx = (float)i - source_x, ddy = (float)j - source_y,
                      ddz = (float)k - source_z;
                if (ddx * ddx + ddy * ddy + ddz * ddz <= radius * radius) {
                    rho[ind] = rho0;
                    temperature[ind] = temperature0;
                }
            }
}

/* Solver.cu:346-368 with BURST 0; launch (ny+1,nz) x nx (:1031) */
void fso_force_uy(float* uy, const float* temperature_field, float dt, float temperature_env,
                  int nx, int ny, int nz)
{
    i3 max_pos = mk_i3(nx, ny, nz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny + 1; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * (ny + 1);
                f3 pos = mk_f3((float)i + 0.5f, (float)j, (float)k + 0.5f);
                float temperature = trilerp(temperature_field, pos, 0.5f, 0.5f, 0.5f, max_pos, 1);
                if (g_fmad) {
                    uy[ind] = fmaf(temperature - temperature_env, dt, uy[ind]);
                    continue;
                }
                float buoyancy = (temperature - temperature_env) * dt;
                uy[ind] += buoyancy;
            }
}

/* Solver.cu:370-403 for dir in 0..3 */
void fso_semilag(const float* field, float* new_field, const float* ux, const float* uy,
                 const float* uz, float dt, int nx, int ny, int nz, int dir, int scheme)
{
    int dx = dir == 1 ? nx + 1 : nx;
    int dy = dir == 2 ? ny + 1 : ny;
    int dz = dir == 3 ? nz + 1 : nz;
    float ox = dir == 1 ? 0.f : 0.5f;
    float oy = dir == 2 ? 0.f : 0.5f;
    float oz = dir == 3 ? 0.f : 0.5f;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int k = 0; k < dz; ++k)
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                size_t ind = (size_t)i + (size_t)j * dx + (size_t)k * dx * dy;
                f3 pos = mk_f3((float)i + ox, (float)j + oy, (float)k + oz);     /* :381-383 */
                f3 coord = scheme == FSO_SCHEME_MACCORMACK
                               ? MacCormack(ux, uy, uz, pos, dt, nx, ny, nz)     /* :394 */
                               : RK4(ux, uy, uz, pos, dt, nx, ny, nz);           /* :396 */
                new_field[ind] = trilerp(field, coord, ox, oy, oz, mk_i3(dx, dy, dz), 0); /* :399 */
            }
}

/* Solver.cu:406-423 */
void fso_divergence(float* div_field, const float* ux, const float* uy, const float* uz,
                    int nx, int ny, int nz)
{
    i3 mux = mk_i3(nx + 1, ny, nz), muy = mk_i3(nx, ny + 1, nz), muz = mk_i3(nx, ny, nz + 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                float ul = sample(ux, mk_i3(i, j, k), mux);
                float ur = sample(ux, mk_i3(i + 1, j, k), mux);
                float ubo = sample(uy, mk_i3(i, j, k), muy);
                float ut = sample(uy, mk_i3(i, j + 1, k), muy);
                float ubh = sample(uz, mk_i3(i, j, k), muz);
                float uf = sample(uz, mk_i3(i, j, k + 1), muz);
                float div = ur + uf + ut - ul - ubh - ubo;                        /* :420 */
                div_field[ind] = div;
            }
}

/* Solver.cu:426-443 -- one sweep; r may be NULL (the residual field is optional) */
void fso_jacobi(const float* field, float* new_field, const float* div_field, float* r,
                int nx, int ny, int nz)
{
    i3 max_pos = mk_i3(nx, ny, nz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                float pl = sample(field, mk_i3(i - 1, j, k), max_pos);
                float pr = sample(field, mk_i3(i + 1, j, k), max_pos);
                float pbo = sample(field, mk_i3(i, j - 1, k), max_pos);
                float pt = sample(field, mk_i3(i, j + 1, k), max_pos);
                float pbh = sample(field, mk_i3(i, j, k - 1), max_pos);
                float pf = sample(field, mk_i3(i, j, k + 1), max_pos);
                float div = sample(div_field, mk_i3(i, j, k), max_pos);
                new_field[ind] = (pl + pr + pbh + pf + pbo + pt - div) / 6.f;     /* :441 */
                if (r)
                    r[ind] = div + 6 * field[ind] - pl - pr - pbh - pf - pbo - pt; /* :442 */
            }
}

/* Solver.cu:446-483 -- the three gradient kernels, in call order :1111-1113 */
void fso_gradient(float* ux, float* uy, float* uz, const float* pressure, int nx, int ny, int nz)
{
    i3 max_pos = mk_i3(nx, ny, nz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx + 1; ++i) {
                size_t ind = (size_t)i + (size_t)j * (nx + 1) + (size_t)k * (nx + 1) * ny;
                float pl = sample(pressure, mk_i3(i - 1, j, k), max_pos);
                float pr = sample(pressure, mk_i3(i, j, k), max_pos);
                ux[ind] -= pr - pl;                                               /* :456 */
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny + 1; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * (ny + 1);
                float pbo = sample(pressure, mk_i3(i, j - 1, k), max_pos);
                float pt = sample(pressure, mk_i3(i, j, k), max_pos);
                uy[ind] -= pt - pbo;                                              /* :469 */
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz + 1; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                float pbh = sample(pressure, mk_i3(i, j, k - 1), max_pos);
                float pf = sample(pressure, mk_i3(i, j, k), max_pos);
                uz[ind] -= pf - pbh;                                              /* :482 */
            }
}

/* --------------------------------------------------- vorticity confinement (optional) */

/* Size of each vorticity array in the defined-behaviour variant.  The reference writes its
 * vorticity arrays with cell strides (nx, nx*ny) into nx*ny*nz floats (Solver.cu:485-525,
 * 894-896) but reads them with the strides of (nx,ny+1,nz+1), (nx+1,ny,nz+1) and (nx+1,ny+1,nz)
 * arrays (Solver.cu:556-558), i.e. beyond the allocation.  Here the arrays hold
 * (nx+1)(ny+1)(nz+1) floats, zero-filled once, so every such read is in bounds and returns either
 * a vorticity value (through the mismatched strides, as in the reference) or zero. */
size_t fso_vort_size(int nx, int ny, int nz)
{
    return (size_t)(nx + 1) * (size_t)(ny + 1) * (size_t)(nz + 1);
}

/* Solver.cu:485-525, launches :1035-1037 <<<(ny,nz), nx>>>: ind = i + j*nx + k*nx*ny */
void fso_find_vort(float* vortx, float* vorty, float* vortz, const float* ux, const float* uy,
                   const float* uz, int nx, int ny, int nz)
{
    const i3 max_pos_ux = mk_i3(nx + 1, ny, nz), max_pos_uy = mk_i3(nx, ny + 1, nz),
             max_pos_uz = mk_i3(nx, ny, nz + 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                {   /* FindVortx :492-496 */
                    float ubh = sample(uy, mk_i3(i, j, k - 1), max_pos_uy);
                    float uf = sample(uy, mk_i3(i, j, k), max_pos_uy);
                    float ubo = sample(uz, mk_i3(i, j - 1, k), max_pos_uz);
                    float ut = sample(uz, mk_i3(i, j, k), max_pos_uz);
                    vortx[ind] = uf - ubh - ut + ubo;
                }
                {   /* FindVorty :506-510 */
                    float ul = sample(uz, mk_i3(i - 1, j, k), max_pos_uz);
                    float ur = sample(uz, mk_i3(i, j, k), max_pos_uz);
                    float ubh = sample(ux, mk_i3(i, j, k - 1), max_pos_ux);
                    float uf = sample(ux, mk_i3(i, j, k), max_pos_ux);
                    vorty[ind] = ur - ul - uf + ubh;
                }
                {   /* FindVortz :520-524 */
                    float ul = sample(uy, mk_i3(i - 1, j, k), max_pos_uy);
                    float ur = sample(uy, mk_i3(i, j, k), max_pos_uy);
                    float ubo = sample(ux, mk_i3(i, j - 1, k), max_pos_ux);
                    float ut = sample(ux, mk_i3(i, j, k), max_pos_ux);
                    vortz[ind] = ut - ubo - ur + ul;
                }
            }
}

/* the three gathers at one point, Solver.cu:556-558 and its repetitions */
static inline f3 vort_at(const float* vortx, const float* vorty, const float* vortz, float px,
                         float py, float pz, int nx, int ny, int nz)
{
    f3 v, pos = mk_f3(px, py, pz);
    v.x = trilerp(vortx, pos, 0.5f, 0.f, 0.f, mk_i3(nx, ny + 1, nz + 1), 1);
    v.y = trilerp(vorty, pos, 0.f, 0.5f, 0.f, mk_i3(nx + 1, ny, nz + 1), 1);
    v.z = trilerp(vortz, pos, 0.f, 0.f, 0.5f, mk_i3(nx + 1, ny + 1, nz), 1);
    return v;
}

/* sqrt of a float argument is the float overload in device code (:559 etc.) */
static inline float vort_norm(f3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

/* Solver.cu:540-605, launches :1078-1080 with blockDim.x = dx, gridDim = (dy, dz) of the
 * component: dir 1 (nx+1,ny,nz), dir 2 (nx,ny+1,nz), dir 3 (nx,ny,nz+1). */
void fso_vort_confine(float* u, const float* vortx, const float* vorty, const float* vortz,
                      int nx, int ny, int nz, int dir, float coeff)
{
    const int dx = nx + (dir == 1), dy = ny + (dir == 2), dz = nz + (dir == 3);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < dz; ++k)
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                /* :548 -- the guard depends on the launch dims of this component */
                if (!(i > 2 && i < dx - 2 && j > 2 && j < dy - 2 && k > 2 && k < dz - 2)) continue;
                size_t ind = (size_t)i + (size_t)j * dx + (size_t)k * dx * dy;
                float px = (float)i + (dir == 1 ? 0.f : 0.5f);          /* :552-554 */
                float py = (float)j + (dir == 2 ? 0.f : 0.5f);
                float pz = (float)k + (dir == 3 ? 0.f : 0.5f);
                /* pos + 0.5 is evaluated in double and narrowed by combine_float3 (:556);
                   halves of small integers are exact either way */
                float vt = vort_norm(vort_at(vortx, vorty, vortz, px, (float)((double)py + 0.5), pz, nx, ny, nz));
                float vbo = vort_norm(vort_at(vortx, vorty, vortz, px, (float)((double)py - 0.5), pz, nx, ny, nz));
                float vr = vort_norm(vort_at(vortx, vorty, vortz, (float)((double)px + 0.5), py, pz, nx, ny, nz));
                float vl = vort_norm(vort_at(vortx, vorty, vortz, (float)((double)px - 0.5), py, pz, nx, ny, nz));
                float vf = vort_norm(vort_at(vortx, vorty, vortz, px, py, (float)((double)pz + 0.5), nx, ny, nz));
                float vbh = vort_norm(vort_at(vortx, vorty, vortz, px, py, (float)((double)pz - 0.5), nx, ny, nz));
                f3 vc = vort_at(vortx, vorty, vortz, px, py, pz, nx, ny, nz);   /* :586-588 */
                float gx = vr - vl;                                      /* :590-592 */
                float gy = vt - vbo;
                float gz = vf - vbh;
                /* :593 -- the float sum is widened by the double literal, double sqrt, narrowed */
                float n = (float)sqrt((double)(gx * gx + gy * gy + gz * gz) + 1e-5);
                gx /= n;                                                 /* :594-596 */
                gy /= n;
                gz /= n;
                if (dir == 1) u[ind] += coeff * (gy * vc.z - gz * vc.y);        /* :598-603 */
                else if (dir == 2) u[ind] += coeff * (gz * vc.x - gx * vc.z);
                else u[ind] += coeff * (gx * vc.y - gy * vc.x);
            }
}

/* What Solver.cu:1185-1187 (Mul + Reduce) is meant to produce: sum of r*r over the grid.
   The reference's fp32 tree drops elements at most sizes and races in place, so the
   defined target is the mathematically plain sum, accumulated in binary64. */
double fso_sum_sq(const float* r, size_t n)
{
    double s = 0.0;
    for (size_t q = 0; q < n; ++q) s += (double)r[q] * (double)r[q];
    return s;
}

/* ------------------------------------------- multigrid-preconditioned CG (optional) */

/* Solver.cu:44-50 -- sum of the six neighbours, zero outside (that association) */
static inline float neibor_sum(const float* field, int i, int j, int k, i3 max_pos)
{
    return cg_sample(field, mk_i3(i - 1, j, k), max_pos) + cg_sample(field, mk_i3(i + 1, j, k), max_pos)
         + cg_sample(field, mk_i3(i, j - 1, k), max_pos) + cg_sample(field, mk_i3(i, j + 1, k), max_pos)
         + cg_sample(field, mk_i3(i, j, k - 1), max_pos) + cg_sample(field, mk_i3(i, j, k + 1), max_pos);
}

/* Solver.cu:1120-1125: levels are stored one after the other, [level][z][y][x] */
size_t fso_mg_space(int nx, int ny, int nz, int levels)
{
    size_t space = 0, t = (size_t)nx * ny * nz;
    for (int i = 0; i < levels; ++i) { space += t; t /= 8; }
    return space;
}

/* Solver.cu:848-860 -- one colour of red/black Gauss-Seidel on the level at `offset` */
void fso_mg_smooth(const float* r, float* z, size_t offset, int phase, int dx, int dy, int dz)
{
    const i3 max_pos = mk_i3(dx, dy, dz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < dz; ++k)
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                if ((((i + j + k) & 1) != 0) != (phase != 0)) continue;
                size_t ind = (size_t)i + (size_t)j * dx + (size_t)k * dx * dy;
                z[offset + ind] = (r[offset + ind] + neibor_sum(z + offset, i, j, k, max_pos)) / 6.f;
            }
}

/* Solver.cu:822-834, 862-877 -- the eight phases run one after the other, each adding half the
 * residual of one child to its parent (the product with the double literal 0.5 and the sum are in
 * double, narrowed on the store) */
void fso_mg_restrict(float* r, const float* z, size_t offset, int dx, int dy, int dz)
{
    const i3 max_pos = mk_i3(dx, dy, dz);
    const int cx = dx / 2, cy = dy / 2, cz = dz / 2;
    const size_t coarse = offset + (size_t)dx * dy * dz;
    for (int phase = 0; phase < 8; ++phase) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int ck = 0; ck < cz; ++ck)
            for (int cj = 0; cj < cy; ++cj)
                for (int ci = 0; ci < cx; ++ci) {
                    size_t new_ind = (size_t)ci + (size_t)cj * cx + (size_t)ck * cx * cy;
                    int i = ci * 2 + ((phase & 1) ? 1 : 0);
                    int j = cj * 2 + ((phase & 2) ? 1 : 0);
                    int k = ck * 2 + ((phase & 4) ? 1 : 0);
                    size_t ind = (size_t)i + (size_t)j * dx + (size_t)k * dx * dy;
                    float res = r[offset + ind] - (6 * z[offset + ind] - neibor_sum(z + offset, i, j, k, max_pos));
                    r[coarse + new_ind] = (float)((double)r[coarse + new_ind] + (double)res * 0.5);
                }
    }
}

/* Solver.cu:836-846 */
void fso_mg_prolongate(float* z, size_t offset, int dx, int dy, int dz)
{
    const size_t coarse = offset + (size_t)dx * dy * dz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < dz; ++k)
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                size_t ind = (size_t)i + (size_t)j * dx + (size_t)k * dx * dy;
                size_t new_ind = (size_t)(i >> 1) + (size_t)(j >> 1) * (dx >> 1) +
                                 (size_t)(k >> 1) * (dx >> 1) * (dy >> 1);
                z[offset + ind] += z[coarse + new_ind];
            }
}

/* Solver.cu:1312-1366 -- one V-cycle: z = M^-1 r (level 0 of r is the input).  The reference
 * clears z and the coarse levels of r with mg_space/1024 (integer division) blocks of 1024
 * threads, which leaves a tail uncleared unless the sizes are multiples of 1024 (they are for its
 * shipped 64x128x64 grid); here everything is cleared. */
void fso_mg_precondition(float* r, float* z, int nx, int ny, int nz, int levels, int init_smooth_steps,
                         int bottom_smooth_steps)
{
    const size_t space = fso_mg_space(nx, ny, nz, levels), n0 = (size_t)nx * ny * nz;
    memset(z, 0, space * sizeof(float));
    memset(r + n0, 0, (space - n0) * sizeof(float));
    int dx = nx, dy = ny, dz = nz;
    size_t offset = 0;
    for (int l = 0; l < levels - 1; ++l) {                                   /* :1330-1343 */
        for (int i = 0; i < (init_smooth_steps << l); ++i) {
            fso_mg_smooth(r, z, offset, 0, dx, dy, dz);
            fso_mg_smooth(r, z, offset, 1, dx, dy, dz);
        }
        fso_mg_restrict(r, z, offset, dx, dy, dz);
        offset += (size_t)dx * dy * dz;
        dx >>= 1; dy >>= 1; dz >>= 1;
    }
    for (int i = 0; i < bottom_smooth_steps; ++i) {                          /* :1346-1350 */
        fso_mg_smooth(r, z, offset, 0, dx, dy, dz);
        fso_mg_smooth(r, z, offset, 1, dx, dy, dz);
    }
    for (int l = levels - 2; l >= 0; --l) {                                  /* :1352-1365 */
        dx <<= 1; dy <<= 1; dz <<= 1;
        offset -= (size_t)dx * dy * dz;
        fso_mg_prolongate(z, offset, dx, dy, dz);
        for (int i = 0; i < (init_smooth_steps << l); ++i) {
            fso_mg_smooth(r, z, offset, 0, dx, dy, dz);
            fso_mg_smooth(r, z, offset, 1, dx, dy, dz);
        }
    }
}

/* What Mul + Reduce (Solver.cu:231-283, 1212-1224) are meant to yield: the sum of the binary32
 * products a[i]*b[i].  The reference's tree drops elements at most sizes and races; the defined
 * target is the plain sum, accumulated in binary64 and narrowed to the float the host reads. */
float fso_dot(const float* a, const float* b, size_t n)
{
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += (double)(a[i] * b[i]);
    return (float)s;
}

/* Solver.cu:789-799 -- A p = 6 p - neighbours, zero outside the box */
void fso_apply_a(float* Ap, const float* p, int nx, int ny, int nz)
{
    const i3 max_pos = mk_i3(nx, ny, nz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t ind = (size_t)i + (size_t)j * nx + (size_t)k * nx * ny;
                float pc = cg_sample(p, mk_i3(i, j, k), max_pos);
                Ap[ind] = 6.f * pc - neibor_sum(p, i, j, k, max_pos);
            }
}

/* --------------------------------------------------------------- whole step */

struct fso_state {
    fso_params p;
    float *ux, *uy, *uz, *new_ux, *new_uy, *new_uz;
    float *rho, *new_rho, *temperature, *new_temperature;
    float *pressure, *new_pressure, *div, *r;
    float *vortx, *vorty, *vortz;   /* only with vort_confine */
    /* MGPCG work arrays (Solver.cu:903-907); cg_x is the member `x`, which Conjugate re-points */
    float *cg_r, *cg_z, *cg_p, *cg_Ap, *cg_x;
    float* owned[24];               /* every allocation, for fso_destroy (pointers get swapped) */
    int n_owned;
    int last_cg_iters;
    float last_rTr;
    int frame;
};

void fso_default_params(fso_params* p, int nx, int ny, int nz)
{
    memset(p, 0, sizeof(*p));
    p->nx = nx; p->ny = ny; p->nz = nz;          /* Solver.h:58-60 */
    p->rho0 = 0.1f;                              /* Solver.h:62 */
    p->temperature0 = 2.f;                       /* Solver.h:64 */
    p->temperature_env = 0.f;                    /* Solver.h:65 */
    p->jacobi_iters = 30;                        /* Solver.h:68 max_iter */
    p->dt = 0.2f;                                /* Solver.h:71 */
    p->source_x = nx / 2.f;                      /* Solver.h:78 */
    p->source_y = 10.f;                          /* Solver.h:79 */
    p->source_z = nz / 2.f;                      /* Solver.h:80 */
    p->source_radius = 7.0f;                     /* Solver.h:81 */
    p->scheme = FSO_SCHEME_MACCORMACK;           /* Solver.cu:11 */
    p->vort_confine = 0;                         /* Solver.cu:13 ships 1; off on the Jacobi hot path */
    p->curl_strength = 0.1f;                     /* Solver.h:73 */
    p->pressure_solver = FSO_PRESSURE_JACOBI;    /* Solver.cu:1178 ships the MGPCG branch */
    p->mg_level = 4;                             /* Solver.h:151 */
    p->init_smooth_steps = 2;                    /* Solver.h:153 */
    p->bottom_smooth_steps = 50;                 /* Solver.h:154 */
    p->iter_precision = 1e-3f;                   /* Solver.h:69 */
}

static float* zalloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

/* Solver.cu:1116-1130 + 879-952 (only the arrays the Jacobi path touches) + 993-995 */
fso_state* fso_create(const fso_params* p)
{
    fso_state* s = (fso_state*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->p = *p;
    size_t nx = p->nx, ny = p->ny, nz = p->nz, n = nx * ny * nz;
    s->ux = zalloc((nx + 1) * ny * nz);  s->new_ux = zalloc((nx + 1) * ny * nz);
    s->uy = zalloc(nx * (ny + 1) * nz);  s->new_uy = zalloc(nx * (ny + 1) * nz);
    s->uz = zalloc(nx * ny * (nz + 1));  s->new_uz = zalloc(nx * ny * (nz + 1));
    s->rho = zalloc(n);          s->new_rho = zalloc(n);
    s->temperature = zalloc(n);  s->new_temperature = zalloc(n);
    s->pressure = zalloc(n);     s->new_pressure = zalloc(n);
    s->div = zalloc(n);          s->r = zalloc(n);
    if (p->vort_confine) {
        size_t nv = fso_vort_size(p->nx, p->ny, p->nz);
        s->vortx = zalloc(nv); s->vorty = zalloc(nv); s->vortz = zalloc(nv);
    }
    if (p->pressure_solver == FSO_PRESSURE_MGPCG) {
        int m = 1 << (p->mg_level - 1);
        if (p->mg_level < 1 || p->nx % m || p->ny % m || p->nz % m) { fso_destroy(s); return NULL; }
        size_t space = fso_mg_space(p->nx, p->ny, p->nz, p->mg_level);
        s->cg_r = zalloc(space); s->cg_z = zalloc(space);                    /* :903-904, 939-940 */
        s->cg_p = zalloc(n); s->cg_Ap = zalloc(n); s->cg_x = zalloc(n);      /* :905-907 */
    }
    {
        float* all[] = {s->ux, s->uy, s->uz, s->new_ux, s->new_uy, s->new_uz, s->rho, s->new_rho,
                        s->temperature, s->new_temperature, s->pressure, s->new_pressure, s->div, s->r,
                        s->vortx, s->vorty, s->vortz, s->cg_r, s->cg_z, s->cg_p, s->cg_Ap, s->cg_x};
        s->n_owned = (int)(sizeof(all) / sizeof(all[0]));
        memcpy(s->owned, all, sizeof(all));
    }
    fso_grid_init(s->temperature, p->temperature_env, p->nx, p->ny, p->nz);
    return s;
}

void fso_destroy(fso_state* s)
{
    if (!s) return;
    if (s->n_owned) {
        for (int i = 0; i < s->n_owned; ++i) free(s->owned[i]);
    } else {   /* creation failed half-way */
        free(s->ux); free(s->uy); free(s->uz); free(s->new_ux); free(s->new_uy); free(s->new_uz);
        free(s->rho); free(s->new_rho); free(s->temperature); free(s->new_temperature);
        free(s->pressure); free(s->new_pressure); free(s->div); free(s->r);
        free(s->vortx); free(s->vorty); free(s->vortz);
    }
    free(s);
}

static void swapp(float** a, float** b) { float* t = *a; *a = *b; *b = t; }   /* :181-186 */

void fso_stage_source(fso_state* s)
{
    const fso_params* p = &s->p;
    fso_source(p->source_x, p->source_y, p->source_z, p->source_radius, s->rho, s->temperature,
               p->rho0, p->temperature0, p->nx, p->ny, p->nz);                  /* :1028 */
}

void fso_stage_force(fso_state* s)
{
    const fso_params* p = &s->p;
    fso_force_uy(s->uy, s->temperature, p->dt, p->temperature_env, p->nx, p->ny, p->nz); /* :1031 */
}

/* Solver.cu:1145-1165 */
void fso_stage_advect(fso_state* s)
{
    const fso_params* p = &s->p;
    int nx = p->nx, ny = p->ny, nz = p->nz, sc = p->scheme;
    fso_semilag(s->ux, s->new_ux, s->ux, s->uy, s->uz, p->dt, nx, ny, nz, 1, sc);
    fso_semilag(s->uy, s->new_uy, s->ux, s->uy, s->uz, p->dt, nx, ny, nz, 2, sc);
    fso_semilag(s->uz, s->new_uz, s->ux, s->uy, s->uz, p->dt, nx, ny, nz, 3, sc);
    swapp(&s->ux, &s->new_ux);
    swapp(&s->uy, &s->new_uy);
    swapp(&s->uz, &s->new_uz);
    fso_semilag(s->temperature, s->new_temperature, s->ux, s->uy, s->uz, p->dt, nx, ny, nz, 0, sc);
    swapp(&s->temperature, &s->new_temperature);
    fso_semilag(s->rho, s->new_rho, s->ux, s->uy, s->uz, p->dt, nx, ny, nz, 0, sc);
    swapp(&s->rho, &s->new_rho);
}

/* Solver.cu:1226-1297 without the prints.  `res` is the caller's f_pressure BY VALUE: the swap
 * at :1295 exchanges this local copy with the member x, so the solution reaches the pressure
 * field only because, from the second call on, member x IS the pressure buffer (the first call
 * leaves the pressure as it was).  Restated literally. */
static void conjugate(fso_state* s, float* res, const float* field)
{
    const fso_params* p = &s->p;
    const int nx = p->nx, ny = p->ny, nz = p->nz;
    const size_t n = (size_t)nx * ny * nz;
    float *r = s->cg_r, *z = s->cg_z, *cp = s->cg_p, *Ap = s->cg_Ap, *x = s->cg_x;
    for (size_t i = 0; i < n; ++i) { r[i] = -field[i]; x[i] = 0.f; }         /* InitConjugate :777-786 */
    float rTr = fso_dot(r, r, n);                                             /* init_rTr :1237-1239 */
    fso_mg_precondition(r, z, nx, ny, nz, p->mg_level, p->init_smooth_steps, p->bottom_smooth_steps);
    for (size_t i = 0; i < n; ++i) cp[i] = z[i] + 0 * cp[i];                  /* UpdateP(p, z, 0) :1250 */
    float old_zTr = fso_dot(z, r, n);                                         /* :1252-1254 */
    int it = 0;
    for (; it < p->jacobi_iters; ++it) {                                      /* max_iter :1256 */
        fso_apply_a(Ap, cp, nx, ny, nz);                                      /* :1259 */
        float pAp = fso_dot(cp, Ap, n);
        float alpha = old_zTr / pAp;                                          /* :1263 */
        for (size_t i = 0; i < n; ++i) {                                      /* UpdateResidual :801-810 */
            x[i] += alpha * cp[i];
            r[i] -= alpha * Ap[i];
        }
        rTr = fso_dot(r, r, n);                                               /* :1269-1271 */
        if (rTr < p->iter_precision || rTr == 0) { ++it; break; }            /* :1275 */
        fso_mg_precondition(r, z, nx, ny, nz, p->mg_level, p->init_smooth_steps, p->bottom_smooth_steps);
        float new_zTr = fso_dot(z, r, n);                                     /* :1285-1287 */
        float beta = new_zTr / old_zTr;
        for (size_t i = 0; i < n; ++i) cp[i] = z[i] + beta * cp[i];           /* UpdateP :812-820 */
        old_zTr = new_zTr;
    }
    s->last_cg_iters = it;
    s->last_rTr = rTr;
    { float* t = res; res = s->cg_x; s->cg_x = t; (void)res; }                /* swap(&res, &x) :1295 */
}

/* Solver.cu:1168-1193: divergence, then the Jacobi branch (`#if 0`, without the per-sweep print)
 * or the shipped MGPCG branch */
void fso_stage_project(fso_state* s)
{
    const fso_params* p = &s->p;
    fso_divergence(s->div, s->ux, s->uy, s->uz, p->nx, p->ny, p->nz);          /* :1176 */
    if (p->pressure_solver == FSO_PRESSURE_MGPCG) {
        conjugate(s, s->pressure, s->div);                                     /* :1191 */
        return;
    }
    for (int i = 0; i < p->jacobi_iters; ++i) {
        fso_jacobi(s->pressure, s->new_pressure, s->div, s->r, p->nx, p->ny, p->nz); /* :1182 */
        swapp(&s->pressure, &s->new_pressure);                                 /* :1183 */
    }
}

void fso_stage_gradient(fso_state* s)
{
    const fso_params* p = &s->p;
    fso_gradient(s->ux, s->uy, s->uz, s->pressure, p->nx, p->ny, p->nz);       /* :1111-1113 */
}

/* u -> omega, Solver.cu:1035-1037 (no-op unless vort_confine) */
void fso_stage_vorticity(fso_state* s)
{
    const fso_params* p = &s->p;
    if (!p->vort_confine) return;
    fso_find_vort(s->vortx, s->vorty, s->vortz, s->ux, s->uy, s->uz, p->nx, p->ny, p->nz);
}

/* Solver.cu:1078-1080 (no-op unless vort_confine) */
void fso_stage_confine(fso_state* s)
{
    const fso_params* p = &s->p;
    if (!p->vort_confine) return;
    fso_vort_confine(s->ux, s->vortx, s->vorty, s->vortz, p->nx, p->ny, p->nz, 1, p->curl_strength);
    fso_vort_confine(s->uy, s->vortx, s->vorty, s->vortz, p->nx, p->ny, p->nz, 2, p->curl_strength);
    fso_vort_confine(s->uz, s->vortx, s->vorty, s->vortz, p->nx, p->ny, p->nz, 3, p->curl_strength);
}

/* Solver.cu:1009-1114 with IVOCK/REFLECT/BURST 0 and VORT_CONFINE = p.vort_confine, then :1136 */
void fso_step(fso_state* s)
{
    fso_stage_source(s);
    fso_stage_force(s);
    fso_stage_vorticity(s);
    fso_stage_advect(s);
    fso_stage_confine(s);
    fso_stage_project(s);
    fso_stage_gradient(s);
    s->frame++;
}

float* fso_field(fso_state* s, int which)
{
    switch (which) {
    case FSO_UX: return s->ux;
    case FSO_UY: return s->uy;
    case FSO_UZ: return s->uz;
    case FSO_RHO: return s->rho;
    case FSO_T: return s->temperature;
    case FSO_P: return s->pressure;
    case FSO_DIV: return s->div;
    case FSO_R: return s->r;
    case FSO_VORTX: return s->vortx;
    case FSO_VORTY: return s->vorty;
    case FSO_VORTZ: return s->vortz;
    default: return NULL;
    }
}

size_t fso_field_size(const fso_state* s, int which)
{
    size_t nx = s->p.nx, ny = s->p.ny, nz = s->p.nz;
    switch (which) {
    case FSO_UX: return (nx + 1) * ny * nz;
    case FSO_UY: return nx * (ny + 1) * nz;
    case FSO_UZ: return nx * ny * (nz + 1);
    case FSO_RHO: case FSO_T: case FSO_P: case FSO_DIV: case FSO_R: return nx * ny * nz;
    case FSO_VORTX: case FSO_VORTY: case FSO_VORTZ:
        return s->p.vort_confine ? fso_vort_size(s->p.nx, s->p.ny, s->p.nz) : 0;
    default: return 0;
    }
}

/* residual of the LAST sweep's input iterate, as :442 leaves it in r */
double fso_residual(const fso_state* s)
{
    return fso_sum_sq(s->r, (size_t)s->p.nx * s->p.ny * s->p.nz);
}

int fso_frame(const fso_state* s) { return s->frame; }

int fso_last_cg_iters(const fso_state* s) { return s->last_cg_iters; }
float fso_last_rtr(const fso_state* s) { return s->last_rTr; }
