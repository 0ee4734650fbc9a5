/*
 * Solver.h -- source-compatible stand-in for the reference's CuSolver/Include/Solver.h,
 * so that the Renderer (Renderer/Source/FluidSolver.cpp:152-156, 210-213; main.cpp:11,17)
 * and the headless programs compile and link against the gfx950 library unchanged.
 *
 * Same public names and meaning as the reference class (Solver.h:13-44):
 *   Solver(), ~Solver(), Initialize(), Update(), Advect(), Project(), GetDensityField(),
 * plus velocity access, which the reference lacks.  The object holds the C-ABI handle of
 * include/fluidsolver.h and a few settings, so constructing it by value on the caller's
 * stack, as the Renderer does, does not depend on the library's internals.
 *
 * Error behaviour is the reference's (CudaUtility.h:12-23): a failed GPU call prints the
 * error to stderr and exits with status 1.
 */
#pragma once

#if defined(_WIN32)
#ifdef SOLVER_EXPORTS
#define SOLVER_API __declspec(dllexport)
#else
#define SOLVER_API __declspec(dllimport)
#endif
#else
#define SOLVER_API __attribute__((visibility("default")))
#endif

/* compile-time grid of the reference (Solver.h:9-11); override with -DNx=.. etc. */
#ifndef Nx
#define Nx 64
#endif
#ifndef Ny
#define Ny 128
#endif
#ifndef Nz
#define Nz 64
#endif

struct fs_solver;

class SOLVER_API Solver
{
public:
	Solver() {}
	~Solver();
	Solver(Solver&& other) noexcept
	    : handle(other.handle), jacobi_iters(other.jacobi_iters), vort_confine(other.vort_confine),
	      mgpcg(other.mgpcg), curl_strength(other.curl_strength) { other.handle = nullptr; }
	Solver& operator=(Solver&& other) noexcept;
	Solver(const Solver&) = delete;
	Solver& operator=(const Solver&) = delete;

	/* allocate + zero all fields, T = T_env (reference Solver.cu:1116-1130).  Inline on purpose:
	   the grid is the Nx, Ny, Nz of the translation unit that includes this header, as with the
	   reference, without rebuilding the library. */
	void Initialize() { InitializeGrid(Nx, Ny, Nz); }
	/* new: the same with a run-time grid */
	void InitializeGrid(int nx, int ny, int nz);
	/* one simulation step; prints "frame: N" like the reference (Solver.cu:1132-1137) */
	void Update();
	/* the two sub-steps the reference also exposes publicly (Solver.h:24,26) */
	void Advect();
	void Project();
	/* blocking copy of the density field into a solver-owned host buffer of
	   Nx*Ny*Nz floats, x fastest (Solver.cu:1368-1372); valid until the next call */
	float* GetDensityField();
	/* new: velocity component c = 0,1,2 -> ux (Nx+1,Ny,Nz), uy (Nx,Ny+1,Nz), uz (Nx,Ny,Nz+1) */
	float* GetVelocityField(int c);
	/* new: iteration limit of the pressure solve = Jacobi sweeps per step on the Jacobi branch
	   (reference max_iter, Solver.h:68); call before Initialize */
	void SetJacobiIterations(int iters) { jacobi_iters = iters; }
	/* The defaults are the configuration the reference is compiled with (Solver.cu:10-15: MGPCG 1,
	   VORT_CONFINE 1; max_iter 30, curl_strength 0.1, Solver.h:68,73), so a caller that only
	   knows Initialize / Update / GetDensityField gets the simulation the reference ships.  The
	   setters below (new; call before Initialize) select the other compile-time branches. */
	/* vorticity confinement (Solver.cu:13, 1035-1037, 1078-1080), in its defined-behaviour variant
	   (see FS_FLAG_VORT_CONFINE in fluidsolver.h); on by default */
	void SetVorticityConfinement(bool on, float strength = 0.1f) { vort_confine = on; curl_strength = strength; }
	/* true (default) = the MGPCG pressure solve the reference enables (Solver.cu:10, 1190-1192; every
	   grid dimension must be a multiple of 8); false = the Jacobi branch it compiles out
	   (Solver.cu:1178-1189), jacobi_iters sweeps per step */
	void SetMultigridSolver(bool on) { mgpcg = on; }

private:
	fs_solver* handle = nullptr;
	int jacobi_iters = 30;
	bool vort_confine = true;
	bool mgpcg = true;
	float curl_strength = 0.1f;
};
