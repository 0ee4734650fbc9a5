/*
 * Solver.h -- source-compatible stand-in for the reference's CuSolver/Include/Solver.h,
 * so that the Renderer (Renderer/Source/FluidSolver.cpp:152-156, 210-213; main.cpp:11,17)
 * and the headless programs compile and link against the gfx950 library unchanged.
 *
 * Same public names and meaning as the reference class (Solver.h:13-44):
 *   Solver(), ~Solver(), Initialize(), Update(), Advect(), Project(), GetDensityField(),
 * and the internal steps it happens to declare public (Solver.h:28-42): Scatter(),
 * ResetGridValue(), Reduce(), Restrict(), Conjugate(), MG_Preconditioner(), PsiVCycle();
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
	      mgpcg(other.mgpcg), fmad(other.fmad), curl_strength(other.curl_strength) { other.handle = nullptr; }
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
	/* The rest of the reference's public surface (Solver.h:28-42).  Its Renderer calls none of
	   them (Renderer/Source/FluidSolver.cpp:152-156, 210-213); they are internal steps of
	   Update() that the reference declares public, and they act on private device arrays, so a
	   caller can observe them only through later results. */
	/* empty in the reference (Solver.cu:1195-1202) */
	void Scatter() {}
	/* density and velocity back to zero (Solver.cu:1204-1210; never called by the reference) */
	void ResetGridValue();
	/* The reference reduces its private `temp` array in place (Solver.cu:1212-1224), a step of
	   its dot products.  Here every dot product is a fused, deterministic reduction inside the
	   kernels of the solve, so there is nothing left to do: no-op. */
	void Reduce() {}
	/* One restriction step between two levels of the private multigrid arrays (Solver.cu:862-877);
	   only meaningful inside MG_Preconditioner(), which performs it itself: no-op. */
	void Restrict(int /*offset*/, int /*max_pos_x*/, int /*max_pos_y*/, int /*max_pos_z*/) {}
	/* The pressure solve of Project() on the current divergence (Solver.cu:1226-1297).  The
	   reference's only call passes its own pressure and divergence arrays (Solver.cu:1191), which
	   are private, so the arguments are ignored and the solver's own fields are used; with
	   SetMultigridSolver(false) the Jacobi sweeps run instead. */
	void Conjugate(float* /*res*/ = nullptr, float* /*field*/ = nullptr);
	/* one V-cycle z = M^-1 r on the private multigrid arrays (Solver.cu:1312-1366); needs the
	   multigrid solver (the default) */
	void MG_Preconditioner();
	/* Stream-function V-cycles of IVOCK (Solver.cu:1299-1310), which the reference compiles out
	   (IVOCK 0, Solver.cu:14) and this library does not contain: no-op. */
	void PsiVCycle() {}
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
	/* new: true = evaluate the a*b + c shaped expressions of the advection and the buoyancy with
	   fused multiply-adds, as a contracting build of the reference (nvcc -fmad=true, its default)
	   does (FS_FLAG_FMAD in fluidsolver.h); false (default) = every product and sum rounded on its own */
	void SetFusedMultiplyAdd(bool on) { fmad = on; }

private:
	fs_solver* handle = nullptr;
	int jacobi_iters = 30;
	bool vort_confine = true;
	bool mgpcg = true;
	bool fmad = false;
	float curl_strength = 0.1f;
};
