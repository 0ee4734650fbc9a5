// fs_headless.cpp -- the Renderer's solver calls without the window: construct the Solver by
// value, Initialize, Update, GetDensityField per frame (reference
// Renderer/Source/FluidSolver.cpp:152-156, 207-214).  Links only the `Solver` class.
#include <Solver.h>

#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char** argv)
{
	int frames = argc > 1 ? std::atoi(argv[1]) : 10;
	// default: the configuration the reference is compiled with (MGPCG + vorticity confinement),
	// which is also the class's default; second argument "jacobi": the Jacobi branch of
	// Solver::Project without confinement
	bool jacobi = argc > 2 && std::string(argv[2]) == "jacobi";
	// third argument "surface": afterwards, call the rest of the public class the reference declares
	// (CuSolver/Include/Solver.h:28-42) and print what can be observed of it
	bool surface = argc > 3 && std::string(argv[3]) == "surface";
	Solver solver = Solver();
	if (jacobi) {
		solver.SetMultigridSolver(false);
		solver.SetVorticityConfinement(false);
	}
	solver.Initialize();
	solver.Update();
	float* rho = solver.GetDensityField();
	for (int f = 1; f < frames; ++f) {
		solver.Update();
		rho = solver.GetDensityField();
	}
	double sum = 0.0;
	float mx = 0.f;
	for (size_t q = 0; q < (size_t)Nx * Ny * Nz; ++q) {
		sum += rho[q];
		if (rho[q] > mx) mx = rho[q];
	}
	float* uy = solver.GetVelocityField(1);
	float umx = 0.f;
	for (size_t q = 0; q < (size_t)Nx * (Ny + 1) * Nz; ++q)
		if (uy[q] > umx) umx = uy[q];
	std::printf("grid %dx%dx%d frames %d  sum(rho) %.6f  max(rho) %.6f  max(uy) %.6f\n", Nx, Ny, Nz,
	            frames, sum, mx, umx);
	if (surface) {
		solver.Scatter();
		solver.Reduce();
		solver.Restrict(0, Nx, Ny, Nz);
		solver.PsiVCycle();
		if (!jacobi) solver.MG_Preconditioner();
		solver.Advect();
		solver.Conjugate();          // the solve of Project() on the divergence as it stands
		solver.Conjugate(nullptr, nullptr);
		solver.Project();
		float before = 0.f;
		rho = solver.GetDensityField();
		for (size_t q = 0; q < (size_t)Nx * Ny * Nz; ++q) before = rho[q] > before ? rho[q] : before;
		solver.ResetGridValue();
		rho = solver.GetDensityField();
		float after = 0.f;
		for (size_t q = 0; q < (size_t)Nx * Ny * Nz; ++q) after = rho[q] != 0.f ? 1.f : after;
		float* ux = solver.GetVelocityField(0);
		for (size_t q = 0; q < (size_t)(Nx + 1) * Ny * Nz; ++q) after = ux[q] != 0.f ? 1.f : after;
		std::printf("surface: max(rho) before reset %.6f, nonzero after reset %d\n", before, (int)after);
	}
	return 0;
}
