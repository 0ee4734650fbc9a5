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
	return 0;
}
