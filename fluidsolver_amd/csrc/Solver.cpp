// Solver.cpp -- the reference-compatible C++ class over the C ABI (include/Solver.h).
#include "Solver.h"
#include "fluidsolver.h"

#include <cstdio>
#include <cstdlib>
#include <initializer_list>

namespace {
// reference error convention: print and exit(1) (CuSolver/Include/CudaUtility.h:12-23)
void check(int rc, const char* what)
{
	if (rc != FS_OK) {
		std::fprintf(stderr, "GPU error at: %s\n%s\n", what, fs_last_error());
		std::exit(1);
	}
}
}

Solver::~Solver()
{
	if (handle) fs_destroy(handle);
}

Solver& Solver::operator=(Solver&& other) noexcept
{
	if (this != &other) {
		if (handle) fs_destroy(handle);
		handle = other.handle;
		jacobi_iters = other.jacobi_iters;
		vort_confine = other.vort_confine;
		mgpcg = other.mgpcg;
		fmad = other.fmad;
		curl_strength = other.curl_strength;
		other.handle = nullptr;
	}
	return *this;
}

void Solver::InitializeGrid(int nx, int ny, int nz)
{
	fs_params p;
	check(fs_default_params(&p, nx, ny, nz), "fs_default_params");
	p.jacobi_iters = jacobi_iters;
	if (vort_confine) p.flags |= FS_FLAG_VORT_CONFINE;
	p.curl_strength = curl_strength;
	if (mgpcg) p.pressure_solver = FS_PRESSURE_MGPCG;
	if (fmad) p.flags |= FS_FLAG_FMAD;
	if (handle) fs_destroy(handle);
	handle = nullptr;
	check(fs_create(&p, &handle), "fs_create");
}

void Solver::Update()
{
	std::printf("frame: %d\n", fs_frame(handle));
	check(fs_step(handle), "fs_step");
}

void Solver::Advect() { check(fs_stage_advect(handle), "fs_stage_advect"); }

void Solver::Project() { check(fs_stage_project(handle), "fs_stage_project"); }

void Solver::ResetGridValue()
{
	for (int f : {FS_RHO, FS_UX, FS_UY, FS_UZ}) check(fs_field_fill(handle, f, 0.f), "fs_field_fill");
}

void Solver::Conjugate(float*, float*) { check(fs_stage_pressure_solve(handle), "fs_stage_pressure_solve"); }

void Solver::MG_Preconditioner() { check(fs_mg_vcycle(handle), "fs_mg_vcycle"); }

float* Solver::GetDensityField()
{
	const float* p = fs_density_host(handle);
	if (!p) check(FS_ERR_HIP, "fs_density_host");
	return const_cast<float*>(p);
}

float* Solver::GetVelocityField(int c)
{
	const float* p = fs_velocity_host(handle, c);
	if (!p) check(FS_ERR_HIP, "fs_velocity_host");
	return const_cast<float*>(p);
}
