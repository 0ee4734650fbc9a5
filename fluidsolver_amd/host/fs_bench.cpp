// fs_bench.cpp -- headless C++ benchmark over the C ABI only (no Python, no torch).
//   fs_bench [n=256] [sweeps=80] [steps=20] [warmup=5] [dump_prefix|-] [flags=0] [pressure_solver=0]
// Prints one JSON line: cell-updates/s of `steps` steps and, from a second profiled run of as
// many steps, the device time per stage.  With dump_prefix, writes
// the final rho/ux/uy/uz as raw little-endian float32, x fastest (dense reference layout), and
// <dump_prefix>.json describing them (read back by fluidsolver_amd.dump.read_dump).
#include "fluidsolver.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static void die(const char* what)
{
	std::fprintf(stderr, "fs_bench: %s: %s\n", what, fs_last_error());
	std::exit(1);
}

int main(int argc, char** argv)
{
	int n = argc > 1 ? std::atoi(argv[1]) : 256;
	int sweeps = argc > 2 ? std::atoi(argv[2]) : 80;
	int steps = argc > 3 ? std::atoi(argv[3]) : 20;
	int warmup = argc > 4 ? std::atoi(argv[4]) : 5;
	const char* dump = argc > 5 && std::string(argv[5]) != "-" ? argv[5] : nullptr;
	unsigned flags = argc > 6 ? (unsigned)std::strtoul(argv[6], nullptr, 0) : 0u;   // FS_FLAG_* bits

	fs_params p;
	if (fs_default_params(&p, n, n, n)) die("fs_default_params");
	p.jacobi_iters = sweeps;
	p.flags = flags;
	if (argc > 7) p.pressure_solver = std::atoi(argv[7]);   // FS_PRESSURE_*
	fs_solver* s = nullptr;
	if (fs_create(&p, &s)) die("fs_create");
	if (fs_steps(s, warmup) || fs_sync(s)) die("warmup");
	// pass 1: the product path (graph replay), wall clock around `steps` steps
	auto t0 = std::chrono::steady_clock::now();
	if (fs_steps(s, steps) || fs_sync(s)) die("fs_steps");
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	// pass 2: the same steps launched eagerly with a pair of events around every stage
	fs_profile_enable(s, 1);
	fs_profile_reset(s);
	if (fs_steps(s, steps) || fs_sync(s)) die("fs_steps (profiled)");
	static const char* names[FS_NUM_STAGES] = {"source", "force", "advect_velocity", "advect_scalars",
	                                           "divergence", "jacobi", "gradient", "vorticity", "confine", "mgpcg"};
	std::printf("{\"n\": %d, \"sweeps\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"cell_updates_per_s\": %.6e, "
	            "\"step_gbs\": %.1f, \"stage_ms_per_step\": {",
	            n, sweeps, steps, sec / steps * 1e3, (double)n * n * n * steps / sec,
	            (144.0 + 12.0 * sweeps) * n * n * n * steps / sec / 1e9);
	for (int q = 0; q < FS_NUM_STAGES; ++q) {
		double ms = 0;
		long long launches = 0;
		fs_profile_read(s, q, &ms, &launches);
		std::printf("%s\"%s\": %.4f", q ? ", " : "", names[q], ms / steps);
	}
	std::printf("}}\n");
	if (dump) {
		// raw little-endian float32, x fastest, plus a JSON sidecar naming dims and parameters
		static const char* fn[4] = {"ux", "uy", "uz", "rho"};
		static const int ids[4] = {FS_UX, FS_UY, FS_UZ, FS_RHO};
		const int dims[4][3] = {{n + 1, n, n}, {n, n + 1, n}, {n, n, n + 1}, {n, n, n}};
		std::string side = std::string(dump) + ".json";
		FILE* js = std::fopen(side.c_str(), "w");
		if (!js) { std::perror(side.c_str()); return 1; }
		std::fprintf(js, "{\"format\": \"fluidsolver-f32-v1\", \"endian\": \"little\", \"order\": \"x-fastest\", "
		                 "\"nx\": %d, \"ny\": %d, \"nz\": %d, \"frame\": %d, \"dt\": %.9g, \"jacobi_iters\": %d, "
		                 "\"scheme\": %d, \"flags\": %u, \"fields\": {",
		             n, n, n, fs_frame(s), p.dt, p.jacobi_iters, p.scheme, p.flags);
		for (int q = 0; q < 4; ++q) {
			const float* h = fs_field_host(s, ids[q]);
			if (!h) die("fs_field_host");
			std::string path = std::string(dump) + "_" + fn[q] + ".f32";
			FILE* f = std::fopen(path.c_str(), "wb");
			if (!f) { std::perror(path.c_str()); return 1; }
			std::fwrite(h, sizeof(float), fs_field_size(s, ids[q]), f);
			std::fclose(f);
			const char* base = std::strrchr(path.c_str(), '/');
			std::fprintf(js, "%s\"%s\": {\"file\": \"%s\", \"dims\": [%d, %d, %d]}", q ? ", " : "", fn[q],
			             base ? base + 1 : path.c_str(), dims[q][0], dims[q][1], dims[q][2]);
		}
		std::fprintf(js, "}}\n");
		std::fclose(js);
	}
	fs_destroy(s);
	return 0;
}
