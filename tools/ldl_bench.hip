// ldl_bench.hip -- diagnostic: the dense bordered LDL^T solvers of the phase-split LM's K2 alone, on a synthetic SPD system, `reps` solves per
// launch (hipEvent timing): per-solve latency of one workgroup and throughput of a full grid, ldl_solve_mfma_tiles against ldl_solve_regtile,
// checked against a host solve.  Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I srba-stereo-slam_amd/csrc tools/ldl_bench.hip -o _ab/ldl_bench [-DLDLM_...]
// Run on the GPU box: _ab/ldl_bench [n = 240] [reps = 200] [workgroups = 1] [index of a diagonal entry to make negative: the failure path]
// Beyond 255 unknowns it also times the leading panels alone (ldl_leading_mfma in a kernel of its own) and the restore copy of the system that every
// timed solve includes; -DLDL_LEAD_STAMPS: clock sums of the leading panels' phases and of a chunk step; -DSSK_K2_LEAD_MFMA=0: the leading columns by
// ldl_leading_panels (the first step of round 5).
#include "ssk_lm_split.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#pragma clang fp contract(fast)
namespace {
#define SSK_LM_NT 512
#undef SSK_LM_WAVES
#define SSK_LM_WAVES 2
#include "ssk_lm_device.h"
constexpr int REC = SSK_ND_REC, STG = SSK_ND_STAGE, LREC = SSK_ND_LREC, SCR = SSK_ND_SCR;
constexpr int R_Q = 9, R_P = 12, R_V = 21;
#include "ssk_lm_node_dev.h"
#include "ssk_lm_ldl_mfma.h"
#include "ssk_lm_ldl_wave.h"

template <int WHICH>
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void bench_kernel(const double* A0, double* Awork, int np, int n, double* x, int reps, int dyn_doubles, int* ok_out)
{
	// (systems beyond the solvers' register / accumulator tiles are factorised partly IN PLACE -- the leading columns --: every workgroup works on its own
	// copy, restored from the pristine system in front of every solve; the copy is inside the timed region: ~1.2 MB from L2 at 396 unknowns)
	const size_t words = (size_t)(n + 1) * np;
	double* A = Awork + blockIdx.x * words;
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];
	const int n6p = (n + 1) & ~1;
	double* s_b = s_dyn;
	double* s_inv = s_b + n6p;
	double* lds = s_inv + n6p;
	const int lds_doubles = dyn_doubles - 2 * n6p;
	bool ok = true;
	for (int r = 0; r < reps; ++r) {
		for (size_t i = threadIdx.x; i < words; i += NT) A[i] = A0[i];
		__syncthreads();
		if constexpr (WHICH == 1) ok &= ldl_solve_mfma_lead(A, np, n, s_b, s_inv, lds, lds_doubles);
		else ok &= ldl_solve_regtile(A, np, n, s_b, s_inv, lds, lds_doubles);
		__syncthreads();
	}
	if (blockIdx.x == 0) {
		for (int i = threadIdx.x; i < n; i += NT) x[i] = s_b[i];
		if (threadIdx.x == 0) *ok_out = ok ? 1 : 0;
	}
}
// the leading panels alone (ldl_leading_mfma in a kernel of its own: its registers without the tile solver's beside them)
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void bench_lead_kernel(const double* A0, double* Awork, int np, int n, int reps, int* ok_out, int copy_only)
{
	const size_t words = (size_t)(n + 1) * np;
	double* A = Awork + blockIdx.x * words;
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];
	const int n6p = (n + 1) & ~1;
	double* s_inv = s_dyn + n6p;
	double* lds = s_inv + n6p;
	const int n1 = n <= LDLM_NMAX ? 0 : 16 * ((n - LDLM_NMAX + 15) >> 4);
	int bad = 0;
	for (int r = 0; r < reps; ++r) {
		for (size_t i = threadIdx.x; i < words; i += NT) A[i] = A0[i];
		__syncthreads();
		if (!copy_only) bad |= ldl_leading_mfma(A, np, n, n1, s_inv, lds);
		__syncthreads();
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) *ok_out = !bad;
}
// the one-wavefront solvers of the persistent node kernel (E <= 10): input = 6 x 6 block accumulators + reduced gradient in LDS; WHICH 2: matrix cores
// (ldl_solve_mfma_wave), 3: a block per lane on the vector ALUs (ldl_solve_blocks6_wave)
template <int WHICH>
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void bench_wave_kernel(const double* Hb_g, const double* g_g, int n, double* x, int reps, int* ok_out)
{
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];
	double* Hb = s_dyn; double* g = Hb + 36 * 55; double* s_b = g + 64; double* lds = s_b + 64;
	for (int i = threadIdx.x; i < 36 * 55; i += NT) Hb[i] = Hb_g[i];
	for (int i = threadIdx.x; i < 64; i += NT) g[i] = i < n ? g_g[i] : 0.0;
	__syncthreads();
	bool ok = true;
	for (int r = 0; r < reps; ++r) {
		if constexpr (WHICH == 2) ok &= ldl_solve_mfma_wave(Hb, g, 0.0, n, s_b, lds);
		else ok &= ldl_solve_blocks6_wave(Hb, g, 0.0, n / 6, s_b, lds);
		__syncthreads();
	}
	if (blockIdx.x == 0) {
		for (int i = threadIdx.x; i < n; i += NT) x[i] = s_b[i];
		if (threadIdx.x == 0) *ok_out = ok ? 1 : 0;
	}
}
}  // namespace

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
	const int n = argc > 1 ? std::atoi(argv[1]) : 240, reps = argc > 2 ? std::atoi(argv[2]) : 200, wgs = argc > 3 ? std::atoi(argv[3]) : 1;
	const int bad_at = argc > 4 ? std::atoi(argv[4]) : -1;      // >= 0: that diagonal entry made negative -- both solvers must report failure (ok 0), not hang
	const int np = n | 1;
	std::vector<double> A((size_t)(n + 1) * np, 0.0), b(n);
	// SPD: M M^T + n I with a banded-ish random M, both triangles; border row = right-hand side
	std::srand(7);
	std::vector<double> M((size_t)n * n);
	for (double& v : M) v = (std::rand() / (double)RAND_MAX - 0.5);
	for (int i = 0; i < n; ++i)
		for (int j = 0; j <= i; ++j) {
			double sacc = 0.0;
			for (int k = 0; k < n; ++k) sacc += M[(size_t)i * n + k] * M[(size_t)j * n + k];
			if (i == j) sacc += 0.05 * n;
			A[(size_t)i * np + j] = sacc; A[(size_t)j * np + i] = sacc;
		}
	for (int i = 0; i < n; ++i) { b[i] = std::rand() / (double)RAND_MAX - 0.5; A[(size_t)n * np + i] = b[i]; }
	if (bad_at >= 0 && bad_at < n) A[(size_t)bad_at * np + bad_at] = -1.0e6;
	// host solve (Cholesky)
	std::vector<double> L((size_t)n * n, 0.0), xr(b);
	for (int j = 0; j < n; ++j) {
		double d = A[(size_t)j * np + j];
		for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
		L[(size_t)j * n + j] = std::sqrt(d);
		for (int i = j + 1; i < n; ++i) {
			double v = A[(size_t)i * np + j];
			for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
			L[(size_t)i * n + j] = v / L[(size_t)j * n + j];
		}
	}
	for (int i = 0; i < n; ++i) { for (int k = 0; k < i; ++k) xr[i] -= L[(size_t)i * n + k] * xr[k]; xr[i] /= L[(size_t)i * n + i]; }
	for (int i = n - 1; i >= 0; --i) { for (int k = i + 1; k < n; ++k) xr[i] -= L[(size_t)k * n + i] * xr[k]; xr[i] /= L[(size_t)i * n + i]; }
	double *dA, *dx; int* dok;
	CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dok, 4));
	CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
	double* dW; CK(hipMalloc(&dW, A.size() * 8 * (size_t)wgs));
	const size_t dyn = 8 * (size_t)(2 * ((n + 1) & ~1) + LDLM_DOUBLES + 256);
	CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
	CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int which = 0; which < 2; ++which) {
		if (which == 1 && n > LDLM_LEAD_NMAX) break;
		float best = 1e30f;
		for (int rep = 0; rep < 4; ++rep) {
			CK(hipEventRecord(e0));
			if (which) hipLaunchKernelGGL(bench_kernel<1>, dim3(wgs), dim3(NT), dyn, 0, dA, dW, np, n, dx, reps, (int)(dyn / 8), dok);
			else hipLaunchKernelGGL(bench_kernel<0>, dim3(wgs), dim3(NT), dyn, 0, dA, dW, np, n, dx, reps, (int)(dyn / 8), dok);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep) best = std::fmin(best, ms);
		}
		std::vector<double> x(n); int ok;
		CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost));
		double err = 0.0, nrm = 0.0;
		for (int i = 0; i < n; ++i) { err = std::fmax(err, std::fabs(x[i] - xr[i])); nrm = std::fmax(nrm, std::fabs(xr[i])); }
		std::printf("%-22s n %d, %d workgroups x %d solves: %.2f us per solve (per workgroup), ok %d, max |x - x_host| / max |x_host| = %.2e\n",
		            which ? "ldl_solve_mfma_lead" : "ldl_solve_regtile", n, wgs, reps, 1e3 * best / reps, ok, err / nrm);
#ifdef LDL_LEAD_STAMPS
		if (which) {
			unsigned long long tt[16];
			CK(hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_lead_t), sizeof tt));
			// (s_memtime: 100 MHz; four launches of `reps` solves)
			std::printf("  stamps, us per solve: tile load %.1f, earlier panels' updates %.1f, diagonal tile %.1f, TRSM + stores %.1f | leading %.1f, trailing %.1f, leading back-substitution %.1f\n",
			            tt[0] / 100.0 / (4 * reps), tt[1] / 100.0 / (4 * reps), tt[2] / 100.0 / (4 * reps), tt[3] / 100.0 / (4 * reps), tt[4] / 100.0 / (4 * reps), tt[5] / 100.0 / (4 * reps), tt[6] / 100.0 / (4 * reps));
			std::printf("  chunk step: first barrier %.1f, commit + fetch issue %.1f, second barrier %.1f, MFMAs %.1f\n", tt[8] / 100.0 / (4 * reps), tt[9] / 100.0 / (4 * reps), tt[10] / 100.0 / (4 * reps), tt[11] / 100.0 / (4 * reps));
		}
#endif
	}
	if (n > LDLM_NMAX && n <= LDLM_LEAD_NMAX) {
		CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_lead_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		float best = 1e30f, best0 = 1e30f;
		for (int rep = 0; rep < 8; ++rep) {
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(bench_lead_kernel, dim3(wgs), dim3(NT), dyn, 0, dA, dW, np, n, reps, dok, rep & 1);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep > 1) { if (rep & 1) best0 = std::fmin(best0, ms); else best = std::fmin(best, ms); }
		}
		std::printf("%-22s n %d: %.2f us per pass, of which the restore copy %.2f us; leading columns %d\n", "ldl_leading_mfma alone", n, 1e3 * best / reps, 1e3 * best0 / reps, 16 * ((n - LDLM_NMAX + 15) >> 4));
	}
	if (n <= 60 && n % 6 == 0) {
		// the node kernel's one-wavefront solvers on the same system given as block accumulators (H(6 e + a, 6 e2 + b), e <= e2, at 36 (e2 (e2 + 1) / 2 + e) + 6 a + b)
		std::vector<double> Hb(36 * 55, 0.0);
		for (int e2 = 0; e2 < n / 6; ++e2)
			for (int e = 0; e <= e2; ++e)
				for (int a = 0; a < 6; ++a)
					for (int c = 0; c < 6; ++c) Hb[(size_t)36 * (e2 * (e2 + 1) / 2 + e) + 6 * a + c] = A[(size_t)(6 * e + a) * np + 6 * e2 + c];
		double *dHb, *dg;
		CK(hipMalloc(&dHb, Hb.size() * 8)); CK(hipMalloc(&dg, 64 * 8));
		CK(hipMemcpy(dHb, Hb.data(), Hb.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, b.data(), n * 8, hipMemcpyHostToDevice));
		const size_t dynw = 8 * (size_t)(36 * 55 + 128 + (LDLW_DOUBLES > LDLB6_DOUBLES ? LDLW_DOUBLES : LDLB6_DOUBLES) + 64);
		for (int which = 2; which < 4; ++which) {
			float best = 1e30f;
			for (int rep = 0; rep < 4; ++rep) {
				CK(hipEventRecord(e0));
				if (which == 2) hipLaunchKernelGGL(bench_wave_kernel<2>, dim3(wgs), dim3(NT), dynw, 0, dHb, dg, n, dx, reps, dok);
				else hipLaunchKernelGGL(bench_wave_kernel<3>, dim3(wgs), dim3(NT), dynw, 0, dHb, dg, n, dx, reps, dok);
				CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep) best = std::fmin(best, ms);
			}
			std::vector<double> x(n); int ok;
			CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost));
			double err = 0.0, nrm = 0.0;
			for (int i = 0; i < n; ++i) { err = std::fmax(err, std::fabs(x[i] - xr[i])); nrm = std::fmax(nrm, std::fabs(xr[i])); }
			std::printf("%-22s n %d, %d workgroups x %d solves: %.2f us per solve (per workgroup), ok %d, max |x - x_host| / max |x_host| = %.2e\n",
			            which == 2 ? "ldl_solve_mfma_wave" : "ldl_solve_blocks6_wave", n, wgs, reps, 1e3 * best / reps, ok, err / nrm);
		}
	}
	return 0;
}
