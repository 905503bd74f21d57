// lm_node_emu.cpp -- CPU emulation of the DATA FLOW of lm_solve_node_kernel (srba-stereo-slam_amd/csrc/ssk_lm_node.hip) on the tables
// the library's own host builder produces (ssk_lm_node_build.h): groups, staging runs, root / node lanes, node records, block-sorted
// items, gradient items, the landmark step.  Test infrastructure (tests/test_lm_node_form.py compares its reduced system with the dense
// textbook form): it checks the symbolic structure and the bit fields without a GPU; the kernel's lane mechanics (DPP sums, LDS
// overlays) are not modelled.  Build: g++ -O2 -shared -fPIC tools/lm_node_emu.cpp -o <out>/liblm_node_emu.so
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
#include "../srba-stereo-slam_amd/csrc/ssk_lm_node_build.h"

namespace {

void cross(const double* a, const double* b, double* c)
{
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

void build_table(const nd_host& H, const nd_win& W, const double* pose, std::vector<double>& tab)
{
	tab.assign(12 * (size_t)W.n_suf, 0.0);
	for (int id = 0; id < W.n_suf; ++id) {
		int code[8], n = 0, cur = id;
		while (cur != 0) { const nd_i2 r = H.suf[(size_t)W.suf0 + cur]; code[n++] = r.x; cur = r.y; }
		double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
		for (int k = n - 1; k >= 0; --k) {
			const double* X = pose + 12 * (code[k] >> 1);
			const int dir = code[k] & 1, si = dir ? 1 : 4, sj = dir ? 4 : 1;
			double tn[3] = {T[3], T[7], T[11]};
			if (dir) { tn[0] -= X[3]; tn[1] -= X[7]; tn[2] -= X[11]; }
			double O[12];
			for (int i = 0; i < 3; ++i) {
				const double r0 = X[si * i], r1 = X[si * i + sj], r2 = X[si * i + 2 * sj];
				for (int j = 0; j < 3; ++j) O[4 * i + j] = r0 * T[j] + r1 * T[4 + j] + r2 * T[8 + j];
				O[4 * i + 3] = r0 * tn[0] + r1 * tn[1] + r2 * tn[2] + (dir ? 0.0 : X[4 * i + 3]);
			}
			std::memcpy(T, O, sizeof T);
		}
		std::memcpy(&tab[12 * (size_t)id], T, sizeof T);
	}
}

}  // namespace

extern "C" {

// Linearisation at the window's state: Hred (n6 x n6, row major, both triangles, lambda on the diagonal) and the reduced gradient g.
// dp != NULL: also the landmark step df (3 per optimised slot) for the given edge step.  pre != 0: the lambda_0 pass (diag blocks of
// H_pp only, vmax = largest diagonal of an optimised landmark's block).  Returns 0, or -1 when the window does not fit the node form.
// form 1: the tables of the phase-split form (ssk_lm_split.hip): (tile, block) slots -> partial sums in "HBM" -> block-major reduction
int nd_emu_linearise_form(const ssk_lm_window* Wn, const ssk_lm_params* P, double lambda, int pre, int waves, double* Hred, double* g, const double* dp, double* df,
                          double* vmax_out, int* stats /* tiles, groups, items, max items per lane (split: most blocks of a tile), partial slots */, int form)
{
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, form == 1)) return -1;
	const nd_win& W = H.win[0];
	const int n6 = 6 * W.E;
	std::vector<double> tab;
	build_table(H, W, Wn->edge_inv_pose, tab);
	const int n_pairs = W.E * (W.E + 1) / 2;
	std::vector<double> Hb(36 * (size_t)n_pairs, 0.0), gacc((size_t)n6 + 1, 0.0);
	std::vector<double> scr(12 * H.lane.size(), 0.0);
	std::vector<double> part(36 * (size_t)(form == 1 ? H.part_slots : 0) + 36, 0.0), gpart(6 * (size_t)(form == 1 ? H.gpart_slots : 0) + 6, 0.0);
	double vmax = 0.0;
	const double* S = P->sensor_pose;
	int n_groups = 0;
	for (int t = 0; t < W.n_tiles; ++t) {
		const nd_tile& T = H.tile[(size_t)W.tile0 + t];
		std::vector<double> rec((size_t)waves * 64 * SSK_ND_REC, 0.0);
		for (int wv = 0; wv < T.n_grp; ++wv) {
			const nd_group& G = H.group[(size_t)W.grp0 + T.grp0 + wv];
			++n_groups;
			double stage[64 * 9];
			for (int lane = 0; lane < G.n_obs; ++lane) {
				const nd_i2 orec = H.obs[(size_t)W.obs0 + G.obs0 + lane];
				const nd_f4 px = H.px[(size_t)W.obs0 + G.obs0 + lane];
				const double* Tp = &tab[12 * (size_t)(orec.y & 0xFFFF)];
				const double* f = Wn->lm_pos + 3 * orec.x;
				double q[3], pc[3];
				for (int i = 0; i < 3; ++i) q[i] = Tp[4 * i] * f[0] + Tp[4 * i + 1] * f[1] + Tp[4 * i + 2] * f[2] + Tp[4 * i + 3];
				const double d0 = q[0] - S[3], d1 = q[1] - S[7], d2 = q[2] - S[11];
				for (int i = 0; i < 3; ++i) pc[i] = S[i] * d0 + S[4 + i] * d1 + S[8 + i] * d2;
				const double iz = 1.0 / pc[2], iz2 = iz * iz;
				const double r[4] = {(double)px.x - (P->cx_l + P->fx_l * pc[0] * iz), (double)px.y - (P->cy_l + P->fy_l * pc[1] * iz),
				                     (double)px.z - (P->cx_r + P->fx_r * (pc[0] - P->baseline) * iz), (double)px.w - (P->cy_r + P->fy_r * pc[1] * iz)};
				const double s2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
				const double w = P->use_robust_kernel ? std::sqrt(1.0 / std::sqrt(1.0 + s2 / (P->kernel_param * P->kernel_param))) : 1.0;
				const double Dm[12] = {P->fx_l * iz, 0, -P->fx_l * pc[0] * iz2, 0, P->fy_l * iz, -P->fy_l * pc[1] * iz2,
				                       P->fx_r * iz, 0, -P->fx_r * (pc[0] - P->baseline) * iz2, 0, P->fy_r * iz, -P->fy_r * pc[1] * iz2};
				double M[12], jf[12];
				for (int a = 0; a < 4; ++a)
					for (int b = 0; b < 3; ++b) M[3 * a + b] = Dm[3 * a] * S[4 * b] + Dm[3 * a + 1] * S[4 * b + 1] + Dm[3 * a + 2] * S[4 * b + 2];
				for (int a = 0; a < 4; ++a)
					for (int b = 0; b < 3; ++b) jf[3 * a + b] = w * (M[3 * a] * Tp[b] + M[3 * a + 1] * Tp[4 + b] + M[3 * a + 2] * Tp[8 + b]);
				double* o = stage + 9 * lane;
				int k = 0;
				for (int a = 0; a < 3; ++a)
					for (int b = a; b < 3; ++b) o[k++] = jf[a] * jf[b] + jf[3 + a] * jf[3 + b] + jf[6 + a] * jf[6 + b] + jf[9 + a] * jf[9 + b];
				for (int b = 0; b < 3; ++b) o[6 + b] = w * (jf[b] * r[0] + jf[3 + b] * r[1] + jf[6 + b] * r[2] + jf[9 + b] * r[3]);
			}
			// the kernel's run sums (ND_SCAN): per-landmark inclusive prefix sums over the observation lanes by six masked lane-shift steps (row_shr 1, 2, 4, 8
			// inside a row of 16 lanes, row_bcast15, row_bcast31; h = orec.y bits 24..30), then P[last] - P[first - 1] per root / node lane
			for (int lane = G.n_obs; lane < 64; ++lane) for (int c = 0; c < 9; ++c) stage[9 * lane + c] = 0.0;
			{
				int hh[64];
				for (int lane = 0; lane < 64; ++lane) hh[lane] = lane < G.n_obs ? (H.obs[(size_t)W.obs0 + G.obs0 + lane].y >> 24) & 0x7F : 0;
				double nxt[64 * 9];
				for (int step = 0; step < 6; ++step) {
					for (int lane = 0; lane < 64; ++lane) {
						const int lr = lane & 15, h = hh[lane];
						int src = -1; bool on = false;
						if (step < 4) { const int d = 1 << step; src = lr >= d ? lane - d : -1; on = h >= d; }
						else if (step == 4) { src = (lane & 16) ? (lane & ~15) - 1 : -1; on = (lane & 16) && h >= lr + 1; }      // (rows 1 and 3: row 2 takes rows 0 + 1 in one piece below)
						else { src = lane >= 32 ? 31 : -1; on = lane >= 32 && h >= lane - 31; }
						for (int c = 0; c < 9; ++c) nxt[9 * lane + c] = stage[9 * lane + c] + ((on && src >= 0) ? stage[9 * src + c] : 0.0);
					}
					std::memcpy(stage, nxt, sizeof nxt);
				}
			}
			double vs[64][9], lrec[64][16];
			for (int lane = 0; lane < G.n_lanes; ++lane) {
				const nd_i4 lr = H.lane[(size_t)W.lane0 + G.lane0 + lane];
				const int a = lr.y & 0xFF, n = (lr.y >> 8) & 0xFF, first = (lr.y >> 16) & 1;
				for (int c = 0; c < 9; ++c) vs[lane][c] = n > 0 ? stage[9 * (a + n - 1) + c] - (first ? 0.0 : stage[9 * (a > 0 ? a - 1 : 0) + c]) : 0.0;
			}
			for (int lane = 0; lane < G.n_lanes; ++lane) {
				const nd_i4 lr = H.lane[(size_t)W.lane0 + G.lane0 + lane];
				if ((lr.w & 0xFF) != 1) continue;
				const bool opt = lr.z >= 0;
				double* rc = lrec[lane];
				const double* v = vs[lane];
				if (opt && !pre) {
					const double a00 = v[0] + lambda, l00 = std::sqrt(a00), i00 = 1.0 / l00;
					const double l10 = v[1] * i00, l20 = v[2] * i00;
					const double a11 = v[3] + lambda - l10 * l10, l11 = std::sqrt(a11), i11 = 1.0 / l11;
					const double l21 = (v[4] - l20 * l10) * i11;
					const double a22 = v[5] + lambda - l20 * l20 - l21 * l21, l22 = std::sqrt(a22), i22 = 1.0 / l22;
					const double i10 = -i11 * l10 * i00, i21 = -i22 * l21 * i11, i20 = -(i21 * l10 + i22 * l20) * i00;
					const double R16[16] = {l00, l10, l11, l20, l21, l22, i00, i10, i11, i20, i21, i22,
					                        i00 * v[6], i10 * v[6] + i11 * v[7], i20 * v[6] + i21 * v[7] + i22 * v[8], 0.0};
					std::memcpy(rc, R16, sizeof R16);
				} else {
					const double R16[16] = {1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1.0};
					std::memcpy(rc, R16, sizeof R16);
				}
				if (pre && opt) vmax = std::fmax(vmax, std::fmax(v[0], std::fmax(v[3], v[5])));
				double* sc = &scr[12 * ((size_t)G.lane0 + lane)];
				for (int c = 0; c < 9; ++c) sc[c] = opt && !pre ? rc[6 + c] : 0.0;
			}
			for (int lane = 0; lane < G.n_lanes; ++lane) {
				const nd_i4 lr = H.lane[(size_t)W.lane0 + G.lane0 + lane];
				if ((lr.w & 0xFF) != 2) continue;
				const double* rc = lrec[(lr.w >> 8) & 0xFF];
				const double* Lm = rc, *Li = rc + 6, *cc = rc + 12;
				const bool fixed = rc[15] != 0.0;
				const double* v9 = vs[lane];
				const double V[9] = {v9[0], v9[1], v9[2], v9[1], v9[3], v9[4], v9[2], v9[4], v9[5]};
				double Z[9], Pm[9], v[3];
				for (int i = 0; i < 3; ++i) {
					Z[3 * i] = V[3 * i] * Li[0];
					Z[3 * i + 1] = V[3 * i] * Li[1] + V[3 * i + 1] * Li[2];
					Z[3 * i + 2] = V[3 * i] * Li[3] + V[3 * i + 1] * Li[4] + V[3 * i + 2] * Li[5];
					v[i] = v9[6 + i] - (Z[3 * i] * cc[0] + Z[3 * i + 1] * cc[1] + Z[3 * i + 2] * cc[2]);
				}
				const double Lf[9] = {Lm[0], 0, 0, Lm[1], Lm[2], 0, Lm[3], Lm[4], Lm[5]};
				for (int c = 0; c < 9; ++c) Pm[c] = fixed ? ((c == 0 || c == 4 || c == 8) ? 1.0 : 0.0) : Lf[c] - Z[c];
				const double* Tn = &tab[12 * (size_t)(lr.z & 0xFFFF)];
				const double sg = ((lr.z >> 24) & 1) ? -1.0 : 1.0;
				const double* f = Wn->lm_pos + 3 * lr.x;
				double* out = &rec[(size_t)SSK_ND_REC * (64 * wv + lane)];
				for (int i = 0; i < 3; ++i) {
					const double r0 = sg * Tn[4 * i], r1 = sg * Tn[4 * i + 1], r2 = sg * Tn[4 * i + 2];
					for (int j = 0; j < 3; ++j) {
						out[3 * i + j] = r0 * Z[j] + r1 * Z[3 + j] + r2 * Z[6 + j];
						out[12 + 3 * i + j] = r0 * Pm[j] + r1 * Pm[3 + j] + r2 * Pm[6 + j];
					}
					out[21 + i] = r0 * v[0] + r1 * v[1] + r2 * v[2];
					out[9 + i] = Tn[4 * i] * f[0] + Tn[4 * i + 1] * f[1] + Tn[4 * i + 2] * f[2] + Tn[4 * i + 3];
				}
				double* sc = &scr[12 * ((size_t)G.lane0 + lane)];
				for (int c = 0; c < 9; ++c) sc[c] = out[c];
				for (int c = 0; c < 3; ++c) sc[9 + c] = out[9 + c];
			}
		}
		const int NQ = SSK_ND_LANE_RUNS ? waves * 64 : waves * 16;      // slots per round
		const int n_slots = form == 1 ? T.n_rounds : T.n_rounds * NQ;
		for (int sl = 0; sl < n_slots; ++sl) {
			{
				int pid, n, first, i0 = 0, istep = 1;
				double* acc;
				if (form == 1) {
					const nd_i4 q = H.sqs[(size_t)H.swin[0].sqs0 + T.qs0 + sl];
					pid = q.x & 0xFFFF; n = q.y & 0xFFFF; first = q.z;
					i0 = (q.y >> 20) & 3; istep = 1 << ((q.y >> 16) & 3);      // a lane slot: every lanes-th item from the lane's rank on
					if (pre && !((q.x >> 30) & 1)) continue;     // (the lambda_0 pass: diagonal blocks only)
					acc = &part[36 * (size_t)q.w];
				} else {
					const nd_i2 q = H.qs[(size_t)W.qs0 + T.qs0 + sl];
					pid = q.x & 0xFFFF; n = (q.x >> 16) & 0x7FFF; first = q.y & 0x0FFFFFFF;
					if (pid == 0xFFFF) continue;                      // (a quad without a block)
					acc = &Hb[36 * (size_t)pid];
				}
				for (int i = i0; i < n; i += istep) {
					const int it = H.item[(size_t)W.item0 + first + i];
					const double* xr = &rec[(size_t)SSK_ND_REC * (it & 511)];
					const double* yc = &rec[(size_t)SSK_ND_REC * ((it >> 10) & 511)];
					const double* X = xr + 12 * ((it >> 9) & 1);
					const double* Y = yc + 12 * ((it >> 19) & 1);
					const double* qr = xr + 9, *qc = yc + 9;
					const double sg = ((it >> 20) & 1) ? (pre ? 0.0 : -1.0) : 1.0;
					double Dm[9], U[9];
					for (int a = 0; a < 3; ++a) {
						for (int b = 0; b < 3; ++b) Dm[3 * a + b] = sg * (X[3 * a] * Y[3 * b] + X[3 * a + 1] * Y[3 * b + 1] + X[3 * a + 2] * Y[3 * b + 2]);
						cross(qc, Dm + 3 * a, U + 3 * a);
						for (int b = 0; b < 3; ++b) { acc[6 * a + b] += Dm[3 * a + b]; acc[6 * a + 3 + b] += U[3 * a + b]; }
					}
					for (int b = 0; b < 3; ++b) {
						const double cd[3] = {Dm[b], Dm[3 + b], Dm[6 + b]}, cu[3] = {U[b], U[3 + b], U[6 + b]};
						double xd[3], xu[3];
						cross(qr, cd, xd); cross(qr, cu, xu);
						for (int a = 0; a < 3; ++a) { acc[6 * (3 + a) + b] += xd[a]; acc[6 * (3 + a) + 3 + b] += xu[a]; }
					}
				}
			}
		}
		if (!pre)
			for (int k = 0; k < T.n_gqs; ++k) {
				const nd_i2 q = H.gqs[(size_t)W.gqs0 + T.gqs0 + k];
				const int e = q.x & 0xFFFF, n = (q.x >> 16) & 0xFFFF;
				double* ga = form == 1 ? &gpart[6 * (size_t)H.gslot[(size_t)H.swin[0].gslot0 + T.gqs0 + k]] : &gacc[6 * (size_t)e];
				for (int i = 0; i < n; ++i) {
					const double* rc = &rec[(size_t)SSK_ND_REC * H.gitem[(size_t)W.gitem0 + q.y + i]];
					double x[3];
					cross(rc + 9, rc + 21, x);
					for (int c = 0; c < 3; ++c) { ga[c] += rc[21 + c]; ga[3 + c] += x[c]; }
				}
			}
	}
	if (form == 1) {
		// the reduction kernel: a block's partial slots are contiguous (tile order), an edge's gradient slots likewise
		const nd_split& S = H.swin[0];
		for (int pid = 0; pid < n_pairs; ++pid)
			for (int s2 = H.poff[(size_t)S.poff0 + pid]; s2 < H.poff[(size_t)S.poff0 + pid + 1]; ++s2)
				for (int c = 0; c < 36; ++c) Hb[36 * (size_t)pid + c] += part[36 * (size_t)s2 + c];
		for (int e = 0; e < W.E; ++e)
			for (int s2 = H.goff[(size_t)S.goff0 + e]; s2 < H.goff[(size_t)S.goff0 + e + 1]; ++s2)
				for (int c = 0; c < 6; ++c) gacc[6 * (size_t)e + c] += gpart[6 * (size_t)s2 + c];
	}
	const int np = n6;
	for (int i = 0; i < n6 * n6; ++i) Hred[i] = 0.0;
	for (int pid = 0; pid < n_pairs; ++pid) {
		int e2 = 0;
		while ((e2 + 1) * (e2 + 2) / 2 <= pid) ++e2;
		const int e = pid - e2 * (e2 + 1) / 2;
		for (int a = 0; a < 6; ++a)
			for (int b = 0; b < 6; ++b) {
				if (e == e2 && a > b) continue;
				const int row = 6 * e + a, col = 6 * e2 + b;
				double v = Hb[36 * (size_t)pid + 6 * a + b];
				if (row == col) v += lambda;
				Hred[row * np + col] = v; Hred[col * np + row] = v;
			}
	}
	for (int i = 0; i < n6; ++i) g[i] = gacc[i];
	if (vmax_out) *vmax_out = vmax;
	if (stats) { stats[0] = W.n_tiles; stats[1] = n_groups; stats[2] = (int)H.item.size(); stats[3] = form == 1 ? H.max_tile_blocks : H.max_items_per_lane; if (form == 1) stats[4] = (int)H.part_slots; }
	if (dp && df) {
		for (int gi = 0; gi < n_groups; ++gi) {
			const nd_group& G = H.group[(size_t)W.grp0 + gi];
			double t3[64][3];
			for (int lane = 0; lane < G.n_lanes; ++lane) {
				const nd_i4 lr = H.lane[(size_t)W.lane0 + G.lane0 + lane];
				if ((lr.w & 0xFF) != 2) continue;
				const double* sc = &scr[12 * ((size_t)G.lane0 + lane)];
				const double* d6 = dp + 6 * ((lr.z >> 16) & 0xFF);
				double x[3];
				cross(d6 + 3, sc + 9, x);
				const double a0 = d6[0] + x[0], a1 = d6[1] + x[1], a2 = d6[2] + x[2];
				for (int j = 0; j < 3; ++j) t3[lane][j] = sc[j] * a0 + sc[3 + j] * a1 + sc[6 + j] * a2;
			}
			for (int lane = 0; lane < G.n_lanes; ++lane) {
				const nd_i4 lr = H.lane[(size_t)W.lane0 + G.lane0 + lane];
				if ((lr.w & 0xFF) != 1 || lr.z < 0) continue;
				const double* sc = &scr[12 * ((size_t)G.lane0 + lane)];
				const int nn = (lr.w >> 16) & 0xFF;
				double rhs[3] = {sc[6], sc[7], sc[8]};
				for (int i = 1; i <= nn; ++i) for (int c = 0; c < 3; ++c) rhs[c] -= t3[lane + i][c];
				df[3 * lr.z] = sc[0] * rhs[0] + sc[1] * rhs[1] + sc[3] * rhs[2];
				df[3 * lr.z + 1] = sc[2] * rhs[1] + sc[4] * rhs[2];
				df[3 * lr.z + 2] = sc[5] * rhs[2];
			}
		}
	}
	return 0;
}

// FNV-1a hashes of every table the host builder emits for the window (tests: two builds of the library must agree; tools: refactors of the builder)
static unsigned long long nd_fnv(const void* p, size_t n, unsigned long long h = 1469598103934665603ull)
{
	const unsigned char* b = (const unsigned char*)p;
	for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}
int nd_emu_tables_hash(const ssk_lm_window* Wn, int waves, int split, unsigned long long* out /* 16 */)
{
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, split != 0)) return -1;
#define HV(i, v) out[i] = nd_fnv(H.v.data(), sizeof(H.v[0]) * H.v.size()) ^ (unsigned long long)H.v.size()
	HV(0, win); HV(1, tile); HV(2, group); HV(3, suf); HV(4, obs); HV(5, px); HV(6, lane); HV(7, item); HV(8, qs); HV(9, gitem); HV(10, gqs);
	HV(11, swin); HV(12, sqs); HV(13, poff); HV(14, goff); HV(15, gslot);
#undef HV
	return 0;
}

// the same with `n_threads` threads building the window's landmark / tile ranges side by side (nd_team): the tables must not depend on it
int nd_emu_tables_hash_team(const ssk_lm_window* Wn, int waves, int split, int n_threads, unsigned long long* out /* 16 */)
{
	nd_team team;
	team.n = n_threads;
	team.run = [n_threads](int n_parts, const std::function<void(int)>& fn) {
		std::atomic<int> next{0};
		auto body = [&] { for (int i = next++; i < n_parts; i = next++) fn(i); };
		std::vector<std::thread> th;
		for (int t = 1; t < n_threads; ++t) th.emplace_back(body);
		body();
		for (std::thread& t : th) t.join();
	};
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, split != 0, &team)) return -1;
#define HV(i, v) out[i] = nd_fnv(H.v.data(), sizeof(H.v[0]) * H.v.size()) ^ (unsigned long long)H.v.size()
	HV(0, win); HV(1, tile); HV(2, group); HV(3, suf); HV(4, obs); HV(5, px); HV(6, lane); HV(7, item); HV(8, qs); HV(9, gitem); HV(10, gqs);
	HV(11, swin); HV(12, sqs); HV(13, poff); HV(14, goff); HV(15, gslot);
#undef HV
	return 0;
}

// host builder alone, `reps` times: milliseconds per build (tools: profiling the symbolic stage without a GPU)
double nd_emu_build_ms(const ssk_lm_window* Wn, int waves, int split, int reps)
{
	const auto t0 = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; ++r) {
		nd_host H;
		nd_win hdr{};
		if (!nd_build_window(*Wn, waves, H, hdr, 20352, split != 0)) return -1.0;
	}
#ifdef ND_BUILD_PROF
	for (int i = 0; i < 6; ++i) std::fprintf(stderr, "section %d: %.1f us\n", i, g_nd_prof[i] / reps), g_nd_prof[i] = 0;
#endif
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
}

// shape statistics of a window's node tables (tools: LDS budgeting): out[0..] = n_edges, E, n_suf, n_groups, n_tiles, n_obs, lanes, nodes, roots,
// max nodes per group, max roots per group, max obs per group, max lanes per group, items, max blocks per tile, lds_doubles
int nd_emu_shape(const ssk_lm_window* Wn, int waves, int* out /* 16 */)
{
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, false)) return -1;
	const nd_win& W = H.win[0];
	int nodes = 0, roots = 0, mxn = 0, mxr = 0, mxo = 0, mxl = 0;
	for (int g = 0; g < W.n_groups; ++g) {
		const nd_group& G = H.group[(size_t)g];
		int gn = 0, gr = 0;
		for (int a = 0; a < G.n_lanes; ++a) { const int k = H.lane[(size_t)G.lane0 + a].w & 0xFF; gn += k == 2; gr += k == 1; }
		nodes += gn; roots += gr;
		mxn = std::max(mxn, gn); mxr = std::max(mxr, gr); mxo = std::max(mxo, G.n_obs); mxl = std::max(mxl, G.n_lanes);
	}
	int mxb = 0;
	for (int t = 0; t < W.n_tiles; ++t) {
		const nd_tile& T = H.tile[(size_t)t];
		int nb = 0;
		for (int k = 0; k < T.n_rounds * waves * 16; ++k) nb += (H.qs[(size_t)T.qs0 + k].x & 0xFFFF) != 0xFFFF;
		mxb = std::max(mxb, nb);
	}
	const int v[16] = {W.n_edges, W.E, W.n_suf, W.n_groups, W.n_tiles, W.n_obs, (int)H.lane.size(), nodes, roots, mxn, mxr, mxo, mxl, (int)H.item.size(), mxb, (int)H.lds_doubles};
	std::memcpy(out, v, sizeof v);
	return 0;
}

// per tile and wavefront of the pair phase: item turns (the longest lane run of the wavefront's 16 quads) and items; out[0] = number of (tile, wave) entries,
// then pairs (turns, items)
int nd_emu_pair_load(const ssk_lm_window* Wn, int waves, int* out, int cap)
{
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, false)) return -1;
	const nd_win& W = H.win[0];
	int k = 0;
	for (int t = 0; t < W.n_tiles; ++t) {
		const nd_tile& T = H.tile[(size_t)t];
		for (int r = 0; r < T.n_rounds; ++r)
			for (int wv = 0; wv < waves; ++wv) {
				int turns = 0, items = 0;
				for (int q = 0; q < 16; ++q) { const int n = (H.qs[(size_t)T.qs0 + r * waves * 16 + wv * 16 + q].x >> 16) & 0x7FFF; turns = std::max(turns, (n + 3) / 4); items += n; }
				if (2 * k + 2 < cap) { out[1 + 2 * k] = turns; out[2 + 2 * k] = items; }
				++k;
			}
	}
	out[0] = k;
	return 0;
}

// item counts of every (tile, block) pair of a window (quad-slot tables; a block cut in two: its parts joined): out[0] = tiles, then per tile: count, counts...
int nd_emu_block_items(const ssk_lm_window* Wn, int waves, int* out, int cap)
{
	nd_host H;
	nd_win hdr{};
	if (!nd_build_window(*Wn, waves, H, hdr, 20352, false)) return -1;
	const nd_win& W = H.win[0];
	int k = 1;
	out[0] = W.n_tiles;
	for (int t = 0; t < W.n_tiles; ++t) {
		const nd_tile& T = H.tile[(size_t)t];
		std::vector<int> cnt((size_t)W.E * (W.E + 1) / 2, 0);
		for (int q = 0; q < T.n_rounds * waves * 16; ++q) { const nd_i2 s = H.qs[(size_t)T.qs0 + q]; if ((s.x & 0xFFFF) != 0xFFFF) cnt[(size_t)(s.x & 0xFFFF)] += (s.x >> 16) & 0x7FFF; }
		int nb = 0;
		for (int c : cnt) nb += c > 0;
		if (k + 1 + nb >= cap) return -2;
		out[k++] = nb;
		for (int c : cnt) if (c > 0) out[k++] = c;
	}
	return k;
}

int nd_emu_linearise(const ssk_lm_window* Wn, const ssk_lm_params* P, double lambda, int pre, int waves, double* Hred, double* g, const double* dp, double* df,
                     double* vmax_out, int* stats /* tiles, groups, items, max items per lane */)
{
	return nd_emu_linearise_form(Wn, P, lambda, pre, waves, Hred, g, dp, df, vmax_out, stats, 0);
}

}  // extern "C"
