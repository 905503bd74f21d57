// ssk_lm_split.h -- the PHASE-SPLIT node form of the relative-BA Levenberg-Marquardt solve (kernels: ssk_lm_split.hip,
// ssk_lm_split_solve.hip; host tables: ssk_lm_node_build.h with split = true).  Reference: rba.define_new_keyframe EST.cpp:782-784 with
// srba_max_optimize_depth = 5 (KITTI.ini:104-106: local areas of ~40 edges / 8 k observations), solver type SRBA.h:34.
//
// Why.  A window whose reduced system (6E x 6E doubles) does not fit LDS cannot keep its block accumulators on one CU: the persistent
// one-workgroup-per-window kernels of rounds 1-3 walked 39 tiles serially per LM pass at two waves per SIMD (91.6 ms per 1 024 depth-5
// windows, 279 spilled VGPRs).  Here one LM pass of the WHOLE BATCH is four launches, each with its own shape and register budget:
//   K1  lm_split_linearise_kernel   grid = all tiles of all windows (4 landmark groups = 4 wavefronts each, ~40 k workgroups): group phase
//       (observations -> run sums -> roots -> node records in LDS) and pair phase of ssk_lm_node.h; a quad of lanes per (tile, block) sums its
//       items in registers and stores ONE 36-double partial to its own slot in HBM (no atomics; slots are numbered block-major by the host),
//       reduced-gradient partials per (tile, edge) likewise.
//   K2a lm_split_reduce_kernel      grid = windows x chunks of 7 blocks: sums a block's partial slots in slot (= tile) order into the dense bordered
//       system -- the upper triangle and, for a matrix-core K2, the lower halves of the diagonal 16 x 16 tiles only (a kernel of its own: the sums are latency-bound loads that want every CU's full occupancy, not the solver's 8 wavefronts)
//   K2  lm_split_solve_kernel       grid = windows, 512 threads: LDL^T with the system in MFMA accumulator tiles (ldl_solve_mfma_tiles, ssk_lm_ldl_mfma.h; beyond 255 unknowns the leading columns first, ldl_leading_mfma: three instances of the kernel, sp_dev::solve_mfma), edge update exp(eps) (+) T, the
//       trial state's suffix table.  The lambda_0 pass ends after the diagonal blocks.
//   K3  lm_split_score_kernel       grid = tiles: landmark steps + robust error of the trial state group by group (nd_score_group), one error /
//       |delta_f|^2 partial per tile.
//   K4  lm_split_decide_kernel      grid = windows: partials summed in tile order, accept / reject, lambda schedule (contract L5 of
//       oracle/oracle_lm.c), trace, state copy, stop flags.  A stopped window's workgroups leave at once in every later launch.
// Every sum has one owner and a fixed order: results are bit-reproducible run to run and independent of the batch around a window.
#pragma once
#include "ssk_lm.h"
#include "ssk_lm_node.h"

struct sp_ctl {                         // LM loop state of one window (device)
	double lambda, nu, err, dpn;
	int iters, accepted, stop, cur;     // cur: which of the two suffix tables holds the current state's transforms
	int active;                         // 1 while the window's loop runs
	int ok;                             // the attempt's factorisation succeeded (K2 -> K3 / K4)
	int pre;                            // 1: the next linearisation is the lambda_0 pass (every window's first: K2 ends it; the host launches no K3 / K4 behind it)
	int pad;
};

// K1 / K3: one workgroup's tile with absolute indices, so that the tile's groups and slots load beside the window's headers instead of behind them
struct alignas(16) sp_tile {
	int wi, tile;                       // window (-1: none), tile of the window
	int grp, n_grp;                     // nd_dev::group[grp ..)
	int sqs, n_slots;                   // sp_dev::sqs[sqs ..): lane slots
	int gqs, n_gqs, gslot;              // nd_dev::gqs[gqs ..), sp_dev::gslot[gslot ..)
	int gtile;                          // the tile's number in the batch (vpart / epart / fpart)
	int item0, gitem0;                  // the window's bases into nd_dev::item / gitem
};

struct sp_dev {
	const nd_split* sw; const nd_i4* sqs; const int* poff; const int* goff; const int* gslot;
	double* Hpart; double* gpart;       // 36 / 6 doubles per partial slot
	double* vpart; double* epart; double* fpart;      // per tile of the batch: largest landmark-block diagonal (lambda_0 pass), robust error, |delta_f|^2
	sp_ctl* ctl;
	int* n_active;                      // windows whose loop still runs (K0 counts them in, K4 counts a stopped one out): the host polls it between groups of passes of a small batch
	const sp_tile* tile_map;            // K1 / K3: workgroup -> tile (wi -1: none): a window's tiles share an XCD (its L2 holds the window's suffix table)
	const int* win_map;                 // K2 / K4: workgroup -> window or -1, the same XCD as its tiles
	int n_tile_wgs, n_win_wgs;
	int max_pairs;                      // most 6x6 blocks of a window's reduced system (K2a's grid)
	int solve_mfma;                     // K2: 2 = ldl_solve_mfma_tiles for every window of the batch, 1 = ldl_solve_mfma_lead (a window beyond 255 unknowns), 0 = ldl_solve_regtile (the host's choice: ssk_sp_solve_mfma + the LDS budget)
};

constexpr int SSK_SP_WAVES = 4;         // wavefronts per K1 / K3 workgroup = landmark groups per tile
// K2: the solver of a batch whose largest window optimises E_max edges (ssk_lm_ldl_mfma.h: the system in MFMA accumulator tiles up to 255 unknowns,
// the leading columns of larger ones -- up to 768 -- by 16-column panels first; SSK_K2_SOLVER=regtile: the register-tile solver)
int ssk_sp_solve_mfma(int E_max, bool force_regtile /* SSK_K2_SOLVER=regtile */);      // 0 register tiles, 1 matrix cores incl. leading columns, 2 matrix cores, every window within 255 unknowns
// K2: dynamic LDS of the solve kernel (doubles): the solver's staging + the window's edge poses and trie records
size_t ssk_sp_solve_lds_doubles(int n_edges, int n_suf, int E, bool mfma);

int ssk_lm_launch_split(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const sp_dev& S, const lm_cam& P, int n_windows, size_t solve_dyn_bytes, size_t init_dyn_bytes);
// (ssk_lm_split_solve.hip)
int ssk_lm_launch_split_solve(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const sp_dev& S, const lm_cam& P, size_t dyn_bytes);
