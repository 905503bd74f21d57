// ssk_lm.h -- device data model of a batch of relative-BA windows (see ssk_lm.hip for the kernel).
#pragma once
#include "ssk_internal.h"

constexpr int SSK_LM_MAX_PATH = 8;     // spanning-tree path length limit (reference max_tree_depth = 4, KITTI.ini:104)
constexpr int SSK_LM_MAX_N6 = 768;     // reduced system limit: 6E <= 768 (E <= 128 optimised edges per window)

// Per-window header.  Every "…0" member is the window's base offset into a batch-wide array family; values
// stored inside the families (offset tables, indices) are window relative.
struct lm_win {
	int n_edges, E, n_lms, L, n_obs, n_path, n_pairs;
	int edge0;     // edges:        edge_slot[], pose*[] x12
	int lm0;       // landmarks:    lm_slot[], lm*[] x3
	int obs0;      // observations: obs_lm[], obs_px x4, obs_r x4, obs_Jf x12
	int opo0;      // obs_path_off[] (n_obs+1 entries per window)
	int path0;     // path entries: path_edge[], path_obs[], path_lme[], path_q x3, path_Je x24
	int us0;       // optimised-edge slots: slot_edge[], gp x6, dp x6
	int uso0;      // per optimised edge offset tables (E+1): el_off[], pe_off[]
	int ls0;       // optimised-landmark slots: slot_lm[], Hff x9, Vinv x9, gf x3, df x3
	int lso0;      // per optimised landmark offset tables (L+1): lmo_off[], lme_off[]
	int lmo0;      // lmo_idx[]  (observations of a landmark)
	int lme0;      // (landmark, touched optimised edge) entries: lme_edge[], lme_lm[], Wb x18, Tb x18
	int pair0;     // block pairs e<=e': pair_ee[]
	int pairo0;    // per pair offset tables (n_pairs+1): ppo_off[], ppl_off[]
	int ppo0;      // ppo[] (path entry of e, path entry of e')
	int ppl0;      // ppl[] (lme entry of e, lme entry of e')
	int el0;       // el_idx[] lme entries per optimised edge
	int pe0;       // pe_idx[] path entries per optimised edge
	int trace0;    // trace entries (window * max_iters)
	long long h0;  // Hpp / Hred (6E)^2 doubles
	long long dp_trace0, df_trace0;
};

struct lm_dev {
	const lm_win* win;
	const int* edge_slot; double* pose; const double* pose0; double* pose_new;
	const int* lm_slot; double* lm; const double* lm0; double* lm_new;
	const int* obs_lm; const float* obs_px; const int* obs_path_off;
	const int* path_edge; const int* path_obs; const int* path_lme;
	const int* slot_edge; const int* slot_lm;
	const int* lmo_off; const int* lmo_idx;
	const int* lme_off; const int* lme_edge; const int* lme_lm;
	const int2* pair_ee; const int* ppo_off; const int2* ppo; const int* ppl_off; const int2* ppl;
	const int* el_off; const int* el_idx; const int* pe_off; const int* pe_idx;
	double* obs_r; double* obs_Jf; double* path_q; double* path_Je;
	double* Hff; double* Vinv; double* gf; double* df; double* Wb; double* Tb;
	double* Hpp; double* Hred; double* gp; double* dp;
	ssk_lm_result* res; ssk_lm_trace_entry* trace; double* dp_trace; double* df_trace;
	int trace_on;
};

struct lm_cam {
	double fx_l, fy_l, cx_l, cy_l, fx_r, fy_r, cx_r, cy_r, baseline;
	double S[12];
	int max_iters, robust;
	double kernel, tau, max_lambda, stop_rmse, conv_rel;
};

int ssk_lm_launch(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, int max_n6);
int ssk_lm_launch_reset(ssk_ctx* ctx, double* pose, const double* pose0, size_t n_pose, double* lm, const double* lm0, size_t n_lm);
