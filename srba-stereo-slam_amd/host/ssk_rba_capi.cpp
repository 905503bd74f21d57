// ssk_rba_capi.cpp -- the C-ABI of include/ssk_rba.h over the facade's mySRBA (exceptions become status codes).
#include <algorithm>
#include <cstring>
#include <memory>
#include <functional>
#include <exception>
#include <vector>
#include "../../include/ssk_rba.h"
#include "srba_stereo_slam.h"

struct ssk_rba {
	std::unique_ptr<ssk::Context> ctx;       // non-owning view of the caller's ssk_ctx
	std::unique_ptr<ssk::mySRBA> rba;
	std::string err;
};

namespace {
template <class F>
int guarded(ssk_rba* r, F f)
{
	try { f(); return SSK_OK; }
	catch (const std::exception& e) { r->err = e.what(); return r->err.find("not positive definite") != std::string::npos ? SSK_ERR_NOT_POSDEF : SSK_ERR_HIP; }
}
ssk::Pose3D to_pose(const double* m) { ssk::Pose3D p; std::memcpy(p.m, m, sizeof p.m); return p; }
}  // namespace

extern "C" {

int ssk_rba_create(ssk_ctx* ctx, const ssk_rba_options* o, ssk_rba** out)
{
	if (!ctx || !o || !out) return SSK_ERR_BAD_ARG;
	if (o->max_tree_depth < 1 || o->max_optimize_depth < 1 || o->submap_size < 2 || !(o->kernel_param > 0)) return SSK_ERR_BAD_ARG;
	ssk_rba* r = new ssk_rba();
	const int rc = guarded(r, [&] {
		r->ctx.reset(new ssk::Context(ctx));
		ssk::TSRBAStereoSLAMOptions opts;
		ssk::TStereoCamera& c = opts.stereo_camera;
		c.fx_l = o->fx_l; c.fy_l = o->fy_l; c.cx_l = o->cx_l; c.cy_l = o->cy_l;
		c.fx_r = o->fx_r; c.fy_r = o->fy_r; c.cx_r = o->cx_r; c.cy_r = o->cy_r; c.baseline = o->baseline;
		opts.camera_pose_on_robot_rvt = to_pose(o->sensor_pose);
		opts.srba_max_tree_depth = (size_t)o->max_tree_depth; opts.srba_max_optimize_depth = (size_t)o->max_optimize_depth;
		opts.srba_submap_size = (size_t)o->submap_size;
		opts.srba_use_robust_kernel = o->use_robust_kernel != 0; opts.srba_use_robust_kernel_stage1 = o->use_robust_kernel_stage1 != 0;
		opts.srba_kernel_param = o->kernel_param;
		opts.srba_literal_lc_edge_init = (o->reserved & 1) != 0;    // flags bit 0: SRBA.h:294 as written
		r->rba.reset(new ssk::mySRBA(*r->ctx, opts));
		if (o->min_obs_to_loop_closure > 0) r->rba->parameters.min_obs_to_loop_closure = (size_t)o->min_obs_to_loop_closure;
	});
	if (rc != SSK_OK) { delete r; return rc; }
	*out = r;
	return SSK_OK;
}

void ssk_rba_destroy(ssk_rba* r) { delete r; }

const char* ssk_rba_last_error(const ssk_rba* r) { return r ? r->err.c_str() : "NULL ssk_rba"; }

int ssk_rba_set_initial_kf_pose(ssk_rba* r, const double pose[12])
{
	if (!r || !pose) return SSK_ERR_BAD_ARG;
	r->rba->setInitialKFPose(to_pose(pose));
	return SSK_OK;
}

int ssk_rba_define_kf(ssk_rba* r, const ssk_rba_obs* obs, int n_obs, int run_local_optimization, ssk_rba_kf_info* out)
{
	if (!r || n_obs < 0 || (n_obs > 0 && !obs) || !out) return SSK_ERR_BAD_ARG;
	return guarded(r, [&] {
		ssk::new_kf_observations_t list((size_t)n_obs);
		for (int i = 0; i < n_obs; ++i) {
			ssk::new_kf_observation_t& o = list[(size_t)i];
			o.is_fixed = false; o.is_unknown_with_init_val = true;
			o.obs.feat_id = obs[i].feat_id;
			o.obs.obs_data.left_px.x = obs[i].left_x; o.obs.obs_data.left_px.y = obs[i].left_y;
			o.obs.obs_data.right_px.x = obs[i].right_x; o.obs.obs_data.right_px.y = obs[i].right_y;
			o.setRelPos(obs[i].rel_pos);
		}
		ssk::TNewKeyFrameInfo info;
		r->rba->define_new_keyframe(list, info, run_local_optimization != 0);
		std::memset(out, 0, sizeof *out);
		out->kf_id = (int32_t)info.kf_id;
		out->n_created_edges = (int32_t)std::min<size_t>(info.created_edge_ids.size(), SSK_RBA_MAX_NEW_EDGES);
		for (int i = 0; i < out->n_created_edges; ++i) out->created_edge_ids[i] = (int32_t)info.created_edge_ids[(size_t)i];
		out->total_sqr_error_init = info.optimize_results.total_sqr_error_init;
		out->total_sqr_error_final = info.optimize_results.total_sqr_error_final;
		out->stg1_obs_rmse = info.optimize_results_stg1.obs_rmse;
		out->num_observations = (int32_t)info.optimize_results.num_observations;
		out->num_iters = (int32_t)info.optimize_results.num_iters;
	});
}

int ssk_rba_define_kf_batch(ssk_rba* const* r, int n, const ssk_rba_obs* const* obs, const int32_t* n_obs, int run_local_optimization,
                            ssk_rba_kf_info* out, ssk_rba_batch_timing* timing)
{
	if (!r || n < 1 || !obs || !n_obs || !out) return SSK_ERR_BAD_ARG;
	for (int i = 0; i < n; ++i) if (!r[i] || n_obs[i] < 0 || (n_obs[i] > 0 && !obs[i])) return SSK_ERR_BAD_ARG;
	return guarded(r[0], [&] {
		std::vector<ssk::mySRBA*> eng((size_t)n);
		for (int i = 0; i < n; ++i) eng[(size_t)i] = r[i]->rba.get();
		// (the records of engine i become its observation list on the host thread that does the engine's bookkeeping)
		const std::function<void(int, ssk::new_kf_observations_t&)> fill = [&](int i, ssk::new_kf_observations_t& list) {
			list.resize((size_t)n_obs[i]);
			for (int k = 0; k < n_obs[i]; ++k) {
				ssk::new_kf_observation_t& o = list[(size_t)k];
				const ssk_rba_obs& s = obs[i][k];
				o.is_fixed = false; o.is_unknown_with_init_val = true;
				o.obs.feat_id = s.feat_id;
				o.obs.obs_data.left_px.x = s.left_x; o.obs.obs_data.left_px.y = s.left_y;
				o.obs.obs_data.right_px.x = s.right_x; o.obs.obs_data.right_px.y = s.right_y;
				o.setRelPos(s.rel_pos);
			}
		};
		std::vector<ssk::TNewKeyFrameInfo> info;
		ssk::mySRBA::TBatchTiming T;
		// (a failure of some engines: the batch goes on for the others, the timing and every healthy engine's record are filled in, then the error
		// -- with the failed engines' indices in its text -- is reported on r[0])
		std::exception_ptr err;
		try { ssk::mySRBA::define_new_keyframe_batch(eng, fill, info, run_local_optimization != 0, &T); }
		catch (...) { err = std::current_exception(); if ((int)info.size() != n) std::rethrow_exception(err); }
		for (int i = 0; i < n; ++i) {
			ssk_rba_kf_info& o = out[i];
			const ssk::TNewKeyFrameInfo& f = info[(size_t)i];
			std::memset(&o, 0, sizeof o);
			o.kf_id = (int32_t)f.kf_id;
			o.n_created_edges = (int32_t)std::min<size_t>(f.created_edge_ids.size(), SSK_RBA_MAX_NEW_EDGES);
			for (int k = 0; k < o.n_created_edges; ++k) o.created_edge_ids[k] = (int32_t)f.created_edge_ids[(size_t)k];
			o.total_sqr_error_init = f.optimize_results.total_sqr_error_init; o.total_sqr_error_final = f.optimize_results.total_sqr_error_final;
			o.stg1_obs_rmse = f.optimize_results_stg1.obs_rmse;
			o.num_observations = (int32_t)f.optimize_results.num_observations; o.num_iters = (int32_t)f.optimize_results.num_iters;
		}
		if (timing) {
			std::memset(timing, 0, sizeof *timing);
			timing->bookkeeping_and_windows_ms = T.bookkeeping_and_windows_ms; timing->stage1_create_ms = T.stage1_create_ms; timing->stage1_solve_ms = T.stage1_solve_ms;
			timing->stage2_refresh_ms = T.stage2_refresh_ms; timing->stage2_create_ms = T.stage2_create_ms; timing->stage2_solve_ms = T.stage2_solve_ms; timing->writeback_ms = T.writeback_ms;
			timing->stage1_symbolic_ms = T.stage1_symbolic_ms; timing->stage2_symbolic_ms = T.stage2_symbolic_ms;
			timing->stage1_mean_obs = T.stage1_mean_obs; timing->stage2_mean_obs = T.stage2_mean_obs; timing->stage2_mean_edges = T.stage2_mean_edges;
			timing->stage1_mean_iters = T.stage1_mean_iters; timing->stage2_mean_iters = T.stage2_mean_iters;
			timing->host_threads = T.host_threads; timing->n_stage1 = T.n_stage1; timing->n_stage2 = T.n_stage2;
			timing->stage2_max_edges = T.stage2_max_edges; timing->stage2_max_obs = T.stage2_max_obs;
			std::memcpy(timing->stage1_kernel, T.stage1_kernel, sizeof timing->stage1_kernel); std::memcpy(timing->stage2_kernel, T.stage2_kernel, sizeof timing->stage2_kernel);
			timing->teardown_ms = T.teardown_ms; timing->total_ms = T.total_ms;
		}
		if (err) std::rethrow_exception(err);
	});
}

int ssk_rba_obs_from_da_batch(int n, int cap, const int32_t* n_matches, const int32_t* status, const int32_t* train_idx, const float* obs_px,
                              const double* init_pos, const int64_t* ids_prev, int64_t* next_id, int64_t* ids_out, ssk_rba_obs* obs_out)
{
	if (n < 1 || cap < 1 || !n_matches || !obs_px || !init_pos || !next_id || !ids_out || !obs_out || (ids_prev && (!status || !train_idx))) return SSK_ERR_BAD_ARG;
	for (int b = 0; b < n; ++b) {
		const int nm = std::min(std::max(n_matches[b], 0), cap);
		const size_t o = (size_t)b * (size_t)cap;
		int64_t nid = next_id[b];
		for (int i = 0; i < nm; ++i) {
			int64_t id = -1;
			if (ids_prev && status[o + i] == SSK_DA_TRACKED && train_idx[o + i] >= 0 && train_idx[o + i] < cap) id = ids_prev[o + (size_t)train_idx[o + i]];
			if (id < 0) id = ((int64_t)b << 40) + nid++;
			ids_out[o + i] = id;
			ssk_rba_obs& r = obs_out[o + i];
			r.feat_id = (uint64_t)id;
			r.left_x = obs_px[4 * (o + i)]; r.left_y = obs_px[4 * (o + i) + 1]; r.right_x = obs_px[4 * (o + i) + 2]; r.right_y = obs_px[4 * (o + i) + 3];
			r.rel_pos[0] = init_pos[3 * (o + i)]; r.rel_pos[1] = init_pos[3 * (o + i) + 1]; r.rel_pos[2] = init_pos[3 * (o + i) + 2];
		}
		for (int i = nm; i < cap; ++i) ids_out[o + i] = -1;
		next_id[b] = nid;
	}
	return SSK_OK;
}

int ssk_rba_num_keyframes(const ssk_rba* r) { return r ? (int)r->rba->num_keyframes() : 0; }
int ssk_rba_num_edges(const ssk_rba* r) { return r ? (int)r->rba->get_rba_state().k2k_edges.size() : 0; }
int ssk_rba_num_landmarks(const ssk_rba* r) { return r ? (int)r->rba->get_rba_state().unknown_lms.size() : 0; }

int ssk_rba_localmap_centers(const ssk_rba* r, int kf, int32_t* out, int cap, int32_t* n)
{
	if (!r || !n || kf < 0 || kf >= ssk_rba_num_keyframes(r)) return SSK_ERR_BAD_ARG;
	if (kf == 0 && ssk_rba_num_keyframes(r) == 1) { *n = 0; return SSK_OK; }      // (keyframe 0 gets its entry when keyframe 1 arrives, SRBA.h:153-158)
	// (a keyframe without an entry makes the map's at() throw: nothing may unwind across the C boundary)
	try {
		const std::set<ssk::TKeyFrameID>& c = r->rba->localmap_centers_of((size_t)kf);
		*n = (int32_t)c.size();
		if ((int)c.size() > cap) return SSK_ERR_CAPACITY;
		if (!c.empty() && !out) return SSK_ERR_BAD_ARG;
		int i = 0;
		for (ssk::TKeyFrameID v : c) out[i++] = (int32_t)v;
	} catch (const std::exception&) {
		*n = 0;
		return SSK_ERR_BAD_ARG;
	}
	return SSK_OK;
}

int ssk_rba_localmap_kfs(const ssk_rba* r, int center, int32_t* out, int cap, int32_t* n)
{
	if (!r || !n || center < 0) return SSK_ERR_BAD_ARG;
	const std::set<ssk::TKeyFrameID>* m = r->rba->localmap_kfs((size_t)center);
	if (!m) { *n = -1; return SSK_OK; }
	*n = (int32_t)m->size();
	if ((int)m->size() > cap) return SSK_ERR_CAPACITY;
	int i = 0;
	for (ssk::TKeyFrameID v : *m) out[i++] = (int32_t)v;
	return SSK_OK;
}

int ssk_rba_get_edge(const ssk_rba* r, int edge, int32_t* from, int32_t* to, double inv_pose[12])
{
	if (!r || edge < 0 || edge >= ssk_rba_num_edges(r)) return SSK_ERR_BAD_ARG;
	const ssk::k2k_edge_t& e = r->rba->get_rba_state().k2k_edges[(size_t)edge];
	if (from) *from = (int32_t)e.from;
	if (to) *to = (int32_t)e.to;
	if (inv_pose) std::memcpy(inv_pose, e.inv_pose.m, sizeof e.inv_pose.m);
	return SSK_OK;
}

int ssk_rba_get_landmark(const ssk_rba* r, uint64_t feat_id, int32_t* base_kf, double pos[3])
{
	if (!r) return SSK_ERR_BAD_ARG;
	const auto& lms = r->rba->get_rba_state().unknown_lms;
	const auto it = lms.find(feat_id);
	if (it == lms.end()) return SSK_ERR_BAD_ARG;
	if (base_kf) *base_kf = (int32_t)it->second.base_kf;
	if (pos) std::memcpy(pos, it->second.pos, 3 * sizeof(double));
	return SSK_OK;
}

int ssk_rba_spanning_tree_pose(const ssk_rba* r, int root, int target, int max_depth, double pose[12], int32_t* found)
{
	if (!r || !pose || !found || root < 0 || target < 0 || root >= ssk_rba_num_keyframes(r) || target >= ssk_rba_num_keyframes(r) || max_depth < 0)
		return SSK_ERR_BAD_ARG;
	ssk::Pose3D p;
	*found = r->rba->get_kf_relative_pose((size_t)root, (size_t)target, (size_t)max_depth, p) ? 1 : 0;
	std::memcpy(pose, p.m, sizeof p.m);
	return SSK_OK;
}

int ssk_rba_global_poses(const ssk_rba* r, int root, double* node_poses, int node_cap, int32_t* edge_ij, double* edge_z, int edge_cap,
                         int32_t* n_nodes, int32_t* n_edges)
{
	if (!r || !n_nodes || !n_edges || root < 0 || root >= std::max(1, ssk_rba_num_keyframes(r))) return SSK_ERR_BAD_ARG;
	std::vector<ssk::Pose3D> nodes, z;
	std::vector<int32_t> ij;
	std::vector<bool> reach;
	r->rba->get_global_graphslam_problem((size_t)root, nodes, ij, z, reach);
	*n_nodes = (int32_t)nodes.size(); *n_edges = (int32_t)z.size();
	if ((int)nodes.size() > node_cap || (int)z.size() > edge_cap) return SSK_ERR_CAPACITY;
	for (size_t k = 0; k < nodes.size(); ++k) std::memcpy(node_poses + 12 * k, nodes[k].m, 12 * sizeof(double));
	for (size_t k = 0; k < z.size(); ++k) { std::memcpy(edge_z + 12 * k, z[k].m, 12 * sizeof(double)); edge_ij[2 * k] = ij[2 * k]; edge_ij[2 * k + 1] = ij[2 * k + 1]; }
	return SSK_OK;
}

}  // extern "C"
