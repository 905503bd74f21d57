// ssk_slam_capi.cpp -- the C-ABI of include/ssk_slam.h over the facade's CSRBAStereoSLAMEstimator.
#include <algorithm>
#include <cstring>
#include <memory>
#include "../../include/ssk_slam.h"
#include "CSRBAStereoSLAMEstimator.h"

struct ssk_slam {
	std::unique_ptr<ssk::Context> ctx;
	ssk::TSRBAStereoSLAMOptions opts;
	std::unique_ptr<ssk::CSRBAStereoSLAMEstimator> est;     // built at the first use: the options are copied at construction
	std::string voc, err;
	ssk::TGeneralOptions gopts;
	ssk::TVectorKfsDaInfo last_da;
	std::vector<size_t> last_da_sorted;
	std::vector<ssk::Pose3D> last_da_init;
	int w = 0, h = 0;
	ssk::CSRBAStereoSLAMEstimator& get()
	{
		if (!est) {
			est.reset(new ssk::CSRBAStereoSLAMEstimator(*ctx, opts, gopts));
			if (!voc.empty()) est->m_bow_manager.loadVocabulary(voc);
		}
		return *est;
	}
};

namespace {
template <class F>
int guarded(ssk_slam* s, F f)
{
	try { f(); return SSK_OK; }
	catch (const std::exception& e) {
		s->err = e.what();
		return s->err.find("less than the minimum") != std::string::npos ? SSK_ERR_TOO_FEW_MATCHES : SSK_ERR_HIP;
	}
}
}  // namespace

extern "C" {

int ssk_slam_create(ssk_ctx* ctx, const char* preset, int width, int height, ssk_slam** out)
{
	if (!ctx || !out || width < 64 || height < 64) return SSK_ERR_BAD_ARG;
	const std::string p = preset ? preset : "kitti";
	if (p != "kitti" && p != "default") return SSK_ERR_BAD_ARG;
	ssk_slam* s = new ssk_slam();
	const int rc = guarded(s, [&] {
		s->ctx.reset(new ssk::Context(ctx));
		s->opts = p == "kitti" ? ssk::TSRBAStereoSLAMOptions::kitti() : ssk::TSRBAStereoSLAMOptions();
		s->opts.stereo_camera.ncols = width; s->opts.stereo_camera.nrows = height;
		s->w = width; s->h = height;
	});
	if (rc != SSK_OK) { delete s; return rc; }
	*out = s;
	return SSK_OK;
}

void ssk_slam_destroy(ssk_slam* s) { delete s; }

const char* ssk_slam_last_error(const ssk_slam* s) { return s ? s->err.c_str() : "NULL ssk_slam"; }

int ssk_slam_set_option(ssk_slam* s, const char* key, double v)
{
	if (!s || !key) return SSK_ERR_BAD_ARG;
	if (s->est) { s->err = "ssk_slam_set_option: options are fixed once the first frame has been processed"; return SSK_ERR_BAD_ARG; }
	const std::string k = key;
	ssk::TSRBAStereoSLAMOptions& o = s->opts;
	if (k == "submap_size") o.srba_submap_size = (size_t)v;
	else if (k == "max_tree_depth") o.srba_max_tree_depth = (size_t)v;
	else if (k == "max_optimize_depth") o.srba_max_optimize_depth = (size_t)v;
	else if (k == "lc_distance") o.lc_distance = (size_t)v;
	else if (k == "max_translation") o.max_translation = v;
	else if (k == "max_rotation") o.max_rotation = v;
	else if (k == "updated_matches_th") o.updated_matches_th = (size_t)v;
	else if (k == "up_matches_th_plus") o.up_matches_th_plus = (size_t)v;
	else if (k == "vo_id_tracking_th") o.vo_id_tracking_th = (int)v;
	else if (k == "n_feats") o.n_feats = (size_t)v;
	else if (k == "detect_fast_th") o.detect_fast_th = (int)v;
	else if (k == "residual_th") o.residual_th = v;
	else if (k == "use_initial_pose") o.use_initial_pose = v != 0;
	else if (k == "literal_lc_edge_init") o.srba_literal_lc_edge_init = v != 0;
	else { s->err = "ssk_slam_set_option: unknown key " + k; return SSK_ERR_BAD_ARG; }
	return SSK_OK;
}

int ssk_slam_set_debug_dir(ssk_slam* s, const char* dir)
{
	if (!s || !dir) return SSK_ERR_BAD_ARG;
	if (s->est) { s->err = "ssk_slam_set_debug_dir: options are fixed once the first frame has been processed"; return SSK_ERR_BAD_ARG; }
	s->gopts.debug = true; s->gopts.out_dir = dir;
	return SSK_OK;
}

int ssk_slam_get_keyframe(const ssk_slam* s, int kf, ssk_kf_out* out, uint64_t* match_ids)
{
	if (!s || !out || !s->est || kf < 0 || (size_t)kf >= s->est->m_keyframes.size()) return SSK_ERR_BAD_ARG;
	const ssk::CStereoSLAMKF& k = s->est->m_keyframes[(size_t)kf];
	const size_t nl = k.m_keypoints_left.size(), nr = k.m_keypoints_right.size(), nm = k.m_matches.size();
	if (nl > (size_t)out->capacity || nr > (size_t)out->capacity || nm > (size_t)out->capacity) return SSK_ERR_CAPACITY;
	if (nl) { std::memcpy(out->kp_left, k.m_keypoints_left.data(), nl * sizeof(ssk_keypoint)); std::memcpy(out->desc_left, k.m_descriptors_left.data.data(), nl * 32); }
	if (nr) { std::memcpy(out->kp_right, k.m_keypoints_right.data(), nr * sizeof(ssk_keypoint)); std::memcpy(out->desc_right, k.m_descriptors_right.data.data(), nr * 32); }
	if (nm) std::memcpy(out->matches, k.m_matches.data(), nm * sizeof(ssk_dmatch));
	out->n_left = (int32_t)nl; out->n_right = (int32_t)nr; out->n_matches = (int32_t)nm;
	if (match_ids) for (size_t m = 0; m < nm; ++m) match_ids[m] = m < k.m_matches_ID.size() ? (uint64_t)k.m_matches_ID[m] : 0;
	return SSK_OK;
}

int ssk_slam_last_da(const ssk_slam* s, int which, int32_t* other_kf, int32_t* tracked, int32_t* sorted_pos,
                     int32_t* tracking_first, double* tracking_dist, int cap, int32_t* n_matches, double* init_pose)
{
	if (!s || which < 0 || (size_t)which >= s->last_da.size() || !n_matches) return SSK_ERR_BAD_ARG;
	const ssk::t_kf_da_info& d = s->last_da[(size_t)which];
	if ((size_t)cap < d.tracking_info.size() && (tracking_first || tracking_dist)) return SSK_ERR_CAPACITY;
	if (other_kf) *other_kf = (int32_t)d.kf_idx;
	if (tracked) *tracked = (int32_t)d.tracked_matches;
	if (sorted_pos) {
		*sorted_pos = -1;
		for (size_t i = 0; i < s->last_da_sorted.size(); ++i) if (s->last_da_sorted[i] == (size_t)which) *sorted_pos = (int32_t)i;
	}
	*n_matches = (int32_t)d.tracking_info.size();
	if (init_pose) {
		const ssk::Pose3D identity;
		std::memcpy(init_pose, ((size_t)which < s->last_da_init.size() ? s->last_da_init[(size_t)which] : identity).m, 12 * sizeof(double));
	}
	for (size_t m = 0; m < d.tracking_info.size(); ++m) {
		if (tracking_first) tracking_first[m] = d.tracking_info[m].first == ssk::INVALID_IDX ? -1 : (int32_t)d.tracking_info[m].first;
		if (tracking_dist) tracking_dist[m] = d.tracking_info[m].second;
	}
	return SSK_OK;
}

int ssk_slam_load_vocabulary(ssk_slam* s, const char* path)
{
	if (!s || !path) return SSK_ERR_BAD_ARG;
	return guarded(s, [&] {
		s->voc = path;
		if (s->est) s->est->m_bow_manager.loadVocabulary(s->voc);
		else (void)ssk::loadVocabFile(s->voc);               // fail now, not at the first frame
	});
}

int ssk_slam_process_frame(ssk_slam* s, const uint8_t* left, const uint8_t* right, int stride, ssk_slam_frame_report* out)
{
	if (!s || !left || !right || !out || stride < s->w) return SSK_ERR_BAD_ARG;
	return guarded(s, [&] {
		ssk::StereoImages im;
		im.left = left; im.right = right; im.width = s->w; im.height = s->h; im.stride = stride;
		const ssk::TFrameReport r = s->get().performStereoSLAM(im);
		std::memset(out, 0, sizeof *out);
		out->frame = (int32_t)r.frame;
		out->vo_valid = r.odometry.valid; out->tracked_from_last_frame = (int32_t)r.odometry.tracked_feats_from_last_frame;
		out->tracked_from_last_kf = (int32_t)r.odometry.tracked_feats_from_last_KF;
		std::memcpy(out->vo_pose, r.odometry.outPose.m, sizeof out->vo_pose);
		out->checked_for_kf = r.checked_for_kf; out->inserted_kf = r.inserted_kf; out->vo_force_new_kf = r.vo_force_new_kf;
		out->vo_force_check_tracking = r.vo_force_check_tracking; out->vo_force_check_distance = r.vo_force_check_distance;
		out->potential_loop_closure = r.potential_loop_closure; out->confirmed_loop_closure = r.confirmed_loop_closure;
		out->lc_kf_id = r.similar.lc_id == ssk::policy::INVALID_KF_ID ? -1 : (int32_t)r.similar.lc_id;
		out->n_query = (int32_t)std::min<size_t>(r.query.size(), 4);
		for (int i = 0; i < out->n_query; ++i) { out->query_ids[i] = (int32_t)r.query[(size_t)i].Id; out->query_scores[i] = r.query[(size_t)i].Score; }
		out->n_similar = (int32_t)std::min<size_t>(r.similar.similar_kfs.size(), 8);
		for (int i = 0; i < out->n_similar; ++i) out->similar_kfs[i] = (int32_t)r.similar.similar_kfs[(size_t)i];
		out->highest_tracked = (int32_t)r.highest_tracked;
		out->kf_id = (int32_t)r.kf_info.kf_id; out->n_created_edges = (int32_t)r.kf_info.created_edge_ids.size();
		out->total_sqr_error_init = r.kf_info.optimize_results.total_sqr_error_init;
		out->total_sqr_error_final = r.kf_info.optimize_results.total_sqr_error_final;
		out->lm_stage2_edges = (int32_t)r.kf_info.optimize_results.num_opt_edges; out->lm_stage2_obs = (int32_t)r.kf_info.optimize_results.num_observations;
		out->n_keyframes = (int32_t)s->get().m_keyframes.size();
		if (r.checked_for_kf) { s->last_da = r.da; s->last_da_sorted = r.da_sorted; s->last_da_init = r.similar.similar_kfs_poses; }
	});
}

int ssk_slam_get_timing(const ssk_slam* s, ssk_slam_timing* out)
{
	if (!s || !out) return SSK_ERR_BAD_ARG;
	std::memset(out, 0, sizeof *out);
	if (!s->est) return SSK_OK;
	const auto& t = s->est->timing;
	const auto& l = s->est->rba.timing;
	out->vo_ms = t.vo_ms; out->query_ms = t.query_ms; out->da_ms = t.da_ms; out->insert_ms = t.insert_ms;
	out->lm_window_build_ms = l.window_build_ms; out->lm_batch_create_ms = l.batch_create_ms; out->lm_solve_ms = l.solve_ms; out->lm_readback_ms = l.readback_ms;
	out->frames = (int32_t)t.frames; out->kf_frames = (int32_t)t.kf_frames; out->lm_solves = (int32_t)l.solves; out->lm_iters = (int32_t)l.iters;
	out->lm_max_opt_edges = (int32_t)l.max_opt_edges; out->lm_max_obs = (int32_t)l.max_obs;
	return SSK_OK;
}

int ssk_slam_num_keyframes(const ssk_slam* s) { return s && s->est ? (int)s->est->m_keyframes.size() : 0; }
int ssk_slam_num_landmarks(const ssk_slam* s) { return s && s->est ? (int)s->est->rba.get_rba_state().unknown_lms.size() : 0; }
int ssk_slam_num_edges(const ssk_slam* s) { return s && s->est ? (int)s->est->rba.get_rba_state().k2k_edges.size() : 0; }

int ssk_slam_optimize_pose_graph(ssk_slam* s, double* poses, int cap, int32_t* n_poses, ssk_pg_result* info, const char* path)
{
	if (!s || !n_poses || cap < 0 || (cap > 0 && !poses)) return SSK_ERR_BAD_ARG;
	return guarded(s, [&] {
		const std::vector<ssk::Pose3D> v = s->get().optimizeGlobalPoseGraph(info, path ? std::string(path) : std::string());
		*n_poses = (int32_t)v.size();
		for (size_t k = 0; k < v.size() && (int)k < cap; ++k) std::memcpy(poses + 12 * k, v[k].m, 12 * sizeof(double));
	});
}

}  // extern "C"
