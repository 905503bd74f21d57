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
	int w = 0, h = 0;
	ssk::CSRBAStereoSLAMEstimator& get()
	{
		if (!est) {
			est.reset(new ssk::CSRBAStereoSLAMEstimator(*ctx, opts));
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
	else { s->err = "ssk_slam_set_option: unknown key " + k; return SSK_ERR_BAD_ARG; }
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
		out->n_keyframes = (int32_t)s->get().m_keyframes.size();
	});
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
