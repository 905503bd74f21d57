// facade_demo.cpp -- drives the C++ facade the way the reference's frame loop drives its estimator for the
// frames it has chosen as keyframes (EST.cpp:44-937 reduced to the hot path): read raw 8-bit stereo pairs,
// processKeyframe() each, print one line per keyframe.  Used by tests/test_facade_gpu.py (configs[0] plumbing
// on synthetic KITTI-shaped pairs) and as the example for INTEGRATION.md.
//   usage: facade_demo <dir> <n_pairs> <width> <height> [debug_out_dir]
//          facade_demo stream <voc.yml[.gz]> <dir> <n_pairs> <width> <height> [key=value ...]
//            the whole frame loop (performStereoSLAM): odometry, keyframe check, bag-of-words query, loop closure; one
//            line per frame.  keys: submap_size max_tree_depth max_optimize_depth lc_distance max_translation max_rotation
//            updated_matches_th up_matches_th_plus vo_id_tracking_th
//   files: <dir>/left_%04d.raw, <dir>/right_%04d.raw   (width*height bytes each)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "CSRBAStereoSLAMEstimator.h"

static bool read_raw(const std::string& path, std::vector<uint8_t>& buf, size_t n)
{
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) return false;
	buf.resize(n);
	const size_t got = std::fread(buf.data(), 1, n, f);
	std::fclose(f);
	return got == n;
}

static int stream_main(int argc, char** argv)
{
	if (argc < 7) { std::fprintf(stderr, "usage: %s stream <voc> <dir> <n_pairs> <width> <height> [key=value ...]\n", argv[0]); return 2; }
	const std::string voc = argv[2], dir = argv[3];
	const int n = std::atoi(argv[4]), w = std::atoi(argv[5]), h = std::atoi(argv[6]);
	try {
		ssk::Context ctx(0);
		ssk::TSRBAStereoSLAMOptions opts = ssk::TSRBAStereoSLAMOptions::kitti();
		opts.stereo_camera.ncols = w; opts.stereo_camera.nrows = h;
		for (int i = 7; i < argc; ++i) {
			const std::string kv = argv[i];
			const size_t eq = kv.find('=');
			if (eq == std::string::npos) { std::fprintf(stderr, "bad option %s\n", argv[i]); return 2; }
			const std::string k = kv.substr(0, eq);
			const double v = std::atof(kv.c_str() + eq + 1);
			if (k == "submap_size") opts.srba_submap_size = (size_t)v;
			else if (k == "max_tree_depth") opts.srba_max_tree_depth = (size_t)v;
			else if (k == "max_optimize_depth") opts.srba_max_optimize_depth = (size_t)v;
			else if (k == "lc_distance") opts.lc_distance = (size_t)v;
			else if (k == "max_translation") opts.max_translation = v;
			else if (k == "max_rotation") opts.max_rotation = v;
			else if (k == "updated_matches_th") opts.updated_matches_th = (size_t)v;
			else if (k == "up_matches_th_plus") opts.up_matches_th_plus = (size_t)v;
			else if (k == "vo_id_tracking_th") opts.vo_id_tracking_th = (int)v;
			else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
		}
		ssk::CSRBAStereoSLAMEstimator est(ctx, opts);
		est.m_bow_manager.loadVocabulary(voc);                       // EST.cpp:1120 loadVocabularyFromConfigFile
		std::vector<uint8_t> left, right;
		double loop_ms = 0, kf_ms = 0;                             // time inside performStereoSLAM (file reads excluded), frame 0 excluded
		size_t timed_frames = 0, timed_kfs = 0;
		for (int k = 0; k < n; ++k) {
			char a[64], b[64];
			std::snprintf(a, sizeof a, "/left_%04d.raw", k);
			std::snprintf(b, sizeof b, "/right_%04d.raw", k);
			if (!read_raw(dir + a, left, (size_t)w * h) || !read_raw(dir + b, right, (size_t)w * h)) { std::fprintf(stderr, "cannot read pair %d\n", k); return 3; }
			ssk::StereoImages im;
			im.left = left.data(); im.right = right.data(); im.width = w; im.height = h; im.stride = w;
			const auto t0 = std::chrono::steady_clock::now();
			const ssk::TFrameReport r = est.performStereoSLAM(im);
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			if (k > 0) { loop_ms += ms; ++timed_frames; if (r.inserted_kf) { kf_ms += ms; ++timed_kfs; } }
			std::printf("FRAME %zu vo_valid %d tracked_frame %zu tracked_kf %zu vo_t %.4f checked %d force_kf %d check_track %d check_dist %d inserted %d pot_lc %d conf_lc %d lc_id %lld highest %zu n_kf %zu",
			            r.frame, (int)r.odometry.valid, r.odometry.tracked_feats_from_last_frame, r.odometry.tracked_feats_from_last_KF,
			            r.odometry.outPose.translationNorm(), (int)r.checked_for_kf, (int)r.vo_force_new_kf, (int)r.vo_force_check_tracking,
			            (int)r.vo_force_check_distance, (int)r.inserted_kf, (int)r.potential_loop_closure, (int)r.confirmed_loop_closure,
			            r.similar.lc_id == ssk::policy::INVALID_KF_ID ? -1LL : (long long)r.similar.lc_id, r.highest_tracked, est.m_keyframes.size());
			std::printf(" query");
			for (const ssk::Result& q : r.query) std::printf(" %u:%.4f", q.Id, q.Score);
			std::printf(" similar");
			for (size_t s : r.similar.similar_kfs) std::printf(" %zu", s);
			if (r.inserted_kf) std::printf(" kf %zu edges %zu err %.4f -> %.4f", r.kf_info.kf_id, r.kf_info.created_edge_ids.size(),
			                               r.kf_info.optimize_results.total_sqr_error_init, r.kf_info.optimize_results.total_sqr_error_final);
			std::printf("\n");
		}
		std::printf("keyframes %zu landmarks %zu edges %zu db %zu\n", est.m_keyframes.size(), est.rba.get_rba_state().unknown_lms.size(),
		            est.rba.get_rba_state().k2k_edges.size(), est.m_bow_manager.size());
		if (timed_frames)
			std::printf("timing frames %zu loop_ms %.2f ms_per_frame %.3f frames_per_s %.1f keyframe_frames %zu ms_per_keyframe_frame %.3f\n", timed_frames, loop_ms,
			            loop_ms / timed_frames, 1e3 * timed_frames / loop_ms, timed_kfs, timed_kfs ? kf_ms / timed_kfs : 0.0);
		// after the loop: the global pose graph (EST.cpp:945-987)
		ssk_pg_result pg;
		const std::vector<ssk::Pose3D> poses = est.optimizeGlobalPoseGraph(&pg, dir + "/out_kf_poses.txt");
		std::printf("posegraph nodes %zu iters %d accepted %d stop %d cost %.6g -> %.6g last %.4f %.4f %.4f\n", poses.size(), pg.iters, pg.accepted,
		            pg.stop_reason, pg.cost_init, pg.cost_final, poses.empty() ? 0.0 : poses.back().x(), poses.empty() ? 0.0 : poses.back().y(),
		            poses.empty() ? 0.0 : poses.back().z());
	} catch (const std::exception& e) {
		std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
		return 1;
	}
	return 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && std::string(argv[1]) == "stream") return stream_main(argc, argv);
	if (argc < 5) { std::fprintf(stderr, "usage: %s <dir> <n_pairs> <width> <height> [debug_out_dir]\n", argv[0]); return 2; }
	const std::string dir = argv[1];
	const int n = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]);
	try {
		ssk::Context ctx(0);
		ssk::TSRBAStereoSLAMOptions opts = ssk::TSRBAStereoSLAMOptions::kitti();
		opts.stereo_camera.ncols = w; opts.stereo_camera.nrows = h;
		ssk::TGeneralOptions g;
		if (argc > 5) { g.debug = true; g.out_dir = argv[5]; }
		ssk::CSRBAStereoSLAMEstimator est(ctx, opts, g);
		std::vector<uint8_t> left, right;
		for (int k = 0; k < n; ++k) {
			char a[64], b[64];
			std::snprintf(a, sizeof a, "/left_%04d.raw", k);
			std::snprintf(b, sizeof b, "/right_%04d.raw", k);
			if (!read_raw(dir + a, left, (size_t)w * h) || !read_raw(dir + b, right, (size_t)w * h)) {
				std::fprintf(stderr, "cannot read pair %d\n", k);
				return 3;
			}
			ssk::StereoImages im;
			im.left = left.data(); im.right = right.data(); im.width = w; im.height = h; im.stride = w;
			const ssk::TNewKeyFrameInfo info = est.processKeyframe(im);
			const ssk::CStereoSLAMKF& kf = est.m_keyframes.back();
			ssk::Pose3D rel;
			const bool have = est.rba.get_kf_relative_pose(0, info.kf_id, 1000, rel);   // pose of this KF seen from KF 0
			size_t shared = 0;
			if (k > 0) {
				const auto& prev = est.m_keyframes[est.m_keyframes.size() - 2].m_matches_ID;
				for (size_t id : kf.m_matches_ID) shared += std::count(prev.begin(), prev.end(), id) ? 1 : 0;
			}
			std::printf("KF %zu nl %zu nr %zu matches %zu shared_ids %zu edges %zu stg1_rmse %.6f err %.6f -> %.6f obs %zu iters %zu pose %d %.6f %.6f %.6f fast_th %d\n",
			            info.kf_id, kf.m_keypoints_left.size(), kf.m_keypoints_right.size(), kf.m_matches.size(), shared,
			            info.created_edge_ids.size(), info.optimize_results_stg1.obs_rmse, info.optimize_results.total_sqr_error_init,
			            info.optimize_results.total_sqr_error_final, info.optimize_results.num_observations, info.optimize_results.num_iters,
			            (int)have, rel.x(), rel.y(), rel.z(), est.currentFASTThreshold());
		}
		std::printf("landmarks %zu edges %zu last_match_id %zu\n", est.rba.get_rba_state().unknown_lms.size(),
		            est.rba.get_rba_state().k2k_edges.size(), est.lastMatchID());
		if (g.debug) {                                             // keyframe section of the reference's state file
			const std::string f = g.out_dir + "/keyframes.kfrec";
			ssk::CSRBAStereoSLAMEstimator back(ctx, opts, g);
			if (!est.saveKeyframes(f) || !back.loadKeyframes(f) || back.m_keyframes.size() != est.m_keyframes.size()) {
				std::fprintf(stderr, "keyframe record round trip failed\n");
				return 4;
			}
			std::printf("saved %zu keyframes to %s\n", est.m_keyframes.size(), f.c_str());
		}
	} catch (const std::exception& e) {     // MAIN.cpp:66-75
		std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
		return 1;
	}
	return 0;
}
