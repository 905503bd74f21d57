// policy_tool.cpp -- CPU-only command-line driver of the facade's host logic (no GPU library), used by the tests to
// compare it with the oracle: the keyframe policy of ssk_kf_policy.h and the vocabulary reader of ssk_bow_vocab_io.h.
//   policy_tool vocab <voc.yml[.gz]> <out.bin>     packed arrays: i32 k, L, scoring, weighting, n_rec, n_words; then
//                                                  node_id[n_rec], parent_id[n_rec] (i32), weight[n_rec] (f64), desc[n_rec*32], word_node[n_words]
//   policy_tool policy < commands                  one command per line, one result line each:
//     vocheck incT_kf incR_kf incT incR tracked maxT_lim maxR_lim maxT maxR vo_th   -> 4 flags
//     similar newKfId submap_size lc_distance nq {id score topo}*nq  (topo -1 = far) -> potential lc_id n ids...
//     decide insert_in potential lc_id voForce updated_th up_plus maxT maxR n {kf_idx tracked}*n
//                                                                                    -> insert confirmed updated highest maxT maxR n sorted...
//     thr x th                                                                        -> translation rotation
//   policy_tool options <kitti|default>            the preset of TSRBAStereoSLAMOptions as `key value` lines (tests/test_kitti_ini.py
//                                                  compares the kitti preset with the reference's demo/config_imgdir_kitti_srba.ini)
#include <cstdio>
#include <iostream>
#include <sstream>
#include "ssk_bow_vocab_io.h"
#include "ssk_kf_policy.h"
#include "srba_stereo_slam_options.h"

using namespace ssk;

static int do_vocab(const char* in, const char* out)
{
	const VocabFile v = loadVocabFile(in);
	FILE* f = std::fopen(out, "wb");
	if (!f) return 2;
	const int32_t head[6] = {v.k, v.L, v.scoring, v.weighting, (int32_t)v.node_id.size(), (int32_t)v.word_node.size()};
	std::fwrite(head, 4, 6, f);
	std::fwrite(v.node_id.data(), 4, v.node_id.size(), f);
	std::fwrite(v.parent_id.data(), 4, v.parent_id.size(), f);
	std::fwrite(v.weight.data(), 8, v.weight.size(), f);
	std::fwrite(v.desc.data(), 1, v.desc.size(), f);
	std::fwrite(v.word_node.data(), 4, v.word_node.size(), f);
	std::fclose(f);
	return 0;
}

static int do_policy()
{
	std::string line;
	while (std::getline(std::cin, line)) {
		std::istringstream in(line);
		std::string cmd;
		if (!(in >> cmd)) continue;
		if (cmd == "vocheck") {
			double a, b, c, d, e, f, g, h; long long tr; int th;
			in >> a >> b >> c >> d >> tr >> e >> f >> g >> h >> th;
			const policy::TVOCheck r = policy::voCheck(a, b, c, d, (size_t)tr, e, f, g, h, th);
			std::printf("%d %d %d %d\n", r.voForceNewKf, r.voForceCheckTracking, r.voForceCheckDistance, r.voForceCheck);
		} else if (cmd == "similar") {
			long long id, sub, lcd; int nq;
			in >> id >> sub >> lcd >> nq;
			std::vector<policy::TQuery> q(nq);
			std::vector<long long> topo(nq);
			for (int i = 0; i < nq; ++i) { long long qi; in >> qi >> q[i].Score >> topo[i]; q[i].Id = (size_t)qi; }
			std::vector<size_t> sim; size_t lc;
			const bool pot = policy::getSimilarKfs((size_t)id, q, (size_t)sub, (size_t)lcd, [&](size_t to) {
				for (int i = 0; i < nq; ++i) if (q[i].Id == to) return topo[i] < 0 ? policy::TOPO_FAR : (size_t)topo[i];
				return policy::TOPO_FAR; }, sim, lc);
			std::printf("%d %lld %zu", pot, lc == policy::INVALID_KF_ID ? -1LL : (long long)lc, sim.size());
			for (size_t s : sim) std::printf(" %zu", s);
			std::printf("\n");
		} else if (cmd == "decide") {
			int ins, pot, vo, n; long long lc, uth, upl; double mt, mr;
			in >> ins >> pot >> lc >> vo >> uth >> upl >> mt >> mr >> n;
			std::vector<policy::TDaSummary> da(n);
			for (int i = 0; i < n; ++i) { long long k, t; in >> k >> t; da[i].kf_idx = (size_t)k; da[i].tracked_matches = (size_t)t; }
			const policy::TInsertDecision d = policy::decideInsertion(ins != 0, da, pot != 0, lc < 0 ? policy::INVALID_KF_ID : (size_t)lc, vo != 0,
			                                                          (size_t)uth, (size_t)upl, mt, mr);
			std::printf("%d %d %d %zu %.17g %.17g %zu", d.insertNewKf, d.confirmedLoopClosure, d.thresholdsUpdated, d.highestNumberTrackedFeats,
			            d.max_translation, d.max_rotation, d.sortedIndices.size());
			for (size_t s : d.sortedIndices) std::printf(" %zu", s);
			std::printf("\n");
		} else if (cmd == "thr") {
			double x, th;
			in >> x >> th;
			std::printf("%.17g %.17g\n", policy::updateTranslationThreshold(x, th), policy::updateRotationThreshold(x, th));
		} else {
			std::printf("? %s\n", cmd.c_str());
		}
	}
	return 0;
}

static int do_options(const std::string& preset)
{
	if (preset != "kitti" && preset != "default") return 1;
	const TSRBAStereoSLAMOptions o = preset == "kitti" ? TSRBAStereoSLAMOptions::kitti() : TSRBAStereoSLAMOptions();
#define P(k, v) std::printf("%s %.17g\n", k, (double)(v))
	P("n_levels", o.n_levels); P("n_feats", o.n_feats); P("detect_fast_th", o.detect_fast_th);
	P("adaptive_th_min_matches", o.adaptive_th_min_matches); P("orb_adaptive_fast_th", o.orb_adaptive_fast_th);
	P("non_maximal_suppression", o.non_maximal_suppression); P("nms_min_distance", o.nms_min_distance);
	P("max_y_diff", o.max_y_diff); P("orb_max_distance", o.orb_max_distance);
	P("ransac_fit_prob", o.ransac_fit_prob); P("max_y_diff_epipolar", o.max_y_diff_epipolar); P("max_orb_distance_da", o.max_orb_distance_da);
	P("da_filter_by_direction", o.da_filter_by_direction); P("da_filter_by_orb_distance", o.da_filter_by_orb_distance);
	P("da_filter_by_fund_matrix", o.da_filter_by_fund_matrix); P("da_filter_by_pose_change", o.da_filter_by_pose_change);
	P("residual_th", o.residual_th);
	P("ls_initial_max_iters", o.ls_initial_max_iters); P("ls_max_iters", o.ls_max_iters); P("ls_max_incr_cost", o.ls_max_incr_cost);
	P("ls_bad_tracking_th", o.ls_bad_tracking_th); P("ls_residual_threshold", o.ls_residual_threshold);
	P("ls_min_mod_out_vector", o.ls_min_mod_out_vector); P("ls_kernel_param", o.ls_kernel_param); P("ls_use_robust_kernel", o.ls_use_robust_kernel);
	P("use_initial_pose", o.use_initial_pose);
	P("srba_kernel_param", o.srba_kernel_param); P("srba_use_robust_kernel", o.srba_use_robust_kernel);
	P("srba_use_robust_kernel_stage1", o.srba_use_robust_kernel_stage1);
	P("srba_submap_size", o.srba_submap_size); P("srba_max_tree_depth", o.srba_max_tree_depth); P("srba_max_optimize_depth", o.srba_max_optimize_depth);
	P("updated_matches_th", o.updated_matches_th); P("up_matches_th_plus", o.up_matches_th_plus); P("lc_distance", o.lc_distance);
	P("max_rotation", o.max_rotation); P("max_translation", o.max_translation); P("vo_id_tracking_th", o.vo_id_tracking_th);
	const TStereoCamera& c = o.stereo_camera;
	P("fx_l", c.fx_l); P("fy_l", c.fy_l); P("cx_l", c.cx_l); P("cy_l", c.cy_l);
	P("fx_r", c.fx_r); P("fy_r", c.fy_r); P("cx_r", c.cx_r); P("cy_r", c.cy_r); P("baseline", c.baseline); P("ncols", c.ncols); P("nrows", c.nrows);
	for (int i = 0; i < 12; ++i) { char k[32]; std::snprintf(k, sizeof k, "camera_pose_on_robot_rvt_%d", i); P(k, o.camera_pose_on_robot_rvt.m[i]); }
#undef P
	return 0;
}

int main(int argc, char** argv)
{
	try {
		if (argc == 3 && std::string(argv[1]) == "options") return do_options(argv[2]);
		if (argc == 4 && std::string(argv[1]) == "vocab") return do_vocab(argv[2], argv[3]);
		if (argc == 2 && std::string(argv[1]) == "policy") return do_policy();
	} catch (const std::exception& e) {
		std::fprintf(stderr, "policy_tool: %s\n", e.what());
		return 3;
	}
	std::fprintf(stderr, "usage: policy_tool vocab <voc.yml[.gz]> <out.bin> | policy_tool policy < commands\n");
	return 1;
}
