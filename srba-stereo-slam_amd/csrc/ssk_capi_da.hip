// ssk_capi_da.hip -- C-ABI entry points of seam B2 (keyframe<->keyframe data association):
// host-pointer drop-ins for the reference call sites plus the device-resident batched forms.
#include "ssk_internal.h"
#include "ssk_frontend.h"

int ssk_da_max_matches();

extern "C" {

int ssk_hamming_match_batched_dev(ssk_ctx* ctx,
                                  const uint8_t* d_desc_q, const int32_t* d_q_off, const int32_t* d_q_gather,
                                  const uint8_t* d_desc_t, const int32_t* d_t_off, const int32_t* d_t_gather,
                                  int n_pairs, int max_query_rows, int32_t* d_train_idx, int32_t* d_dist)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_pairs < 0 || max_query_rows < 0 || !d_q_off || !d_t_off || !d_train_idx || !d_dist ||
	    (n_pairs > 0 && (!d_desc_q || !d_desc_t)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_hamming_match_batched_dev: bad argument");
	return ssk_launch_hamming(ctx, d_desc_q, d_q_off, d_q_off + 1, d_q_gather, d_desc_t, d_t_off, d_t_off + 1, d_t_gather,
	                          nullptr, n_pairs, max_query_rows, d_train_idx, d_dist);
}

int ssk_hamming_match(ssk_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt,
                      ssk_dmatch* out, int* n_out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (nq < 0 || nt < 0 || !n_out || (nq > 0 && (!query || !out)) || (nt > 0 && !train))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_hamming_match: bad argument");
	*n_out = 0;
	if (nq == 0 || nt == 0) return SSK_OK;          // BFMatcher clears the output
	if (nt >= (1 << 22)) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_hamming_match: nt >= 2^22 rows");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const size_t qb = (size_t)nq * 32, tb = (size_t)nt * 32;
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad(qb) + ssk_pad(tb) + 2 * ssk_pad((size_t)nq * 4) + 4 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, 2 * ssk_pad((size_t)nq * 4) + 2 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	uint8_t* d_q = ssk_arena_take<uint8_t>(ctx->dev, qb);
	uint8_t* d_t = ssk_arena_take<uint8_t>(ctx->dev, tb);
	int32_t* d_off = ssk_arena_take<int32_t>(ctx->dev, 4);
	int32_t* d_idx = ssk_arena_take<int32_t>(ctx->dev, nq);
	int32_t* d_dist = ssk_arena_take<int32_t>(ctx->dev, nq);
	int32_t* h_idx = ssk_arena_take<int32_t>(ctx->pin, nq);
	int32_t* h_dist = ssk_arena_take<int32_t>(ctx->pin, nq);
	const int32_t off[4] = {0, nq, 0, nt};
	SSK_HIP(ctx, hipMemcpyAsync(d_q, query, qb, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_t, train, tb, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, ctx->stream));
	if ((rc = ssk_launch_hamming(ctx, d_q, d_off, d_off + 1, nullptr, d_t, d_off + 2, d_off + 3, nullptr, nullptr, 1, nq, d_idx, d_dist))) return rc;
	SSK_HIP(ctx, hipMemcpyAsync(h_idx, d_idx, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(h_dist, d_dist, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int q = 0; q < nq; ++q) {
		out[q].queryIdx = q; out[q].trainIdx = h_idx[q]; out[q].imgIdx = 0; out[q].distance = (float)h_dist[q];
	}
	*n_out = nq;
	return SSK_OK;
}

// shared tail of ssk_da_filters / ssk_data_association: device arrays already in place.
static int run_filters_and_fetch(ssk_ctx* ctx, const int32_t* d_idx, const int32_t* d_dist, const int32_t* d_off,
                                 const float* d_axy, const float* d_bxy, const ssk_da_params* p, int na,
                                 int32_t* d_status, int32_t* d_tracked, int32_t* h_status, int32_t* h_tracked)
{
	int rc;
	if ((rc = ssk_launch_da_filters(ctx, d_idx, d_dist, nullptr, d_off, d_off + 1, d_off + 2, d_off + 3, d_axy, d_bxy, 1, *p, d_status, d_tracked))) return rc;
	SSK_HIP(ctx, hipMemcpyAsync(h_status, d_status, (size_t)na * 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(h_tracked, d_tracked, 4, hipMemcpyDeviceToHost, ctx->stream));
	return SSK_OK;
}

int ssk_da_filters(ssk_ctx* ctx, const ssk_dmatch* matches, int n_this, int n_other,
                   const float* this_xy, const float* other_xy, const ssk_da_params* params,
                   int32_t* status, int32_t* out_tracked)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_this < 0 || n_other < 0 || !params || !out_tracked ||
	    (n_this > 0 && (!matches || !this_xy || !status)) || (n_other > 0 && !other_xy))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_filters: bad argument");
	*out_tracked = 0;
	if (n_this == 0) return SSK_OK;
	if (n_this > ssk_da_max_matches() || n_other > ssk_da_max_matches())
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_da_filters: more than %d matches per keyframe", ssk_da_max_matches());
	for (int k = 0; k < n_this; ++k)
		if (matches[k].queryIdx != k || (n_other > 0 && (matches[k].trainIdx < 0 || matches[k].trainIdx >= n_other)))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_filters: matches[%d] is not BFMatcher-shaped", k);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t na = (size_t)n_this, nb = (size_t)(n_other > 0 ? n_other : 1);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, 3 * ssk_pad(na * 4) + ssk_pad(na * 8) + ssk_pad(nb * 8) + 8 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, 3 * ssk_pad(na * 4) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	int32_t* d_idx = ssk_arena_take<int32_t>(ctx->dev, na);
	int32_t* d_dist = ssk_arena_take<int32_t>(ctx->dev, na);
	int32_t* d_status = ssk_arena_take<int32_t>(ctx->dev, na);
	float* d_axy = ssk_arena_take<float>(ctx->dev, 2 * na);
	float* d_bxy = ssk_arena_take<float>(ctx->dev, 2 * nb);
	int32_t* d_off = ssk_arena_take<int32_t>(ctx->dev, 4);
	int32_t* d_tracked = ssk_arena_take<int32_t>(ctx->dev, 1);
	int32_t* h_idx = ssk_arena_take<int32_t>(ctx->pin, na);
	int32_t* h_dist = ssk_arena_take<int32_t>(ctx->pin, na);
	int32_t* h_status = ssk_arena_take<int32_t>(ctx->pin, na);
	int32_t* h_tracked = ssk_arena_take<int32_t>(ctx->pin, 1);
	for (int k = 0; k < n_this; ++k) { h_idx[k] = matches[k].trainIdx; h_dist[k] = (int32_t)matches[k].distance; }
	const int32_t off[4] = {0, n_this, 0, n_other};
	SSK_HIP(ctx, hipMemcpyAsync(d_idx, h_idx, na * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_dist, h_dist, na * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_axy, this_xy, na * 8, hipMemcpyHostToDevice, ctx->stream));
	if (n_other > 0) SSK_HIP(ctx, hipMemcpyAsync(d_bxy, other_xy, (size_t)n_other * 8, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, ctx->stream));
	if ((rc = run_filters_and_fetch(ctx, d_idx, d_dist, d_off, d_axy, d_bxy, params, n_this, d_status, d_tracked, h_status, h_tracked))) return rc;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::memcpy(status, h_status, na * 4);
	*out_tracked = *h_tracked;
	return SSK_OK;
}

int ssk_data_association(ssk_ctx* ctx, const ssk_kf_view* a, const ssk_kf_view* b,
                         const ssk_da_params* params, ssk_dmatch* out_matches, int32_t* out_status,
                         int32_t* out_tracked)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!a || !b || !params || !out_tracked || a->n_matches < 0 || b->n_matches < 0 ||
	    (a->n_matches > 0 && (!out_matches || !out_status || !a->matches || !a->desc_left || !a->kp_left)) ||
	    (b->n_matches > 0 && (!b->matches || !b->desc_left || !b->kp_left)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association: bad argument");
	*out_tracked = 0;
	const int na = a->n_matches, nb = b->n_matches;
	if (na == 0) return SSK_OK;
	if (na > ssk_da_max_matches() || nb > ssk_da_max_matches())
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_data_association: more than %d matches per keyframe", ssk_da_max_matches());
	for (int k = 0; k < na; ++k)
		if (a->matches[k].queryIdx < 0 || a->matches[k].queryIdx >= a->n_left)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association: this_kf match %d out of range", k);
	for (int k = 0; k < nb; ++k)
		if (b->matches[k].queryIdx < 0 || b->matches[k].queryIdx >= b->n_left)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association: other_kf match %d out of range", k);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t sa = (size_t)na, sb = (size_t)(nb > 0 ? nb : 1);
	const size_t nLa = (size_t)a->n_left, nLb = (size_t)(b->n_left > 0 ? b->n_left : 1);
	const size_t in_bytes = ssk_pad(nLa * 32) + ssk_pad(nLb * 32) + ssk_pad(sa * 4) + ssk_pad(sb * 4) + ssk_pad(sa * 8) + ssk_pad(sb * 8) + ssk_pad(16);
	const size_t out_bytes = 3 * ssk_pad(sa * 4) + ssk_pad(4);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, in_bytes + out_bytes + 4 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, in_bytes + out_bytes + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	// device: whole left descriptor blocks of both KFs + gather lists (match -> left keypoint row), so the
	// "matched-only" matrices curLDesc / preLDesc (EST.cpp:1351-1358,1432-1439) are never built.  One transfer in, one out.
	ssk_xfer_pack in; in.begin(ctx, in_bytes + out_bytes);
	int32_t *h_ga, *h_gb, *h_off, *h_idx, *h_dist, *h_status, *h_tracked;
	float *h_axy, *h_bxy;
	const uint8_t* d_da = in.put(a->desc_left, nLa * 32);
	const uint8_t* d_db = in.put(b->desc_left, b->n_left > 0 ? (size_t)b->n_left * 32 : 0);
	in.off = ssk_pad(nLa * 32) + ssk_pad(nLb * 32);          // (an empty other keyframe still owns its padded slot)
	int32_t* d_ga = in.slot<int32_t>(sa, &h_ga);
	int32_t* d_gb = in.slot<int32_t>(sb, &h_gb);
	float* d_axy = in.slot<float>(2 * sa, &h_axy);
	float* d_bxy = in.slot<float>(2 * sb, &h_bxy);
	int32_t* d_off = in.slot<int32_t>(4, &h_off);
	const size_t in_end = in.off;
	int32_t* d_idx = in.slot<int32_t>(sa, &h_idx);
	int32_t* d_dist = in.slot<int32_t>(sa, &h_dist);
	int32_t* d_status = in.slot<int32_t>(sa, &h_status);
	int32_t* d_tracked = in.slot<int32_t>(1, &h_tracked);
	for (int k = 0; k < na; ++k) {
		const int i = a->matches[k].queryIdx;
		h_ga[k] = i; h_axy[2 * k] = a->kp_left[i].x; h_axy[2 * k + 1] = a->kp_left[i].y;
	}
	for (int k = 0; k < nb; ++k) {
		const int i = b->matches[k].queryIdx;
		h_gb[k] = i; h_bxy[2 * k] = b->kp_left[i].x; h_bxy[2 * k + 1] = b->kp_left[i].y;
	}
	h_off[0] = 0; h_off[1] = na; h_off[2] = 0; h_off[3] = nb;
	SSK_HIP(ctx, in.upload(ctx->stream, 0, in_end));
	if ((rc = ssk_launch_hamming(ctx, d_da, d_off, d_off + 1, d_ga, d_db, d_off + 2, d_off + 3, d_gb, nullptr, 1, na, d_idx, d_dist))) return rc;
	if ((rc = ssk_launch_da_filters(ctx, d_idx, d_dist, nullptr, d_off, d_off + 1, d_off + 2, d_off + 3, d_axy, d_bxy, 1, *params, d_status, d_tracked))) return rc;
	SSK_HIP(ctx, in.download(ctx->stream, in_end, in.off));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int k = 0; k < na; ++k) {
		out_matches[k].queryIdx = k; out_matches[k].trainIdx = h_idx[k]; out_matches[k].imgIdx = 0;
		out_matches[k].distance = nb > 0 ? (float)h_dist[k] : 3.402823466e+38f;
		out_status[k] = h_status[k];
	}
	*out_tracked = *h_tracked;
	return SSK_OK;
}

int ssk_data_association_staged_multi(ssk_ctx* ctx, const ssk_kf_view* a, const ssk_kf_view* others, int n_others, const ssk_da_params* params,
                                      const ssk_fm_params* fmp, const uint64_t* fm_seeds, const ssk_cip_params* cpp, const double* init_poses,
                                      ssk_dmatch* out_matches, int32_t* out_status, ssk_da_staged_out* outs)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!a || n_others < 0 || (n_others > 0 && (!others || !outs)) || !params || a->n_matches < 0 ||
	    (a->n_matches > 0 && n_others > 0 && (!out_matches || !out_status || !a->matches || !a->desc_left || !a->kp_left || !a->kp_right)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: bad argument");
	if (fmp && (!(fmp->threshold > 0) || !(fmp->norm_scale > 0) || fmp->max_iters < 1))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: threshold, norm_scale and max_iters must be positive");
	if (cpp && cpp->use_robust_kernel && !(cpp->kernel_param > 0)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: kernel_param must be > 0");
	if (n_others == 0) return SSK_OK;
	std::memset(outs, 0, sizeof(ssk_da_staged_out) * (size_t)n_others);
	const int na = a->n_matches, S = n_others;
	if (na == 0) return SSK_OK;
	size_t nb_tot = 0, nLb_tot = 0;
	for (int p = -1; p < S; ++p) {
		const ssk_kf_view* kf = p < 0 ? a : others + p;
		if (kf->n_matches < 0 || (kf->n_matches > 0 && (!kf->matches || !kf->desc_left || !kf->kp_left || !kf->kp_right)))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: keyframe %d: bad view", p);
		if (kf->n_matches > ssk_da_max_matches())
			return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_data_association_staged: more than %d matches per keyframe", ssk_da_max_matches());
		for (int k = 0; k < kf->n_matches; ++k)
			if (kf->matches[k].queryIdx < 0 || kf->matches[k].queryIdx >= kf->n_left || kf->matches[k].trainIdx < 0 || kf->matches[k].trainIdx >= kf->n_right)
				return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: keyframe %d: match %d outside the keypoint lists", p, k);
		if (p >= 0) { nb_tot += (size_t)kf->n_matches; nLb_tot += (size_t)(kf->n_matches > 0 ? kf->n_left : 0); }
	}
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t sa = (size_t)na, sb = nb_tot > 0 ? nb_tot : 1, nLa = (size_t)a->n_left, nLb = nLb_tot > 0 ? nLb_tot : 1, sS = (size_t)S;
	const size_t total = ssk_pad(nLa * 32) + ssk_pad(nLb * 32) + ssk_pad(sa * 4) + ssk_pad(sb * 4) + ssk_pad(sa * 8) + ssk_pad(sb * 8) +
	                     ssk_pad(sa * 16) + ssk_pad(sb * 16) + ssk_pad(96 * sS) + ssk_pad(8 * sS) + ssk_pad(20 * sS) +
	                     3 * ssk_pad(sS * sa * 4) + ssk_pad(12 * sS) + ssk_pad(sizeof(ssk_fm_result) * sS) + ssk_pad(sizeof(ssk_cip_result) * sS);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, total + 2 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, total + 2 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	// pair p: query rows [0, na) of this keyframe, train rows [t_beg[p], t_end[p]) of the others laid end to end, output row p * na
	ssk_xfer_pack pk; pk.begin(ctx, total);
	uint8_t *h_da, *h_db;
	int32_t *h_ga, *h_gb, *h_rng, *h_idx, *h_dist, *h_status, *h_tr;
	float *h_axy, *h_bxy, *h_aobs, *h_bobs;
	double* h_T;
	unsigned long long* h_seed;
	ssk_fm_result* h_fm;
	ssk_cip_result* h_cip;
	const uint8_t* d_da = pk.slot<uint8_t>(nLa * 32, &h_da);
	const uint8_t* d_db = pk.slot<uint8_t>(nLb * 32, &h_db);
	int32_t* d_ga = pk.slot<int32_t>(sa, &h_ga);
	int32_t* d_gb = pk.slot<int32_t>(sb, &h_gb);
	float* d_axy = pk.slot<float>(2 * sa, &h_axy);
	float* d_bxy = pk.slot<float>(2 * sb, &h_bxy);
	float* d_aobs = pk.slot<float>(4 * sa, &h_aobs);
	float* d_bobs = pk.slot<float>(4 * sb, &h_bobs);
	double* d_T = pk.slot<double>(12 * sS, &h_T);
	unsigned long long* d_seed = pk.slot<unsigned long long>(sS, &h_seed);
	int32_t* d_rng = pk.slot<int32_t>(5 * sS, &h_rng);           // q_beg | q_end | t_beg | t_end | out_beg
	const size_t in_end = pk.off;
	int32_t* d_idx = pk.slot<int32_t>(sS * sa, &h_idx);
	int32_t* d_dist = pk.slot<int32_t>(sS * sa, &h_dist);
	int32_t* d_status = pk.slot<int32_t>(sS * sa, &h_status);
	int32_t* d_tr = pk.slot<int32_t>(3 * sS, &h_tr);
	ssk_fm_result* d_fm = pk.slot<ssk_fm_result>(sS, &h_fm);
	ssk_cip_result* d_cip = pk.slot<ssk_cip_result>(sS, &h_cip);
	auto fill = [](const ssk_kf_view* kf, int row0, int32_t* g, float* xy, float* ob) {
		for (int k = 0; k < kf->n_matches; ++k) {
			const int l = kf->matches[k].queryIdx, r = kf->matches[k].trainIdx;
			g[k] = row0 + l;
			xy[2 * k] = kf->kp_left[l].x; xy[2 * k + 1] = kf->kp_left[l].y;
			ob[4 * k] = kf->kp_left[l].x; ob[4 * k + 1] = kf->kp_left[l].y; ob[4 * k + 2] = kf->kp_right[r].x; ob[4 * k + 3] = kf->kp_right[r].y;
		}
	};
	std::memcpy(h_da, a->desc_left, nLa * 32);
	fill(a, 0, h_ga, h_axy, h_aobs);
	static const double identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
	size_t row = 0, m0 = 0;
	for (int p = 0; p < S; ++p) {
		const ssk_kf_view* kf = others + p;
		if (kf->n_matches > 0) {
			std::memcpy(h_db + row * 32, kf->desc_left, (size_t)kf->n_left * 32);
			fill(kf, (int)row, h_gb + m0, h_bxy + 2 * m0, h_bobs + 4 * m0);
		}
		h_rng[p] = 0; h_rng[S + p] = na; h_rng[2 * S + p] = (int32_t)m0; h_rng[3 * S + p] = (int32_t)(m0 + (size_t)kf->n_matches); h_rng[4 * S + p] = p * na;
		std::memcpy(h_T + 12 * p, init_poses ? init_poses + 12 * p : identity, sizeof identity);
		h_seed[p] = fm_seeds ? fm_seeds[p] : (fmp ? fmp->seed + (uint64_t)p : 0);
		if (kf->n_matches > 0) { row += (size_t)kf->n_left; m0 += (size_t)kf->n_matches; }
	}
	const int32_t *q_beg = d_rng, *q_end = d_rng + S, *t_beg = d_rng + 2 * S, *t_end = d_rng + 3 * S, *o_beg = d_rng + 4 * S;
	SSK_HIP(ctx, pk.upload(ctx->stream, 0, in_end));
	if ((rc = ssk_launch_hamming(ctx, d_da, q_beg, q_end, d_ga, d_db, t_beg, t_end, d_gb, o_beg, S, na, d_idx, d_dist))) return rc;
	if ((rc = ssk_launch_da_filters(ctx, d_idx, d_dist, o_beg, q_beg, q_end, t_beg, t_end, d_axy, d_bxy, S, *params, d_status, d_tr))) return rc;
	if (fmp && (rc = ssk_launch_fm(ctx, d_idx, d_status, o_beg, q_beg, q_end, t_beg, reinterpret_cast<const float2*>(d_axy),
	                               reinterpret_cast<const float2*>(d_bxy), S, na, *fmp, d_fm, d_tr + S, d_seed))) return rc;
	if (cpp && (rc = ssk_launch_cip(ctx, d_idx, d_status, o_beg, q_beg, q_end, t_beg, reinterpret_cast<const float4*>(d_aobs),
	                                reinterpret_cast<const float4*>(d_bobs), d_T, S, na, *cpp, nullptr, d_cip, d_tr + 2 * S))) return rc;
	SSK_HIP(ctx, pk.download(ctx->stream, in_end, pk.off));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int p = 0; p < S; ++p) {
		const int nb = others[p].n_matches;
		for (int k = 0; k < na; ++k) {
			ssk_dmatch& m = out_matches[(size_t)p * sa + k];
			m.queryIdx = k; m.trainIdx = h_idx[(size_t)p * sa + k]; m.imgIdx = 0;
			m.distance = nb > 0 ? (float)h_dist[(size_t)p * sa + k] : 3.402823466e+38f;
			out_status[(size_t)p * sa + k] = h_status[(size_t)p * sa + k];
		}
		ssk_da_staged_out& o = outs[p];
		o.tracked_da = h_tr[p];
		o.tracked_fm = fmp ? h_tr[S + p] : o.tracked_da;
		o.tracked_final = cpp ? h_tr[2 * S + p] : o.tracked_fm;
		if (fmp) o.fm = h_fm[p];
		if (cpp) o.cip = h_cip[p];
	}
	return SSK_OK;
}

int ssk_data_association_staged(ssk_ctx* ctx, const ssk_kf_view* a, const ssk_kf_view* b, const ssk_da_params* params,
                                const ssk_fm_params* fmp, const ssk_cip_params* cpp, const double* init_pose,
                                ssk_dmatch* out_matches, int32_t* out_status, ssk_da_staged_out* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !out) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_data_association_staged: bad argument");
	return ssk_data_association_staged_multi(ctx, a, b, 1, params, fmp, nullptr, cpp, init_pose, out_matches, out_status, out);
}

}  // extern "C"
