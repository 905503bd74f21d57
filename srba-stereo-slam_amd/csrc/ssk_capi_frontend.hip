// ssk_capi_frontend.hip -- C-ABI entry points of seam B1 (stereo front-end) and the device-resident
// keyframe batch, plus the batched data association that runs directly on batch-resident features.
#include "ssk_frontend.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace {

int validate_params(ssk_ctx* ctx, const ssk_frontend_params* p, int max_feats, const char* who)
{
	if (!p) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "%s: params is NULL", who);
	if (p->n_levels != 1)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "%s: n_levels must be 1 (the reference forces nlevels=1, EST.cpp:1141)", who);
	if (p->n_feats < 1 || p->n_feats > max_feats || p->n_feats > SSK_MAX_FEATS)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "%s: n_feats %d outside [1,%d]", who, p->n_feats, max_feats < SSK_MAX_FEATS ? max_feats : SSK_MAX_FEATS);
	if (p->fast_threshold < 1 || p->fast_threshold > 254)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "%s: fast_threshold %d outside [1,254]", who, p->fast_threshold);
	if (p->nms_radius < 0 || p->orb_max_distance < 0)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "%s: negative nms_radius / orb_max_distance", who);
	return SSK_OK;
}

// beg/end/out rows of every data-association pair, computed on the device from the match counts
__global__ void da_ranges_kernel(const int32_t* __restrict__ pairs, int n_pairs, int max_feats,
                                 const int32_t* __restrict__ nm_this, const int32_t* __restrict__ nm_other,
                                 int32_t* __restrict__ r /* 5 x n_pairs: q_beg q_end t_beg t_end out_beg */)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	const int a = pairs[2 * p], b = pairs[2 * p + 1];
	r[p] = a * max_feats;
	r[n_pairs + p] = a * max_feats + min(nm_this[a], max_feats);
	r[2 * n_pairs + p] = b * max_feats;
	r[3 * n_pairs + p] = b * max_feats + min(nm_other[b], max_feats);
	r[4 * n_pairs + p] = p * max_feats;
}

template <typename T>
int dev_alloc(ssk_ctx* ctx, T** p, size_t count)
{
	SSK_HIP(ctx, hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
	return SSK_OK;
}

int batch_create(ssk_ctx* ctx, int n_kf, int w, int h, int max_feats, ssk_kfbatch** out)
{
	ssk_kfbatch* b = new ssk_kfbatch();
	b->ctx = ctx; b->n_kf = n_kf; b->w = w; b->h = h; b->max_feats = max_feats;
	b->pitch = (w + 8 + 127) & ~127;
	const int ew = w - 2 * SSK_EDGE, eh = h - 2 * SSK_EDGE;
	b->cand_cap = ((ew + 1) / 2 + 1) * ((eh + 1) / 2 + 1);   // strict 3x3 maxima: at most one per 2x2 block
	const size_t ni = (size_t)n_kf * 2, nf = ni * max_feats, nk = (size_t)n_kf * max_feats;
	int rc = SSK_OK;
	if (!rc) rc = dev_alloc(ctx, &b->img, ni * h * b->pitch + 256);
	if (!rc) rc = dev_alloc(ctx, &b->cand, ni * b->cand_cap);
	if (!rc) rc = dev_alloc(ctx, &b->ncand, ni + SSK_FAST_CTR_INTS);      // + the tile counters of ssk_launch_fast
	if (!rc) rc = dev_alloc(ctx, &b->kp_xy, nf);
	if (!rc) rc = dev_alloc(ctx, &b->kp_resp, nf);
	if (!rc) rc = dev_alloc(ctx, &b->kp_angle, nf);
	if (!rc) rc = dev_alloc(ctx, &b->nkp, ni);
	if (!rc) rc = dev_alloc(ctx, &b->desc, nf * 32);
	if (!rc) rc = dev_alloc(ctx, &b->m_q, nk);
	if (!rc) rc = dev_alloc(ctx, &b->m_t, nk);
	if (!rc) rc = dev_alloc(ctx, &b->m_d, nk);
	if (!rc) rc = dev_alloc(ctx, &b->nm, (size_t)n_kf);
	if (!rc) rc = dev_alloc(ctx, &b->m_row, nk);
	if (!rc) rc = dev_alloc(ctx, &b->m_xy, nk);
	if (!rc) rc = dev_alloc(ctx, &b->m_obs, nk);
	if (rc) { ssk_kfbatch_destroy(b); return rc; }
	hipError_t e = hipMemsetAsync(b->img, 0, ni * h * b->pitch + 256, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(b->nkp, 0, ni * 4, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(b->nm, 0, (size_t)n_kf * 4, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(b->ncand, 0, (ni + SSK_FAST_CTR_INTS) * 4, ctx->stream);
	// the storage is the caller's to write when create returns -- through ssk_kfbatch_images on ANY stream (bench.py fills it with torch on the null
	// stream): a fill still queued on the context's non-blocking stream raced such a write and zeroed the first images of a batch now and then
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) { ssk_kfbatch_destroy(b); return ssk_set_error(ctx, SSK_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)); }
	*out = b;
	return SSK_OK;
}

}  // namespace

void ssk_kfbatch_destroy_internal(ssk_kfbatch* b) { ssk_kfbatch_destroy(b); }

extern "C" {

int ssk_set_brief_pattern(ssk_ctx* ctx, const int8_t* pattern1024)
{
	if (!ctx || !pattern1024) return SSK_ERR_BAD_ARG;
	for (int i = 0; i < 1024; ++i)
		if (pattern1024[i] < -13 || pattern1024[i] > 13)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_set_brief_pattern: |coordinate| > 13 at %d", i);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	std::memcpy(ctx->pattern, pattern1024, 1024);
	SSK_HIP(ctx, ssk_upload_pattern(ctx));
	return SSK_OK;
}

int ssk_set_orb_steering(ssk_ctx* ctx, int mode)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (mode != 0 && mode != 1) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_set_orb_steering: mode %d (0 = centroid direction, 1 = cv::ORB)", mode);
	ctx->orb_steering = mode;
	return SSK_OK;
}

int ssk_get_brief_pattern(ssk_ctx* ctx, int8_t* pattern1024)
{
	if (!ctx || !pattern1024) return SSK_ERR_BAD_ARG;
	std::memcpy(pattern1024, ctx->pattern, 1024);
	return SSK_OK;
}

int ssk_kfbatch_create(ssk_ctx* ctx, int n_kf, int width, int height, int max_feats, ssk_kfbatch** out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!out || n_kf < 1 || n_kf > 16384 || max_feats < 1 || max_feats > SSK_MAX_FEATS || width > 4096 || height > 4096 ||
	    width < 2 * SSK_EDGE + 1 || height < 2 * SSK_EDGE + 1)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_create: bad argument (image %dx%d must lie in [63,4096]^2, max_feats <= %d)",
		                     width, height, SSK_MAX_FEATS);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	return batch_create(ctx, n_kf, width, height, max_feats, out);
}

void ssk_kfbatch_destroy(ssk_kfbatch* b)
{
	if (!b) return;
	if (b->ctx) { (void)hipSetDevice(b->ctx->device); (void)hipStreamSynchronize(b->ctx->stream); }
	void* ptrs[] = {b->img, b->cand, b->ncand, b->kp_xy, b->kp_resp, b->kp_angle, b->kp_cs, b->nkp, b->desc,
	                b->m_q, b->m_t, b->m_d, b->nm, b->m_row, b->m_xy, b->m_obs};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	ssk_kfbatch_bow_free(b);
	delete b;
}

int ssk_kfbatch_images(ssk_kfbatch* b, uint8_t** d_images, int* pitch)
{
	if (!b || !d_images || !pitch) return SSK_ERR_BAD_ARG;
	*d_images = b->img;
	*pitch = b->pitch;
	return SSK_OK;
}

int ssk_kfbatch_upload(ssk_ctx* ctx, ssk_kfbatch* b, int kf, const uint8_t* left, const uint8_t* right, int stride)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || kf < 0 || kf >= b->n_kf || !left || !right || stride < b->w)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_upload: bad argument");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	uint8_t* dl = b->img + (size_t)(kf * 2) * b->h * b->pitch;
	// the rows are re-pitched on the host into pinned staging and go over as ONE contiguous DMA for both eyes (a 2-D
	// copy from pageable memory is issued row by row: 6 ms for a 1241x376 pair instead of ~0.1 ms)
	const size_t eye = (size_t)b->h * b->pitch;
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, ssk_pad(2 * eye) + 256))) return rc;
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // staging may still feed an earlier copy
	ssk_arena_reset(ctx->pin);
	uint8_t* st = ssk_arena_take<uint8_t>(ctx->pin, 2 * eye);
	for (int e = 0; e < 2; ++e) {
		const uint8_t* src = e ? right : left;
		uint8_t* dst = st + e * eye;
		for (int y = 0; y < b->h; ++y) {
			std::memcpy(dst + (size_t)y * b->pitch, src + (size_t)y * stride, (size_t)b->w);
			std::memset(dst + (size_t)y * b->pitch + b->w, 0, (size_t)(b->pitch - b->w));
		}
	}
	SSK_HIP(ctx, hipMemcpyAsync(dl, st, 2 * eye, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the staging arena is shared with every other host entry
	return SSK_OK;
}

int ssk_frontend_run_batch(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_frontend_params* params)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || kf_begin < 0 || kf_end > b->n_kf || kf_begin > kf_end)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_frontend_run_batch: keyframe range [%d,%d) outside the batch", kf_begin, kf_end);
	int rc = validate_params(ctx, params, b->max_feats, "ssk_frontend_run_batch");
	if (rc) return rc;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const int i0 = kf_begin * 2, i1 = kf_end * 2;
	if ((rc = ssk_launch_fast(ctx, b, i0, i1, params->fast_threshold))) return rc;
	if ((rc = ssk_launch_select(ctx, b, i0, i1, *params))) return rc;
	if ((rc = ssk_launch_describe(ctx, b, i0, i1, params->n_feats))) return rc;
	if ((rc = ssk_launch_stereo(ctx, b, kf_begin, kf_end, *params))) return rc;
	return SSK_OK;
}

int ssk_kfbatch_counts(ssk_ctx* ctx, ssk_kfbatch* b, int32_t* h_n_kp, int32_t* h_n_matches)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_counts: batch is NULL");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (h_n_kp) SSK_HIP(ctx, hipMemcpy(h_n_kp, b->nkp, sizeof(int32_t) * 2 * (size_t)b->n_kf, hipMemcpyDeviceToHost));
	if (h_n_matches) SSK_HIP(ctx, hipMemcpy(h_n_matches, b->nm, sizeof(int32_t) * (size_t)b->n_kf, hipMemcpyDeviceToHost));
	return SSK_OK;
}

int ssk_kfbatch_download(ssk_ctx* ctx, ssk_kfbatch* b, int kf, ssk_kf_out* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || !out || kf < 0 || kf >= b->n_kf || !out->kp_left || !out->kp_right || !out->desc_left || !out->desc_right || !out->matches)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_download: bad argument");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	// every array of the keyframe into the pinned arena behind one synchronisation (ten blocking copies cost 0.15 ms per frame)
	const int N = b->max_feats;
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, ssk_pad(16) + ssk_pad(sizeof(short2) * 2 * N) + 2 * ssk_pad(sizeof(float) * 2 * N) + ssk_pad((size_t)2 * N * 32) +
	                                               3 * ssk_pad(sizeof(int32_t) * N) + 2 * 256))) return rc;
	ssk_arena_reset(ctx->pin);
	int32_t* cnt = ssk_arena_take<int32_t>(ctx->pin, 4);
	short2* xy = ssk_arena_take<short2>(ctx->pin, (size_t)2 * N);
	float* resp = ssk_arena_take<float>(ctx->pin, (size_t)2 * N);
	float* ang = ssk_arena_take<float>(ctx->pin, (size_t)2 * N);
	uint8_t* desc = ssk_arena_take<uint8_t>(ctx->pin, (size_t)2 * N * 32);
	int32_t* mq = ssk_arena_take<int32_t>(ctx->pin, N);
	int32_t* mt = ssk_arena_take<int32_t>(ctx->pin, N);
	int32_t* md = ssk_arena_take<int32_t>(ctx->pin, N);
	const size_t o = (size_t)kf * 2 * N;
	SSK_HIP(ctx, hipMemcpyAsync(cnt, b->nkp + 2 * kf, 8, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(cnt + 2, b->nm + kf, 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(xy, b->kp_xy + o, sizeof(short2) * 2 * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(resp, b->kp_resp + o, sizeof(float) * 2 * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(ang, b->kp_angle + o, sizeof(float) * 2 * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(desc, b->desc + o * 32, (size_t)2 * N * 32, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(mq, b->m_q + (size_t)kf * N, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(mt, b->m_t + (size_t)kf * N, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(md, b->m_d + (size_t)kf * N, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const int nl = cnt[0], nr = cnt[1], nmm = cnt[2];
	if (nl > out->capacity || nr > out->capacity || nmm > out->capacity)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_kfbatch_download: capacity %d < %d/%d/%d", out->capacity, nl, nr, nmm);
	if (nl) std::memcpy(out->desc_left, desc, (size_t)nl * 32);
	if (nr) std::memcpy(out->desc_right, desc + (size_t)N * 32, (size_t)nr * 32);
	for (int e = 0; e < 2; ++e) {
		ssk_keypoint* kp = e ? out->kp_right : out->kp_left;
		const int n = e ? nr : nl;
		for (int i = 0; i < n; ++i) {
			const size_t s = (size_t)e * N + i;
			kp[i].x = (float)xy[s].x; kp[i].y = (float)xy[s].y; kp[i].size = 31.f;   // cv::ORB patchSize at level 0
			kp[i].angle = ang[s]; kp[i].response = resp[s]; kp[i].octave = 0; kp[i].class_id = -1;
		}
	}
	for (int i = 0; i < nmm; ++i) {
		out->matches[i].queryIdx = mq[i]; out->matches[i].trainIdx = mt[i]; out->matches[i].imgIdx = 0;
		out->matches[i].distance = (float)md[i];
	}
	out->n_left = nl; out->n_right = nr; out->n_matches = nmm;
	return SSK_OK;
}

int ssk_kfbatch_debug_candidates(ssk_ctx* ctx, ssk_kfbatch* b, int kf, int eye, uint32_t* keys, int cap, int* n)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || kf < 0 || kf >= b->n_kf || eye < 0 || eye > 1 || !keys || !n)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_debug_candidates: bad argument");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const int img = kf * 2 + eye;
	int32_t cnt = 0;
	SSK_HIP(ctx, hipMemcpy(&cnt, b->ncand + img, 4, hipMemcpyDeviceToHost));
	*n = cnt;
	const int m = cnt < cap ? cnt : cap;
	std::vector<uint32_t> all((size_t)(cnt > 0 ? cnt : 1));
	if (cnt > 0) SSK_HIP(ctx, hipMemcpy(all.data(), b->cand + (size_t)img * b->cand_cap, sizeof(uint32_t) * (size_t)cnt, hipMemcpyDeviceToHost));
	std::sort(all.begin(), all.begin() + cnt, [](uint32_t x, uint32_t y) { return x > y; });
	std::memcpy(keys, all.data(), sizeof(uint32_t) * (size_t)m);
	return SSK_OK;
}

int ssk_frontend_run(ssk_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height,
                     int stride, const ssk_frontend_params* params, ssk_kf_out* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!left || !right || !out || !params) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_frontend_run: NULL argument");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	ssk_kfbatch* b = ctx->single;
	if (!b || b->w != width || b->h != height || b->max_feats < params->n_feats) {
		if (b) { ssk_kfbatch_destroy(b); ctx->single = nullptr; }
		const int mf = params->n_feats > 512 ? params->n_feats : 512;
		int rc = ssk_kfbatch_create(ctx, 1, width, height, mf > SSK_MAX_FEATS ? SSK_MAX_FEATS : mf, &b);
		if (rc) return rc;
		ctx->single = b;
	}
	int rc;
	static const bool timing = std::getenv("SSK_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	if ((rc = ssk_kfbatch_upload(ctx, b, 0, left, right, stride))) return rc;
	if (timing) (void)hipStreamSynchronize(ctx->stream);
	const auto t1 = std::chrono::steady_clock::now();
	if ((rc = ssk_frontend_run_batch(ctx, b, 0, 1, params))) return rc;
	if (timing) (void)hipStreamSynchronize(ctx->stream);
	const auto t2 = std::chrono::steady_clock::now();
	rc = ssk_kfbatch_download(ctx, b, 0, out);
	if (timing) {
		const auto t3 = std::chrono::steady_clock::now();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
		std::fprintf(stderr, "[ssk timing] ssk_frontend_run: upload %.3f ms, kernels %.3f ms, download %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
	}
	return rc;
}

int ssk_da_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                     const ssk_da_params* params, int32_t* d_train_idx, int32_t* d_dist, int32_t* d_status,
                     int32_t* d_tracked)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!this_b || !other_b || n_pairs < 0 || !params || (n_pairs > 0 && (!pairs || !d_train_idx || !d_dist || !d_status || !d_tracked)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_run_batch: bad argument");
	if (this_b->max_feats != other_b->max_feats)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_run_batch: batches differ in max_feats");
	if (n_pairs == 0) return SSK_OK;
	for (int p = 0; p < n_pairs; ++p)
		if (pairs[2 * p] < 0 || pairs[2 * p] >= this_b->n_kf || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= other_b->n_kf)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_run_batch: pair %d references a keyframe outside its batch", p);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad((size_t)n_pairs * 8) + ssk_pad((size_t)n_pairs * 20) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	int32_t* d_pairs = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 2);
	int32_t* d_r = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 5);
	SSK_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, ctx->stream));
	const int N = this_b->max_feats;
	hipLaunchKernelGGL(da_ranges_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, d_pairs, n_pairs, N,
	                   this_b->nm, other_b->nm, d_r);
	SSK_LAUNCH_CHECK(ctx, "da_ranges_kernel");
	const int32_t *q_beg = d_r, *q_end = d_r + n_pairs, *t_beg = d_r + 2 * n_pairs, *t_end = d_r + 3 * n_pairs, *o_beg = d_r + 4 * n_pairs;
	// virtual rows are match slots (kf*N + k); m_row maps a slot to the left keypoint's descriptor row (Q3)
	if ((rc = ssk_launch_hamming(ctx, this_b->desc, q_beg, q_end, this_b->m_row, other_b->desc, t_beg, t_end, other_b->m_row,
	                             o_beg, n_pairs, N, d_train_idx, d_dist))) return rc;
	return ssk_launch_da_filters(ctx, d_train_idx, d_dist, o_beg, q_beg, q_end, t_beg, t_end,
	                             reinterpret_cast<const float*>(this_b->m_xy), reinterpret_cast<const float*>(other_b->m_xy),
	                             n_pairs, *params, d_status, d_tracked);
}

int ssk_cip_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                      const ssk_cip_params* params, const double* init_poses, const int32_t* d_train_idx,
                      int32_t* d_status, int32_t* d_tracked, ssk_cip_result* d_results)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!this_b || !other_b || n_pairs < 0 || !params || (n_pairs > 0 && (!pairs || !d_train_idx || !d_status || !d_tracked)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_run_batch: bad argument");
	if (this_b->max_feats != other_b->max_feats)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_run_batch: batches differ in max_feats");
	if (n_pairs == 0) return SSK_OK;
	for (int p = 0; p < n_pairs; ++p)
		if (pairs[2 * p] < 0 || pairs[2 * p] >= this_b->n_kf || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= other_b->n_kf)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_run_batch: pair %d references a keyframe outside its batch", p);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad((size_t)n_pairs * 8) + ssk_pad((size_t)n_pairs * 20) + ssk_pad((size_t)n_pairs * 96) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	int32_t* d_pairs = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 2);
	int32_t* d_r = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 5);
	double* d_init = nullptr;
	SSK_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, ctx->stream));
	if (init_poses) {
		d_init = ssk_arena_take<double>(ctx->dev, (size_t)n_pairs * 12);
		SSK_HIP(ctx, hipMemcpyAsync(d_init, init_poses, sizeof(double) * 12 * (size_t)n_pairs, hipMemcpyHostToDevice, ctx->stream));
	}
	const int N = this_b->max_feats;
	hipLaunchKernelGGL(da_ranges_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, d_pairs, n_pairs, N,
	                   this_b->nm, other_b->nm, d_r);
	SSK_LAUNCH_CHECK(ctx, "da_ranges_kernel");
	return ssk_launch_cip(ctx, d_train_idx, d_status, d_r + 4 * n_pairs, d_r, d_r + n_pairs, d_r + 2 * n_pairs, this_b->m_obs, other_b->m_obs,
	                      d_init, n_pairs, N, *params, nullptr, d_results, d_tracked);
}

int ssk_fm_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                     const ssk_fm_params* params, const int32_t* d_train_idx, int32_t* d_status, int32_t* d_tracked,
                     ssk_fm_result* d_results)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!this_b || !other_b || n_pairs < 0 || !params || (n_pairs > 0 && (!pairs || !d_train_idx || !d_status || !d_tracked)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_run_batch: bad argument");
	if (this_b->max_feats != other_b->max_feats)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_run_batch: batches differ in max_feats");
	if (!(params->threshold > 0) || !(params->norm_scale > 0) || params->max_iters < 1)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_run_batch: threshold, norm_scale and max_iters must be positive");
	if (n_pairs == 0) return SSK_OK;
	for (int p = 0; p < n_pairs; ++p)
		if (pairs[2 * p] < 0 || pairs[2 * p] >= this_b->n_kf || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= other_b->n_kf)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_run_batch: pair %d references a keyframe outside its batch", p);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad((size_t)n_pairs * 8) + ssk_pad((size_t)n_pairs * 20) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	int32_t* d_pairs = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 2);
	int32_t* d_r = ssk_arena_take<int32_t>(ctx->dev, (size_t)n_pairs * 5);
	SSK_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, ctx->stream));
	const int N = this_b->max_feats;
	hipLaunchKernelGGL(da_ranges_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, d_pairs, n_pairs, N,
	                   this_b->nm, other_b->nm, d_r);
	SSK_LAUNCH_CHECK(ctx, "da_ranges_kernel");
	return ssk_launch_fm(ctx, d_train_idx, d_status, d_r + 4 * n_pairs, d_r, d_r + n_pairs, d_r + 2 * n_pairs, this_b->m_xy, other_b->m_xy,
	                     n_pairs, N, *params, d_results, d_tracked);
}

}  // extern "C"
