/* oracle_edge_policy.c -- TEST INFRASTRUCTURE (CPU oracle; never linked into the product).
 *
 * Literal restatement of mySRBA::edge_creation_policy, reference src/srba-stereo-slam.h:144-308 (SRBA.h below), the fixed-size
 * submap topology the reference plugs into SRBA's define_new_keyframe: which kf-to-kf edges a new keyframe gets, the local-map
 * bookkeeping (m_kf_localmap_center_ID SRBA.h:67, m_localmap_kf_IDs SRBA.h:68) and the INITIAL inv_pose of every created edge.
 * What the policy reads of SRBA itself (absent: header-only library, no version pinned) is restated from its published behaviour:
 *   - create_kf2kf_edge appends the edge (from, to) and makes it part of the spanning trees at once, so the distance lookups of later
 *     iterations of the same call see it (SRBA.h:253-269 reads spanning_tree.sym.next_edge[from][to].distance: the topological
 *     distance when it is <= max_tree_depth, absent otherwise) -- here a breadth-first search limited to max_tree_depth;
 *   - make_ordered_list_base_kfs (SRBA.h:220-221): for the observations whose feature id is already a known landmark, the number of
 *     observations per base keyframe, as a multimap ordered by descending count (equal counts: insertion order = ascending base id);
 *   - define_new_keyframe registers a feature id seen for the first time as a landmark based at the new keyframe, AFTER the policy ran.
 * Poses are row-major 3x4 [R | t]; inv_pose of an edge = pose of `from` seen from `to`.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ssk_oracle.h"

#define ECP_MAX_KF 4096
#define ECP_MAX_EDGE 8192
#define ECP_MAX_FEAT (1 << 20)

typedef struct { int from, to; double inv_pose[12]; } ecp_edge;

struct ssko_ecp {
	int submap_size, min_obs_lc, max_tree_depth, max_optimize_depth;
	int n_kf, n_edges;
	ecp_edge* edges;
	/* m_kf_localmap_center_ID: per keyframe a SET of centre ids; m_localmap_kf_IDs: per centre a set of keyframes (present = is a centre) */
	unsigned char* center_of;      /* [kf][centre] membership, ECP_MAX_KF x ECP_MAX_KF bits would be large: store lists instead */
	int* kf_centers; int* kf_n_centers;        /* up to 8 centres per keyframe */
	unsigned char* is_center;
	int* lm_members; int* lm_n_members;        /* per centre up to 64 member keyframes */
	int* feat_base;                            /* landmark id -> base keyframe, -1 unknown */
	double init_pose[12];
};

static void pose_identity(double* T) { memset(T, 0, 12 * sizeof(double)); T[0] = T[5] = T[10] = 1.0; }
static void pose_inverse(const double* T, double* O)
{
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) O[4 * i + j] = T[4 * j + i];
		O[4 * i + 3] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
	}
}
static void pose_compose(const double* A, const double* B, double* O)   /* O = A (+) B */
{
	double R[12];
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) R[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
		R[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
	}
	memcpy(O, R, sizeof R);
}

ssko_ecp* ssko_ecp_create(int submap_size, int min_obs_to_loop_closure, int max_tree_depth, int max_optimize_depth)
{
	ssko_ecp* s = (ssko_ecp*)calloc(1, sizeof *s);
	s->submap_size = submap_size; s->min_obs_lc = min_obs_to_loop_closure; s->max_tree_depth = max_tree_depth; s->max_optimize_depth = max_optimize_depth;
	s->edges = (ecp_edge*)calloc(ECP_MAX_EDGE, sizeof(ecp_edge));
	s->kf_centers = (int*)calloc(ECP_MAX_KF * 8, sizeof(int)); s->kf_n_centers = (int*)calloc(ECP_MAX_KF, sizeof(int));
	s->is_center = (unsigned char*)calloc(ECP_MAX_KF, 1);
	s->lm_members = (int*)calloc(ECP_MAX_KF * 64, sizeof(int)); s->lm_n_members = (int*)calloc(ECP_MAX_KF, sizeof(int));
	s->feat_base = (int*)malloc(sizeof(int) * ECP_MAX_FEAT);
	for (int i = 0; i < ECP_MAX_FEAT; ++i) s->feat_base[i] = -1;
	pose_identity(s->init_pose);
	return s;
}
void ssko_ecp_destroy(ssko_ecp* s)
{
	if (!s) return;
	free(s->edges); free(s->kf_centers); free(s->kf_n_centers); free(s->is_center); free(s->lm_members); free(s->lm_n_members); free(s->feat_base); free(s);
}
void ssko_ecp_set_initial_kf_pose(ssko_ecp* s, const double* pose) { memcpy(s->init_pose, pose, sizeof s->init_pose); }   /* SRBA.h:95-97 */
void ssko_ecp_set_edge_pose(ssko_ecp* s, int e, const double* pose) { memcpy(s->edges[e].inv_pose, pose, 12 * sizeof(double)); }
int ssko_ecp_num_edges(const ssko_ecp* s) { return s->n_edges; }
int ssko_ecp_get_edge(const ssko_ecp* s, int e, int* from, int* to, double* pose)
{
	if (e < 0 || e >= s->n_edges) return -1;
	*from = s->edges[e].from; *to = s->edges[e].to; memcpy(pose, s->edges[e].inv_pose, 12 * sizeof(double));
	return 0;
}
/* std::set semantics: sorted ascending, no duplicates */
static void set_insert(int* v, int* n, int x)
{
	int i = 0;
	while (i < *n && v[i] < x) ++i;
	if (i < *n && v[i] == x) return;
	for (int k = *n; k > i; --k) v[k] = v[k - 1];
	v[i] = x; ++*n;
}
int ssko_ecp_kf_centers(const ssko_ecp* s, int kf, int* out) { for (int i = 0; i < s->kf_n_centers[kf]; ++i) out[i] = s->kf_centers[8 * kf + i]; return s->kf_n_centers[kf]; }
int ssko_ecp_localmap_kfs(const ssko_ecp* s, int c, int* out)
{
	if (!s->is_center[c]) return -1;
	for (int i = 0; i < s->lm_n_members[c]; ++i) out[i] = s->lm_members[64 * c + i];
	return s->lm_n_members[c];
}
static int is_localmap_center(const ssko_ecp* s, int kf) { return s->is_center[kf]; }                                   /* SRBA.h:100-102 */
static int get_localmap_center(const ssko_ecp* s, int kf) { return s->kf_centers[8 * kf + s->kf_n_centers[kf] - 1]; }   /* SRBA.h:105-108: *rbegin() */

static int create_kf2kf_edge(ssko_ecp* s, int from, int to)
{
	s->edges[s->n_edges].from = from; s->edges[s->n_edges].to = to;
	pose_identity(s->edges[s->n_edges].inv_pose);
	return s->n_edges++;
}
/* spanning_tree.sym.next_edge[a][b].distance, or -1 when b is not within max_tree_depth of a */
static int topo_distance(const ssko_ecp* s, int a, int b, int n_kf)
{
	int* dist = (int*)malloc(sizeof(int) * (size_t)n_kf);
	int* queue = (int*)malloc(sizeof(int) * (size_t)n_kf);
	for (int i = 0; i < n_kf; ++i) dist[i] = -1;
	int qh = 0, qt = 0, res = -1;
	dist[a] = 0; queue[qt++] = a;
	while (qh < qt) {
		const int u = queue[qh++];
		if (u == b) { res = dist[u]; break; }
		if (dist[u] >= s->max_tree_depth) continue;
		for (int e = 0; e < s->n_edges; ++e) {
			const int v = s->edges[e].from == u ? s->edges[e].to : (s->edges[e].to == u ? s->edges[e].from : -1);
			if (v >= 0 && v < n_kf && dist[v] < 0) { dist[v] = dist[u] + 1; queue[qt++] = v; }
		}
	}
	free(dist); free(queue);
	return res;
}

/* define_new_keyframe as far as the topology goes: the policy for keyframes >= 1, then the new landmarks.  Returns the number of
 * created edges (their ids are the last ones), -1 for "the node becomes isolated" (SRBA.h:304). */
int ssko_ecp_define_kf(ssko_ecp* s, const uint64_t* feat_ids, int n_obs)
{
	const int new_kf_id = s->n_kf;
	int created = 0;
	if (new_kf_id >= 1) {                                                          /* SRBA.h:150 */
		if (new_kf_id == 1) {                                                      /* :153-158 */
			s->kf_n_centers[0] = 0; set_insert(s->kf_centers, &s->kf_n_centers[0], 0);
			s->is_center[0] = 1; s->lm_n_members[0] = 0;
		}
		const int MINIMUM_OBS_TO_LOOP_CLOSURE = s->min_obs_lc, SUBMAP_SIZE = s->submap_size;            /* :166-167 */
		const int currentLocalmapBaseId = is_localmap_center(s, new_kf_id - 1) ? new_kf_id - 1 : get_localmap_center(s, new_kf_id - 1);   /* :170-173 */
		const int NUM_KFS_LOCALMAP = s->is_center[currentLocalmapBaseId] ? s->lm_n_members[currentLocalmapBaseId] : -1;               /* :176 */
		if (NUM_KFS_LOCALMAP >= 0 && NUM_KFS_LOCALMAP < SUBMAP_SIZE - 1) {          /* :178 (size_t comparison: -1 is huge) */
			s->kf_n_centers[new_kf_id] = 0; set_insert(s->kf_centers + 8 * new_kf_id, &s->kf_n_centers[new_kf_id], currentLocalmapBaseId);   /* :182-183 */
			const int id = create_kf2kf_edge(s, currentLocalmapBaseId, new_kf_id);   /* :186 */
			if (NUM_KFS_LOCALMAP == 0) memcpy(s->edges[id].inv_pose, s->init_pose, sizeof s->init_pose);                                     /* :194 */
			else {                                                                 /* :203 inverseComposeFrom(prev, initial) = initial^-1 (+) prev */
				double inv[12];
				pose_inverse(s->init_pose, inv);
				pose_compose(inv, s->edges[id - 1].inv_pose, s->edges[id].inv_pose);
			}
			++created;
			set_insert(s->lm_members + 64 * currentLocalmapBaseId, &s->lm_n_members[currentLocalmapBaseId], new_kf_id);                       /* :209 */
		} else {
			/* :220-221 observations per base keyframe of the already known landmarks, descending */
			int* per_base = (int*)calloc((size_t)new_kf_id + 1, sizeof(int));
			for (int o = 0; o < n_obs; ++o) if (feat_ids[o] < ECP_MAX_FEAT && s->feat_base[feat_ids[o]] >= 0) per_base[s->feat_base[feat_ids[o]]]++;
			/* :224-232 votes per area (the multimap is walked in descending count; the sums do not depend on the order) */
			int* per_area = (int*)calloc((size_t)new_kf_id + 1, sizeof(int));
			for (int b = 0; b < new_kf_id; ++b) if (per_base[b] > 0) per_area[get_localmap_center(s, b)] += per_base[b];
			/* :235-239 sorted by votes, descending; equal votes keep the order of insertion = ascending centre id */
			int n_area = 0;
			int* area = (int*)malloc(sizeof(int) * ((size_t)new_kf_id + 1));
			for (int c = 0; c < new_kf_id; ++c) if (per_area[c] > 0) area[n_area++] = c;
			for (int i = 1; i < n_area; ++i) {                                     /* stable insertion sort */
				const int c = area[i];
				int k = i;
				while (k > 0 && per_area[area[k - 1]] < per_area[c]) { area[k] = area[k - 1]; --k; }
				area[k] = c;
			}
			for (int i = 0; i < n_area; ++i) {                                      /* :242 */
				const int num_obs_this_base = per_area[area[i]], central_kf_id = area[i];
				const int d = topo_distance(s, new_kf_id, central_kf_id, new_kf_id + 1);   /* :253-269, -1 = infinity */
				if (d < 0 || d >= s->max_optimize_depth) {                          /* :271 */
					if (num_obs_this_base >= MINIMUM_OBS_TO_LOOP_CLOSURE) {         /* :273 */
						const int id = create_kf2kf_edge(s, central_kf_id, new_kf_id);   /* :278 */
						++created;
						if (s->kf_n_centers[new_kf_id] == 0 && created > 1) { /* unreachable: the first created edge always sets a centre */ }
						set_insert(s->kf_centers + 8 * new_kf_id, &s->kf_n_centers[new_kf_id], central_kf_id);   /* :284-290 (insert or push_back: the same set) */
						set_insert(s->lm_members + 64 * central_kf_id, &s->lm_n_members[central_kf_id], new_kf_id);   /* :293 */
						if (id == 0) pose_identity(s->edges[id].inv_pose);          /* :294 */
						else memcpy(s->edges[id].inv_pose, s->edges[id - 1].inv_pose, 12 * sizeof(double));
					}
				}
			}
			free(per_base); free(per_area); free(area);
			if (created < 1) return -1;                                             /* :304 */
			s->is_center[new_kf_id] = 1; s->lm_n_members[new_kf_id] = 0;            /* :305 */
		}
	}
	for (int o = 0; o < n_obs; ++o)                                                 /* add_observation: new ids become landmarks based here */
		if (feat_ids[o] < ECP_MAX_FEAT && s->feat_base[feat_ids[o]] < 0) s->feat_base[feat_ids[o]] = new_kf_id;
	s->n_kf++;
	return created;
}
