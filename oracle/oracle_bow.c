/*
 * oracle_bow.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h): the bag-of-words place
 * recognition behind CBoWManager (BOW.h:59-109): BriefVocabulary::transform, BriefDatabase::add / ::query.
 *
 * The arithmetic lives in DBoW2 1.0.2 (README.md:13), absent from the image; restated from its published algorithm
 * (Galvez-Lopez & Tardos, "Bags of binary words for fast place recognition in image sequences", T-RO 2012; the
 * TemplatedVocabulary / TemplatedDatabase procedure) for the configuration the reference uses: a k-ary tree of
 * 256-bit BRIEF nodes loaded from the vocabulary file (demo/voc.yml.gz: k=8, L=5, weightingType 0 = TF_IDF,
 * scoringType 0 = L1_NORM), database with the inverted file, query by L1 score.  In-tree parts restated literally:
 * the descriptor bit order (bit m + 8k of the DBoW2 bitset = bit m of byte k, BOW.h:95-107), which descriptors are
 * used (all rows of m_descriptors_left, BOW.h:73,86) and the number of results asked for (4, EST.cpp:426).
 * "Bag-of-words contract":
 *   W1  vocabulary: nodes in file order (nodeId, parentId, weight, descriptor); children of a node are in file
 *       order; a node without children is a leaf (word); wordId <-> nodeId from the file's word list.
 *   W2  word of a descriptor: from the root, at every level the child with the smallest Hamming distance, the FIRST
 *       one on ties (strict <), until a leaf; weight = the leaf's stored weight (idf).
 *   W3  bag-of-words vector of a keyframe: per word the sum of the weights of its occurrences (added one occurrence
 *       at a time, in f64; occurrences of weight <= 0 are skipped), then L1 normalisation: norm = sum of |v| in
 *       ascending word order, v /= norm when norm > 0.  Stored sorted by word id.
 *   W4  database entry ids are 0, 1, 2 ... in order of insertion (= keyframe ids, EST.cpp:118,919).
 *   W5  L1 query: for each entry with >= 1 common word, s = sum over the common words in ascending word order of
 *       |q - d| - |q| - |d| (f64); entries with id >= max_id are skipped unless max_id == -1; results sorted by s
 *       ascending, ties by ascending entry id (std::sort leaves ties open; fixed here), cut to max_results (> 0),
 *       score = -s / 2.
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ssko_bow_vocab {
	int k, L, n_nodes, n_words;      /* n_nodes counts the root (node 0) */
	int32_t* first;                  /* [n_nodes + 1] CSR of children, in file order */
	int32_t* child;
	int32_t* word_of;                /* [n_nodes] word id or -1 */
	double* weight;                  /* [n_nodes] */
	double* word_weight;             /* [n_words] weight of the word's leaf */
	uint8_t* desc;                   /* [n_nodes][32] */
};

ssko_bow_vocab* ssko_bow_vocab_create(int k, int L, int n_rec, const int32_t* node_id, const int32_t* parent_id,
                                      const double* weight, const uint8_t* desc, int n_words, const int32_t* word_node)
{
	int max_id = 0;
	for (int i = 0; i < n_rec; ++i) { if (node_id[i] <= 0 || parent_id[i] < 0) return NULL; if (node_id[i] > max_id) max_id = node_id[i]; }
	const int n = max_id + 1;
	for (int i = 0; i < n_rec; ++i) if (parent_id[i] >= n) return NULL;
	ssko_bow_vocab* v = (ssko_bow_vocab*)calloc(1, sizeof *v);
	v->k = k; v->L = L; v->n_nodes = n; v->n_words = n_words;
	v->first = (int32_t*)calloc((size_t)n + 2, sizeof(int32_t));
	v->child = (int32_t*)calloc((size_t)(n_rec > 0 ? n_rec : 1), sizeof(int32_t));
	v->word_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
	v->weight = (double*)calloc((size_t)n, sizeof(double));
	v->desc = (uint8_t*)calloc((size_t)n, 32);
	for (int i = 0; i < n; ++i) v->word_of[i] = -1;
	for (int i = 0; i < n_rec; ++i) v->first[parent_id[i] + 2]++;             /* counting sort by parent keeps file order (W1) */
	for (int i = 0; i < n; ++i) v->first[i + 2] += v->first[i + 1];
	for (int i = 0; i < n_rec; ++i) {
		v->child[v->first[parent_id[i] + 1]++] = node_id[i];
		v->weight[node_id[i]] = weight[i];
		memcpy(v->desc + 32 * (size_t)node_id[i], desc + 32 * (size_t)i, 32);
	}
	v->word_weight = (double*)calloc((size_t)(n_words > 0 ? n_words : 1), sizeof(double));
	for (int w = 0; w < n_words; ++w) if (word_node[w] > 0 && word_node[w] < n) { v->word_of[word_node[w]] = w; v->word_weight[w] = v->weight[word_node[w]]; }
	return v;
}

void ssko_bow_vocab_destroy(ssko_bow_vocab* v)
{
	if (!v) return;
	free(v->first); free(v->child); free(v->word_of); free(v->weight); free(v->word_weight); free(v->desc); free(v);
}

static int hamming256(const uint8_t* a, const uint8_t* b)
{
	int d = 0;
	for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
	return d;
}

/* W2 */
int ssko_bow_word(const ssko_bow_vocab* v, const uint8_t* feat, double* weight)
{
	int node = 0;
	if (v->first[1] == v->first[0]) { if (weight) *weight = 0; return -1; }
	do {
		const int c0 = v->first[node], c1 = v->first[node + 1];
		int best = v->child[c0], best_d = hamming256(feat, v->desc + 32 * (size_t)best);
		for (int c = c0 + 1; c < c1; ++c) {
			const int id = v->child[c], d = hamming256(feat, v->desc + 32 * (size_t)id);
			if (d < best_d) { best_d = d; best = id; }
		}
		node = best;
	} while (v->first[node + 1] > v->first[node]);
	if (weight) *weight = v->weight[node];
	return v->word_of[node];
}

static int cmp_i32(const void* a, const void* b) { const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

/* W3; out arrays have capacity n; returns the number of distinct words */
int ssko_bow_transform(const ssko_bow_vocab* v, const uint8_t* feats, int n, int32_t* out_words, double* out_values)
{
	int32_t* w = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
	int m = 0;
	for (int i = 0; i < n; ++i) {
		double wt;
		const int id = ssko_bow_word(v, feats + 32 * (size_t)i, &wt);
		if (id >= 0 && wt > 0) w[m++] = id;                       /* "if (w > 0) v.addWeight(id, w)" */
	}
	qsort(w, (size_t)m, sizeof(int32_t), cmp_i32);
	int nw = 0;
	for (int i = 0; i < m;) {
		const double wt = v->word_weight[w[i]];
		double acc = wt;                                         /* first occurrence inserts, the others add */
		int j = i + 1;
		for (; j < m && w[j] == w[i]; ++j) acc += wt;
		out_words[nw] = w[i]; out_values[nw] = acc; ++nw;
		i = j;
	}
	double norm = 0;
	for (int i = 0; i < nw; ++i) norm += fabs(out_values[i]);
	if (norm > 0) for (int i = 0; i < nw; ++i) out_values[i] /= norm;
	free(w);
	return nw;
}

/* ---- database: the inverted file of TemplatedDatabase (one row per word, entries in insertion order) ---- */
typedef struct if_row { int32_t n, cap; int32_t* entry; double* value; } if_row;
struct ssko_bow_db { int n_words, n_entries; if_row* row; };

ssko_bow_db* ssko_bow_db_create(int n_words)
{
	ssko_bow_db* d = (ssko_bow_db*)calloc(1, sizeof *d);
	d->n_words = n_words;
	d->row = (if_row*)calloc((size_t)(n_words > 0 ? n_words : 1), sizeof(if_row));
	return d;
}

void ssko_bow_db_destroy(ssko_bow_db* d)
{
	if (!d) return;
	for (int w = 0; w < d->n_words; ++w) { free(d->row[w].entry); free(d->row[w].value); }
	free(d->row); free(d);
}

int ssko_bow_db_size(const ssko_bow_db* d) { return d->n_entries; }

/* W4; returns the entry id, -1 on a word id outside the vocabulary */
int ssko_bow_db_add(ssko_bow_db* d, const int32_t* words, const double* values, int n)
{
	for (int i = 0; i < n; ++i) if (words[i] < 0 || words[i] >= d->n_words) return -1;
	const int id = d->n_entries++;
	for (int i = 0; i < n; ++i) {
		if_row* r = d->row + words[i];
		if (r->n == r->cap) {
			r->cap = r->cap ? 2 * r->cap : 4;
			r->entry = (int32_t*)realloc(r->entry, sizeof(int32_t) * (size_t)r->cap);
			r->value = (double*)realloc(r->value, sizeof(double) * (size_t)r->cap);
		}
		r->entry[r->n] = id; r->value[r->n] = values[i]; r->n++;
	}
	return id;
}

typedef struct res_t { double s; int32_t id; } res_t;
static int cmp_res(const void* a, const void* b)
{
	const res_t* x = (const res_t*)a; const res_t* y = (const res_t*)b;
	if (x->s != y->s) return x->s < y->s ? -1 : 1;
	return (x->id > y->id) - (x->id < y->id);
}

/* W5; out arrays have capacity max_results (or n_entries when max_results <= 0); returns the number of results */
int ssko_bow_db_query(const ssko_bow_db* d, const int32_t* words, const double* values, int n, int max_results, int max_id,
                      int32_t* out_ids, double* out_scores)
{
	const int ne = d->n_entries;
	double* s = (double*)calloc((size_t)(ne > 0 ? ne : 1), sizeof(double));
	unsigned char* hit = (unsigned char*)calloc((size_t)(ne > 0 ? ne : 1), 1);
	for (int i = 0; i < n; ++i) {
		if (words[i] < 0 || words[i] >= d->n_words) continue;
		const if_row* r = d->row + words[i];
		const double q = values[i];
		for (int j = 0; j < r->n; ++j) {
			const int e = r->entry[j];
			if (!(e < max_id || max_id == -1)) continue;
			const double dv = r->value[j];
			const double val = fabs(q - dv) - fabs(q) - fabs(dv);
			if (hit[e]) s[e] += val; else { s[e] = val; hit[e] = 1; }
		}
	}
	res_t* res = (res_t*)malloc(sizeof(res_t) * (size_t)(ne > 0 ? ne : 1));
	int nr = 0;
	for (int e = 0; e < ne; ++e) if (hit[e]) { res[nr].s = s[e]; res[nr].id = e; ++nr; }
	qsort(res, (size_t)nr, sizeof(res_t), cmp_res);
	if (max_results > 0 && nr > max_results) nr = max_results;
	for (int i = 0; i < nr; ++i) { out_ids[i] = res[i].id; out_scores[i] = -res[i].s / 2.0; }
	free(s); free(hit); free(res);
	return nr;
}
