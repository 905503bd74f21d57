// CBoWManager.h -- the reference's bag-of-words manager (src/CBoWManager.h:44-111) with the same method names;
// DBoW2's BriefVocabulary / BriefDatabase are replaced by the device vocabulary tree and database of the C-ABI
// (ssk_bow_*, include/ssk_capi.h; kernels in csrc/ssk_bow.hip).  QueryResults / Result keep DBoW2's field names
// (Id, Score), which is all the estimator reads (EST.cpp:437-441, 1748-1826).
#pragma once
#include "CStereoSLAMKF.h"
#include "ssk_bow_vocab_io.h"

namespace ssk {

struct Result {                                            // DBoW2::Result
	unsigned int Id = 0;
	double Score = 0;
	Result() {}
	Result(unsigned int id, double s) : Id(id), Score(s) {}
};
typedef std::vector<Result> QueryResults;                  // DBoW2::QueryResults

class CBoWManager {
public:
	explicit CBoWManager(Context& ctx, int db_capacity = 16384, int max_words_per_kf = 2048)
		: m_ctx(ctx), m_capacity(db_capacity), m_max_words(max_words_per_kf) {}
	~CBoWManager() { ssk_bow_db_destroy(m_db); ssk_bow_vocab_destroy(m_voc); }
	CBoWManager(const CBoWManager&) = delete;
	CBoWManager& operator=(const CBoWManager&) = delete;

	// BOW.h:59-66: loadVocabularyFromConfigFile reads the file name from the ini ([SRBA_GENERAL] vocabulary) and calls
	// m_voc.load(name); m_db.setVocabulary(m_voc, true, 5).  The direct index (use_di) is never read by the reference.
	void loadVocabulary(const std::string& voc_filename) { setVocabulary(loadVocabFile(voc_filename)); }

	void setVocabulary(const VocabFile& v)
	{
		ssk_bow_db_destroy(m_db); m_db = nullptr;
		ssk_bow_vocab_destroy(m_voc); m_voc = nullptr;
		m_ctx.check(ssk_bow_vocab_create(m_ctx.get(), v.k, v.L, v.weighting, v.scoring, (int)v.node_id.size(), v.node_id.data(), v.parent_id.data(),
		                                 v.weight.data(), v.desc.data(), (int)v.word_node.size(), v.word_node.data(), &m_voc), "ssk_bow_vocab_create");
		m_ctx.check(ssk_bow_db_create(m_ctx.get(), m_voc, m_capacity, m_max_words, &m_db), "ssk_bow_db_create");
	}

	bool hasVocabulary() const { return m_voc != nullptr; }
	size_t size() const { return (size_t)ssk_bow_db_size(m_db); }

	// BOW.h:71-77: inserts the keyframe's left descriptors; the entry id is the keyframe's index (EST.cpp:118,919)
	inline unsigned int insertIntoDB(const CStereoSLAMKF& kf)
	{
		transform(kf);
		int id = -1;
		m_ctx.check(ssk_bow_db_add(m_ctx.get(), m_db, m_words.data(), m_values.data(), m_n, &id), "ssk_bow_db_add");
		return (unsigned int)id;
	}

	// BOW.h:84-90
	inline void queryDB(const CStereoSLAMKF& kf, QueryResults& ret, unsigned int num_results = 1)
	{
		transform(kf);
		int32_t ids[SSK_BOW_MAX_RESULTS];
		double scores[SSK_BOW_MAX_RESULTS];
		int n = 0;
		m_ctx.check(ssk_bow_db_query(m_ctx.get(), m_db, m_words.data(), m_values.data(), m_n, (int)num_results, -1, ids, scores, &n), "ssk_bow_db_query");
		ret.clear();
		for (int i = 0; i < n; ++i) ret.push_back(Result((unsigned int)ids[i], scores[i]));
	}

private:
	void transform(const CStereoSLAMKF& kf)
	{
		if (!m_voc) throw std::runtime_error("CBoWManager: no vocabulary loaded");
		const int n = kf.m_descriptors_left.rows;
		m_words.resize((size_t)n + 1); m_values.resize((size_t)n + 1);
		m_n = 0;
		m_ctx.check(ssk_bow_transform(m_ctx.get(), m_voc, kf.m_descriptors_left.data.data(), n, m_words.data(), m_values.data(), &m_n), "ssk_bow_transform");
	}
	Context& m_ctx;
	int m_capacity, m_max_words;
	ssk_bow_vocab* m_voc = nullptr;
	ssk_bow_db* m_db = nullptr;
	std::vector<int32_t> m_words;
	std::vector<double> m_values;
	int m_n = 0;
};

}  // namespace ssk
