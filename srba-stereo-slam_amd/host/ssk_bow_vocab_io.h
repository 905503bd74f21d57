// ssk_bow_vocab_io.h -- reader of the vocabulary file the reference loads with BriefVocabulary::load
// (src/CBoWManager.h:59-66; demo/voc.yml.gz): DBoW2's OpenCV-FileStorage YAML, plain or gzip.
//   vocabulary: { k, L, scoringType, weightingType,
//                 nodes: [ { nodeId, parentId, weight, descriptor:"<256 chars 0/1>" } ... ],   (root excluded)
//                 words: [ { wordId, nodeId } ... ] }
// The descriptor text is boost::dynamic_bitset's: the FIRST character is the HIGHEST bit.  Bit m + 8k of that bitset is
// bit m of descriptor byte k (m_change_structure_binary, BOW.h:95-107), so character c lands in byte (255-c)/8, bit (255-c)%8.
// Pure host code (no device); host/CBoWManager.h hands the arrays to ssk_bow_vocab_create.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace ssk {

struct VocabFile {
	int k = 0, L = 0, scoring = 0, weighting = 0;
	std::vector<int32_t> node_id, parent_id;      // file order
	std::vector<double> weight;
	std::vector<uint8_t> desc;                    // 32 bytes per record
	std::vector<int32_t> word_node;               // word id -> node id
};

namespace vocab_io {

inline std::string slurp(const std::string& path)
{
	gzFile f = gzopen(path.c_str(), "rb");        // reads plain files too
	if (!f) throw std::runtime_error("vocabulary file not found: " + path);
	std::string s;
	char buf[1 << 16];
	int n;
	while ((n = gzread(f, buf, sizeof buf)) > 0) s.append(buf, (size_t)n);
	gzclose(f);
	if (n < 0) throw std::runtime_error("vocabulary file unreadable: " + path);
	return s;
}

// value after `key` (e.g. "nodeId:") searched in [pos, end); returns npos when absent
inline size_t after(const std::string& s, const char* key, size_t pos, size_t end)
{
	const size_t p = s.find(key, pos);
	return p == std::string::npos || p >= end ? std::string::npos : p + std::strlen(key);
}

inline long header_int(const std::string& s, const char* key, size_t end)
{
	// keys of the header are at the start of a line ("   k: 8"); "L:" must not match "%YAML:1.0"
	size_t p = 0;
	while ((p = s.find(key, p)) != std::string::npos && p < end) {
		size_t b = p;
		while (b > 0 && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
		if (b == 0 || s[b - 1] == '\n') return std::strtol(s.c_str() + p + std::strlen(key), nullptr, 10);
		++p;
	}
	throw std::runtime_error(std::string("vocabulary file: missing header field ") + key);
}

}  // namespace vocab_io

inline VocabFile loadVocabFile(const std::string& path)
{
	using namespace vocab_io;
	const std::string s = slurp(path);
	const size_t p_nodes = s.find("nodes:"), p_words = s.find("words:");
	if (s.find("vocabulary:") == std::string::npos || p_nodes == std::string::npos)
		throw std::runtime_error("not a DBoW2 vocabulary file: " + path);
	const size_t nodes_end = p_words == std::string::npos ? s.size() : p_words;
	VocabFile v;
	v.k = (int)header_int(s, "k:", p_nodes);
	v.L = (int)header_int(s, "L:", p_nodes);
	v.scoring = (int)header_int(s, "scoringType:", p_nodes);
	v.weighting = (int)header_int(s, "weightingType:", p_nodes);
	size_t p = p_nodes;
	while (true) {
		const size_t a = after(s, "nodeId:", p, nodes_end);
		if (a == std::string::npos) break;
		const size_t b = after(s, "parentId:", a, nodes_end), c = after(s, "weight:", a, nodes_end), d = after(s, "descriptor:", a, nodes_end);
		if (b == std::string::npos || c == std::string::npos || d == std::string::npos) throw std::runtime_error("vocabulary file: truncated node record");
		const size_t q0 = s.find('"', d), q1 = q0 == std::string::npos ? q0 : s.find('"', q0 + 1);
		if (q1 == std::string::npos || q1 > nodes_end) throw std::runtime_error("vocabulary file: unterminated descriptor");
		if (q1 - q0 - 1 != 256) throw std::runtime_error("vocabulary file: descriptor is not 256 bits (BRIEF-256 / ORB expected)");
		v.node_id.push_back((int32_t)std::strtol(s.c_str() + a, nullptr, 10));
		v.parent_id.push_back((int32_t)std::strtol(s.c_str() + b, nullptr, 10));
		v.weight.push_back(std::strtod(s.c_str() + c, nullptr));
		uint8_t bytes[32] = {0};
		for (int ch = 0; ch < 256; ++ch) {
			const char t = s[q0 + 1 + ch];
			if (t != '0' && t != '1') throw std::runtime_error("vocabulary file: descriptor holds a character other than 0/1");
			if (t == '1') { const int bit = 255 - ch; bytes[bit >> 3] |= (uint8_t)(1u << (bit & 7)); }
		}
		v.desc.insert(v.desc.end(), bytes, bytes + 32);
		p = q1 + 1;
	}
	if (p_words != std::string::npos) {
		p = p_words;
		while (true) {
			const size_t a = after(s, "wordId:", p, s.size());
			if (a == std::string::npos) break;
			const size_t b = after(s, "nodeId:", a, s.size());
			if (b == std::string::npos) throw std::runtime_error("vocabulary file: truncated word record");
			const long w = std::strtol(s.c_str() + a, nullptr, 10), n = std::strtol(s.c_str() + b, nullptr, 10);
			if (w < 0 || w > (1 << 26)) throw std::runtime_error("vocabulary file: bad wordId");
			if ((size_t)w >= v.word_node.size()) v.word_node.resize((size_t)w + 1, 0);
			v.word_node[(size_t)w] = (int32_t)n;
			p = b;
		}
	}
	return v;
}

}  // namespace ssk
