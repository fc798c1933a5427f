// Batched sparse ("BM25") text provider on the host cores -- the data contract of
// EmbeddingHandler.encode_sparse (app/core/embedding/embedding_handler.py:101-142), which the
// reference delegates, one text per call, to fastembed's Qdrant/bm25 (:41, :123; its own TODO
// :100 asks for batching).  This is the native twin of rag_application_amd/bm25.py -- same
// pipeline, same arithmetic, results identical to it (tests/test_host_logic.py) -- for texts of
// ASCII plus typographic punctuation; a text holding any other non-ASCII code point is flagged
// and the Python path (Unicode \w and lower-casing) handles it.  PARITY UNPINNED like bm25.py: fastembed cannot run offline.
//
//   lower-case -> non-word characters ([^A-Za-z0-9_]) become separators -> drop "_", English stop
//   words, tokens longer than 40 -> Snowball-English (Porter2) stem -> term id =
//   abs(int32(murmur3_x86_32(token, 0))) -> value = tf*(k+1) / (tf + k*(1 - b + b*len/avg_len))
//   in fp64; ids ascending; two tokens hashing to one id keep the larger value.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

const char* const STOPWORDS[] = {
    "i", "me", "my", "myself", "we", "our", "ours", "ourselves", "you", "you're", "you've", "you'll", "you'd", "your",
    "yours", "yourself", "yourselves", "he", "him", "his", "himself", "she", "she's", "her", "hers", "herself", "it",
    "it's", "its", "itself", "they", "them", "their", "theirs", "themselves", "what", "which", "who", "whom", "this",
    "that", "that'll", "these", "those", "am", "is", "are", "was", "were", "be", "been", "being", "have", "has", "had",
    "having", "do", "does", "did", "doing", "a", "an", "the", "and", "but", "if", "or", "because", "as", "until",
    "while", "of", "at", "by", "for", "with", "about", "against", "between", "into", "through", "during", "before",
    "after", "above", "below", "to", "from", "up", "down", "in", "out", "on", "off", "over", "under", "again",
    "further", "then", "once", "here", "there", "when", "where", "why", "how", "all", "any", "both", "each", "few",
    "more", "most", "other", "some", "such", "no", "nor", "not", "only", "own", "same", "so", "than", "too", "very",
    "s", "t", "can", "will", "just", "don", "don't", "should", "should've", "now", "d", "ll", "m", "o", "re", "ve",
    "y", "ain", "aren", "aren't", "couldn", "couldn't", "didn", "didn't", "doesn", "doesn't", "hadn", "hadn't", "hasn",
    "hasn't", "haven", "haven't", "isn", "isn't", "ma", "mightn", "mightn't", "mustn", "mustn't", "needn", "needn't",
    "shan", "shan't", "shouldn", "shouldn't", "wasn", "wasn't", "weren", "weren't", "won", "won't", "wouldn",
    "wouldn't"};

const std::unordered_set<std::string>& stopwords() {
  static const std::unordered_set<std::string> s(std::begin(STOPWORDS), std::end(STOPWORDS));
  return s;
}

uint32_t murmur3_x86_32(const uint8_t* data, size_t n, uint32_t seed) {
  const uint32_t c1 = 0xCC9E2D51u, c2 = 0x1B873593u;
  uint32_t h = seed;
  const size_t nb = n - (n & 3);
  for (size_t i = 0; i < nb; i += 4) {
    uint32_t k = (uint32_t)data[i] | ((uint32_t)data[i + 1] << 8) | ((uint32_t)data[i + 2] << 16) |
                 ((uint32_t)data[i + 3] << 24);
    k *= c1;
    k = (k << 15) | (k >> 17);
    k *= c2;
    h ^= k;
    h = (h << 13) | (h >> 19);
    h = h * 5 + 0xE6546B64u;
  }
  uint32_t k = 0;
  const size_t rem = n & 3;
  if (rem == 3) k ^= (uint32_t)data[nb + 2] << 16;
  if (rem >= 2) k ^= (uint32_t)data[nb + 1] << 8;
  if (rem >= 1) {
    k ^= data[nb];
    k *= c1;
    k = (k << 15) | (k >> 17);
    k *= c2;
    h ^= k;
  }
  h ^= (uint32_t)n;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

int32_t term_id(const std::string& tok) {
  const int64_t v = (int32_t)murmur3_x86_32((const uint8_t*)tok.data(), tok.size(), 0);
  return (int32_t)(v < 0 ? -v : v);   // abs(int32); |INT32_MIN| does not fit and wraps like numpy would not: see bm25.py
}

// ---- Porter2, statement for statement as bm25.py stem() ------------------------------------------
inline bool is_vowel(char c) { return c == 'a' || c == 'e' || c == 'i' || c == 'o' || c == 'u' || c == 'y'; }
inline bool ends(const std::string& w, const char* suf) {
  const size_t m = strlen(suf);
  return w.size() >= m && memcmp(w.data() + w.size() - m, suf, m) == 0;
}
inline bool starts(const std::string& w, const char* pre) {
  const size_t m = strlen(pre);
  return w.size() >= m && memcmp(w.data(), pre, m) == 0;
}
bool has_vowel(const std::string& w, size_t n) {   // any vowel in w[0:n)
  for (size_t i = 0; i < n && i < w.size(); ++i)
    if (is_vowel(w[i])) return true;
  return false;
}
bool short_syllable_end(const std::string& w) {
  const size_t n = w.size();
  if (n == 2) return is_vowel(w[0]) && !is_vowel(w[1]);
  if (n >= 3) {
    const char c = w[n - 1];
    return !is_vowel(w[n - 3]) && is_vowel(w[n - 2]) && !is_vowel(c) && c != 'w' && c != 'x' && c != 'Y';
  }
  return false;
}
void regions(const std::string& w, size_t& r1, size_t& r2) {
  r1 = w.size();
  bool special = false;
  for (const char* pre : {"gener", "commun", "arsen"})
    if (starts(w, pre)) {
      r1 = strlen(pre);
      special = true;
      break;
    }
  if (!special)
    for (size_t i = 1; i < w.size(); ++i)
      if (!is_vowel(w[i]) && is_vowel(w[i - 1])) {
        r1 = i + 1;
        break;
      }
  r2 = w.size();
  for (size_t i = r1 + 1; i < w.size(); ++i)
    if (!is_vowel(w[i]) && is_vowel(w[i - 1])) {
      r2 = i + 1;
      break;
    }
}
std::string unY(std::string w) {
  for (char& c : w)
    if (c == 'Y') c = 'y';
  return w;
}

std::string stem(const std::string& word) {
  static const std::unordered_map<std::string, std::string> EXC1 = {
      {"skis", "ski"}, {"skies", "sky"}, {"dying", "die"}, {"lying", "lie"}, {"tying", "tie"}, {"idly", "idl"},
      {"gently", "gentl"}, {"ugly", "ugli"}, {"early", "earli"}, {"only", "onli"}, {"singly", "singl"}, {"sky", "sky"},
      {"news", "news"}, {"howe", "howe"}, {"atlas", "atlas"}, {"cosmos", "cosmos"}, {"bias", "bias"}, {"andes", "andes"}};
  static const std::unordered_set<std::string> EXC2 = {"inning", "outing", "canning", "herring", "earring", "proceed",
                                                       "exceed", "succeed"};
  static const std::pair<const char*, const char*> STEP2[] = {
      {"ization", "ize"}, {"ational", "ate"}, {"fulness", "ful"}, {"ousness", "ous"}, {"iveness", "ive"},
      {"tional", "tion"}, {"biliti", "ble"}, {"lessli", "less"}, {"entli", "ent"}, {"ation", "ate"}, {"alism", "al"},
      {"aliti", "al"}, {"ousli", "ous"}, {"iviti", "ive"}, {"fulli", "ful"}, {"enci", "ence"}, {"anci", "ance"},
      {"abli", "able"}, {"izer", "ize"}, {"ator", "ate"}, {"alli", "al"}, {"bli", "ble"}, {"ogi", nullptr},
      {"li", nullptr}};
  static const std::pair<const char*, const char*> STEP3[] = {
      {"ational", "ate"}, {"tional", "tion"}, {"alize", "al"}, {"icate", "ic"}, {"iciti", "ic"}, {"ative", nullptr},
      {"ical", "ic"}, {"ness", ""}, {"ful", ""}};
  static const char* const STEP4[] = {"ement", "ance", "ence", "able", "ible", "ment", "ant", "ent", "ism", "ate",
                                      "iti", "ous", "ive", "ize", "ion", "al", "er", "ic"};
  static const char* const DOUBLES[] = {"bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt"};
  std::string w = word;
  if (w.size() <= 2) return w;
  auto e1 = EXC1.find(w);
  if (e1 != EXC1.end()) return e1->second;
  if (w[0] == '\'') w = w.substr(1);
  if (w.empty()) return w;
  if (w[0] == 'y') w[0] = 'Y';
  for (size_t i = 1; i < w.size(); ++i)
    if (w[i] == 'y' && is_vowel(w[i - 1])) w[i] = 'Y';   // (a 'Y' set earlier is not a vowel, as in bm25.py)
  size_t r1, r2;
  regions(w, r1, r2);
  // step 0
  for (const char* suf : {"'s'", "'s", "'"})
    if (ends(w, suf)) {
      w.resize(w.size() - strlen(suf));
      break;
    }
  // step 1a
  if (ends(w, "sses")) {
    w.resize(w.size() - 2);
  } else if (ends(w, "ied") || ends(w, "ies")) {
    w.resize(w.size() > 4 ? w.size() - 2 : w.size() - 1);
  } else if (ends(w, "us") || ends(w, "ss")) {
  } else if (ends(w, "s")) {
    if (w.size() >= 2 && has_vowel(w, w.size() - 2)) w.resize(w.size() - 1);
  }
  if (EXC2.count(w)) return unY(w);
  // step 1b
  bool done1b = false;
  for (const char* suf : {"eedly", "eed"})
    if (ends(w, suf)) {
      const size_t m = strlen(suf);
      if (w.size() - m >= r1) {
        w.resize(w.size() - m);
        w += "ee";
      }
      done1b = true;
      break;
    }
  if (!done1b)
    for (const char* suf : {"ingly", "edly", "ing", "ed"})
      if (ends(w, suf)) {
        const size_t m = strlen(suf);
        if (has_vowel(w, w.size() - m)) {
          w.resize(w.size() - m);
          bool dbl = false;
          for (const char* d : DOUBLES) dbl = dbl || ends(w, d);
          if (ends(w, "at") || ends(w, "bl") || ends(w, "iz")) w += "e";
          else if (dbl) w.resize(w.size() - 1);
          else if (short_syllable_end(w) && r1 >= w.size()) w += "e";
        }
        break;
      }
  // step 1c
  if (w.size() > 2 && (w.back() == 'y' || w.back() == 'Y') && !is_vowel(w[w.size() - 2])) w.back() = 'i';
  // step 2
  for (const auto& sr : STEP2)
    if (ends(w, sr.first)) {
      const size_t m = strlen(sr.first);
      if (w.size() - m >= r1) {
        if (!strcmp(sr.first, "ogi")) {
          if (ends(w, "logi")) w.resize(w.size() - 1);
        } else if (!strcmp(sr.first, "li")) {
          if (w.size() >= 3 && strchr("cdeghkmnrt", w[w.size() - 3])) w.resize(w.size() - 2);
        } else {
          w.resize(w.size() - m);
          w += sr.second;
        }
      }
      break;
    }
  // step 3
  for (const auto& sr : STEP3)
    if (ends(w, sr.first)) {
      const size_t m = strlen(sr.first);
      if (w.size() - m >= r1) {
        if (!strcmp(sr.first, "ative")) {
          if (w.size() - m >= r2) w.resize(w.size() - m);
        } else {
          w.resize(w.size() - m);
          w += sr.second;
        }
      }
      break;
    }
  // step 4
  for (const char* suf : STEP4)
    if (ends(w, suf)) {
      const size_t m = strlen(suf);
      if (w.size() - m >= r2) {
        if (!strcmp(suf, "ion")) {
          if (w.size() > 3 && (w[w.size() - 4] == 's' || w[w.size() - 4] == 't')) w.resize(w.size() - 3);
        } else {
          w.resize(w.size() - m);
        }
      }
      break;
    }
  // step 5
  if (ends(w, "e")) {
    const std::string w1 = w.substr(0, w.size() - 1);
    if (w.size() - 1 >= r2 || (w.size() - 1 >= r1 && !short_syllable_end(w1))) w.resize(w.size() - 1);
  } else if (ends(w, "l")) {
    if (w.size() - 1 >= r2 && w.size() > 1 && w[w.size() - 2] == 'l') w.resize(w.size() - 1);
  }
  return unY(w);
}

struct Row {
  std::vector<std::pair<int32_t, double>> terms;   // ascending term id
  bool fallback = false;
};

void embed_one(const char* text, int64_t len, double k, double b, double avg_len, Row& out) {
  out.terms.clear();
  out.fallback = false;
  // Non-ASCII: U+0080..U+00FF and U+2000..U+206F code points that Python's \w does NOT match
  // (typographic dashes, quotes, bullets, section / multiplication signs ...) are separators
  // like any ASCII punctuation; the bitmaps below were generated with
  //   re.match(r"\w", chr(cp)) is None     (CPython 3.10, unicodedata 13)
  // Anything else (letters that need Unicode lower-casing, other scripts, malformed UTF-8)
  // sends the whole text to the Python path.
  static const uint64_t NONWORD_LATIN1[2] = {0x89d3fbffffffffffull, 0x0080000000800000ull};   // U+0080 + bit
  auto nonword_cp = [](uint32_t cp) -> bool {
    if (cp >= 0x80 && cp < 0x100) return ((NONWORD_LATIN1[(cp - 0x80) >> 6] >> ((cp - 0x80) & 63)) & 1ull) != 0;
    return cp >= 0x2000 && cp < 0x2070;   // General Punctuation: none is a word character
  };
  for (int64_t i = 0; i < len;) {
    const unsigned char c = (unsigned char)text[i];
    if (c < 0x80) {
      ++i;
      continue;
    }
    uint32_t cp = 0;
    int nb = 0;
    if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; nb = 2; }
    else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; nb = 3; }
    bool ok = nb != 0 && i + nb <= len;
    for (int j = 1; ok && j < nb; ++j) {
      const unsigned char d = (unsigned char)text[i + j];
      ok = (d & 0xC0) == 0x80;
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok || !nonword_cp(cp)) {
      out.fallback = true;
      return;
    }
    i += nb;
  }
  const auto& stop = stopwords();
  std::unordered_map<std::string, int> tf;
  std::vector<std::string> order;   // first-seen order of the stems (as collections.Counter iterates)
  int64_t doc_len = 0;
  std::string tok;
  auto flush = [&]() {
    if (tok.empty()) return;
    const bool punct = tok.size() == 1 && tok[0] == '_';
    if (!punct && !stop.count(tok) && tok.size() <= 40) {
      const std::string s = stem(tok);
      if (!s.empty()) {
        ++doc_len;
        auto it = tf.find(s);
        if (it == tf.end()) {
          tf.emplace(s, 1);
          order.push_back(s);
        } else {
          ++it->second;
        }
      }
    }
    tok.clear();
  };
  for (int64_t i = 0; i < len; ++i) {
    unsigned char c = (unsigned char)text[i];
    if (c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
    const bool word = (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_';
    if (word) tok.push_back((char)c);
    else flush();   // ASCII punctuation / space, or a byte of a vetted non-word code point
  }
  flush();
  if (doc_len == 0) return;
  std::unordered_map<int32_t, double> acc;
  for (const std::string& s : order) {
    const double n = (double)tf[s];
    const double w = n * (k + 1.0) / (n + k * (1.0 - b + b * (double)doc_len / avg_len));
    const int32_t id = term_id(s);
    auto it = acc.find(id);
    if (it == acc.end()) acc.emplace(id, w > 0.0 ? w : 0.0);   // max(acc.get(id, 0.0), w)
    else if (w > it->second) it->second = w;
  }
  out.terms.assign(acc.begin(), acc.end());
  std::sort(out.terms.begin(), out.terms.end());
}

}  // namespace

extern "C" {

// texts[i] = UTF-8 bytes of text i (lens[i] bytes, no terminator needed).  Writes the CSR of the
// batch: indptr[n+1], idx / val (capacity `cap` entries; sum over i of lens[i]/2 + 1 always
// suffices) and flags[i] = 1 where text i holds non-ASCII bytes (its row is left empty: use the
// Python provider for it).  Returns 0, or 1 when `cap` is too small.  `threads` <= 0: one per core
// (at most 16).
int hx_bm25_embed_batch(const char* const* texts, const int64_t* lens, int64_t n, double k, double b, double avg_len,
                        int32_t threads, int64_t* indptr, int32_t* idx, double* val, int64_t cap, int32_t* flags) {
  if (n < 0 || !indptr || (n > 0 && (!texts || !lens || !flags))) return 1;
  std::vector<Row> rows((size_t)n);
  int nt = threads > 0 ? threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(n, 1));
  std::atomic<int64_t> next(0);
  auto work = [&]() {
    for (;;) {
      const int64_t i0 = next.fetch_add(64);
      if (i0 >= n) break;
      for (int64_t i = i0; i < std::min(n, i0 + 64); ++i) embed_one(texts[i], lens[i], k, b, avg_len, rows[(size_t)i]);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  int64_t o = 0;
  indptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const Row& r = rows[(size_t)i];
    flags[i] = r.fallback ? 1 : 0;
    if (o + (int64_t)r.terms.size() > cap) return 1;
    for (const auto& t : r.terms) {
      idx[o] = t.first;
      val[o] = t.second;
      ++o;
    }
    indptr[i + 1] = o;
  }
  return 0;
}

}  // extern "C"
