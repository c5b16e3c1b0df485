// Native index build: everything MatchMaker.__init__ derives from the two title
// collections (doppelspeller/match_maker.py:84-181, common.py:145-151), computed from the transformed titles
// themselves instead of from per-title Python sets:
//
//   get_n_grams            common.py:150-151      distinct substrings of n characters of a title
//   get_n_grams_counter    common.py:145-147      document frequency of every n-gram (data and truth)
//   _get_idf_s_mapping     match_maker.py:135-142 idf = log(N_truth / df_truth) in float64; n-grams unseen in the truth
//                                                 set get the largest idf (`_get_encoding_values`, :149-153)
//   _get_encoding_mappings match_maker.py:144-147 n-gram -> column id
//   _construct_*_matrix    match_maker.py:155-178 float32 matrix entries; sums_matrix_truth[t] = float32 sum of the
//                                                 title's idf values; explicit zeros vanish in `.nonzero()` (:118,:128)
//   get_closest_matches    match_maker.py:196-197 ascending column ids per query row; max_intersection_possible =
//                                                 float64 sum of the idf values in that order
//
// Host code only (no kernel: the build is a handful of counting passes over ~22 n-grams per title; the time goes into
// the tiling + upload of ds_index_create).  Two orders the reference leaves to Python's set iteration (hash seed
// dependent) are fixed here: column ids ascend with the n-gram's byte string, and a title's idf values are
// summed in the order in which its n-grams first occur in the title.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ds_common.h"

struct ds_problem {
    int32_t n_gram = 0;
    int64_t n_truth = 0, n_queries = 0, n_columns = 0;
    std::vector<uint32_t> vocabulary;  // [V] n-gram key of every column (big-endian bytes of the n-gram), ascending
    std::vector<float> idf32;          // [V]
    std::vector<double> idf64;         // [V]
    std::vector<int64_t> rowptr;       // [V + 1] truth inverted index
    std::vector<int32_t> truth_idx;    // [nnz]
    std::vector<float> sums32;         // [N]
    std::vector<int64_t> q_rowptr;     // [Q + 1]
    std::vector<int32_t> q_cols;       // [q_nnz] ascending per row
    std::vector<double> q_maxint;      // [Q]
};

namespace {

// Distinct n-gram keys of one title in order of first occurrence (a title has at most 255 - n + 1 of them).
inline int title_keys(const uint8_t *chars, int64_t length, int n, uint32_t *keys)
{
    int count = 0;
    for (int64_t i = 0; i + n <= length; ++i) {
        uint32_t key = 0;
        for (int c = 0; c < n; ++c) key = (key << 8) | chars[i + c];
        bool seen = false;
        for (int j = 0; j < count && !seen; ++j) seen = keys[j] == key;
        if (!seen) keys[count++] = key;
    }
    return count;
}

}  // namespace

extern "C" {

int ds_problem_create(const uint8_t *truth_chars, const int64_t *truth_offsets, int64_t n_truth,
                      const uint8_t *query_chars, const int64_t *query_offsets, int64_t n_queries, int32_t n_gram,
                      ds_problem **out)
{
    DS_REQUIRE(out != nullptr, "ds_problem_create: out is null");
    *out = nullptr;
    DS_REQUIRE(n_gram >= 1 && n_gram <= 3, "ds_problem_create: n_gram=%d outside 1..3", n_gram);
    DS_REQUIRE(n_truth > 0 && n_queries >= 0, "ds_problem_create: bad counts");
    DS_REQUIRE(truth_offsets && (n_queries == 0 || query_offsets), "ds_problem_create: null offsets");
    const size_t table = size_t(1) << (8 * n_gram);
    constexpr int64_t kMaxTitle = 4096;
    for (int64_t t = 0; t < n_truth; ++t)
        DS_REQUIRE(truth_offsets[t + 1] >= truth_offsets[t] && truth_offsets[t + 1] - truth_offsets[t] <= kMaxTitle,
                   "ds_problem_create: bad truth title %lld", (long long)t);
    for (int64_t q = 0; q < n_queries; ++q)
        DS_REQUIRE(query_offsets[q + 1] >= query_offsets[q] && query_offsets[q + 1] - query_offsets[q] <= kMaxTitle,
                   "ds_problem_create: bad query title %lld", (long long)q);

    ds_problem *problem = new ds_problem();
    problem->n_gram = n_gram;
    problem->n_truth = n_truth;
    problem->n_queries = n_queries;
    std::vector<uint32_t> keys(kMaxTitle);

    // document frequencies (get_n_grams_counter): truth counts, data presence
    std::vector<int32_t> df_truth(table, 0);
    std::vector<uint8_t> present(table, 0);
    int64_t truth_entries = 0, query_entries = 0;
    for (int64_t t = 0; t < n_truth; ++t) {
        const int count = title_keys(truth_chars + truth_offsets[t], truth_offsets[t + 1] - truth_offsets[t], n_gram,
                                     keys.data());
        for (int j = 0; j < count; ++j) {
            ++df_truth[keys[j]];
            present[keys[j]] = 1;
        }
        truth_entries += count;
    }
    for (int64_t q = 0; q < n_queries; ++q) {
        const int count = title_keys(query_chars + query_offsets[q], query_offsets[q + 1] - query_offsets[q], n_gram,
                                     keys.data());
        for (int j = 0; j < count; ++j) present[keys[j]] = 1;
        query_entries += count;
    }

    // vocabulary (_get_encoding_mappings) in ascending key order, idf (_get_idf_s_mapping / _get_idf_given_index)
    std::vector<int32_t> column_of(table, -1);
    for (size_t key = 0; key < table; ++key) {
        if (!present[key]) continue;
        column_of[key] = static_cast<int32_t>(problem->vocabulary.size());
        problem->vocabulary.push_back(static_cast<uint32_t>(key));
    }
    const int64_t V = static_cast<int64_t>(problem->vocabulary.size());
    problem->n_columns = V;
    problem->idf64.assign(static_cast<size_t>(V), 0.0);
    double max_idf = -INFINITY;
    for (int64_t g = 0; g < V; ++g) {
        const int32_t df = df_truth[problem->vocabulary[static_cast<size_t>(g)]];
        if (df > 0) {
            const double value = std::log(static_cast<double>(n_truth) / static_cast<double>(df));  // :139
            problem->idf64[static_cast<size_t>(g)] = value;
            max_idf = std::max(max_idf, value);
        }
    }
    for (int64_t g = 0; g < V; ++g)
        if (df_truth[problem->vocabulary[static_cast<size_t>(g)]] == 0) problem->idf64[static_cast<size_t>(g)] = max_idf;
    problem->idf32.resize(static_cast<size_t>(V));
    for (int64_t g = 0; g < V; ++g)
        problem->idf32[static_cast<size_t>(g)] = static_cast<float>(problem->idf64[static_cast<size_t>(g)]);

    // truth inverted index: counting sort by column, rows ascending inside a column; zero entries are not stored
    problem->rowptr.assign(static_cast<size_t>(V) + 1, 0);
    for (int64_t g = 0; g < V; ++g) {
        const int32_t df = df_truth[problem->vocabulary[static_cast<size_t>(g)]];
        problem->rowptr[static_cast<size_t>(g) + 1] =
            problem->rowptr[static_cast<size_t>(g)] + (problem->idf32[static_cast<size_t>(g)] != 0.f ? df : 0);
    }
    problem->truth_idx.resize(static_cast<size_t>(problem->rowptr[static_cast<size_t>(V)]));
    problem->sums32.assign(static_cast<size_t>(n_truth), 0.f);
    {
        std::vector<int64_t> cursor(problem->rowptr.begin(), problem->rowptr.end() - 1);
        for (int64_t t = 0; t < n_truth; ++t) {
            const int count = title_keys(truth_chars + truth_offsets[t], truth_offsets[t + 1] - truth_offsets[t],
                                         n_gram, keys.data());
            float sum = 0.f;  // sum(uniqueness_values), float32, one addition per n-gram (:174)
            for (int j = 0; j < count; ++j) {
                const int32_t g = column_of[keys[j]];
                const float value = problem->idf32[static_cast<size_t>(g)];
                sum = sum + value;
                if (value != 0.f) problem->truth_idx[static_cast<size_t>(cursor[static_cast<size_t>(g)]++)] = static_cast<int32_t>(t);
            }
            problem->sums32[static_cast<size_t>(t)] = sum;
        }
    }

    // query rows: ascending non-zero column ids, max_intersection_possible in that order (float64)
    problem->q_rowptr.assign(static_cast<size_t>(n_queries) + 1, 0);
    problem->q_cols.reserve(static_cast<size_t>(query_entries));
    problem->q_maxint.assign(static_cast<size_t>(n_queries), 0.0);
    std::vector<int32_t> row;
    for (int64_t q = 0; q < n_queries; ++q) {
        const int count = title_keys(query_chars + query_offsets[q], query_offsets[q + 1] - query_offsets[q], n_gram,
                                     keys.data());
        row.clear();
        for (int j = 0; j < count; ++j) {
            const int32_t g = column_of[keys[j]];
            if (problem->idf32[static_cast<size_t>(g)] != 0.f) row.push_back(g);
        }
        std::sort(row.begin(), row.end());
        double total = 0.0;
        for (int32_t g : row) total = total + problem->idf64[static_cast<size_t>(g)];
        problem->q_maxint[static_cast<size_t>(q)] = total;
        problem->q_cols.insert(problem->q_cols.end(), row.begin(), row.end());
        problem->q_rowptr[static_cast<size_t>(q) + 1] = static_cast<int64_t>(problem->q_cols.size());
    }
    (void)truth_entries;
    *out = problem;
    return DS_OK;
}

void ds_problem_destroy(ds_problem *problem) { delete problem; }

int ds_problem_info(const ds_problem *problem, int64_t info[8])
{
    DS_REQUIRE(problem && info, "ds_problem_info: null argument");
    info[0] = problem->n_truth;
    info[1] = problem->n_queries;
    info[2] = problem->n_columns;
    info[3] = static_cast<int64_t>(problem->truth_idx.size());
    info[4] = static_cast<int64_t>(problem->q_cols.size());
    info[5] = problem->n_gram;
    info[6] = info[7] = 0;
    return DS_OK;
}

int ds_problem_arrays(const ds_problem *problem, const uint32_t **vocabulary, const float **idf32, const double **idf64,
                      const int64_t **rowptr, const int32_t **truth_idx, const float **sums32, const int64_t **q_rowptr,
                      const int32_t **q_cols, const double **q_maxint)
{
    DS_REQUIRE(problem != nullptr, "ds_problem_arrays: null problem");
    if (vocabulary) *vocabulary = problem->vocabulary.data();
    if (idf32) *idf32 = problem->idf32.data();
    if (idf64) *idf64 = problem->idf64.data();
    if (rowptr) *rowptr = problem->rowptr.data();
    if (truth_idx) *truth_idx = problem->truth_idx.data();
    if (sums32) *sums32 = problem->sums32.data();
    if (q_rowptr) *q_rowptr = problem->q_rowptr.data();
    if (q_cols) *q_cols = problem->q_cols.data();
    if (q_maxint) *q_maxint = problem->q_maxint.data();
    return DS_OK;
}

}  // extern "C"

// ---- transform_title (doppelspeller/common.py:20-47) for a batch of titles ----------------------------------------
// The caller has already applied the Unicode part (NFD normalisation, ASCII encoding with 'ignore': Python's
// unicodedata, identity for ASCII titles); what remains is byte work: lower case, '-' -> ' ', keep [a-zA-Z0-9\s],
// collapse runs of ' ' (other white space is kept as it is, as SUBSTITUTE_REGEX = ' +' does), strip, cut to
// max_characters and strip again, left-pad with '0' to n_gram characters when the title was shorter than that.
extern "C" int ds_transform_titles(const uint8_t *chars, const int64_t *offsets, int64_t n, int32_t max_characters,
                                   int32_t n_gram, uint8_t *out_chars, int64_t *out_offsets)
{
    DS_REQUIRE(offsets && out_offsets && (n == 0 || (chars && out_chars)), "ds_transform_titles: null pointer");
    DS_REQUIRE(max_characters >= 1 && n_gram >= 0 && n_gram <= max_characters, "ds_transform_titles: bad limits");
    auto is_space = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); };  // str.isspace, ASCII
    std::vector<uint8_t> text;
    int64_t write = 0;
    out_offsets[0] = 0;
    for (int64_t t = 0; t < n; ++t) {
        DS_REQUIRE(offsets[t + 1] >= offsets[t], "ds_transform_titles: bad offsets at %lld", (long long)t);
        text.clear();
        for (int64_t i = offsets[t]; i < offsets[t + 1]; ++i) {
            uint8_t c = chars[i];
            DS_REQUIRE(c < 128, "ds_transform_titles: title %lld is not ASCII (apply the Unicode step first)", (long long)t);
            if (c >= 'A' && c <= 'Z') c = static_cast<uint8_t>(c + 32);
            if (c == '-') c = ' ';
            const bool keep = (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || is_space(c);
            if (!keep) continue;
            if (c == ' ' && !text.empty() && text.back() == ' ') continue;  // ' +' -> ' '
            text.push_back(c);
        }
        size_t first = 0, last = text.size();
        while (first < last && is_space(text[first])) ++first;
        while (last > first && is_space(text[last - 1])) --last;
        const size_t number_of_characters = last - first;                  // :30
        if (last - first > static_cast<size_t>(max_characters)) last = first + static_cast<size_t>(max_characters);
        while (first < last && is_space(text[first])) ++first;             // .strip() after the cut (:31)
        while (last > first && is_space(text[last - 1])) --last;
        if (number_of_characters < static_cast<size_t>(n_gram))            // :33-37 rjust(n_gram, '0')
            for (size_t pad = last - first; pad < static_cast<size_t>(n_gram); ++pad) out_chars[write++] = '0';
        for (size_t i = first; i < last; ++i) out_chars[write++] = text[i];
        out_offsets[t + 1] = write;
    }
    return DS_OK;
}
