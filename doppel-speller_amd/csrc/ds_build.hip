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
#include "ds_host.h"

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

constexpr int64_t kMaxTitle = 4096;

// The n-grams of the two collections live in a DENSE key space: the bytes that occur at all (37 after
// transform_title: [a-z0-9 ]) are numbered in ascending byte order, an n-gram's key is its base-`symbols` number.
// Dense keys ascend exactly as the byte strings do, and the per-thread count tables of the threaded passes stay in
// the L2 cache (37^3 = 50,653 entries instead of 2^24).
struct KeySpace {
    int n = 0, symbols = 0;
    size_t size = 0;
    uint16_t dense[256];
    uint8_t byte_of[256];

    uint32_t bytes_of_key(uint32_t key) const  // big-endian bytes of the n-gram in a uint32 (ds_problem_arrays)
    {
        uint32_t digits[3] = {0, 0, 0};
        for (int c = n - 1; c >= 0; --c) {
            digits[c] = key % static_cast<uint32_t>(symbols);
            key /= static_cast<uint32_t>(symbols);
        }
        uint32_t out = 0;
        for (int c = 0; c < n; ++c) out = (out << 8) | byte_of[digits[c]];
        return out;
    }
};

// Distinct n-gram keys of one title in order of first occurrence.  `stamp` (one entry per key, private to the calling
// thread) holds the last title that produced the key: membership in O(1) instead of a scan of the keys so far.
struct TitleKeys {
    const KeySpace &space;
    std::vector<uint32_t> stamp;
    uint32_t current = 0;
    uint32_t keys[kMaxTitle];

    explicit TitleKeys(const KeySpace &key_space) : space(key_space), stamp(key_space.size, 0u) {}

    int collect(const uint8_t *chars, int64_t length)
    {
        if (++current == 0) {  // the stamp counter wrapped: forget everything
            std::fill(stamp.begin(), stamp.end(), 0u);
            current = 1;
        }
        const int n = space.n;
        const uint32_t symbols = static_cast<uint32_t>(space.symbols);
        int count = 0;
        for (int64_t i = 0; i + n <= length; ++i) {
            uint32_t key = 0;
            for (int c = 0; c < n; ++c) key = key * symbols + space.dense[chars[i + c]];
            if (stamp[key] == current) continue;
            stamp[key] = current;
            keys[count++] = key;
        }
        return count;
    }
};

}  // namespace

extern "C" {

// Threaded (ds::host_threads()): the titles are cut into one contiguous range per thread; a range counts its n-grams
// in a private table, the tables are combined into per-(thread, column) write positions, and the same ranges then fill
// the inverted index -- thread t's rows precede thread t + 1's in every column, so posting lists come out ascending
// whatever the number of threads.
int ds_problem_create(const uint8_t *truth_chars, const int64_t *truth_offsets, int64_t n_truth,
                      const uint8_t *query_chars, const int64_t *query_offsets, int64_t n_queries, int32_t n_gram,
                      ds_problem **out)
{
    DS_REQUIRE(out != nullptr, "ds_problem_create: out is null");
    *out = nullptr;
    DS_REQUIRE(n_gram >= 1 && n_gram <= 3, "ds_problem_create: n_gram=%d outside 1..3", n_gram);
    DS_REQUIRE(n_truth > 0 && n_queries >= 0, "ds_problem_create: bad counts");
    DS_REQUIRE(n_truth < (int64_t(1) << 31), "ds_problem_create: more than 2^31 truth titles");
    DS_REQUIRE(truth_offsets && (n_queries == 0 || query_offsets), "ds_problem_create: null offsets");
    const int threads = ds::host_threads();
    ds::FirstError error;
    auto validate = [&](const int64_t *offsets, int64_t count, const char *what) {
        ds::parallel_ranges(count, threads, [&](int, int64_t begin, int64_t end) {
            for (int64_t t = begin; t < end; ++t)
                if (!(offsets[t + 1] >= offsets[t] && offsets[t + 1] - offsets[t] <= kMaxTitle)) {
                    error.raise("ds_problem_create: bad %s title %lld", what, (long long)t);
                    return;
                }
        });
    };
    validate(truth_offsets, n_truth, "truth");
    if (!error.failed()) validate(query_offsets, n_queries, "query");
    DS_REQUIRE(!error.failed(), "%s", error.message());
    DS_REQUIRE(truth_offsets[n_truth] == truth_offsets[0] || truth_chars, "ds_problem_create: null truth characters");
    DS_REQUIRE(n_queries == 0 || query_offsets[n_queries] == query_offsets[0] || query_chars,
               "ds_problem_create: null query characters");

    // ---- the alphabet: which bytes occur
    KeySpace space;
    space.n = n_gram;
    {
        std::vector<uint8_t> seen(static_cast<size_t>(threads) * 256, 0);
        auto scan = [&](const uint8_t *chars, int64_t first, int64_t last) {
            ds::parallel_ranges(last - first, threads, [&](int thread, int64_t begin, int64_t end) {
                uint8_t *mine = seen.data() + static_cast<size_t>(thread) * 256;
                for (int64_t i = first + begin; i < first + end; ++i) mine[chars[i]] = 1;
            });
        };
        scan(truth_chars, truth_offsets[0], truth_offsets[n_truth]);
        if (n_queries > 0) scan(query_chars, query_offsets[0], query_offsets[n_queries]);
        for (int byte = 0; byte < 256; ++byte) {
            bool any = false;
            for (int t = 0; t < threads; ++t) any = any || seen[static_cast<size_t>(t) * 256 + byte];
            space.dense[byte] = static_cast<uint16_t>(space.symbols);
            if (any) space.byte_of[space.symbols++] = static_cast<uint8_t>(byte);
        }
        if (space.symbols == 0) space.byte_of[space.symbols++] = 0;
        space.size = 1;
        for (int c = 0; c < n_gram; ++c) space.size *= static_cast<size_t>(space.symbols);
    }
    const size_t table = space.size;
    // private tables: 9 bytes per (worker, key) -- int64 counts / write positions + one presence byte -- capped at 2.25 GiB
    // in all (a binary alphabet of 256 symbols has 2^24 tri-grams: 16 workers; the 37-symbol alphabet of transform_title
    // has 50,653: every thread gets its tables)
    const int workers = static_cast<int>(std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(threads),
                                                                              (size_t(1) << 28) / std::max<size_t>(table, 1))));

    ds_problem *problem = new ds_problem();
    problem->n_gram = n_gram;
    problem->n_truth = n_truth;
    problem->n_queries = n_queries;

    // ---- document frequencies (get_n_grams_counter): truth counts per thread range, data presence
    std::vector<int64_t> counts(static_cast<size_t>(workers) * table, 0);   // [worker][key], later: write positions
    std::vector<uint8_t> present(table, 0);
    {
        std::vector<uint8_t> present_by(static_cast<size_t>(workers) * table, 0);
        ds::parallel_ranges(n_truth, workers, [&](int worker, int64_t begin, int64_t end) {
            TitleKeys title(space);
            int64_t *mine = counts.data() + static_cast<size_t>(worker) * table;
            for (int64_t t = begin; t < end; ++t) {
                const int count = title.collect(truth_chars + truth_offsets[t], truth_offsets[t + 1] - truth_offsets[t]);
                for (int j = 0; j < count; ++j) ++mine[title.keys[j]];
            }
        });
        ds::parallel_ranges(n_queries, workers, [&](int worker, int64_t begin, int64_t end) {
            TitleKeys title(space);
            uint8_t *mine = present_by.data() + static_cast<size_t>(worker) * table;
            for (int64_t q = begin; q < end; ++q) {
                const int count = title.collect(query_chars + query_offsets[q], query_offsets[q + 1] - query_offsets[q]);
                for (int j = 0; j < count; ++j) mine[title.keys[j]] = 1;
            }
        });
        for (size_t key = 0; key < table; ++key) {
            bool any = false;
            for (int w = 0; w < workers && !any; ++w)
                any = present_by[static_cast<size_t>(w) * table + key] || counts[static_cast<size_t>(w) * table + key] > 0;
            present[key] = any;
        }
    }

    // ---- vocabulary (_get_encoding_mappings) in ascending key order, idf (_get_idf_s_mapping / _get_idf_given_index)
    std::vector<int32_t> column_of(table, -1);
    std::vector<int64_t> df_truth;
    for (size_t key = 0; key < table; ++key) {
        if (!present[key]) continue;
        column_of[key] = static_cast<int32_t>(problem->vocabulary.size());
        problem->vocabulary.push_back(space.bytes_of_key(static_cast<uint32_t>(key)));
        int64_t df = 0;
        for (int w = 0; w < workers; ++w) df += counts[static_cast<size_t>(w) * table + key];
        df_truth.push_back(df);
    }
    const int64_t V = static_cast<int64_t>(problem->vocabulary.size());
    problem->n_columns = V;
    problem->idf64.assign(static_cast<size_t>(V), 0.0);
    double max_idf = -INFINITY;
    for (int64_t g = 0; g < V; ++g) {
        const int64_t df = df_truth[static_cast<size_t>(g)];
        if (df > 0) {
            const double value = std::log(static_cast<double>(n_truth) / static_cast<double>(df));  // :139
            problem->idf64[static_cast<size_t>(g)] = value;
            max_idf = std::max(max_idf, value);
        }
    }
    for (int64_t g = 0; g < V; ++g)
        if (df_truth[static_cast<size_t>(g)] == 0) problem->idf64[static_cast<size_t>(g)] = max_idf;
    problem->idf32.resize(static_cast<size_t>(V));
    for (int64_t g = 0; g < V; ++g)
        problem->idf32[static_cast<size_t>(g)] = static_cast<float>(problem->idf64[static_cast<size_t>(g)]);

    // ---- truth inverted index: counting sort by column, rows ascending inside a column; zero entries are not stored
    problem->rowptr.assign(static_cast<size_t>(V) + 1, 0);
    for (int64_t g = 0; g < V; ++g)
        problem->rowptr[static_cast<size_t>(g) + 1] =
            problem->rowptr[static_cast<size_t>(g)] +
            (problem->idf32[static_cast<size_t>(g)] != 0.f ? df_truth[static_cast<size_t>(g)] : 0);
    for (size_t key = 0; key < table; ++key) {  // counts -> first write position of every (worker, key)
        const int32_t g = column_of[key];
        int64_t position = g >= 0 ? problem->rowptr[static_cast<size_t>(g)] : 0;
        for (int w = 0; w < workers; ++w) {
            const int64_t here = counts[static_cast<size_t>(w) * table + key];
            counts[static_cast<size_t>(w) * table + key] = position;
            position += here;
        }
    }
    problem->truth_idx.resize(static_cast<size_t>(problem->rowptr[static_cast<size_t>(V)]));
    problem->sums32.assign(static_cast<size_t>(n_truth), 0.f);
    ds::parallel_ranges(n_truth, workers, [&](int worker, int64_t begin, int64_t end) {
        TitleKeys title(space);
        int64_t *cursor = counts.data() + static_cast<size_t>(worker) * table;
        for (int64_t t = begin; t < end; ++t) {
            const int count = title.collect(truth_chars + truth_offsets[t], truth_offsets[t + 1] - truth_offsets[t]);
            float sum = 0.f;  // sum(uniqueness_values), float32, one addition per n-gram (:174)
            for (int j = 0; j < count; ++j) {
                const uint32_t key = title.keys[j];
                const float value = problem->idf32[static_cast<size_t>(column_of[key])];
                sum = sum + value;
                if (value != 0.f) problem->truth_idx[static_cast<size_t>(cursor[key]++)] = static_cast<int32_t>(t);
            }
            problem->sums32[static_cast<size_t>(t)] = sum;
        }
    });

    // ---- query rows: ascending non-zero column ids, max_intersection_possible in that order (float64)
    problem->q_rowptr.assign(static_cast<size_t>(n_queries) + 1, 0);
    problem->q_maxint.assign(static_cast<size_t>(n_queries), 0.0);
    {
        std::vector<std::vector<int32_t>> parts(static_cast<size_t>(workers));
        ds::parallel_ranges(n_queries, workers, [&](int worker, int64_t begin, int64_t end) {
            TitleKeys title(space);
            std::vector<int32_t> &mine = parts[static_cast<size_t>(worker)];
            mine.reserve(static_cast<size_t>(end - begin) * 24);
            std::vector<int32_t> row;
            for (int64_t q = begin; q < end; ++q) {
                const int count = title.collect(query_chars + query_offsets[q], query_offsets[q + 1] - query_offsets[q]);
                row.clear();
                for (int j = 0; j < count; ++j) {
                    const int32_t g = column_of[title.keys[j]];
                    if (problem->idf32[static_cast<size_t>(g)] != 0.f) row.push_back(g);
                }
                std::sort(row.begin(), row.end());
                double total = 0.0;
                for (int32_t g : row) total = total + problem->idf64[static_cast<size_t>(g)];
                problem->q_maxint[static_cast<size_t>(q)] = total;
                mine.insert(mine.end(), row.begin(), row.end());
                problem->q_rowptr[static_cast<size_t>(q) + 1] = static_cast<int64_t>(row.size());  // length for now
            }
        });
        for (int64_t q = 0; q < n_queries; ++q) problem->q_rowptr[static_cast<size_t>(q) + 1] += problem->q_rowptr[static_cast<size_t>(q)];
        problem->q_cols.resize(static_cast<size_t>(problem->q_rowptr[static_cast<size_t>(n_queries)]));
        size_t at = 0;
        for (const std::vector<int32_t> &part : parts) {
            if (!part.empty()) std::memcpy(problem->q_cols.data() + at, part.data(), part.size() * sizeof(int32_t));
            at += part.size();
        }
    }
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

// ---- the encoders feeding construct_features, for whole collections -------------------
// FeatureEngineering.encode_title (doppelspeller/feature_engineering.py:298-307) for n titles: out_enc[t] = the title's
// characters mapped through code_of (NULL: the bytes as they are), right-padded with 0 to `stride` bytes (255 in the
// reference); out_len[t] = the number of characters (predict.py:195-197 `.str.len()` as uint8).  Threaded.
extern "C" int ds_encode_titles(const uint8_t *chars, const int64_t *offsets, int64_t n, const uint8_t *code_of,
                                int64_t stride, uint8_t *out_enc, uint8_t *out_len)
{
    DS_REQUIRE(offsets && out_enc && out_len && n >= 0, "ds_encode_titles: null pointer");
    DS_REQUIRE(stride >= 1 && stride <= 65535, "ds_encode_titles: stride=%lld outside 1..65535", (long long)stride);
    ds::FirstError error;
    ds::parallel_ranges(n, ds::host_threads(), [&](int, int64_t begin, int64_t end) {
        for (int64_t t = begin; t < end; ++t) {
            const int64_t length = offsets[t + 1] - offsets[t];
            if (length < 0 || length > stride || (length > 0 && !chars)) {
                error.raise("ds_encode_titles: title %lld has %lld characters (stride %lld)", (long long)t,
                            (long long)length, (long long)stride);
                return;
            }
            uint8_t *row = out_enc + t * stride;
            const uint8_t *source = chars + offsets[t];
            if (code_of) for (int64_t i = 0; i < length; ++i) row[i] = code_of[source[i]];
            else if (length) std::memcpy(row, source, static_cast<size_t>(length));
            std::memset(row + length, 0, static_cast<size_t>(stride - length));
            out_len[t] = static_cast<uint8_t>(length > 255 ? 255 : length);
        }
    });
    DS_REQUIRE(!error.failed(), "%s", error.message());
    return DS_OK;
}

namespace {

// One word occurrence, first of its title (duplicates inside a title count once: common.py:140-142 `set(y)`).
struct WordSlot {
    uint64_t hash;
    uint32_t title;
    uint16_t start, length;
};

inline uint64_t word_hash(const uint8_t *word, int length)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < length; ++i) h = (h ^ word[i]) * 0x100000001b3ull;
    return ds::mix64(h ^ (static_cast<uint64_t>(length) << 56));
}

// Exact string -> document frequency table of one hash bucket: open addressing on the 64-bit hash, every hit verified
// against the bytes of the entry's representative occurrence.
struct WordTable {
    struct Entry {
        uint64_t hash;
        uint32_t title;
        uint16_t start, length;
        uint32_t count;
    };
    std::vector<Entry> slots;
    size_t used = 0;
    const uint8_t *chars;
    const int64_t *offsets;

    WordTable(const uint8_t *c, const int64_t *o) : slots(1024, Entry{0, 0, 0, 0, 0}), chars(c), offsets(o) {}

    bool same(const Entry &entry, const uint8_t *word, int length) const
    {
        return entry.length == length &&
               std::memcmp(chars + offsets[entry.title] + entry.start, word, static_cast<size_t>(length)) == 0;
    }
    Entry *find(uint64_t hash, const uint8_t *word, int length)
    {
        size_t at = static_cast<size_t>(hash >> 8) & (slots.size() - 1);
        while (slots[at].count != 0 && !(slots[at].hash == hash && same(slots[at], word, length)))
            at = (at + 1) & (slots.size() - 1);
        return &slots[at];
    }
    void add(const WordSlot &slot)
    {
        if ((used + 1) * 2 > slots.size()) {
            std::vector<Entry> old(slots.size() * 2, Entry{0, 0, 0, 0, 0});
            old.swap(slots);
            for (const Entry &entry : old) {
                if (entry.count == 0) continue;
                size_t at = static_cast<size_t>(entry.hash >> 8) & (slots.size() - 1);
                while (slots[at].count != 0) at = (at + 1) & (slots.size() - 1);
                slots[at] = entry;
            }
        }
        Entry *entry = find(slot.hash, chars + offsets[slot.title] + slot.start, slot.length);
        if (entry->count == 0) {
            *entry = Entry{slot.hash, slot.title, slot.start, slot.length, 1};
            ++used;
        } else {
            ++entry->count;
        }
    }
};

// Calls word(start, length) for every maximal run of non-separator bytes of a title (str.split()).
template <typename F>
inline void for_each_word(const uint8_t *title, int64_t length, const uint8_t *separator, F word)
{
    int64_t i = 0;
    while (i < length) {
        while (i < length && separator[title[i]]) ++i;
        const int64_t start = i;
        while (i < length && !separator[title[i]]) ++i;
        if (i > start) word(start, i - start);
    }
}

}  // namespace

// FeatureEngineering.get_truth_words_counts (feature_engineering.py:309-319) over the whole truth collection, with the
// counter of common.py:140-142 built on the way: out_counts[t][j] = in how many truth titles the j-th word of title t
// occurs (a word repeated inside a title counts once), first DS_WORDS words, 0-padded.  separator[256]: non-zero for the
// bytes str.split() splits on.  Words are compared as byte strings (exact); threaded by hash bucket.
extern "C" int ds_truth_word_counts(const uint8_t *chars, const int64_t *offsets, int64_t n, const uint8_t *separator,
                                    uint32_t *out_counts)
{
    DS_REQUIRE(offsets && separator && out_counts && n >= 0, "ds_truth_word_counts: null pointer");
    DS_REQUIRE(n < (int64_t(1) << 32), "ds_truth_word_counts: more than 2^32 titles");
    const int threads = ds::host_threads();
    ds::FirstError error;
    // pass 1: the distinct words of every title, routed to the bucket (= owning thread) of their hash
    std::vector<std::vector<WordSlot>> routed(static_cast<size_t>(threads) * static_cast<size_t>(threads));
    ds::parallel_ranges(n, threads, [&](int thread, int64_t begin, int64_t end) {
        std::vector<WordSlot> mine;
        for (int64_t t = begin; t < end; ++t) {
            const int64_t length = offsets[t + 1] - offsets[t];
            if (length < 0 || length > kMaxTitle || (length > 0 && !chars)) {
                error.raise("ds_truth_word_counts: bad title %lld", (long long)t);
                return;
            }
            const uint8_t *title = chars + offsets[t];
            mine.clear();
            for_each_word(title, length, separator, [&](int64_t start, int64_t size) {
                const uint64_t hash = word_hash(title + start, static_cast<int>(size));
                for (const WordSlot &earlier : mine)
                    if (earlier.hash == hash && earlier.length == size &&
                        std::memcmp(title + earlier.start, title + start, static_cast<size_t>(size)) == 0)
                        return;
                mine.push_back(WordSlot{hash, static_cast<uint32_t>(t), static_cast<uint16_t>(start),
                                        static_cast<uint16_t>(size)});
            });
            for (const WordSlot &slot : mine)
                routed[static_cast<size_t>(thread) * threads + static_cast<size_t>(slot.hash % static_cast<uint64_t>(threads))]
                    .push_back(slot);
        }
    });
    DS_REQUIRE(!error.failed(), "%s", error.message());
    // pass 2: every bucket counts its words
    std::vector<WordTable> tables;
    tables.reserve(static_cast<size_t>(threads));
    for (int b = 0; b < threads; ++b) tables.emplace_back(chars, offsets);
    ds::parallel_ranges(threads, threads, [&](int, int64_t begin, int64_t end) {
        for (int64_t b = begin; b < end; ++b)
            for (int source = 0; source < threads; ++source) {
                std::vector<WordSlot> &list = routed[static_cast<size_t>(source) * threads + static_cast<size_t>(b)];
                for (const WordSlot &slot : list) tables[static_cast<size_t>(b)].add(slot);
                std::vector<WordSlot>().swap(list);
            }
    });
    // pass 3: the counts of every title's first DS_WORDS words
    ds::parallel_ranges(n, threads, [&](int, int64_t begin, int64_t end) {
        for (int64_t t = begin; t < end; ++t) {
            const uint8_t *title = chars + offsets[t];
            uint32_t *row = out_counts + t * DS_WORDS;
            int filled = 0;
            for_each_word(title, offsets[t + 1] - offsets[t], separator, [&](int64_t start, int64_t size) {
                if (filled >= DS_WORDS) return;
                const uint64_t hash = word_hash(title + start, static_cast<int>(size));
                row[filled++] = tables[static_cast<size_t>(hash % static_cast<uint64_t>(threads))]
                                    .find(hash, title + start, static_cast<int>(size))->count;
            });
            for (; filled < DS_WORDS; ++filled) row[filled] = 0;
        }
    });
    return DS_OK;
}
