// Synthetic title generator behind doppel-speller_amd/synth.py: NOT part of libdoppel_amd.so --
// test / benchmark infrastructure, compiled with g++ into libdoppel_synth.so.  It exists because the C5 configuration
// needs 50M truth titles: the NumPy generator of rounds 1-2 took 316 s there, this one a few seconds.
//
// Every title has its own random stream (a splitmix64 sequence keyed by (seed, title index)), so the output does not
// depend on the number of threads and a title can be generated twice (once for its length, once for its characters).
// All distributions arrive as tables from synth.py (cumulative probabilities computed with NumPy): the native side only
// draws uniform numbers and searches tables -- no libm call whose last bit could differ between machines.
//
// Titles are sequences of character CODES (feature_engineering.py:200-203: '-' = 0 fill, ' ' = 1, a-z = 2..27,
// 0-9 = 28..37), words joined by single spaces, at most 255 characters (settings.py:68).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kWordWidth = 20;      // longest word
constexpr int kMaxTitle = 255;      // MAX_CHARACTERS_ALLOWED_IN_THE_TITLE
constexpr uint8_t kSpace = 1, kZeroDigit = 28;

struct Stream {
    uint64_t state;
    Stream(uint64_t seed, uint64_t index, uint64_t purpose)
    {
        state = seed * 0x9e3779b97f4a7c15ull + index * 0xd1342543de82ef95ull + purpose * 0xaf251af3b0f025b5ull;
        next();
        next();
    }
    uint64_t next()
    {
        uint64_t z = (state += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
    int64_t below(int64_t n) { return static_cast<int64_t>(uniform() * static_cast<double>(n)); }
};

// first index whose cumulative probability is >= u (np.searchsorted, side='left'), clipped to the table
inline int64_t draw(const double *cumulative, int64_t size, double u)
{
    const int64_t at = std::lower_bound(cumulative, cumulative + size, u) - cumulative;
    return at < size ? at : size - 1;
}

struct Parameters {
    const uint8_t *word_chars;       // [n_vocabulary][kWordWidth], 0-padded
    const uint8_t *word_lengths;     // [n_vocabulary]
    int64_t n_vocabulary;
    int64_t n_suffixes;              // the first words of the vocabulary: "limited", "ltd", ...
    const double *zipf_cumulative;   // [n_vocabulary - n_suffixes] over the non-suffix words
    const double *words_cumulative;  // [20] P(words per title <= i + 1)
    const double *suffix_cumulative; // [n_suffixes]
    const double *hapax_cumulative;  // [17] P(hapax length <= 4 + i)
    double suffix_share, hapax_fraction, hapax_digit_share;
};

// One title into `out` (room for kMaxTitle bytes); returns its length.  Recipe of synth.py's docstring: 1 + Poisson(2.5)
// words drawn Zipf-Mandelbrot from the vocabulary, a company suffix in the last slot of `suffix_share` of the titles
// with more than one word, `hapax_fraction` of the other slots spelled from random characters, leading words kept while
// they fit into 255 characters.
int make_title(const Parameters &p, uint64_t seed, uint64_t index, uint64_t purpose, uint8_t *out)
{
    Stream rng(seed, index, purpose);
    const int n_words = static_cast<int>(draw(p.words_cumulative, 20, rng.uniform())) + 1;
    const bool has_suffix = rng.uniform() < p.suffix_share;
    const int64_t suffix = draw(p.suffix_cumulative, p.n_suffixes, rng.uniform());
    int length = 0;
    for (int slot = 0; slot < n_words; ++slot) {
        uint8_t word[kWordWidth];
        int word_length;
        const double u_word = rng.uniform(), u_hapax = rng.uniform();
        if (has_suffix && n_words > 1 && slot == n_words - 1) {
            word_length = p.word_lengths[suffix];
            std::memcpy(word, p.word_chars + suffix * kWordWidth, kWordWidth);
        } else if (u_hapax < p.hapax_fraction) {
            word_length = 4 + static_cast<int>(draw(p.hapax_cumulative, 17, rng.uniform()));
            for (int i = 0; i < word_length; ++i)
                word[i] = rng.uniform() < p.hapax_digit_share ? static_cast<uint8_t>(28 + rng.below(10))
                                                               : static_cast<uint8_t>(2 + rng.below(26));
        } else {
            const int64_t id = p.n_suffixes + draw(p.zipf_cumulative, p.n_vocabulary - p.n_suffixes, u_word);
            word_length = p.word_lengths[id];
            std::memcpy(word, p.word_chars + id * kWordWidth, kWordWidth);
        }
        const int needed = length + (slot > 0 ? 1 : 0) + word_length;
        if (needed > kMaxTitle) break;  // the leading words that fit
        if (slot > 0) out[length++] = kSpace;
        std::memcpy(out + length, word, static_cast<size_t>(word_length));
        length += word_length;
    }
    return length;
}

// " ".join(title.split())[:255].strip(), then rjust(3, '0') (common.py:28-38); returns the new length
int normalise(uint8_t *title, int length)
{
    int write = 0;
    for (int i = 0; i < length; ++i) {
        if (title[i] == kSpace && (write == 0 || title[write - 1] == kSpace)) continue;
        title[write++] = title[i];
    }
    while (write > 0 && title[write - 1] == kSpace) --write;
    if (write > kMaxTitle) write = kMaxTitle;
    while (write > 0 && title[write - 1] == kSpace) --write;
    if (write < 3) {
        uint8_t padded[3] = {kZeroDigit, kZeroDigit, kZeroDigit};
        std::memcpy(padded + (3 - write), title, static_cast<size_t>(write));
        std::memcpy(title, padded, 3);
        write = 3;
    }
    return write;
}

// 1-2 keyboard-style edits of a truth title (the recipe of feature_engineering_prepare.py:90-173): delete a character,
// insert / substitute a keyboard neighbour, insert / remove a space, swap two adjacent words.  `title` has room for
// kMaxTitle + 8 bytes.  neighbours[code][0..1] = the keys left and right of a character (0 = none).
int misspell(Stream &rng, uint8_t *title, int length, const uint8_t *neighbours)
{
    auto neighbour = [&](uint8_t code) -> uint8_t {
        const uint8_t a = neighbours[2 * code], b = neighbours[2 * code + 1];
        if (a && b) return rng.below(2) ? b : a;
        if (a || b) return a ? a : b;
        return 6;  // 'e'
    };
    const int edits = 1 + (rng.uniform() < 0.4 ? 1 : 0);
    for (int edit = 0; edit < edits && length > 0; ++edit) {
        const int kind = static_cast<int>(rng.below(6));
        const int at = static_cast<int>(rng.below(length));
        if (kind == 0) {
            if (length > 4) {
                std::memmove(title + at, title + at + 1, static_cast<size_t>(length - at - 1));
                --length;
            }
        } else if (kind == 1) {
            if (length < kMaxTitle + 4) {
                const uint8_t inserted = neighbour(title[at]);
                std::memmove(title + at + 1, title + at, static_cast<size_t>(length - at));
                title[at] = inserted;
                ++length;
            }
        } else if (kind == 2) {
            if (title[at] != kSpace) title[at] = neighbour(title[at]);
        } else if (kind == 3) {
            if (at > 0 && at < length - 1 && title[at] != kSpace && title[at - 1] != kSpace && length < kMaxTitle + 4) {
                std::memmove(title + at + 1, title + at, static_cast<size_t>(length - at));
                title[at] = kSpace;
                ++length;
            }
        } else {
            int spaces = 0;
            for (int i = 0; i < length; ++i) spaces += title[i] == kSpace;
            if (spaces == 0) continue;
            int chosen = static_cast<int>(rng.below(spaces)), cut = 0;
            for (int i = 0; i < length; ++i)
                if (title[i] == kSpace && chosen-- == 0) cut = i;
            if (kind == 4) {
                std::memmove(title + cut, title + cut + 1, static_cast<size_t>(length - cut - 1));
                --length;
            } else {  // swap the words on both sides of the chosen space
                int left = cut, right = cut + 1;
                while (left > 0 && title[left - 1] != kSpace) --left;
                int right_end = right;
                while (right_end < length && title[right_end] != kSpace) ++right_end;
                uint8_t swapped[kMaxTitle + 8];
                int write = 0;
                std::memcpy(swapped + write, title + right, static_cast<size_t>(right_end - right));
                write += right_end - right;
                swapped[write++] = kSpace;
                std::memcpy(swapped + write, title + left, static_cast<size_t>(cut - left));
                write += cut - left;
                std::memcpy(title + left, swapped, static_cast<size_t>(write));
            }
        }
    }
    return normalise(title, length);
}

template <typename F>
void fan_out(int64_t n, int threads, F fn)
{
    if (threads <= 1 || n < 1024) {
        fn(int64_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back([=] { fn(n * t / threads, n * (t + 1) / threads); });
    for (std::thread &thread : pool) thread.join();
}

Parameters parameters(const uint8_t *word_chars, const uint8_t *word_lengths, int64_t n_vocabulary, int64_t n_suffixes,
                      const double *zipf_cumulative, const double *words_cumulative, const double *suffix_cumulative,
                      const double *hapax_cumulative, const double *shares)
{
    return Parameters{word_chars, word_lengths, n_vocabulary, n_suffixes, zipf_cumulative, words_cumulative,
                      suffix_cumulative, hapax_cumulative, shares[0], shares[1], shares[2]};
}

}  // namespace

extern "C" {

int synth_version(void) { return 3; }

// Fresh titles `first .. first + count` of stream (seed, purpose).  out_offsets == NULL: out_lengths[count] only.
// Otherwise out_flat receives the characters at out_offsets[i] (the caller's cumulative sum of the lengths).
int synth_titles(uint64_t seed, uint64_t purpose, int64_t first, int64_t count, const uint8_t *word_chars,
                 const uint8_t *word_lengths, int64_t n_vocabulary, int64_t n_suffixes, const double *zipf_cumulative,
                 const double *words_cumulative, const double *suffix_cumulative, const double *hapax_cumulative,
                 const double *shares, int threads, int32_t *out_lengths, const int64_t *out_offsets, uint8_t *out_flat)
{
    if (!word_chars || !word_lengths || n_vocabulary <= n_suffixes || n_suffixes < 1 || count < 0) return -1;
    const Parameters p = parameters(word_chars, word_lengths, n_vocabulary, n_suffixes, zipf_cumulative, words_cumulative,
                                    suffix_cumulative, hapax_cumulative, shares);
    fan_out(count, threads, [&](int64_t begin, int64_t end) {
        uint8_t title[kMaxTitle + 8];
        for (int64_t i = begin; i < end; ++i) {
            const int length = make_title(p, seed, static_cast<uint64_t>(first + i), purpose, title);
            if (out_offsets) std::memcpy(out_flat + out_offsets[i], title, static_cast<size_t>(length));
            else out_lengths[i] = length;
        }
    });
    return 0;
}

// Queries: source[j] >= 0 -> the truth title of that row with 1-2 edits, source[j] < 0 -> a fresh title of stream
// (seed, purpose 2).  Two passes like synth_titles (lengths, then characters).
int synth_queries(uint64_t seed, int64_t count, const int64_t *source, const uint8_t *truth_flat,
                  const int64_t *truth_offsets, const uint8_t *neighbours, const uint8_t *word_chars,
                  const uint8_t *word_lengths, int64_t n_vocabulary, int64_t n_suffixes, const double *zipf_cumulative,
                  const double *words_cumulative, const double *suffix_cumulative, const double *hapax_cumulative,
                  const double *shares, int threads, int32_t *out_lengths, const int64_t *out_offsets, uint8_t *out_flat)
{
    if (!source || !truth_flat || !truth_offsets || !neighbours || count < 0) return -1;
    const Parameters p = parameters(word_chars, word_lengths, n_vocabulary, n_suffixes, zipf_cumulative, words_cumulative,
                                    suffix_cumulative, hapax_cumulative, shares);
    fan_out(count, threads, [&](int64_t begin, int64_t end) {
        uint8_t title[kMaxTitle + 16];
        for (int64_t j = begin; j < end; ++j) {
            int length;
            if (source[j] >= 0) {
                const int64_t from = truth_offsets[source[j]];
                length = static_cast<int>(truth_offsets[source[j] + 1] - from);
                std::memcpy(title, truth_flat + from, static_cast<size_t>(length));
                Stream rng(seed, static_cast<uint64_t>(j), 1);
                length = misspell(rng, title, length, neighbours);
            } else {
                length = normalise(title, make_title(p, seed, static_cast<uint64_t>(j), 2, title));
            }
            if (out_offsets) std::memcpy(out_flat + out_offsets[j], title, static_cast<size_t>(length));
            else out_lengths[j] = length;
        }
    });
    return 0;
}

}  // extern "C"
