// Host-side text plumbing of the hot path's input side (no GPU work): byte-level BPE merges and the token-batch builder.
// Replaces the native parts of mlx_whisper.tokenizer (tiktoken's CoreBPE) behind get_tokenizer().encode / .decode and
// IPADataset._tokenize_ipa_batch (scripts/ipa_data_loader.py:102-131, 146-152).  The GPT-2 pre-tokeniser regex (Unicode
// classes) stays in Python; what arrives here is one pre-split piece at a time.
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "wipa.h"

void wipa_set_error(const char* fmt, ...);

struct wipa_bpe {
    std::unordered_map<std::string, int32_t> rank;
    std::vector<std::string> token;  // rank -> bytes ("" if the rank is unused)
};

extern "C" wipa_bpe* wipa_bpe_create(const uint8_t* blob, const int32_t* lens, const int32_t* ranks, int n) {
    if (!blob || !lens || !ranks || n <= 0) {
        wipa_set_error("wipa_bpe_create: null table");
        return nullptr;
    }
    wipa_bpe* b = new wipa_bpe();
    int32_t max_rank = -1;
    for (int i = 0; i < n; ++i) max_rank = ranks[i] > max_rank ? ranks[i] : max_rank;
    b->token.assign((size_t)max_rank + 1, std::string());
    b->rank.reserve((size_t)n * 2);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (lens[i] <= 0 || ranks[i] < 0) {
            delete b;
            wipa_set_error("wipa_bpe_create: bad entry %d (len %d, rank %d)", i, lens[i], ranks[i]);
            return nullptr;
        }
        std::string t(reinterpret_cast<const char*>(blob + off), (size_t)lens[i]);
        off += (size_t)lens[i];
        b->rank.emplace(t, ranks[i]);
        b->token[(size_t)ranks[i]] = t;
    }
    return b;
}

extern "C" void wipa_bpe_free(wipa_bpe* b) { delete b; }

// tiktoken's byte_pair_merge: start from single bytes, repeatedly merge the adjacent pair with the LOWEST rank (leftmost
// on ties), stop when no adjacent pair is in the table.  Every final part must itself be a token.
extern "C" int wipa_bpe_encode_piece(const wipa_bpe* b, const uint8_t* piece, int len, int32_t* out, int max_out) {
    if (!b || (!piece && len > 0) || !out) {
        wipa_set_error("wipa_bpe_encode_piece: null pointer");
        return WIPA_ERR_ARG;
    }
    if (len <= 0) return 0;
    const std::string s(reinterpret_cast<const char*>(piece), (size_t)len);
    auto whole = b->rank.find(s);
    if (whole != b->rank.end()) {
        if (max_out < 1) return WIPA_ERR_ARG;
        out[0] = whole->second;
        return 1;
    }
    std::vector<int> start((size_t)len + 1);  // part i = bytes [start[i], start[i+1])
    for (int i = 0; i <= len; ++i) start[(size_t)i] = i;
    auto pair_rank = [&](size_t i) -> int32_t {  // rank of parts i and i+1 joined, or -1
        auto it = b->rank.find(s.substr((size_t)start[i], (size_t)(start[i + 2] - start[i])));
        return it == b->rank.end() ? -1 : it->second;
    };
    while (start.size() > 2) {
        int32_t best = -1;
        size_t best_i = 0;
        for (size_t i = 0; i + 2 < start.size(); ++i) {
            const int32_t r = pair_rank(i);
            if (r >= 0 && (best < 0 || r < best)) {
                best = r;
                best_i = i;
            }
        }
        if (best < 0) break;
        start.erase(start.begin() + (long)best_i + 1);
    }
    const int n = (int)start.size() - 1;
    if (n > max_out) {
        wipa_set_error("wipa_bpe_encode_piece: %d ids do not fit into %d", n, max_out);
        return WIPA_ERR_ARG;
    }
    for (int i = 0; i < n; ++i) {
        auto it = b->rank.find(s.substr((size_t)start[(size_t)i], (size_t)(start[(size_t)i + 1] - start[(size_t)i])));
        if (it == b->rank.end()) {
            wipa_set_error("wipa_bpe_encode_piece: byte sequence without a token (table lacks single bytes?)");
            return WIPA_ERR_STATE;
        }
        out[i] = it->second;
    }
    return n;
}

// ids below the table size -> their bytes, concatenated; returns the byte count, or the negative index-1 of the first id the
// table does not hold (special tokens are rendered by the caller).
extern "C" int wipa_bpe_decode(const wipa_bpe* b, const int32_t* ids, int n, uint8_t* out, int max_out) {
    if (!b || (!ids && n > 0) || !out) {
        wipa_set_error("wipa_bpe_decode: null pointer");
        return WIPA_ERR_ARG;
    }
    int w = 0;
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || (size_t)ids[i] >= b->token.size() || b->token[(size_t)ids[i]].empty()) return -(i + 1) - 100;
        const std::string& t = b->token[(size_t)ids[i]];
        if (w + (int)t.size() > max_out) {
            wipa_set_error("wipa_bpe_decode: output buffer too small");
            return WIPA_ERR_ARG;
        }
        memcpy(out + w, t.data(), t.size());
        w += (int)t.size();
    }
    return w;
}

// IPADataset._tokenize_ipa_batch (ipa_data_loader.py:102-131): row r = prefix (SOT sequence incl. <|notimestamps|>) + the
// r-th id list + eot, padded with eot to the longest row.  out is int32 [n_rows, ld_out]; returns the common row width.
extern "C" int wipa_build_token_batch(const int32_t* ids, const int32_t* row_lens, int n_rows, const int32_t* prefix, int n_prefix,
                                      int32_t eot, int32_t* out, int64_t ld_out) {
    if ((!ids && n_rows > 0) || !row_lens || (!prefix && n_prefix > 0) || !out || n_rows <= 0) {
        wipa_set_error("wipa_build_token_batch: bad arguments");
        return WIPA_ERR_ARG;
    }
    int width = 0;
    for (int r = 0; r < n_rows; ++r) {
        if (row_lens[r] < 0) return WIPA_ERR_ARG;
        const int w = n_prefix + row_lens[r] + 1;
        width = w > width ? w : width;
    }
    if (width > ld_out) {
        wipa_set_error("wipa_build_token_batch: rows of %d ids do not fit into ld_out = %lld", width, (long long)ld_out);
        return WIPA_ERR_ARG;
    }
    size_t off = 0;
    for (int r = 0; r < n_rows; ++r) {
        int32_t* row = out + (int64_t)r * ld_out;
        int c = 0;
        for (int i = 0; i < n_prefix; ++i) row[c++] = prefix[i];
        for (int i = 0; i < row_lens[r]; ++i) row[c++] = ids[off + (size_t)i];
        off += (size_t)row_lens[r];
        while (c < width) row[c++] = eot;
    }
    return width;
}
