// Host-side native data plumbing in front of the hot path: TFRecord framing,
// tf.train.Example decoding and vocabulary-file string -> id lookup, i.e. what the reference gets
// from TensorFlow's C++ tf.data / parse_example / HashTable kernels
// (/root/reference algorithm/utils.py:18-24, algorithm/DeepFM/deepfm.py:102-117,56-64;
//  record format written by dataset/wechat_algo_data1/DataGenerator.py:390-447).
// Plain C ABI (include/recalgo_host.h), no dependencies, built with g++ into librecalgo_host.so.
//
//   record  = uint64 len | uint32 masked_crc32c(len) | bytes | uint32 masked_crc32c(bytes)
//   Example = {1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}
//   SequenceExample: field 1 (context) is read like Example.features, field 2 (feature_lists) is
//   skipped — exactly what tf.parse_example does.
//   vocabulary id = 0-based line number of the key, -1 when absent (A-2).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <pthread.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>
#include <algorithm>

#include "../../include/recalgo_host.h"

namespace {

// ---- crc32c (Castagnoli), slicing-by-8 -----------------------------------------------------------
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
const CrcTables kCrc;

uint32_t crc32c(const uint8_t* p, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = kCrc.t[7][lo & 0xFF] ^ kCrc.t[6][(lo >> 8) & 0xFF] ^ kCrc.t[5][(lo >> 16) & 0xFF] ^ kCrc.t[4][lo >> 24] ^
            kCrc.t[3][hi & 0xFF] ^ kCrc.t[2][(hi >> 8) & 0xFF] ^ kCrc.t[1][(hi >> 16) & 0xFF] ^ kCrc.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = kCrc.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ---- protobuf wire helpers -------------------------------------------------------------------------
struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
};
inline bool read_varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    if (p < end && !(*p & 0x80)) {                         // one byte: every tag and almost every length of an Example
        v = *p++;
        return true;
    }
    v = 0;
    for (int shift = 0; p < end && shift < 70; shift += 7) {
        uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}
// iterate the fields of a message; calls f(field, wire_type, payload_span_or_value)
template <class Fn>
bool for_fields(Span s, Fn&& f) {
    const uint8_t* p = s.p;
    const uint8_t* end = s.p + s.n;
    while (p < end) {
        uint64_t key;
        if (!read_varint(p, end, key)) return false;
        const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (wt == 2) {
            uint64_t n;
            if (!read_varint(p, end, n) || n > (uint64_t)(end - p)) return false;
            f(field, wt, Span{p, (size_t)n}, (uint64_t)0);
            p += n;
        } else if (wt == 0) {
            uint64_t v;
            if (!read_varint(p, end, v)) return false;
            f(field, wt, Span{}, v);
        } else if (wt == 5) {
            if (end - p < 4) return false;
            f(field, wt, Span{p, 4}, (uint64_t)0);
            p += 4;
        } else if (wt == 1) {
            if (end - p < 8) return false;
            f(field, wt, Span{p, 8}, (uint64_t)0);
            p += 8;
        } else {
            return false;
        }
    }
    return true;
}

// Canonical encodings, recognised by a few byte compares (what every protobuf writer emits for an Example): the generic
// field walks below cost ~60 ns per map entry, 27 entries per record — two thirds of a record's decode time once the
// vocabulary probes are prefetched.  Anything else (other field order, repeated fields, lengths >= 128) -> false, and the
// caller takes the generic walk, which yields the same (name, feature) / (first value, count).
//   map entry  = 0x0A klen key 0x12 vlen Feature                       (key = 1, value = 2, both length-delimited)
inline bool fast_map_entry(Span entry, std::string_view& name, Span& feat) {
    const uint8_t* e = entry.p;
    const size_t n = entry.n;
    if (n < 4 || e[0] != 0x0A || (e[1] & 0x80)) return false;
    const size_t kl = e[1];
    if (kl + 4 > n || e[2 + kl] != 0x12 || (e[3 + kl] & 0x80) || kl + 4 + (size_t)e[3 + kl] != n) return false;
    name = std::string_view((const char*)e + 2, kl);
    feat = Span{e + 4 + kl, (size_t)e[3 + kl]};
    return true;
}
//   Feature    = 0x0A llen BytesList ;  BytesList = 0x0A slen bytes    (exactly one value) | empty
inline bool fast_single_bytes(Span feat, Span& first, int& count) {
    const uint8_t* v = feat.p;
    if (feat.n < 2 || v[0] != 0x0A || (v[1] & 0x80) || 2 + (size_t)v[1] != feat.n) return false;
    const uint8_t* l = v + 2;
    const size_t ln = v[1];
    if (ln == 0) {
        count = 0;
        return true;
    }
    if (ln < 2 || l[0] != 0x0A || (l[1] & 0x80) || 2 + (size_t)l[1] != ln) return false;
    first = Span{l + 2, (size_t)l[1]};
    count = 1;
    return true;
}

// a == b over n bytes without a libc call for the short strings of an Example (feature names, vocabulary keys: 4..24
// bytes): overlapping 8-byte loads, every load inside [x, x + n).
inline uint64_t load64(const char* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t load32(const char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline bool bytes_equal(const char* a, const char* b, size_t n) {
    if (n >= 8) {
        if (load64(a) != load64(b) || load64(a + n - 8) != load64(b + n - 8)) return false;
        if (n <= 16) return true;
        if (n <= 24) return load64(a + 8) == load64(b + 8);
        return memcmp(a + 8, b + 8, n - 16) == 0;
    }
    if (n >= 4) return load32(a) == load32(b) && load32(a + n - 4) == load32(b + n - 4);
    for (size_t i = 0; i < n; ++i)
        if (a[i] != b[i]) return false;
    return true;
}
inline bool sv_equal(std::string_view a, std::string_view b) { return a.size() == b.size() && bytes_equal(a.data(), b.data(), a.size()); }

// Vocabulary: open-addressing hash table over the file's bytes (keys are views into `blob`).  A lookup is one
// 64-bit hash of the key (the keys are short: "userid_12345"), a probe that compares stored hashes first and
// memcmp only on a hash match — about 3x faster than std::unordered_map<string_view> on 10^5..10^6-key
// vocabularies, and the lookups are where a record's decode time goes (26 of them per record).
struct Vocab {
    std::string blob;                                      // the file's bytes (keys longer than kInline are compared here)
    static constexpr uint32_t kInline = 22;
    struct Slot {                                          // 32 bytes: half a cache line, no second miss for short keys
        int32_t id = -1;                                   // -1 = empty slot
        uint32_t tag = 0;                                  // upper half of the key's hash
        uint8_t len = 0;                                   // key length when <= kInline, else 255 and key = {off, len} in blob
        char key[23] = {0};
    };
    static_assert(sizeof(Slot) == 32, "vocabulary slot layout");
    // The table of a 10^6-key vocabulary is 64 MB and every probe lands on a random slot: with 4 KB pages each probe is a
    // TLB miss on top of the cache miss (and a software prefetch that misses the TLB may be dropped).  Tables of 2 MB and
    // more are 2 MB-aligned and advised MADV_HUGEPAGE (transparent huge pages: "madvise" or "always" mode; a no-op otherwise).
    struct SlotTable {
        Slot* p = nullptr;
        size_t n = 0;
        SlotTable() = default;
        SlotTable(const SlotTable&) = delete;
        SlotTable& operator=(const SlotTable&) = delete;
        ~SlotTable() { std::free(p); }
        void assign(size_t cap, const Slot& v) {
            std::free(p);
            p = nullptr;
            n = 0;
            constexpr size_t kHuge = size_t(1) << 21;
            const size_t bytes = cap * sizeof(Slot);
            if (bytes >= kHuge) {
                const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
                p = static_cast<Slot*>(std::aligned_alloc(kHuge, rounded));
#ifdef MADV_HUGEPAGE
                if (p) (void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
            } else {
                p = static_cast<Slot*>(std::malloc(bytes ? bytes : sizeof(Slot)));
            }
            if (!p) throw std::bad_alloc();
            for (size_t i = 0; i < cap; ++i) p[i] = v;
            n = cap;
        }
        bool empty() const { return n == 0; }
        Slot& operator[](size_t i) { return p[i]; }
        const Slot& operator[](size_t i) const { return p[i]; }
    };
    SlotTable slots;
    uint64_t mask = 0;
    size_t count = 0;

    // 64-bit hash of a key: the first and the last (up to) 8 bytes folded by one 64 x 64 -> 128 multiplication (wyhash's
    // "mum"), 8-byte words in between for keys longer than 16.  No byte loop and no variable-length memcpy: the FNV form
    // before cost 20 ns per value — as much as the probe it feeds.  (The table lives in memory only: the function can change.)
    static uint64_t mum(uint64_t a, uint64_t b) {
        const unsigned __int128 m = (unsigned __int128)a * b;
        return (uint64_t)m ^ (uint64_t)(m >> 64);
    }
    static uint64_t hash_of(const char* p, size_t n) {
        constexpr uint64_t k1 = 0xe7037ed1a0b428dbull, k2 = 0x8ebc6af09c88c6e3ull;
        uint64_t a, b, seed = k2 ^ (n * 0x9E3779B97F4A7C15ull);
        if (n <= 16) {
            if (n >= 8) { a = load64(p); b = load64(p + n - 8); }
            else if (n >= 4) { a = load32(p); b = load32(p + n - 4); }
            else if (n > 0) { a = ((uint64_t)(uint8_t)p[0] << 16) | ((uint64_t)(uint8_t)p[n >> 1] << 8) | (uint8_t)p[n - 1]; b = 0; }
            else { a = b = 0; }
        } else {
            size_t i = n;
            const char* q = p;
            while (i > 16) {
                seed = mum(load64(q) ^ k1, load64(q + 8) ^ seed);
                q += 16;
                i -= 16;
            }
            a = load64(p + n - 16);
            b = load64(p + n - 8);
        }
        return mum(k1 ^ n, mum(a ^ k1, b ^ seed));
    }
    void reserve(size_t keys) {
        size_t cap = 16;
        while (cap < keys * 2 + 2) cap <<= 1;
        slots.assign(cap, Slot{});
        mask = cap - 1;
    }
    bool equal(const Slot& s, std::string_view k) const {
        if (s.len != 255) return s.len == k.size() && bytes_equal(s.key, k.data(), s.len);
        uint32_t off, len;
        memcpy(&off, s.key, 4);
        memcpy(&len, s.key + 4, 4);
        return len == k.size() && memcmp(blob.data() + off, k.data(), len) == 0;
    }
    void insert_first(std::string_view k, int64_t id) {   // the first occurrence of a key wins
        const uint64_t h = hash_of(k.data(), k.size());
        const uint32_t tag = (uint32_t)(h >> 32);
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            Slot& s = slots[i];
            if (s.id < 0) {
                s.id = (int32_t)id;
                s.tag = tag;
                if (k.size() <= kInline) {
                    s.len = (uint8_t)k.size();
                    memcpy(s.key, k.data(), k.size());
                } else {
                    const uint32_t off = (uint32_t)(k.data() - blob.data()), len = (uint32_t)k.size();
                    s.len = 255;
                    memcpy(s.key, &off, 4);
                    memcpy(s.key + 4, &len, 4);
                }
                ++count;
                return;
            }
            if (s.tag == tag && equal(s, k)) return;
        }
    }
    int64_t find_hashed(std::string_view k, uint64_t h) const {
        if (slots.empty()) return -1;
        const uint32_t tag = (uint32_t)(h >> 32);
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            const Slot& s = slots[i];
            if (s.id < 0) return -1;
            if (s.tag == tag && equal(s, k)) return s.id;
        }
    }
    int64_t find(std::string_view k) const { return find_hashed(k, hash_of(k.data(), k.size())); }
};

// ---- a small persistent worker pool: decode is per-record independent ------------------------------
// parallel_for(n, f) runs f(chunk) for chunk in [0, n) on the workers and the calling thread; chunks
// are handed out by an atomic counter.  RECALGO_READER_THREADS (default: hardware threads, at most
// 16; 1 = everything inline) sizes it.  One job at a time (callers are serialised by the mutex).
class Pool {
  public:
    // Heap singleton, never destroyed.  After fork() the child has none of the worker threads: the
    // pthread_atfork child handler swaps in a pool without workers (everything inline), so a forked
    // data-loader worker keeps decoding instead of waiting for threads that do not exist.
    static Pool& get() {
        static Pool* p = [] {
            inst() = new Pool(false);
            pthread_atfork(nullptr, nullptr, [] { inst() = new Pool(true); });
            return inst();
        }();
        (void)p;
        return *inst();
    }
    size_t threads() const { return workers_.size() + 1; }
    void parallel_for(size_t n, const std::function<void(size_t)>& f) {
        if (n == 0) return;
        if (workers_.empty() || n == 1) {
            for (size_t i = 0; i < n; ++i) f(i);
            return;
        }
        std::lock_guard<std::mutex> serial(call_);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &f;
            n_ = n;
            next_.store(0);
            busy_ = workers_.size();
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return busy_ == 0; });
        job_ = nullptr;
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }

  private:
    static Pool*& inst() {
        static Pool* p = nullptr;
        return p;
    }
    explicit Pool(bool inline_only) {
        if (inline_only) return;
        size_t n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        if (n > 16) n = 16;
        if (const char* e = std::getenv("RECALGO_READER_THREADS")) {
            long v = std::atol(e);
            if (v >= 1 && v <= 256) n = (size_t)v;
        }
        for (size_t i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    void work() {
        for (size_t i = next_.fetch_add(1); i < n_; i = next_.fetch_add(1)) (*job_)(i);
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
            std::lock_guard<std::mutex> lk(m_);
            if (--busy_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, call_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t n_ = 0, busy_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

constexpr size_t kChunk = 128;                             // records per parallel_for chunk

struct Entry {                                             // one (feature name -> Feature message) pair
    std::string_view name;
    Span feat;
};

struct Reader {
    int fd = -1;                                           // the file is mapped: records are views, never copied
    const uint8_t* map = nullptr;
    size_t size = 0, pos = 0;
    bool verify = false;
    int64_t epochs = 1, epoch = 0;                         // dataset.repeat(epochs); epochs < 0: forever
    size_t shuffle = 0;                                    // dataset.shuffle(buffer_size), 0 = off
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    std::vector<Span> pool;                                // shuffle buffer
    bool source_done = false;                              // every pass of dataset.repeat() consumed
    bool pass_done = false;                                // current pass read to its end (shuffle buffer draining)
    std::vector<Span> recs;                                // payloads of the current batch
    // per record: its (name, Feature span) pairs in wire order (built lazily per batch).  Records of
    // one writer list their features in one order, so a lookup starts at the position the key had
    // in the previous record and is O(1) in practice; a repeated map key keeps its LAST entry
    // (protobuf map semantics; resolved while the index is built).
    std::vector<std::vector<Entry>> index;
    bool indexed = false;
    int scans = 0;                                         // accessor calls served by a direct scan since the batch was read
    std::string error;
};

// cheap hash of a feature name: length, first and last (up to) 8 bytes
inline uint64_t name_hash(std::string_view k) {
    uint64_t a = 0, b = 0;
    const size_t n = k.size(), m = n < 8 ? n : 8;
    if (m) {                                               // (an empty name has a null data pointer)
        memcpy(&a, k.data(), m);
        memcpy(&b, k.data() + n - m, m);
    }
    uint64_t h = (a ^ (b * 0x9E3779B97F4A7C15ull) ^ (n * 0xff51afd7ed558ccdull)) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

const Span* find_feature(const std::vector<Entry>& idx, std::string_view k, size_t& hint) {
    const size_t n = idx.size();
    if (hint < n && idx[hint].name == k) return &idx[hint].feat;
    for (size_t j = 0; j < n; ++j)
        if (idx[j].name == k) {
            hint = j;
            return &idx[j].feat;
        }
    return nullptr;
}

// One feature of one record WITHOUT the index: walk the map entries, compare names (length first), keep the LAST match
// (protobuf map semantics).  ~10 ns per entry — cheaper than building the index when a batch is asked for one or two
// features besides the id matrix (the label); from the third such call on the index is built (feature_of).
bool scan_feature(Span rec, std::string_view k, Span& out, bool& ok) {
    bool found = false;
    ok = for_fields(rec, [&](uint32_t field, uint32_t wt, Span features, uint64_t) {
        if (field != 1 || wt != 2) return;
        for_fields(features, [&](uint32_t f2, uint32_t w2, Span entry, uint64_t) {
            if (f2 != 1 || w2 != 2) return;
            bool is_key = false;
            Span feat;
            for_fields(entry, [&](uint32_t f3, uint32_t w3, Span pl, uint64_t) {
                if (w3 != 2) return;
                if (f3 == 1) is_key = pl.n == k.size() && memcmp(pl.p, k.data(), pl.n) == 0;
                else if (f3 == 2) feat = pl;
            });
            if (is_key) {
                out = feat;
                found = true;
            }
        });
    });
    return found && ok;
}

bool index_batch(Reader& r);

// how an accessor reads a feature of the current batch: direct scans for the first two calls, the index afterwards
// -> false when the index had to be built and the batch is malformed
bool prepare_access(Reader& r, bool& use_index) {
    use_index = r.indexed || ++r.scans > 2;
    return !use_index || r.indexed || index_batch(r);
}

bool index_batch(Reader& r) {
    const size_t B = r.recs.size();
    r.index.resize(B);
    std::atomic<long> bad{-1};
    Pool::get().parallel_for((B + kChunk - 1) / kChunk, [&](size_t c) {
        for (size_t i = c * kChunk; i < std::min(B, (c + 1) * kChunk); ++i) {
            const Span rec = r.recs[i];
            auto& idx = r.index[i];
            idx.clear();
            uint64_t seen = 0;                                 // bloom over the names met so far in this record
            bool ok = for_fields(rec, [&](uint32_t field, uint32_t wt, Span features, uint64_t) {
                if (field != 1 || wt != 2) return;             // SequenceExample.feature_lists (2) is skipped
                for_fields(features, [&](uint32_t f2, uint32_t w2, Span entry, uint64_t) {
                    if (f2 != 1 || w2 != 2) return;
                    Entry e;
                    for_fields(entry, [&](uint32_t f3, uint32_t w3, Span pl, uint64_t) {
                        if (w3 != 2) return;
                        if (f3 == 1) e.name = std::string_view((const char*)pl.p, pl.n);
                        else if (f3 == 2) e.feat = pl;
                    });
                    bool dup = false;                          // a repeated map key: the last entry wins
                    const uint64_t bit = 1ull << (name_hash(e.name) & 63);
                    if (seen & bit)                            // (only then can the name have occurred before)
                        for (auto& old : idx)
                            if (old.name == e.name) {
                                old.feat = e.feat;
                                dup = true;
                                break;
                            }
                    seen |= bit;
                    if (!dup) idx.push_back(e);
                });
            });
            if (!ok) bad.store((long)i);
        }
    });
    if (bad.load() >= 0) {
        r.error = "malformed Example in record " + std::to_string(bad.load());
        return false;
    }
    r.indexed = true;
    return true;
}

uint64_t next_rand(Reader& r) {                            // xorshift64*
    r.rng ^= r.rng >> 12;
    r.rng ^= r.rng << 25;
    r.rng ^= r.rng >> 27;
    return r.rng * 0x2545F4914F6CDD1Dull;
}

// next record of the CURRENT pass over the file: 1 = ok, 0 = end of the pass, -1 = error
int read_one(Reader& r, Span& out) {
    if (r.pos >= r.size) return 0;
    if (r.size - r.pos < 12) {
        r.error = "truncated record header";
        return -1;
    }
    const uint8_t* hdr = r.map + r.pos;
    uint64_t len;
    uint32_t hcrc;
    memcpy(&len, hdr, 8);
    memcpy(&hcrc, hdr + 8, 4);
    if (r.verify && masked(crc32c(hdr, 8)) != hcrc) { r.error = "length crc mismatch"; return -1; }
    if (len > r.size - r.pos - 12 || r.size - r.pos - 12 - len < 4) {
        r.error = "truncated record";
        return -1;
    }
    const uint8_t* payload = hdr + 12;
    if (r.verify) {
        uint32_t dcrc;
        memcpy(&dcrc, payload + len, 4);
        if (masked(crc32c(payload, (size_t)len)) != dcrc) { r.error = "data crc mismatch"; return -1; }
    }
    out = Span{payload, (size_t)len};
    r.pos += 16 + (size_t)len;
    return 1;
}

// start the next pass if dataset.repeat() has one left: true = rewound
bool next_epoch(Reader& r) {
    ++r.epoch;
    if ((r.epochs >= 0 && r.epoch >= r.epochs) || r.size == 0) return false;
    r.pos = 0;
    return true;
}

// next record of dataset.shuffle(buffer).repeat(epochs) — the reference's order (algorithm/utils.py:19-21: shuffle
// BEFORE repeat): within a pass tf.data fills the buffer, emits a random slot and refills it from the source; at
// the end of the pass the buffer is drained in random order, and only then does the next pass start, so records of
// different epochs never mix.
int next_record(Reader& r, Span& out) {
    for (;;) {
        if (r.source_done) return 0;
        if (r.shuffle == 0) {
            int rc = read_one(r, out);
            if (rc != 0) return rc;
            if (!next_epoch(r)) { r.source_done = true; return 0; }
            continue;
        }
        while (!r.pass_done && r.pool.size() < r.shuffle) {
            Span rec;
            int rc = read_one(r, rec);
            if (rc < 0) return -1;
            if (rc == 0) { r.pass_done = true; break; }
            r.pool.push_back(rec);
        }
        if (r.pool.empty()) {                              // pass drained: on to the next one
            if (!next_epoch(r)) { r.source_done = true; return 0; }
            r.pass_done = false;
            continue;
        }
        const size_t j = (size_t)(next_rand(r) % r.pool.size());
        out = r.pool[j];
        if (!r.pass_done) {
            int rc = read_one(r, r.pool[j]);
            if (rc < 0) return -1;
            if (rc == 0) r.pass_done = true;
        }
        if (r.pass_done) {                                 // slot j is stale: close the gap
            r.pool[j] = r.pool.back();
            r.pool.pop_back();
        }
        return 1;
    }
}


// =====================================================================================================
// Asynchronous batch pipeline: TFRecordDataset(file)[.shuffle].repeat().batch(B).map(parse).prefetch(depth) as ONE object.
// A producer thread frames (and shuffles) the records of batch after batch into a ring of `depth` slots; every slot's
// decode is cut into chunks of kPipeChunk records that a set of worker threads take from one queue — chunks of SEVERAL
// batches are in the queue together, so the workers never idle at a batch boundary (the synchronous accessors above
// fork and join the pool once or twice per batch, and the consumer's own work sits between two batches).  One pass over
// a record's wire bytes serves every requested feature: single-valued string ids (vocabulary lookups prefetched in
// groups, as recalgo_reader_id_matrix does) and fixed-length float features (FixedLenFeature((n,), float32, default)).
// =====================================================================================================
constexpr size_t kPipeChunk = 64;

struct PipeFloat {
    std::string key;
    int n;
    float def;
    bool has_def;
    size_t off;                                            // first column in a row of the float matrix
};

struct PipeSlot {
    std::vector<Span> recs;
    int64_t* ids = nullptr;                                // [B][F]
    float* floats = nullptr;                               // [B][NF]
    std::vector<int64_t> own_ids;
    std::vector<float> own_floats;
    int state = 0;                                         // 0 free, 1 decoding, 2 ready, 3 end of data, 4 error
    size_t pending = 0;                                    // chunks not yet decoded (guarded by Pipeline::m)
    bool multi = false;
    bool missing = false;                                  // the error is a required feature without a value (not bad bytes)
    std::string error;
    std::string multi_msg;                                 // names the multi-valued column (NOT an error: the batch is state 2,
                                                           // recalgo_pipeline_next answers -2 and the caller re-reads the column ragged)
};

struct Pipeline {
    Reader* r = nullptr;
    size_t B = 0, F = 0, NF = 0, depth = 0;
    std::vector<std::string> keys;                         // id columns
    std::vector<const Vocab*> vocabs;
    std::vector<PipeFloat> floats;
    std::vector<std::string_view> names;                   // ids then floats: the requested feature names
    std::vector<int> ntab;                                 // name -> index into `names` (open addressing)
    size_t tcap = 16;
    std::vector<PipeSlot> slots;
    std::mutex m;
    std::condition_variable cv_free, cv_ready, cv_task;
    std::deque<std::pair<size_t, size_t>> tasks;           // (slot, chunk)
    std::thread producer;
    std::vector<std::thread> workers;
    bool stop = false;
    uint64_t next_out = 0;                                 // sequence number of the batch the consumer takes next
    std::string error;

    int lookup(std::string_view name) const {
        size_t t = name_hash(name) & (tcap - 1);
        while (ntab[t] >= 0 && names[(size_t)ntab[t]] != name) t = (t + 1) & (tcap - 1);
        return ntab[t];
    }

    // records [i0, i1) of a slot: every requested feature from one pass over the wire bytes
    void decode_chunk(PipeSlot& S, size_t i0, size_t i1) {
        constexpr size_t kGroup = 8;                       // records whose vocabulary probes are in flight together
        struct Pending {
            const char* p;
            uint32_t len;
            uint32_t col;
            uint64_t h;
            const Vocab* vm;
        };
        std::vector<Pending> pend;
        pend.reserve(kGroup * F);
        std::vector<int> slot_of(F);
        std::vector<char> got(floats.size());
        std::vector<int> order;                            // feature index of the j-th map entry of the previous record
        auto resolve = [&]() {
            for (auto& q : pend)
                if (q.vm) S.ids[q.col] = q.vm->find_hashed(std::string_view(q.p, q.len), q.h);
            pend.clear();
        };
        bool bad = false, multi = false;
        size_t multi_col = 0;
        std::string missing;
        for (size_t i = i0; i < i1; ++i) {
            for (size_t f = 0; f < F; ++f) {
                S.ids[i * F + f] = -1;
                slot_of[f] = -1;
            }
            std::fill(got.begin(), got.end(), 0);
            size_t j = 0;
            bool ok = true, inner = true;
            ok = for_fields(S.recs[i], [&](uint32_t field, uint32_t wt, Span features, uint64_t) {
                if (field != 1 || wt != 2) return;
                inner = for_fields(features, [&](uint32_t f2, uint32_t w2, Span entry, uint64_t) {
                    if (f2 != 1 || w2 != 2) return;
                    std::string_view name;
                    Span feat;
                    if (!fast_map_entry(entry, name, feat))
                        for_fields(entry, [&](uint32_t f3, uint32_t w3, Span pl, uint64_t) {
                            if (w3 != 2) return;
                            if (f3 == 1) name = std::string_view((const char*)pl.p, pl.n);
                            else if (f3 == 2) feat = pl;
                        });
                    const size_t pos = j++;
                    int col;
                    if (pos < order.size() && order[pos] >= 0 && sv_equal(names[(size_t)order[pos]], name)) col = order[pos];
                    else {
                        col = lookup(name);
                        if (pos >= order.size()) order.resize(pos + 1, -1);
                        order[pos] = col;
                    }
                    if (col < 0) return;
                    if ((size_t)col >= F) {                // a float feature (a repeated map key: the last entry wins)
                        const PipeFloat& pf = floats[(size_t)col - F];
                        float* o = S.floats + i * NF + pf.off;
                        int filled = 0;
                        for_fields(feat, [&](uint32_t f4, uint32_t w4, Span list, uint64_t) {
                            if (f4 != 2 || w4 != 2) return;                       // FloatList
                            for_fields(list, [&](uint32_t f5, uint32_t w5, Span pl, uint64_t) {
                                if (f5 != 1) return;
                                if (w5 == 2) {
                                    for (size_t b = 0; b + 4 <= pl.n && filled < pf.n; b += 4) memcpy(&o[filled++], pl.p + b, 4);
                                } else if (w5 == 5 && filled < pf.n) {
                                    memcpy(&o[filled++], pl.p, 4);
                                }
                            });
                        });
                        got[(size_t)col - F] = filled >= pf.n ? 1 : (filled == 0 ? 0 : 2);
                        return;
                    }
                    const size_t f = (size_t)col;
                    int n = 0;
                    Span first;
                    if (!fast_single_bytes(feat, first, n))
                        for_fields(feat, [&](uint32_t f4, uint32_t w4, Span list, uint64_t) {
                            if (f4 != 1 || w4 != 2) return;                       // BytesList
                            for_fields(list, [&](uint32_t f5, uint32_t w5, Span pl, uint64_t) {
                                if (f5 != 1 || w5 != 2) return;
                                if (n++ == 0) first = pl;
                            });
                        });
                    if (n > 1 && !multi) { multi = true; multi_col = f; }
                    if (n == 0) {
                        if (slot_of[f] >= 0) pend[(size_t)slot_of[f]].vm = nullptr;
                        slot_of[f] = -1;
                        S.ids[i * F + f] = -1;
                        return;
                    }
                    const Vocab* vm = vocabs[f];
                    Pending q{(const char*)first.p, (uint32_t)first.n, (uint32_t)(i * F + f),
                              Vocab::hash_of((const char*)first.p, first.n), vm};
                    if (!vm->slots.empty()) __builtin_prefetch(&vm->slots[q.h & vm->mask]);
                    if (slot_of[f] >= 0) pend[(size_t)slot_of[f]] = q;
                    else {
                        slot_of[f] = (int)pend.size();
                        pend.push_back(q);
                    }
                }) && inner;
            });
            if (!ok || !inner) bad = true;
            for (size_t k = 0; k < floats.size(); ++k) {
                if (got[k] == 1) continue;
                const PipeFloat& pf = floats[k];
                if (got[k] == 0 && pf.has_def) {
                    for (int c = 0; c < pf.n; ++c) S.floats[i * NF + pf.off + (size_t)c] = pf.def;
                } else if (missing.empty()) {
                    missing = "feature " + pf.key + " is required but missing in a record";
                }
            }
            if ((i + 1 - i0) % kGroup == 0) resolve();
        }
        resolve();
        if (bad || multi || !missing.empty()) {
            std::lock_guard<std::mutex> lk(m);
            if (multi && !S.multi) {
                S.multi = true;
                S.multi_msg = "feature " + keys[multi_col] + " holds more than one value in a record";
            }
            if (bad && S.error.empty()) S.error = "malformed Example in a record";
            if (!missing.empty() && S.error.empty()) { S.error = missing; S.missing = true; }
        }
    }

    void work_loop() {
        for (;;) {
            std::pair<size_t, size_t> t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_task.wait(lk, [&] { return stop || !tasks.empty(); });
                if (stop) return;
                t = tasks.front();
                tasks.pop_front();
            }
            PipeSlot& S = slots[t.first];
            decode_chunk(S, t.second * kPipeChunk, std::min(S.recs.size(), (t.second + 1) * kPipeChunk));
            std::lock_guard<std::mutex> lk(m);
            if (--S.pending == 0) {
                S.state = S.error.empty() ? 2 : 4;
                cv_ready.notify_all();
            }
        }
    }

    void produce_loop() {
        for (uint64_t seq = 0;; ++seq) {
            PipeSlot& S = slots[seq % depth];
            {
                std::unique_lock<std::mutex> lk(m);
                cv_free.wait(lk, [&] { return stop || S.state == 0; });
                if (stop) return;
            }
            // (only this thread touches the reader and a free slot's record list)
            S.recs.clear();
            S.error.clear();
            S.multi = S.missing = false;
            S.multi_msg.clear();
            int rc = 1;
            for (size_t i = 0; i < B; ++i) {
                Span rec;
                rc = next_record(*r, rec);
                if (rc <= 0) break;
                S.recs.push_back(rec);
            }
            std::lock_guard<std::mutex> lk(m);
            if (rc < 0) {
                S.error = r->error;
                S.state = 4;
                cv_ready.notify_all();
                return;
            }
            if (S.recs.empty()) {
                S.state = 3;
                cv_ready.notify_all();
                return;
            }
            const size_t chunks = (S.recs.size() + kPipeChunk - 1) / kPipeChunk;
            S.pending = chunks;
            S.state = 1;
            for (size_t c = 0; c < chunks; ++c) tasks.emplace_back((size_t)(seq % depth), c);
            cv_task.notify_all();
            if (rc == 0) {                                 // the data ended inside this batch: the next slot reports the end
                // (falls through to the next iteration, whose first next_record returns 0 again)
            }
        }
    }

    ~Pipeline() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_free.notify_all();
        cv_task.notify_all();
        cv_ready.notify_all();
        if (producer.joinable()) producer.join();
        for (auto& w : workers)
            if (w.joinable()) w.join();
    }
};

}  // namespace

extern "C" {

#define EXPORT __attribute__((visibility("default")))

EXPORT int recalgo_host_abi_version(void) { return RECALGO_HOST_ABI_VERSION; }

EXPORT uint32_t recalgo_crc32c(const void* data, uint64_t n) { return crc32c((const uint8_t*)data, (size_t)n); }

// ---- vocabulary ------------------------------------------------------------------------------------
EXPORT void* recalgo_vocab_open(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return nullptr;
    auto* v = new Vocab();
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v->blob.resize((size_t)n);
    if (n && fread(v->blob.data(), 1, (size_t)n, f) != (size_t)n) {
        fclose(f);
        delete v;
        return nullptr;
    }
    fclose(f);
    size_t start = 0;
    int64_t line = 0;
    const std::string& b = v->blob;
    if (b.size() >= (1ull << 32)) {                        // slot offsets are 32-bit
        delete v;
        return nullptr;
    }
    size_t lines = 1;
    for (char c : b) lines += c == '\n';
    v->reserve(lines);
    while (start < b.size()) {
        size_t e = b.find('\n', start);
        if (e == std::string::npos) e = b.size();
        v->insert_first(std::string_view(b.data() + start, e - start), line++);
        start = e + 1;
    }
    return v;
}
EXPORT int64_t recalgo_vocab_size(const void* vocab) { return vocab ? (int64_t)((const Vocab*)vocab)->count : -1; }
EXPORT int64_t recalgo_vocab_lookup(const void* vocab, const char* key, uint64_t len) {
    return ((const Vocab*)vocab)->find(std::string_view(key, (size_t)len));
}
EXPORT void recalgo_vocab_close(void* vocab) { delete (Vocab*)vocab; }

// ---- TFRecord reader -------------------------------------------------------------------------------
EXPORT void* recalgo_reader_open(const char* path, int verify_crc) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return nullptr;
    }
    auto* r = new Reader();
    r->fd = fd;
    r->size = (size_t)st.st_size;
    r->verify = verify_crc != 0;
    if (r->size) {
        void* m = mmap(nullptr, r->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            delete r;
            return nullptr;
        }
        madvise(m, r->size, MADV_SEQUENTIAL);
        r->map = (const uint8_t*)m;
    }
    return r;
}
EXPORT void recalgo_reader_close(void* reader) {
    auto* r = (Reader*)reader;
    if (!r) return;
    if (r->map) munmap((void*)r->map, r->size);
    if (r->fd >= 0) close(r->fd);
    delete r;
}
EXPORT const char* recalgo_reader_error(const void* reader) { return ((const Reader*)reader)->error.c_str(); }
EXPORT int recalgo_reader_rewind(void* reader) {
    auto* r = (Reader*)reader;
    r->epoch = 0;
    r->source_done = false;
    r->pass_done = false;
    r->pool.clear();
    r->pos = 0;
    return 0;
}

// Reads up to max_records records into the reader's batch buffer.  Returns the number read
// (0 at end of file), -1 on a framing / crc error (see recalgo_reader_error).
EXPORT int64_t recalgo_reader_next_batch(void* reader, int64_t max_records) {
    auto* r = (Reader*)reader;
    r->recs.clear();
    r->indexed = false;
    r->scans = 0;
    r->error.clear();
    for (int64_t i = 0; i < max_records; ++i) {
        Span rec;
        int rc = next_record(*r, rec);
        if (rc < 0) return -1;
        if (rc == 0) break;
        r->recs.push_back(rec);
    }
    return (int64_t)r->recs.size();
}

// dataset.repeat(num_epochs) (num_epochs < 0: forever) and dataset.shuffle(buffer_size) with a seed;
// call before the first recalgo_reader_next_batch.
EXPORT void recalgo_reader_configure(void* reader, int64_t num_epochs, int64_t shuffle_buffer_size, uint64_t seed) {
    auto* r = (Reader*)reader;
    r->epochs = num_epochs;
    r->shuffle = shuffle_buffer_size > 0 ? (size_t)shuffle_buffer_size : 0;
    r->rng = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    if (r->rng == 0) r->rng = 1;
}

// FixedLenFeature((n,), float32, default): out[B, n].  Returns 0, or -1 when a record lacks the
// feature (or holds fewer than n values) and has_default == 0.
EXPORT int recalgo_reader_float_feature(void* reader, const char* key, int n, float default_value, int has_default,
                                        float* out) {
    auto* r = (Reader*)reader;
    bool use_index;
    if (!prepare_access(*r, use_index)) return -1;
    const std::string_view k(key);
    const size_t B = r->recs.size();
    std::atomic<long> missing{-1}, bad{-1};
    Pool::get().parallel_for((B + kChunk - 1) / kChunk, [&](size_t c) {
        size_t hint = 0;
        for (size_t i = c * kChunk; i < std::min(B, (c + 1) * kChunk); ++i) {
            float* o = out + i * (size_t)n;
            int filled = 0;
            Span scanned;
            const Span* feat = nullptr;
            if (use_index) feat = find_feature(r->index[i], k, hint);
            else {
                bool ok = true;
                if (scan_feature(r->recs[i], k, scanned, ok)) feat = &scanned;
                if (!ok) bad.store((long)i);
            }
            if (feat) {
                for_fields(*feat, [&](uint32_t field, uint32_t wt, Span list, uint64_t) {
                    if (field != 2 || wt != 2) return;                       // FloatList
                    for_fields(list, [&](uint32_t f2, uint32_t w2, Span pl, uint64_t) {
                        if (f2 != 1) return;
                        if (w2 == 2) {                                       // packed
                            for (size_t b = 0; b + 4 <= pl.n && filled < n; b += 4) memcpy(&o[filled++], pl.p + b, 4);
                        } else if (w2 == 5 && filled < n) {
                            memcpy(&o[filled++], pl.p, 4);
                        }
                    });
                });
            }
            if (filled < n) {
                if (!(filled == 0 && has_default)) {
                    long want = -1;
                    missing.compare_exchange_strong(want, (long)i);
                    continue;
                }
                for (int j = 0; j < n; ++j) o[j] = default_value;
            }
        }
    });
    if (bad.load() >= 0) {
        r->error = "malformed Example in record " + std::to_string(bad.load());
        return -1;
    }
    if (missing.load() >= 0) {
        r->error = std::string("feature ") + key + " is required but missing in record " + std::to_string(missing.load());
        return -1;
    }
    return 0;
}

// VarLenFeature(string) -> vocabulary ids: offsets[B+1], values[nnz] (-1 = key not in vocabulary).
// Returns nnz; when values_cap < nnz only the offsets are written (call again with a larger buffer).
EXPORT int64_t recalgo_reader_id_feature(void* reader, const char* key, const void* vocab, int64_t* offsets,
                                         int64_t* values, int64_t values_cap) {
    auto* r = (Reader*)reader;
    bool use_index;
    if (!prepare_access(*r, use_index)) return -1;
    const Vocab& vm = *(const Vocab*)vocab;
    const std::string_view k(key);
    const size_t B = r->recs.size();
    const size_t chunks = (B + kChunk - 1) / kChunk;
    std::atomic<long> bad{-1};
    // visit the byte strings of record i's feature
    auto each_value = [&](size_t i, size_t& hint, auto&& fn) {
        Span scanned;
        const Span* feat = nullptr;
        if (use_index) feat = find_feature(r->index[i], k, hint);
        else {
            bool ok = true;
            if (scan_feature(r->recs[i], k, scanned, ok)) feat = &scanned;
            if (!ok) bad.store((long)i);
        }
        if (feat) {
            for_fields(*feat, [&](uint32_t field, uint32_t wt, Span list, uint64_t) {
                if (field != 1 || wt != 2) return;                           // BytesList
                for_fields(list, [&](uint32_t f2, uint32_t w2, Span pl, uint64_t) {
                    if (f2 == 1 && w2 == 2) fn(pl);
                });
            });
        }
    };
    // pass 1: value counts per record -> offsets (serial prefix sum); pass 2: lookups, in place
    offsets[0] = 0;
    Pool::get().parallel_for(chunks, [&](size_t c) {
        size_t hint = 0;
        for (size_t i = c * kChunk; i < std::min(B, (c + 1) * kChunk); ++i) {
            int64_t cnt = 0;
            each_value(i, hint, [&](Span) { ++cnt; });
            offsets[i + 1] = cnt;
        }
    });
    if (bad.load() >= 0) {
        r->error = "malformed Example in record " + std::to_string(bad.load());
        return -1;
    }
    for (size_t i = 0; i < B; ++i) offsets[i + 1] += offsets[i];
    const int64_t nnz = offsets[B];
    if (nnz > values_cap) return nnz;
    Pool::get().parallel_for(chunks, [&](size_t c) {
        size_t hint = 0;
        for (size_t i = c * kChunk; i < std::min(B, (c + 1) * kChunk); ++i) {
            int64_t at = offsets[i];
            each_value(i, hint, [&](Span pl) {
                values[at++] = vm.find(std::string_view((const char*)pl.p, pl.n));
            });
        }
    });
    return nnz;
}

// All single-valued id features of the current batch in ONE parallel pass over the records:
// out [B, n_keys] int64 row-major (column f = keys[f] looked up in vocabs[f]); a record without a value for
// a key gives -1.  A feature that holds MORE than one value in some record sets multi[f] = 1 (its column
// then holds the first value; use recalgo_reader_id_feature for that key).  Returns 0, -1 on a malformed batch.
EXPORT int recalgo_reader_id_matrix(void* reader, int n_keys, const char* const* keys, const void* const* vocabs,
                                    int64_t* out, int32_t* multi) {
    // ONE pass over the wire bytes of every record (no per-record index): each map entry's name is matched against
    // the requested keys through a small hash table, the first byte string of its BytesList is hashed and the
    // vocabulary slot it maps to is PREFETCHED; the probes of a group of records are resolved afterwards, so the
    // cache misses of the 26 lookups x kGroup records overlap instead of queueing (the slot holds the key bytes
    // inline, so a lookup in a 10^6-key vocabulary is ONE miss).
    auto* r = (Reader*)reader;
    if (n_keys < 0 || (n_keys > 0 && (!keys || !vocabs || !out || !multi))) return -1;
    const size_t B = r->recs.size(), F = (size_t)n_keys;
    std::vector<std::string_view> ks(F);
    for (size_t f = 0; f < F; ++f) {
        ks[f] = std::string_view(keys[f]);
        multi[f] = 0;
    }
    // name -> column: open addressing, load <= 1/4; a key given twice maps to its first column (the others stay -1)
    size_t tcap = 16;
    while (tcap < 4 * F + 4) tcap <<= 1;
    std::vector<int> ntab(tcap, -1);
    for (size_t f = 0; f < F; ++f) {
        size_t i = name_hash(ks[f]) & (tcap - 1);
        while (ntab[i] >= 0 && ks[(size_t)ntab[i]] != ks[f]) i = (i + 1) & (tcap - 1);
        if (ntab[i] < 0) ntab[i] = (int)f;
    }
    std::vector<std::atomic<int>> mflag(F);
    for (auto& m : mflag) m.store(0);
    std::atomic<long> bad{-1};
    constexpr size_t kGroup = 8;                               // records whose lookups are in flight together
    struct Pending {
        const char* p;
        uint32_t len;
        uint32_t col;                                          // out index (record * F + column)
        uint64_t h;
        const Vocab* vm;
    };
    Pool::get().parallel_for((B + kChunk - 1) / kChunk, [&](size_t c) {
        std::vector<Pending> pend;
        pend.reserve(kGroup * F);
        std::vector<int> slot_of(F);                           // this record's pending entry per column (-1 none)
        std::vector<int> order;                                // column of the j-th map entry of the previous record (-1: not
                                                               // requested): one writer lists its features in one order
        const size_t i_end = std::min(B, (c + 1) * kChunk);
        auto resolve = [&]() {                               // the slots were prefetched while the records were parsed
            for (auto& q : pend) out[q.col] = q.vm->find_hashed(std::string_view(q.p, q.len), q.h);
            pend.clear();
        };
        for (size_t i = c * kChunk; i < i_end; ++i) {
            for (size_t f = 0; f < F; ++f) {
                out[i * F + f] = -1;
                slot_of[f] = -1;
            }
            const uint8_t* p = r->recs[i].p;
            const uint8_t* end = p + r->recs[i].n;
            bool ok = true;
            size_t j = 0;                                      // map entry counter of this record
            // Example { features = 1 }  (SequenceExample.feature_lists = 2 and anything else is skipped)
            ok = for_fields(Span{p, (size_t)(end - p)}, [&](uint32_t field, uint32_t wt, Span features, uint64_t) {
                if (field != 1 || wt != 2) return;
                ok = for_fields(features, [&](uint32_t f2, uint32_t w2, Span entry, uint64_t) {
                    if (f2 != 1 || w2 != 2) return;
                    std::string_view name;
                    Span feat;
                    if (!fast_map_entry(entry, name, feat))
                        for_fields(entry, [&](uint32_t f3, uint32_t w3, Span pl, uint64_t) {
                            if (w3 != 2) return;
                            if (f3 == 1) name = std::string_view((const char*)pl.p, pl.n);
                            else if (f3 == 2) feat = pl;
                        });
                    const size_t pos = j++;
                    int col;
                    if (pos < order.size() && order[pos] >= 0 && sv_equal(ks[(size_t)order[pos]], name)) {
                        col = order[pos];
                    } else {
                        size_t t = name_hash(name) & (tcap - 1);
                        while (ntab[t] >= 0 && ks[(size_t)ntab[t]] != name) t = (t + 1) & (tcap - 1);
                        col = ntab[t];
                        if (pos >= order.size()) order.resize(pos + 1, -1);
                        order[pos] = col;
                    }
                    if (col < 0) return;                       // not a requested key
                    const size_t f = (size_t)col;
                    int n = 0;
                    Span first;
                    if (!fast_single_bytes(feat, first, n))
                        for_fields(feat, [&](uint32_t f4, uint32_t w4, Span list, uint64_t) {
                            if (f4 != 1 || w4 != 2) return;                       // BytesList
                            for_fields(list, [&](uint32_t f5, uint32_t w5, Span pl, uint64_t) {
                                if (f5 != 1 || w5 != 2) return;
                                if (n++ == 0) first = pl;
                            });
                        });
                    if (n > 1) mflag[f].store(1, std::memory_order_relaxed);
                    // a repeated map key: the last entry wins (protobuf map semantics) -> overwrite this column's entry
                    if (n == 0) {
                        if (slot_of[f] >= 0) pend[(size_t)slot_of[f]].vm = nullptr;
                        slot_of[f] = -1;
                        out[i * F + f] = -1;
                        return;
                    }
                    const Vocab* vm = (const Vocab*)vocabs[f];
                    Pending q{(const char*)first.p, (uint32_t)first.n, (uint32_t)(i * F + f),
                              Vocab::hash_of((const char*)first.p, first.n), vm};
                    if (!vm->slots.empty()) __builtin_prefetch(&vm->slots[q.h & vm->mask]);
                    if (slot_of[f] >= 0) pend[(size_t)slot_of[f]] = q;
                    else {
                        slot_of[f] = (int)pend.size();
                        pend.push_back(q);
                    }
                }) && ok;
            }) && ok;
            if (!ok) bad.store((long)i);
            if ((i + 1 - c * kChunk) % kGroup == 0) {
                pend.erase(std::remove_if(pend.begin(), pend.end(), [](const Pending& q) { return q.vm == nullptr; }), pend.end());
                resolve();
            }
        }
        pend.erase(std::remove_if(pend.begin(), pend.end(), [](const Pending& q) { return q.vm == nullptr; }), pend.end());
        resolve();
    });
    if (bad.load() >= 0) {
        r->error = "malformed Example in record " + std::to_string(bad.load());
        return -1;
    }
    for (size_t f = 0; f < F; ++f) multi[f] = mflag[f].load();
    return 0;
}


// ---- asynchronous batch pipeline (see struct Pipeline) -------------------------------------------------------------
// Opens `path` and starts decoding ahead.  id_keys[n_ids] / vocabs[n_ids]: single-valued string features -> the int64
// [B, n_ids] matrix of a slot (column order as given); float_keys[n_floats] with float_n / float_default /
// float_has_default: FixedLenFeature((n,), float32[, default]) -> the float32 [B, sum n] matrix.  depth >= 2 slots;
// ids_slots / float_slots: depth caller-owned buffers each (e.g. pinned), or NULL: the pipeline's own.  threads <= 0:
// RECALGO_READER_THREADS, else half the hardware threads (2 .. 32).  NULL on error.
EXPORT void* recalgo_pipeline_open(const char* path, int verify_crc, int64_t num_epochs, int64_t shuffle_buffer_size,
                                   uint64_t seed, int64_t batch_size, int n_ids, const char* const* id_keys,
                                   const void* const* vocabs, int n_floats, const char* const* float_keys,
                                   const int32_t* float_n, const float* float_default, const int32_t* float_has_default,
                                   int depth, int64_t* const* ids_slots, float* const* float_slots, int threads) {
    if (!path || batch_size < 1 || n_ids < 0 || n_floats < 0 || depth < 2 || depth > 64) return nullptr;
    if ((n_ids > 0 && (!id_keys || !vocabs)) || (n_floats > 0 && (!float_keys || !float_n || !float_default || !float_has_default)))
        return nullptr;
    void* rh = recalgo_reader_open(path, verify_crc);
    if (!rh) return nullptr;
    recalgo_reader_configure(rh, num_epochs, shuffle_buffer_size, seed);
    auto* P = new Pipeline();
    P->r = (Reader*)rh;
    P->B = (size_t)batch_size;
    P->F = (size_t)n_ids;
    P->depth = (size_t)depth;
    for (int f = 0; f < n_ids; ++f) {
        P->keys.emplace_back(id_keys[f]);
        P->vocabs.push_back((const Vocab*)vocabs[f]);
    }
    size_t off = 0;
    for (int k = 0; k < n_floats; ++k) {
        if (float_n[k] < 1) { delete P; recalgo_reader_close(rh); return nullptr; }
        P->floats.push_back(PipeFloat{float_keys[k], float_n[k], float_default[k], float_has_default[k] != 0, off});
        off += (size_t)float_n[k];
    }
    P->NF = off;
    for (auto& k : P->keys) P->names.emplace_back(k);
    for (auto& pf : P->floats) P->names.emplace_back(pf.key);
    while (P->tcap < 4 * P->names.size() + 4) P->tcap <<= 1;
    P->ntab.assign(P->tcap, -1);
    for (size_t f = 0; f < P->names.size(); ++f) {           // a name given twice maps to its first entry
        size_t i = name_hash(P->names[f]) & (P->tcap - 1);
        while (P->ntab[i] >= 0 && P->names[(size_t)P->ntab[i]] != P->names[f]) i = (i + 1) & (P->tcap - 1);
        if (P->ntab[i] < 0) P->ntab[i] = (int)f;
    }
    P->slots.resize(P->depth);
    for (size_t d = 0; d < P->depth; ++d) {
        PipeSlot& S = P->slots[d];
        if (ids_slots && ids_slots[d]) S.ids = ids_slots[d];
        else {
            S.own_ids.resize(P->B * std::max<size_t>(P->F, 1));
            S.ids = S.own_ids.data();
        }
        if (float_slots && float_slots[d]) S.floats = float_slots[d];
        else {
            S.own_floats.resize(P->B * std::max<size_t>(P->NF, 1));
            S.floats = S.own_floats.data();
        }
    }
    size_t n = 0;
    if (threads > 0) n = (size_t)threads;
    else if (const char* e = std::getenv("RECALGO_READER_THREADS")) {
        long v = std::atol(e);
        if (v >= 1 && v <= 256) n = (size_t)v;
    }
    if (n == 0) {
        // (measured on a 256-thread host feeding one GPU: 32 decode threads 10.6 M examples/s end to end, 64 threads 9.7 M —
        // the decoders then crowd the training loop's own thread)
        n = std::thread::hardware_concurrency() / 2;
        n = n < 2 ? 2 : (n > 32 ? 32 : n);
    }
    for (size_t i = 0; i < n; ++i) P->workers.emplace_back([P] { P->work_loop(); });
    P->producer = std::thread([P] { P->produce_loop(); });
    return P;
}

// Waits for the next batch (in order).  Returns its number of records (> 0) with *slot = the ring slot holding it
// (ids at ids_slots[slot] / recalgo_pipeline_ids, floats likewise) — valid until recalgo_pipeline_release(slot);
// 0 at the end of the data; -1 on a framing error or a malformed Example, -3 when a required float feature is missing
// (recalgo_pipeline_error says which); -2 when some id feature holds more than one value in a record (not a single-valued
// feature: decode it with the synchronous reader instead).
EXPORT int64_t recalgo_pipeline_next(void* pipeline, int* slot) {
    auto* P = (Pipeline*)pipeline;
    const size_t d = (size_t)(P->next_out % P->depth);
    PipeSlot& S = P->slots[d];
    std::unique_lock<std::mutex> lk(P->m);
    P->cv_ready.wait(lk, [&] { return S.state >= 2; });
    if (S.state == 3) return 0;
    if (S.state == 4) {
        P->error = S.error;
        return S.missing ? -3 : -1;
    }
    if (S.multi) {
        P->error = S.multi_msg;                                 // (names the column)
        return -2;
    }
    ++P->next_out;
    if (slot) *slot = (int)d;
    return (int64_t)S.recs.size();
}

EXPORT void recalgo_pipeline_release(void* pipeline, int slot) {
    auto* P = (Pipeline*)pipeline;
    if (slot < 0 || (size_t)slot >= P->depth) return;
    {
        std::lock_guard<std::mutex> lk(P->m);
        if (P->slots[(size_t)slot].state == 2) P->slots[(size_t)slot].state = 0;
    }
    P->cv_free.notify_all();
}

EXPORT const int64_t* recalgo_pipeline_ids(void* pipeline, int slot) { return ((Pipeline*)pipeline)->slots[(size_t)slot].ids; }
EXPORT const float* recalgo_pipeline_floats(void* pipeline, int slot) { return ((Pipeline*)pipeline)->slots[(size_t)slot].floats; }
EXPORT const char* recalgo_pipeline_error(void* pipeline) { return ((Pipeline*)pipeline)->error.c_str(); }
EXPORT int recalgo_pipeline_threads(void* pipeline) { return (int)((Pipeline*)pipeline)->workers.size(); }

EXPORT void recalgo_pipeline_close(void* pipeline) {
    auto* P = (Pipeline*)pipeline;
    if (!P) return;
    Reader* r = P->r;
    delete P;                                                // (joins the threads)
    recalgo_reader_close(r);
}

}  // extern "C"
