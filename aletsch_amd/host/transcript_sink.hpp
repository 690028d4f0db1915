// transcript_sink.hpp -- host-side result sink with the merge semantics of the reference's transcript_set
// (rnacore/transcript_set.cc:38-175; ordering and hashing from gtf/transcript.cc:183-308, util/util.cc:38-46).
//
// Layout is this library's own: a transcript's exons are ONE flat coordinate list (l0 r0 l1 r1 ...), the same words the record
// and transcript streams carry, so hashing and comparing walk a contiguous span; the numeric attributes that a merge only ever
// raises live in one `peak` block with a single raise(); per-sample state is a small sorted array.  What is dictated by bit-exact
// parity, and therefore kept: which coordinates a comparison looks at (the reference never looks at the last-but-one exon of a
// multi-exon chain, transcript.cc:236-245), the FORWARD walk of a bucket merge (the single-exon overlap test is not transitive,
// so the sequence of comparisons is behaviour), coverage adding for multi-exon and taking the maximum for single-exon
// transcripts, the first inserted transcript keeping its id.
//
// Pinned against the reference's own transcript_set.cc built from source (oracle/_ref/ref_tset, tests/golden/ref_tset.json).
// In the multi-GPU flow rank 0 feeds the gathered transcript streams in ascending global graph id, so the result does not depend
// on the number of ranks.
#pragma once
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <cstdint>
#include <cstddef>
#include <utility>
#include <iterator>
#include <memory>
#include <new>

namespace aletsch {

// Containers of the sink.  A merge into a persistent set of a million items is bound by cache and TLB misses, not by arithmetic: with
// std::unordered_map<key, std::vector<item>> and std::vector members an item was five separate heap blocks (map node, bucket array,
// item, exon list, sample list), i.e. five dependent misses per transcript.  Here the first N elements of a list live INSIDE their
// owner (a bucket holds one item, an item its first sample and up to eight exons: the common case), and the table is an index array
// over entries that never move: one miss for the slot, one or two for the entry.
template<class T, unsigned N> class small_vec {
public:
    typedef T value_type; typedef T *iterator; typedef const T *const_iterator;
    small_vec() {}
    small_vec(const small_vec &o) { append_copy(o.p_, o.n_); }
    small_vec(small_vec &&o) noexcept { take(o); }
    small_vec &operator=(const small_vec &o) { if(this != &o) { clear(); append_copy(o.p_, o.n_); } return *this; }
    small_vec &operator=(small_vec &&o) noexcept { if(this != &o) { clear(); release(); take(o); } return *this; }
    ~small_vec() { clear(); release(); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; } const T *data() const { return p_; }
    iterator begin() { return p_; } iterator end() { return p_ + n_; }
    const_iterator begin() const { return p_; } const_iterator end() const { return p_ + n_; }
    T &operator[](size_t i) { return p_[i]; } const T &operator[](size_t i) const { return p_[i]; }
    T &front() { return p_[0]; } const T &front() const { return p_[0]; }
    T &back() { return p_[n_ - 1]; } const T &back() const { return p_[n_ - 1]; }
    void clear() { for(uint32_t i = 0; i < n_; i++) p_[i].~T(); n_ = 0; }
    void reserve(size_t want) { if(want > cap_) grow(want); }
    void push_back(const T &v) { emplace_back(v); }
    void push_back(T &&v) { emplace_back(std::move(v)); }
    template<class... A> T &emplace_back(A &&... a) {
        if(n_ == cap_) { T fresh(std::forward<A>(a)...); grow((size_t)cap_ * 2); T *q = new(p_ + n_) T(std::move(fresh)); n_++; return *q; }   // built BEFORE the storage moves: v.push_back(v[i]) must work as it does for std::vector
        T *q = new(p_ + n_) T(std::forward<A>(a)...); n_++; return *q;
    }
    template<class It> void assign(It a, It b) { clear(); const size_t k = (size_t)(b - a); reserve(k); for(size_t i = 0; i < k; i++, ++a) new(p_ + i) T(*a); n_ = (uint32_t)k; }
    // a new element in front of `pos`: the tail moves up by one
    template<class... A> iterator emplace(iterator pos, A &&... a) {
        const size_t at = (size_t)(pos - p_);
        T fresh(std::forward<A>(a)...);                                       // built first: the arguments may refer to elements of this list
        if(n_ == cap_) grow((size_t)cap_ * 2);
        if(at == n_) new(p_ + n_) T(std::move(fresh));
        else { new(p_ + n_) T(std::move(p_[n_ - 1])); for(size_t i = n_ - 1; i > at; i--) p_[i] = std::move(p_[i - 1]); p_[at] = std::move(fresh); }
        n_++;
        return p_ + at;
    }
    iterator insert(iterator pos, const T &v) { return emplace(pos, v); }
    void swap(small_vec &o) { small_vec t(std::move(o)); o = std::move(*this); *this = std::move(t); }
private:
    T *inl() { return reinterpret_cast<T*>(buf_); }
    bool on_heap() const { return p_ != reinterpret_cast<const T*>(buf_); }
    void release() { if(on_heap()) ::operator delete((void*)p_); p_ = inl(); cap_ = N; }
    void grow(size_t want) {
        if(want < (size_t)N * 2) want = (size_t)N * 2;
        T *q = static_cast<T*>(::operator new(want * sizeof(T)));
        for(uint32_t i = 0; i < n_; i++) { new(q + i) T(std::move(p_[i])); p_[i].~T(); }
        if(on_heap()) ::operator delete((void*)p_);
        p_ = q; cap_ = (uint32_t)want;
    }
    void append_copy(const T *q, uint32_t k) { reserve(k); for(uint32_t i = 0; i < k; i++) new(p_ + i) T(q[i]); n_ = k; }
    void take(small_vec &o) {                                                 // *this is empty and inline
        if(o.on_heap()) { p_ = o.p_; cap_ = o.cap_; n_ = o.n_; o.p_ = o.inl(); o.cap_ = N; o.n_ = 0; }
        else { for(uint32_t i = 0; i < o.n_; i++) { new(p_ + i) T(std::move(o.p_[i])); o.p_[i].~T(); } n_ = o.n_; o.n_ = 0; }
    }
    alignas(T) unsigned char buf_[sizeof(T) * N];
    T *p_ = reinterpret_cast<T*>(buf_); uint32_t n_ = 0, cap_ = N;
};

// key -> bucket: an open-addressing index (4 bytes per slot, at most half full) over entries kept in fixed chunks, so an entry never
// moves and a reference to a bucket stays valid while others are added.  Iteration is in insertion order (callers that need the
// reference's ascending-key order sort the keys: transcript_sink::sorted_keys).  Nothing is allocated before the first insertion: the
// per-graph sets of the merge loop come and go by the hundred thousand.
template<class B> class chain_table {
public:
    struct entry { size_t first; B second; entry(size_t k) : first(k) {} entry(size_t k, B &&b) : first(k), second(std::move(b)) {} };
    class iterator {
    public:
        iterator(chain_table *t, size_t i) : t_(t), i_(i) {}
        entry &operator*() const { return t_->at(i_); } entry *operator->() const { return &t_->at(i_); }
        iterator &operator++() { ++i_; return *this; }
        bool operator==(const iterator &o) const { return i_ == o.i_; } bool operator!=(const iterator &o) const { return i_ != o.i_; }
    private:
        chain_table *t_; size_t i_;
    };
    chain_table() {}
    chain_table(const chain_table &o) { for(size_t i = 0; i < o.n_; i++) { const entry &e = const_cast<chain_table&>(o).at(i); B copy(e.second); emplace(e.first, std::move(copy)); } }
    chain_table(chain_table &&o) noexcept : chunks_(std::move(o.chunks_)), slot_(std::move(o.slot_)), n_(o.n_) { o.n_ = 0; o.chunks_.clear(); o.slot_.clear(); }
    chain_table &operator=(chain_table o) { clear(); chunks_.swap(o.chunks_); slot_.swap(o.slot_); std::swap(n_, o.n_); return *this; }
    ~chain_table() { clear(); }
    size_t size() const { return n_; }
    iterator begin() { return iterator(this, 0); } iterator end() { return iterator(this, n_); }
    iterator begin() const { return iterator(const_cast<chain_table*>(this), 0); } iterator end() const { return iterator(const_cast<chain_table*>(this), n_); }
    iterator find(size_t key) const { const size_t i = probe(key); return iterator(const_cast<chain_table*>(this), i == NONE ? n_ : i); }
    B &operator[](size_t key) { const size_t i = probe(key); return i != NONE ? at(i).second : append(key)->second; }
    void emplace(size_t key, B &&b) { if(probe(key) == NONE) append(key)->second = std::move(b); }
    // a merge loop that knows the keys it will ask for next can have the slot, then the entry, on their way into the cache before it
    // needs them (two dependent misses per lookup otherwise): prefetch_slot(key) some iterations ahead, prefetch_entry(key) a few
    void prefetch_slot(size_t key) const { if(!slot_.empty()) __builtin_prefetch(&slot_[mix(key) & (slot_.size() - 1)]); }
    void prefetch_entry(size_t key) const {
        if(slot_.empty()) return;
        const size_t i = probe(key); if(i == NONE) return;
        const char *e = reinterpret_cast<const char*>(&const_cast<chain_table*>(this)->at(i));
        __builtin_prefetch(e); __builtin_prefetch(e + 64); __builtin_prefetch(e + 128); __builtin_prefetch(e + 192);
    }
    void clear() {
        for(size_t i = 0; i < n_; i++) at(i).~entry();
        for(auto &c : chunks_) ::operator delete((void*)c);
        chunks_.clear(); slot_.clear(); n_ = 0;
    }
private:
    enum : size_t { NONE = ~(size_t)0, CHUNK_FIRST = 4 };
    // chunk k holds CHUNK_FIRST << k entries (a one-graph set costs one small block, a million-item table twenty blocks)
    static void locate(size_t i, size_t &c, size_t &o) { const size_t q = i / CHUNK_FIRST + 1; c = 63 - (size_t)__builtin_clzll((unsigned long long)q); o = i - (((size_t)1 << c) - 1) * CHUNK_FIRST; }
    entry &at(size_t i) { size_t c, o; locate(i, c, o); return chunks_[c][o]; }
    static size_t mix(size_t k) { k *= 0x9E3779B97F4A7C15ull; return k ^ (k >> 29); }
    size_t probe(size_t key) const {
        if(slot_.empty()) { for(size_t i = 0; i < n_; i++) if(const_cast<chain_table*>(this)->at(i).first == key) return i; return NONE; }     // a handful of entries: no index yet
        const size_t mask = slot_.size() - 1;
        for(size_t h = mix(key) & mask; ; h = (h + 1) & mask) { const uint32_t s = slot_[h]; if(s == 0) return NONE; if(const_cast<chain_table*>(this)->at(s - 1).first == key) return s - 1; }
    }
    entry *append(size_t key) {
        size_t c, o; locate(n_, c, o);
        if(c == chunks_.size()) chunks_.push_back(static_cast<entry*>(::operator new((CHUNK_FIRST << c) * sizeof(entry))));
        entry *e = new(&chunks_[c][o]) entry(key);
        n_++;
        if(n_ > 8) {
            if(2 * n_ > slot_.size()) { size_t cap = slot_.empty() ? 32 : slot_.size() * 2; slot_.assign(cap, 0); for(size_t i = 0; i < n_; i++) place(at(i).first, i); }
            else place(key, n_ - 1);
        }
        return e;
    }
    void place(size_t key, size_t i) { const size_t mask = slot_.size() - 1; size_t h = mix(key) & mask; while(slot_[h]) h = (h + 1) & mask; slot_[h] = (uint32_t)(i + 1); }
    std::vector<entry*> chunks_; std::vector<uint32_t> slot_; size_t n_ = 0;
};

// attributes a merge can only raise (trans_item::merge takes the maximum of each: transcript_set.cc:47-50, 63-66)
struct peak {
    double cov2 = 0, conf = 0, abd = 0; int count1 = 0;
    void raise(const peak &o) { if(cov2 < o.cov2) cov2 = o.cov2; if(conf < o.conf) conf = o.conf; if(abd < o.abd) abd = o.abd; if(count1 < o.count1) count1 = o.count1; }
};

// the reference's comparators answer +1 when the left operand sorts first, -1 when the right one does
template<class T> inline int first_is(const T &a, const T &b) { return a < b ? +1 : (b < a ? -1 : 0); }

struct sink_transcript {
    char strand = '.';
    double coverage = 0;
    peak top;                                    // cov2 / conf / abd / count1
    int count2 = 0;
    int64_t tid = 0;                             // transcript_id (the reference keeps a string "chr<chrm>.<gid>.<i>")
    small_vec<int32_t, 16> xs;                   // exon coordinates, flat: l0 r0 l1 r1 ... (up to eight exons inside the item)

    size_t n_exons() const { return xs.size() / 2; }
    void add_exon(int32_t l, int32_t r) { xs.push_back(l); xs.push_back(r); }
    int32_t span_len() const { int32_t s = 0; for(size_t k = 0; k + 1 < xs.size(); k += 2) s += xs[k + 1] - xs[k]; return s; }

    // bucket key (transcript.cc:183-201 over util.cc:38-46): single exon -> its mid-point bin; otherwise a boost-style hash_combine
    // over the inner coordinates r0 l1 r1 ... l_last, i.e. the flat list without its first and last word
    static size_t chain_key(const int32_t *x, size_t n_words) {
        if(n_words < 2) return 0;
        if(n_words == 2) return (size_t)((x[0] + x[1]) / 10000) + 1;
        size_t h = n_words - 2;
        for(size_t k = 1; k + 1 < n_words; k++) h ^= (size_t)(x[k]) + 0x9e3779b9 + (h << 6) + (h >> 2);
        return (h & 0x7FFFFFFF) + 1;
    }
    size_t chain_key() const { return chain_key(xs.data(), xs.size()); }

    // transcript::compare1 (transcript.cc:269-300) for one seqname.  Coordinates looked at for a chain of m >= 2 exons, in this
    // order: word 1, words 2 .. 2m-5, word 2m-2 -- the reference's loop stops one exon early (transcript.cc:236-245).
    int order_against(const sink_transcript &o, double single_exon_overlap) const {
        if(int c = first_is(xs.size(), o.xs.size())) return c;
        if(int c = first_is(strand, o.strand)) return c;
        const size_t w = xs.size();
        if(w < 2) return 0;
        if(w == 2) {
            const int32_t lo = std::max(xs[0], o.xs[0]), hi = std::min(xs[1], o.xs[1]);
            const int32_t shared = hi - lo;
            if(shared >= single_exon_overlap * span_len() || shared >= single_exon_overlap * o.span_len()) return 0;
            if(int c = first_is(xs[0], o.xs[0])) return c;
            return first_is(xs[1], o.xs[1]);
        }
        if(int c = first_is(xs[1], o.xs[1])) return c;
        for(size_t k = 2; k + 5 <= w; k++) if(int c = first_is(xs[k], o.xs[k])) return c;
        return first_is(xs[w - 2], o.xs[w - 2]);
    }
    void widen_to(const sink_transcript &o) {    // transcript::extend_bounds (transcript.cc:302-308)
        if(xs.empty()) return;
        xs.front() = std::min(xs.front(), o.xs.front());
        xs.back() = std::max(xs.back(), o.xs.back());
    }
};

struct sink_sample { double coverage = 0; peak top; int count2 = 0; };    // the fields of trans_item::samples[sid] that a merge reads or writes

// a handful of samples per item: sorted array with the find / insert / ordered walk a std::map<int, ...> gives
struct sample_map {
    typedef std::pair<int, sink_sample> value_type;
    typedef small_vec<value_type, 1> list;       // the first sample inside the item
    list v;
    size_t size() const { return v.size(); }
    list::iterator begin() { return v.begin(); }
    list::iterator end() { return v.end(); }
    list::const_iterator begin() const { return v.begin(); }
    list::const_iterator end() const { return v.end(); }
    // the slot of sample `sid`, created from `fresh` when absent; `was_new` tells which
    sink_sample &slot(int sid, const sink_sample &fresh, bool &was_new) {
        auto it = std::lower_bound(v.begin(), v.end(), sid, [](const value_type &a, int key) { return a.first < key; });
        was_new = (it == v.end() || it->first != sid);
        if(was_new) it = v.insert(it, value_type(sid, fresh));
        return it->second;
    }
};

struct sink_item {                               // trans_item (transcript_set.h:20-33)
    sink_transcript trst; int count = 0;
    sample_map samples;
    sink_item() {}
    sink_item(const sink_transcript &t, int c, int sid) : trst(t), count(c) { bool fresh; samples.slot(sid, as_sample(t), fresh); }

    static sink_sample as_sample(const sink_transcript &t) { sink_sample s; s.coverage = t.coverage; s.top = t.top; s.count2 = 1; return s; }

    // trans_item::merge in TRANSCRIPT_COUNT_ADD_COVERAGE_ADD mode (transcript_set.cc:38-75), as two steps shared by both entry points
    void absorb_head(const sink_transcript &t, int c) {
        if(trst.n_exons() >= 2) trst.coverage += t.coverage;
        else trst.coverage = std::max(trst.coverage, t.coverage);
        trst.widen_to(t);
        count += c;
        trst.top.raise(t.top);
    }
    void absorb_sample(int sid, const sink_sample &s) { bool fresh; sink_sample &mine = samples.slot(sid, s, fresh); if(!fresh) mine.top.raise(s.top); }
    void settle() {                              // every per-sample copy mirrors the merged coverage and the number of samples
        const int k = (int)samples.size();
        trst.count2 = k;
        for(auto &x : samples) { x.second.coverage = trst.coverage; x.second.count2 = k; }
    }
    void merge(const sink_item &o) { absorb_head(o.trst, o.count); for(const auto &x : o.samples) absorb_sample(x.first, x.second); settle(); }
    // == merge(sink_item(t, c, sid)) without the temporary (what transcript_set::add(t, count, sid) does when an equal item exists)
    void merge_transcript(const sink_transcript &t, int c, int sid) { absorb_head(t, c); absorb_sample(sid, as_sample(t)); settle(); }
};

class transcript_sink {                          // transcript_set (transcript_set.h:37-59), one chromosome / region per sink
public:
    typedef small_vec<sink_item, 1> bucket;      // nearly every bucket holds one item (an intron chain), which then lives in the table entry
    explicit transcript_sink(double single_exon_overlap = 0.8) : overlap_(single_exon_overlap) {}
    // the reference's map<size_t, vector<trans_item>> (transcript_set.h:43) is only ever probed by key and walked in key order at the
    // end: a hash table for the probes (a million buckets make every tree descent a chain of cache misses), keys sorted on demand
    chain_table<bucket> mt;
    std::vector<size_t> sorted_keys() const { std::vector<size_t> k; k.reserve(mt.size()); for(auto &x : mt) k.push_back(x.first); std::sort(k.begin(), k.end()); return k; }

    // transcript_set::add(t, count, sid) (transcript_set.cc:149-154): a one-item set merged in -- walk the bucket while its items sort
    // first, then merge into the item that compares equal or insert right there
    void add(const sink_transcript &t, int count, int sid) { add_hashed(t, t.chain_key(), count, sid); }
    void add_hashed(const sink_transcript &t, size_t key, int count, int sid) {       // key == t.chain_key(), computed by the caller
        bucket &b = mt[key];
        size_t at = 0; int c = +1;
        while(at < b.size() && (c = b[at].trst.order_against(t, overlap_)) == +1) at++;
        if(at < b.size() && c == 0) b[at].merge_transcript(t, count, sid);
        else b.emplace(b.begin() + at, t, count, sid);
    }
    void add(transcript_sink &ts) { for(auto &x : ts.mt) add_bucket(x.first, x.second); }      // transcript_set.cc:156-175
    void add_bucket(size_t key, bucket &incoming) {                                           // one iteration of that loop
        auto z = mt.find(key);
        if(z == mt.end()) mt.emplace(key, std::move(incoming));
        else if(incoming.size() == 1) place(z->second, incoming[0]);
        else zip(z->second, incoming);
    }
    // zip() when the incoming bucket holds ONE item (nearly always): the forward walk passes the items that sort first and stops at the
    // one that compares equal (merge) or sorts behind (the item goes in front of it) -- in place, nothing else moves
    void place(bucket &b, sink_item &it) {
        size_t at = 0; int c = +1;
        while(at < b.size() && (c = b[at].trst.order_against(it.trst, overlap_)) == +1) at++;
        if(at < b.size() && c == 0) b[at].merge(it);
        else b.emplace(b.begin() + at, std::move(it));
    }
    size_t size() const { size_t n = 0; for(auto &x : mt) n += x.second.size(); return n; }
    void clear() { mt.clear(); }
    double single_exon_overlap() const { return overlap_; }
private:
    double overlap_;
    bucket zipped_;                              // scratch of zip(), kept across calls
    // merge_sorted_trans_items (transcript_set.cc:83-120): forward walk of both sorted buckets; equal heads fuse into `mine`'s item
    void zip(bucket &mine, bucket &theirs) {
        bucket &out = zipped_; out.clear(); out.reserve(mine.size() + theirs.size());
        size_t i = 0, j = 0;
        while(i < mine.size() && j < theirs.size()) {
            const int c = mine[i].trst.order_against(theirs[j].trst, overlap_);
            if(c == -1) { out.push_back(std::move(theirs[j++])); continue; }
            if(c == 0) mine[i].merge(theirs[j++]);
            out.push_back(std::move(mine[i++]));
        }
        std::move(mine.begin() + i, mine.end(), std::back_inserter(out));
        std::move(theirs.begin() + j, theirs.end(), std::back_inserter(out));
        mine.swap(out);
    }
};

} // namespace aletsch
