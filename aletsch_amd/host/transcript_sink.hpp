// transcript_sink.hpp -- host-side result sink: the reference's transcript_set (rnacore/transcript_set.cc:38-175) restated over
// plain records.  Transcripts are bucketed by intron-chain hash (gtf/transcript.cc:183-201, util/util.cc:38-46), each bucket is
// kept sorted by transcript::compare1 (gtf/transcript.cc:269-300), equal transcripts merge (trans_item::merge,
// transcript_set.cc:38-81: multi-exon coverage adds, single-exon takes the max, bounds widen, per-sample records merge,
// count2 = number of samples).  Quirks are kept: intron_chain_compare skips the last-but-one exon (transcript.cc:236-245),
// the first inserted transcript keeps its id.
//
// Pinned against the reference's own transcript_set.cc built from source (oracle/_ref/ref_tset, tests/golden/ref_tset.json).
// In the multi-GPU flow rank 0 feeds the gathered records in ascending global graph id, so the result does not depend on the
// number of ranks.
#pragma once
#include <vector>
#include <map>
#include <unordered_map>
#include <algorithm>
#include <cstdint>
#include <cstddef>
#include <utility>

namespace aletsch {

struct sink_transcript {
    char strand = '.';
    double coverage = 0, cov2 = 0, conf = 0, abd = 0;
    int count1 = 0, count2 = 0;
    int64_t tid = 0;                            // transcript_id (the reference keeps a string "chr<chrm>.<gid>.<i>")
    std::vector<std::pair<int32_t, int32_t>> exons;

    int length() const { int s = 0; for(auto &e : exons) s += e.second - e.first; return s; }
    size_t intron_chain_hashing() const {       // transcript.cc:183-201 + util.cc:38-46 (vector_hash)
        if(exons.empty()) return 0;
        if(exons.size() == 1) { size_t p = (size_t)((exons[0].first + exons[0].second) / 10000); return p + 1; }
        std::vector<int32_t> vv;
        int32_t p = exons[0].second;
        for(size_t k = 1; k < exons.size(); k++) { vv.push_back(p); vv.push_back(exons[k].first); p = exons[k].second; }
        size_t seed = vv.size();
        for(size_t i = 0; i < vv.size(); i++) seed ^= (size_t)(vv[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return (seed & 0x7FFFFFFF) + 1;
    }
    int intron_chain_compare(const sink_transcript &t) const {      // transcript.cc:226-246
        if(exons.size() < t.exons.size()) return +1;
        if(exons.size() > t.exons.size()) return -1;
        if(exons.size() <= 1) return 0;
        int n = (int)exons.size() - 1;
        if(exons[0].second < t.exons[0].second) return +1;
        if(exons[0].second > t.exons[0].second) return -1;
        for(int k = 1; k < n - 1; k++) {
            if(exons[k].first < t.exons[k].first) return +1;
            if(exons[k].first > t.exons[k].first) return -1;
            if(exons[k].second < t.exons[k].second) return +1;
            if(exons[k].second > t.exons[k].second) return -1;
        }
        if(exons[n].first < t.exons[n].first) return +1;
        if(exons[n].first > t.exons[n].first) return -1;
        return 0;
    }
    int compare1(const sink_transcript &t, double single_exon_overlap) const {     // transcript.cc:269-300 (one seqname per sink)
        if(exons.size() < t.exons.size()) return +1;
        if(exons.size() > t.exons.size()) return -1;
        if(strand < t.strand) return +1;
        if(strand > t.strand) return -1;
        if(exons.size() == 1) {
            int32_t p2 = exons[0].first < t.exons[0].first ? t.exons[0].first : exons[0].first;
            int32_t q2 = exons[0].second > t.exons[0].second ? t.exons[0].second : exons[0].second;
            int32_t overlap = q2 - p2;
            if(overlap >= single_exon_overlap * length()) return 0;
            if(overlap >= single_exon_overlap * t.length()) return 0;
            if(exons[0].first < t.exons[0].first) return +1;
            if(exons[0].first > t.exons[0].first) return -1;
            if(exons[0].second < t.exons[0].second) return +1;
            if(exons[0].second > t.exons[0].second) return -1;
        }
        return intron_chain_compare(t);
    }
    void extend_bounds(const sink_transcript &t) {                  // transcript.cc:302-308
        if(exons.empty()) return;
        if(t.exons.front().first < exons.front().first) exons.front().first = t.exons.front().first;
        if(t.exons.back().second > exons.back().second) exons.back().second = t.exons.back().second;
    }
};

struct sink_sample {                             // what the reference keeps of the per-sample transcript copy in trans_item::samples:
    double coverage = 0, cov2 = 0, conf = 0, abd = 0; int count1 = 0, count2 = 0;     // the fields merge() reads and writes (exons are those of trst)
};
// map<int, sink_sample> for a handful of samples: one sorted array (find / insert / ordered iteration as std::map gives them)
struct sample_map {
    typedef std::pair<int, sink_sample> value_type;
    std::vector<value_type> v;
    typedef std::vector<value_type>::iterator iterator;
    iterator begin() { return v.begin(); } iterator end() { return v.end(); }
    std::vector<value_type>::const_iterator begin() const { return v.begin(); } std::vector<value_type>::const_iterator end() const { return v.end(); }
    size_t size() const { return v.size(); }
    iterator find(int k) { iterator it = std::lower_bound(v.begin(), v.end(), k, [](const value_type &a, int key) { return a.first < key; }); return (it != v.end() && it->first == k) ? it : v.end(); }
    void insert(const value_type &x) { iterator it = std::lower_bound(v.begin(), v.end(), x.first, [](const value_type &a, int key) { return a.first < key; }); if(it == v.end() || it->first != x.first) v.insert(it, x); }
};
struct sink_item {                               // trans_item (transcript_set.h:20-33)
    sink_transcript trst; int count = 0;
    sample_map samples;
    sink_item() {}
    sink_item(const sink_transcript &t, int c, int s) : trst(t), count(c) {
        sink_sample x; x.coverage = t.coverage; x.cov2 = t.cov2; x.conf = t.conf; x.abd = t.abd; x.count1 = t.count1; x.count2 = 1;
        samples.insert(std::make_pair(s, x));
    }
    void merge(sink_item &ti) {                  // TRANSCRIPT_COUNT_ADD_COVERAGE_ADD (transcript_set.cc:38-75)
        if(trst.exons.size() >= 2) trst.coverage += ti.trst.coverage;
        else if(trst.coverage < ti.trst.coverage) trst.coverage = ti.trst.coverage;
        trst.extend_bounds(ti.trst);
        count += ti.count;
        if(trst.cov2 < ti.trst.cov2) trst.cov2 = ti.trst.cov2;
        if(trst.conf < ti.trst.conf) trst.conf = ti.trst.conf;
        if(trst.abd < ti.trst.abd) trst.abd = ti.trst.abd;
        if(trst.count1 < ti.trst.count1) trst.count1 = ti.trst.count1;
        for(auto &x : ti.samples) {
            auto f = samples.find(x.first);
            if(f == samples.end()) samples.insert(x);
            else {
                if(f->second.cov2 < x.second.cov2) f->second.cov2 = x.second.cov2;
                if(f->second.conf < x.second.conf) f->second.conf = x.second.conf;
                if(f->second.abd < x.second.abd) f->second.abd = x.second.abd;
                if(f->second.count1 < x.second.count1) f->second.count1 = x.second.count1;
            }
        }
        trst.count2 = (int)samples.size();
        for(auto &x : samples) { x.second.coverage = trst.coverage; x.second.count2 = (int)samples.size(); }
    }
    // merge(sink_item(t, c, s)) without building the temporary item (what transcript_set::add(t, count, sid) amounts to when an equal
    // transcript is already there)
    void merge_transcript(const sink_transcript &t, int c, int s) {
        if(trst.exons.size() >= 2) trst.coverage += t.coverage;
        else if(trst.coverage < t.coverage) trst.coverage = t.coverage;
        trst.extend_bounds(t);
        count += c;
        if(trst.cov2 < t.cov2) trst.cov2 = t.cov2;
        if(trst.conf < t.conf) trst.conf = t.conf;
        if(trst.abd < t.abd) trst.abd = t.abd;
        if(trst.count1 < t.count1) trst.count1 = t.count1;
        auto f = samples.find(s);
        if(f == samples.end()) { sink_sample x; x.coverage = t.coverage; x.cov2 = t.cov2; x.conf = t.conf; x.abd = t.abd; x.count1 = t.count1; x.count2 = 1; samples.insert(std::make_pair(s, x)); }
        else {
            if(f->second.cov2 < t.cov2) f->second.cov2 = t.cov2;
            if(f->second.conf < t.conf) f->second.conf = t.conf;
            if(f->second.abd < t.abd) f->second.abd = t.abd;
            if(f->second.count1 < t.count1) f->second.count1 = t.count1;
        }
        trst.count2 = (int)samples.size();
        for(auto &x : samples) { x.second.coverage = trst.coverage; x.second.count2 = (int)samples.size(); }
    }
};

class transcript_sink {                          // transcript_set (transcript_set.h:37-59), one chromosome / region per sink
public:
    explicit transcript_sink(double single_exon_overlap = 0.8) : overlap_(single_exon_overlap) {}
    // the reference's map<size_t, vector<trans_item>> (transcript_set.h:43) is only ever probed by key and walked in key order at the
    // end: a hash table for the probes (a million buckets make every tree descent a chain of cache misses), keys sorted on demand
    std::unordered_map<size_t, std::vector<sink_item>> mt;
    std::vector<size_t> sorted_keys() const { std::vector<size_t> k; k.reserve(mt.size()); for(auto &x : mt) k.push_back(x.first); std::sort(k.begin(), k.end()); return k; }

    // transcript_set::add(t, count, sid) (transcript_set.cc:149-154) = a one-item set merged in.  merge_sorted_trans_items with a
    // single y walks the bucket while compare1 says "x first", then merges into the first equal item or inserts y right there.
    void add(const sink_transcript &t, int count, int sid) { add_hashed(t, t.intron_chain_hashing(), count, sid); }
    void add_hashed(const sink_transcript &t, size_t key, int count, int sid) {       // key == t.intron_chain_hashing(), computed by the caller
        auto z = mt.find(key);
        if(z == mt.end()) { std::vector<sink_item> v; v.emplace_back(t, count, sid); mt.emplace(key, std::move(v)); return; }
        std::vector<sink_item> &vx = z->second;
        size_t kx = 0; int b = +1;
        while(kx < vx.size() && (b = vx[kx].trst.compare1(t, overlap_)) == +1) kx++;
        if(kx < vx.size() && b == 0) vx[kx].merge_transcript(t, count, sid);
        else vx.insert(vx.begin() + kx, sink_item(t, count, sid));
    }
    void add(transcript_sink &ts) {              // transcript_set.cc:156-175
        for(auto &x : ts.mt) add_bucket(x.first, x.second);
    }
    void add_bucket(size_t key, std::vector<sink_item> &vy) {       // one iteration of that loop
        auto z = mt.find(key);
        if(z == mt.end()) mt.emplace(key, std::move(vy));
        else merge_sorted(z->second, vy);
    }
    size_t size() const { size_t n = 0; for(auto &x : mt) n += x.second.size(); return n; }
    void clear() { mt.clear(); }
    double single_exon_overlap() const { return overlap_; }
private:
    double overlap_;
    void merge_sorted(std::vector<sink_item> &vx, std::vector<sink_item> &vy) {    // transcript_set.cc:83-120
        std::vector<sink_item> vz; vz.reserve(vx.size() + vy.size());
        size_t kx = 0, ky = 0;
        while(kx < vx.size() && ky < vy.size()) {
            int b = vx[kx].trst.compare1(vy[ky].trst, overlap_);
            if(b == 0) { vx[kx].merge(vy[ky]); vz.emplace_back(std::move(vx[kx])); kx++; ky++; }
            else if(b == 1) { vz.emplace_back(std::move(vx[kx])); kx++; }
            else { vz.emplace_back(std::move(vy[ky])); ky++; }
        }
        for(size_t i = kx; i < vx.size(); i++) vz.emplace_back(std::move(vx[i]));
        for(size_t i = ky; i < vy.size(); i++) vz.emplace_back(std::move(vy[i]));
        vx.swap(vz);
    }
};

} // namespace aletsch
