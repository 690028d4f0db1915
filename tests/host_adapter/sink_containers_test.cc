// sink_containers_test.cc -- the containers of aletsch_amd/host/transcript_sink.hpp under AddressSanitizer + UBSan on the CPU, and the sink
// itself against a plain model of the reference's transcript_set (std::map<size_t, std::vector<item>>, rnacore/transcript_set.cc:38-175)
// on random transcripts.  Built and run by tests/test_tset_cpu.py::test_sink_containers_under_sanitizers; header only, no GPU, no library.
#include "../../aletsch_amd/host/transcript_sink.hpp"
#include <map>
#include <string>
#include <random>
#include <cstdio>
#include <cstdlib>
using namespace aletsch;

static int fails = 0;
#define CHECK(c) do { if(!(c)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c); fails++; } } while(0)

struct tracked {                                   // counts live objects: every construction must meet its destruction
    static long live; std::string s; int v;
    tracked(int x = 0) : s(40, 'a' + x % 26), v(x) { live++; }
    tracked(const tracked &o) : s(o.s), v(o.v) { live++; }
    tracked(tracked &&o) noexcept : s(std::move(o.s)), v(o.v) { o.v = -1; live++; }
    tracked &operator=(const tracked &o) { s = o.s; v = o.v; return *this; }
    tracked &operator=(tracked &&o) noexcept { s = std::move(o.s); v = o.v; o.v = -1; return *this; }
    ~tracked() { live--; }
};
long tracked::live = 0;

static void test_small_vec()
{
    {
        small_vec<tracked, 2> a;
        std::vector<int> model;
        std::mt19937 rng(7);
        for(int step = 0; step < 4000; step++) {
            const int op = (int)(rng() % 7);
            if(op <= 2) { const int x = (int)(rng() % 1000); const size_t at = model.empty() ? 0 : rng() % (model.size() + 1); a.emplace(a.begin() + at, x); model.insert(model.begin() + (long)at, x); }
            else if(op == 3) { const int x = (int)(rng() % 1000); a.push_back(tracked(x)); model.push_back(x); }
            else if(op == 4 && model.size() > 6) { a.clear(); model.clear(); }
            else if(op == 5) { small_vec<tracked, 2> b(a); small_vec<tracked, 2> c(std::move(b)); a = c; a.swap(c); a = std::move(c); }          // copies, moves, swap: inline and heap
            else if(op == 6) { std::vector<tracked> src; for(int k = 0; k < (int)(rng() % 6); k++) src.push_back(tracked(k)); a.assign(src.begin(), src.end()); model.clear(); for(auto &t : src) model.push_back(t.v); }
            CHECK(a.size() == model.size());
            for(size_t i = 0; i < model.size(); i++) CHECK(a[i].v == model[i] && a[i].s.size() == 40);
        }
        small_vec<int32_t, 16> xs; for(int i = 0; i < 40; i++) xs.push_back(i); CHECK(xs.size() == 40 && xs.front() == 0 && xs.back() == 39);
        const int32_t few[3] = {5, 6, 7}; xs.assign(few, few + 3); CHECK(xs.size() == 3 && xs.data()[2] == 7);
        // an element of the list handed to push_back while the list is full: std::vector guarantees it, so does small_vec (the new
        // element is built before the storage moves) -- run under ASan, which sees a read of the freed block if it is not
        small_vec<tracked, 2> sv; std::vector<int> sm;
        for(int i = 0; i < 3; i++) { sv.push_back(tracked(10 + i)); sm.push_back(10 + i); }
        for(int i = 0; i < 70; i++) { const size_t k = (size_t)i % sv.size(); sv.push_back(sv[k]); sm.push_back(sm[k]); sv.push_back(sv.back()); sm.push_back(sm.back()); }
        CHECK(sv.size() == sm.size());
        for(size_t i = 0; i < sm.size(); i++) CHECK(sv[i].v == sm[i] && sv[i].s.size() == 40);
    }
    CHECK(tracked::live == 0);
}

static void test_chain_table()
{
    {
        chain_table<small_vec<tracked, 1>> t; std::map<size_t, std::vector<int>> model;
        std::mt19937_64 rng(11);
        for(int step = 0; step < 30000; step++) {
            const size_t key = (rng() % 3 == 0) ? rng() % 64 : rng() % 5000;          // dense and sparse keys
            if(rng() % 4 == 0) { auto z = t.find(key); CHECK((z == t.end()) == (model.find(key) == model.end())); if(z != t.end()) CHECK(z->first == key && z->second.size() == model[key].size()); }
            else { const int x = (int)(rng() % 100); t[key].push_back(tracked(x)); model[key].push_back(x); }
            if(step == 15000) { chain_table<small_vec<tracked, 1>> copy(t); CHECK(copy.size() == t.size()); chain_table<small_vec<tracked, 1>> moved(std::move(copy)); CHECK(moved.size() == t.size()); }
        }
        CHECK(t.size() == model.size());
        size_t seen = 0;
        for(auto &e : t) { seen++; auto m = model.find(e.first); CHECK(m != model.end()); if(m != model.end()) { CHECK(e.second.size() == m->second.size()); for(size_t i = 0; i < m->second.size(); i++) CHECK(e.second[i].v == m->second[i]); } }
        CHECK(seen == model.size());
        small_vec<tracked, 1> b; b.push_back(tracked(1)); t.emplace(999999, std::move(b)); CHECK(t.find(999999) != t.end() && t.find(999999)->second.size() == 1);
        t.clear(); CHECK(t.size() == 0 && t.find(3) == t.end());
    }
    CHECK(tracked::live == 0);
}

// ---- the reference's transcript_set, restated with standard containers (the model) ----
struct model_item { sink_transcript t; int count; std::map<int, sink_sample> samples; };
struct model_set {
    double ov; std::map<size_t, std::vector<model_item>> mt;
    static model_item make(const sink_transcript &t, int c, int sid) { model_item z; z.t = t; z.count = c; z.samples[sid] = sink_item::as_sample(t); return z; }
    static void merge(model_item &a, const model_item &b) {                     // trans_item::merge, ADD / ADD mode (transcript_set.cc:38-75)
        if(a.t.n_exons() >= 2) a.t.coverage += b.t.coverage; else a.t.coverage = std::max(a.t.coverage, b.t.coverage);
        a.t.widen_to(b.t); a.count += b.count; a.t.top.raise(b.t.top);
        for(auto &x : b.samples) { auto f = a.samples.find(x.first); if(f == a.samples.end()) a.samples[x.first] = x.second; else f->second.top.raise(x.second.top); }
        a.t.count2 = (int)a.samples.size(); for(auto &x : a.samples) { x.second.coverage = a.t.coverage; x.second.count2 = a.t.count2; }
    }
    void add_set(model_set &o) {                                                // transcript_set::add(transcript_set&) (transcript_set.cc:156-175, 83-120)
        for(auto &x : o.mt) {
            auto z = mt.find(x.first);
            if(z == mt.end()) { mt[x.first] = x.second; continue; }
            std::vector<model_item> &mine = z->second, &theirs = x.second, out; size_t i = 0, j = 0;
            while(i < mine.size() && j < theirs.size()) {
                const int c = mine[i].t.order_against(theirs[j].t, ov);
                if(c == -1) { out.push_back(theirs[j++]); continue; }
                if(c == 0) merge(mine[i], theirs[j++]);
                out.push_back(mine[i++]);
            }
            for(; i < mine.size(); i++) out.push_back(mine[i]);
            for(; j < theirs.size(); j++) out.push_back(theirs[j]);
            mine.swap(out);
        }
    }
    void add(const sink_transcript &t, int c, int sid) { model_set one; one.ov = ov; one.mt[t.chain_key()].push_back(make(t, c, sid)); add_set(one); }      // transcript_set.cc:149-154
};

static sink_transcript random_transcript(std::mt19937 &rng)
{
    sink_transcript t; t.strand = "+-."[rng() % 3];
    const int kind = (int)(rng() % 10);
    const int ne = kind < 2 ? 1 : (kind < 9 ? 2 + (int)(rng() % 4) : 9 + (int)(rng() % 6));          // single exon / short chains / chains beyond the inline eight exons
    int32_t p = 1000 + 50 * (int32_t)(rng() % 40);
    for(int k = 0; k < ne; k++) { const int32_t len = 20 + 10 * (int32_t)(rng() % 4); t.add_exon(p, p + len); p += len + 100 * (1 + (int32_t)(rng() % 3)); }
    t.coverage = 1.0 + (double)(rng() % 1000) / 7.0; t.top.cov2 = t.coverage; t.top.conf = (double)(rng() % 100) / 3.0; t.top.abd = (double)(rng() % 50); t.top.count1 = (int)(rng() % 9); t.count2 = 1;
    t.tid = (int64_t)rng();
    return t;
}

static void test_sink_against_model()
{
    std::mt19937 rng(2026);
    transcript_sink S(0.8); model_set M; M.ov = 0.8;
    for(int round = 0; round < 300; round++) {
        if(rng() % 3 == 0) { for(int k = 0; k < 20; k++) { sink_transcript t = random_transcript(rng); const int sid = (int)(rng() % 12); S.add(t, 1, sid); M.add(t, 1, sid); } }
        else {                                                                  // a per-graph set merged in (assembler.cc:1105-1133)
            transcript_sink ts(0.8); model_set ms; ms.ov = 0.8; const int sid = (int)(rng() % 12);
            for(int k = 0; k < (int)(rng() % 30); k++) { sink_transcript t = random_transcript(rng); ts.add(t, 1, sid); ms.add(t, 1, sid); }
            S.add(ts); M.add_set(ms);
        }
    }
    CHECK(S.mt.size() == M.mt.size());
    size_t items = 0;
    for(size_t key : S.sorted_keys()) {
        auto m = M.mt.find(key); CHECK(m != M.mt.end()); if(m == M.mt.end()) continue;
        const transcript_sink::bucket &b = S.mt.find(key)->second; CHECK(b.size() == m->second.size());
        for(size_t i = 0; i < b.size() && i < m->second.size(); i++) {
            const sink_item &a = b[i]; const model_item &z = m->second[i]; items++;
            CHECK(a.count == z.count && a.trst.strand == z.t.strand && a.trst.coverage == z.t.coverage && a.trst.tid == z.t.tid && a.trst.count2 == z.t.count2);
            CHECK(a.trst.top.cov2 == z.t.top.cov2 && a.trst.top.conf == z.t.top.conf && a.trst.top.abd == z.t.top.abd && a.trst.top.count1 == z.t.top.count1);
            CHECK(a.trst.xs.size() == z.t.xs.size()); for(size_t k = 0; k < a.trst.xs.size() && k < z.t.xs.size(); k++) CHECK(a.trst.xs[k] == z.t.xs[k]);
            CHECK(a.samples.size() == z.samples.size());
            auto q = z.samples.begin();
            for(auto &x : a.samples) { if(q == z.samples.end()) break; CHECK(x.first == q->first && x.second.coverage == q->second.coverage && x.second.count2 == q->second.count2 && x.second.top.conf == q->second.top.conf && x.second.top.abd == q->second.top.abd); ++q; }
        }
    }
    CHECK(items > 1500 && items == S.size());
    printf("sink vs model: %zu items in %zu buckets\n", items, S.mt.size());
}

int main()
{
    test_small_vec(); test_chain_table(); test_sink_against_model();
    if(fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
    printf("sink containers ok\n");
    return 0;
}
