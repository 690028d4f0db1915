// gpu_dispatch.hpp -- queue + flusher in place of the reference's inline per-graph calls.
//
// The reference runs one pool task per cluster (meta/incubator.cc:553-577, 609-637); inside a task every sample graph and the
// merged graph go through assembler::assemble(gx, px, sid) (meta/assembler.cc:296-347, 370, 1075-1136), which builds a scallop,
// decomposes ONE graph and merges its transcripts into the shared transcript_set under `mylock` (assembler.cc:1127-1132).
//
// aletsch::gpu_assembly_queue keeps that call shape for the pool tasks -- submit(gx, hx, sid) where the reference has
// `scallop sx(gx, hx, pa); sx.assemble(); ... tm.add(ts)` -- and batches ACROSS tasks and clusters:
//
//   pool threads --submit()--> per-thread chunk --full--> ready chunks --> [pack threads: ald_batch_add_packed into a free batch]
//        --> [GPU thread: upload, kernel, download] --> [merge thread: ald_tset_add_batch] --> sink
//
// A submitting thread converts its graph and appends it to ITS OWN chunk of packed arrays (no shared lock on that path); a full
// chunk is handed over with one short critical section and gets the next run of tickets.  A pack thread gathers chunks into a
// batch of `batch_graphs`; `slots` batches rotate through the pack, GPU and merge stages, so staging, kernel and merge overlap.  Graphs
// are merged into the sink in ticket order, batch after batch -- the same per-bucket sequence of trans_item::merge calls as a
// serial run over the tickets.  With one submitting thread tickets are the submission order and the result is bit-identical to
// the serial loop; with several, chunks interleave as the pool produced them, exactly as the reference's merge order is whatever
// `mylock` produced.
//
// Several MI355X in ONE process (the reference's own shape: one process, one pool -- meta/incubator.cc:609-637): give the queue a device
// list.  Every device gets its own slots and its own GPU thread; a pack thread deals a batch to the least loaded device; the merge
// thread takes finished batches in the order they were CUT (= ticket order), whichever device finishes first, so the merged set does
// not depend on the number of devices.  No collective is involved: bundles are independent, the result set lives in host memory.
//
// C++11, header only; compiles inside the reference tree with its own types (see gpu_scallop.hpp for the members used).
#pragma once
#include "gpu_scallop.hpp"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <memory>
#include <chrono>
#include <atomic>

#ifndef ALD_CONVERT_AHEAD_EDGES
#define ALD_CONVERT_AHEAD_EDGES 12     /* edge objects asked for this many edges ahead of the walk over gr.edges() */
#endif
#ifndef ALD_CONVERT_AHEAD_NODES
#define ALD_CONVERT_AHEAD_NODES 6      /* first nodes of an edge's sample set / abundance map, this many edges ahead of the copy */
#endif

namespace aletsch {

// many staged graphs back to back, in the layout of ald_batch_add_packed
struct packed_chunk {
    std::vector<int32_t> g_nv, g_ne, g_np, vertex_offset, edge_target, edge_sample_offset, sample_id, vertex_lpos, vertex_rpos, vertex_type,
                         phasing_offset, phasing_vertex, phasing_count, edge_count, edge_rank, sid;
    std::vector<double> edge_weight, edge_abd, sample_abd, vertex_weight; std::vector<uint8_t> edge_strand; std::vector<char> graph_strand;
    // raw graphs (submit_raw): per graph -1 or the max_group_boundary_distance of a graph whose pre-steps run on the device, its phases
    std::vector<int32_t> raw_dist, nphase, rphase_offset, rphase_coord, rphase_count; bool any_raw = false;
    long first = 0;                                               // ticket of the first graph, set when the chunk is handed over
    int n() const { return (int)g_nv.size(); }
    template<class T> static void cat(std::vector<T> &d, const std::vector<T> &s) { d.insert(d.end(), s.begin(), s.end()); }
    void append(const staged_graph &s, int sample)
    {
        g_nv.push_back((int32_t)s.vertex_weight.size()); g_ne.push_back((int32_t)s.edge_target.size()); g_np.push_back((int32_t)s.phasing_count.size());
        graph_strand.push_back(s.strand); sid.push_back((int32_t)sample); raw_dist.push_back(-1); nphase.push_back(0); rphase_offset.push_back(0);
        cat(vertex_offset, s.vertex_offset); cat(edge_target, s.edge_target); cat(edge_weight, s.edge_weight); cat(edge_strand, s.edge_strand); cat(edge_abd, s.edge_abd);
        cat(edge_sample_offset, s.edge_sample_offset); cat(sample_id, s.sample_id); cat(sample_abd, s.sample_abd);
        cat(vertex_weight, s.vertex_weight); cat(vertex_lpos, s.vertex_lpos); cat(vertex_rpos, s.vertex_rpos); cat(vertex_type, s.vertex_type);
        cat(phasing_offset, s.phasing_offset); cat(phasing_vertex, s.phasing_vertex); cat(phasing_count, s.phasing_count); cat(edge_count, s.edge_count); cat(edge_rank, s.edge_rank);
    }
    // The same as stage_graph() + append(), without the staged_graph in between: ONE walk over gr.edges(), a counting sort into CSR
    // rows (rows keep the walk's order, i.e. creation order; inside a row a stable insertion by target: rows hold a handful of edges),
    // every field written straight behind the chunk's last graph.  `tmp` is the calling thread's scratch, kept across graphs.
    // The walk is bound by cache misses on the reference's objects -- an edge object, its edge_info, the first node of its sample
    // set and of its abundance map are four heap blocks --, so every loop asks for what it will read a few edges ahead: the edge
    // objects while it walks gr.edges() (a second iterator runs in front), the edge_info records (ONE get_edge_info per edge, the
    // addresses are kept) while it sorts, the container nodes while it copies.
    struct scratch { std::vector<int32_t> row, at, ord, dst, perm; std::vector<const void*> eh, inf; };
    template<class SpliceGraph, class HyperSet>
    void append_graph(SpliceGraph &gr, const HyperSet &hs, int sample, scratch &tmp)
    {
        typedef typename std::decay<decltype(*gr.edges().first)>::type edge_t;       // edge_descriptor (a pointer in the reference)
        static_assert(std::is_pointer<edge_t>::value, "edge_descriptor is expected to be a pointer (graph/edge_base.h)");
        typedef typename std::decay<decltype(gr.get_edge_info(*gr.edges().first))>::type einfo_t;
        const int V = (int)gr.num_vertices();
        std::vector<const void*> &eh = tmp.eh, &inf = tmp.inf;
        eh.clear(); inf.clear(); tmp.dst.clear(); tmp.row.assign((size_t)V + 1, 0);
        std::vector<int32_t> &src = tmp.ord;                                         // source of edge k (walk order)
        src.clear();
        {
            auto pe = gr.edges(); auto ahead = pe.first;
            for(int i = 0; i < ALD_CONVERT_AHEAD_EDGES && ahead != pe.second; i++, ++ahead) __builtin_prefetch((const void*)*ahead);
            for(auto it = pe.first; it != pe.second; ++it) {
                if(ahead != pe.second) { __builtin_prefetch((const void*)*ahead); ++ahead; }
                const int a = (*it)->source(), b = (*it)->target();
                const einfo_t *p = &gr.get_edge_info(*it); __builtin_prefetch((const void*)p); __builtin_prefetch((const char*)(const void*)p + 64);
                eh.push_back((const void*)*it); inf.push_back((const void*)p); src.push_back(a); tmp.dst.push_back(b); tmp.row[(size_t)a + 1]++;
            }
        }
        const int E = (int)eh.size();
        for(int i = 0; i < V; i++) tmp.row[(size_t)i + 1] += tmp.row[(size_t)i];
        tmp.at.assign(tmp.row.begin(), tmp.row.end() - 1);
        tmp.perm.resize((size_t)E);                                                  // CSR position -> walk position (= scallop's edge index, graph_base.cc:139-153)
        int32_t *perm = tmp.perm.data(); const int32_t *dstp = tmp.dst.data();
        for(int k = 0; k < E; k++) {
            const int a = src[(size_t)k]; int pos = tmp.at[(size_t)a]++;
            const int lo = tmp.row[(size_t)a];
            while(pos > lo && dstp[perm[pos - 1]] > dstp[k]) { perm[pos] = perm[pos - 1]; pos--; }      // stable: equal targets keep walk order
            perm[pos] = k;
        }
        g_nv.push_back(V); g_ne.push_back(E); graph_strand.push_back(gr.strand); sid.push_back((int32_t)sample); raw_dist.push_back(-1); nphase.push_back(0); rphase_offset.push_back(0);
        vertex_offset.insert(vertex_offset.end(), tmp.row.begin(), tmp.row.end());
        const size_t e0 = edge_target.size(), s0 = edge_sample_offset.size();
        edge_target.resize(e0 + (size_t)E); edge_weight.resize(e0 + (size_t)E); edge_strand.resize(e0 + (size_t)E); edge_abd.resize(e0 + (size_t)E); edge_count.resize(e0 + (size_t)E); edge_rank.resize(e0 + (size_t)E);
        edge_sample_offset.resize(s0 + (size_t)E + 1); edge_sample_offset[s0] = 0;
        const size_t smp0 = sample_id.size();
        int32_t *o_tgt = edge_target.data() + e0, *o_cnt = edge_count.data() + e0, *o_rank = edge_rank.data() + e0, *o_so = edge_sample_offset.data() + s0 + 1;
        double *o_w = edge_weight.data() + e0, *o_abd = edge_abd.data() + e0; uint8_t *o_st = edge_strand.data() + e0;
        const int AHEAD = ALD_CONVERT_AHEAD_NODES;
        for(int q = 0; q < AHEAD && q < E; q++) { const einfo_t &ea = *(const einfo_t*)inf[(size_t)perm[q]]; if(!ea.samples.empty()) __builtin_prefetch((const void*)&*ea.samples.begin()); if(!ea.spAbd.empty()) __builtin_prefetch((const void*)&*ea.spAbd.begin()); }
        for(int q = 0; q < E; q++) {
            if(q + AHEAD < E) { const einfo_t &ea = *(const einfo_t*)inf[(size_t)perm[q + AHEAD]]; if(!ea.samples.empty()) __builtin_prefetch((const void*)&*ea.samples.begin()); if(!ea.spAbd.empty()) __builtin_prefetch((const void*)&*ea.spAbd.begin()); }
            const int k = perm[q]; const edge_t e = (edge_t)const_cast<void*>(eh[(size_t)k]);
            const einfo_t &ei = *(const einfo_t*)inf[(size_t)k];
            o_tgt[q] = dstp[k]; o_w[q] = gr.get_edge_weight(e); o_st[q] = (uint8_t)ei.strand; o_abd[q] = ei.abd;
            o_cnt[q] = (int32_t)ei.count; o_rank[q] = k;
            if(ei.samples.size() == 1 && ei.spAbd.size() == 1 && ei.spAbd.begin()->first == *ei.samples.begin()) { sample_id.push_back(*ei.samples.begin()); sample_abd.push_back(ei.spAbd.begin()->second); }     // the common case without a hash lookup
            else for(int sp : ei.samples) { sample_id.push_back(sp); auto f = ei.spAbd.find(sp); sample_abd.push_back(f == ei.spAbd.end() ? 0.0 : f->second); }
            o_so[q] = (int32_t)(sample_id.size() - smp0);
        }
        const size_t v0 = vertex_weight.size();
        vertex_weight.resize(v0 + (size_t)V); vertex_lpos.resize(v0 + (size_t)V); vertex_rpos.resize(v0 + (size_t)V); vertex_type.resize(v0 + (size_t)V);
        for(int i = 0; i < V; i++) { const auto &vi = gr.get_vertex_info(i); vertex_weight[v0 + (size_t)i] = gr.get_vertex_weight(i); vertex_lpos[v0 + (size_t)i] = vi.lpos; vertex_rpos[v0 + (size_t)i] = vi.rpos; vertex_type[v0 + (size_t)i] = vi.type; }
        const size_t pv0 = phasing_vertex.size(); int np = 0;
        phasing_offset.push_back(0);
        for(const auto &kv : hs.nodes) { for(int x : kv.first) phasing_vertex.push_back(x); phasing_offset.push_back((int32_t)(phasing_vertex.size() - pv0)); phasing_count.push_back(kv.second); np++; }
        g_np.push_back(np);
    }
    // the graph appended last is RAW: its pre-steps (meta/assembler.cc:1075-1086) run on the device; PhaseSet: .pmap (map<vector<int32_t>, int>)
    template<class PhaseSet>
    void last_is_raw(const PhaseSet &px, int max_group_boundary_distance)
    {
        raw_dist.back() = max_group_boundary_distance; any_raw = true;
        const size_t c0 = rphase_coord.size(); int np = 0;
        for(const auto &kv : px.pmap) { rphase_coord.insert(rphase_coord.end(), kv.first.begin(), kv.first.end()); rphase_offset.push_back((int32_t)(rphase_coord.size() - c0)); rphase_count.push_back((int32_t)kv.second); np++; }
        nphase.back() = np;
    }
    void reserve_like(const packed_chunk &o)                      // the next chunk of a thread is about as large as its last one
    {
        g_nv.reserve(o.g_nv.size()); g_ne.reserve(o.g_ne.size()); g_np.reserve(o.g_np.size()); graph_strand.reserve(o.graph_strand.size()); sid.reserve(o.sid.size());
        vertex_offset.reserve(o.vertex_offset.size()); edge_target.reserve(o.edge_target.size()); edge_weight.reserve(o.edge_weight.size()); edge_strand.reserve(o.edge_strand.size());
        edge_abd.reserve(o.edge_abd.size()); edge_sample_offset.reserve(o.edge_sample_offset.size()); sample_id.reserve(o.sample_id.size()); sample_abd.reserve(o.sample_abd.size());
        vertex_weight.reserve(o.vertex_weight.size()); vertex_lpos.reserve(o.vertex_lpos.size()); vertex_rpos.reserve(o.vertex_rpos.size()); vertex_type.reserve(o.vertex_type.size());
        phasing_offset.reserve(o.phasing_offset.size()); phasing_vertex.reserve(o.phasing_vertex.size()); phasing_count.reserve(o.phasing_count.size()); edge_count.reserve(o.edge_count.size()); edge_rank.reserve(o.edge_rank.size());
    }
    int add_to(ald_batch *b) const
    {
        static const int32_t zero_i = 0; static const double zero_d = 0;     // empty arrays still need valid pointers
        auto pi = [](const std::vector<int32_t> &v) { return v.empty() ? &zero_i : v.data(); };
        auto pd = [](const std::vector<double> &v) { return v.empty() ? &zero_d : v.data(); };
        if(!any_raw)
        return ald_batch_add_packed(b, n(), g_nv.data(), g_ne.data(), g_np.data(), vertex_offset.data(), pi(edge_target), pd(edge_weight),
                                    edge_strand.empty() ? (const uint8_t*)&zero_i : edge_strand.data(), pd(edge_abd), edge_sample_offset.data(), pi(sample_id), pd(sample_abd),
                                    pd(vertex_weight), pi(vertex_lpos), pi(vertex_rpos), pi(vertex_type), phasing_offset.data(), pi(phasing_vertex), pi(phasing_count),
                                    graph_strand.data(), pi(edge_count), edge_rank.size() == edge_target.size() && !edge_rank.empty() ? edge_rank.data() : nullptr);
        return ald_batch_add_packed_raw(b, n(), g_nv.data(), g_ne.data(), g_np.data(), vertex_offset.data(), pi(edge_target), pd(edge_weight),
                                    edge_strand.empty() ? (const uint8_t*)&zero_i : edge_strand.data(), pd(edge_abd), edge_sample_offset.data(), pi(sample_id), pd(sample_abd),
                                    pd(vertex_weight), pi(vertex_lpos), pi(vertex_rpos), pi(vertex_type), phasing_offset.data(), pi(phasing_vertex), pi(phasing_count),
                                    graph_strand.data(), pi(edge_count), edge_rank.size() == edge_target.size() && !edge_rank.empty() ? edge_rank.data() : nullptr,
                                    raw_dist.data(), nphase.data(), rphase_offset.data(), pi(rphase_coord), pi(rphase_count));
    }
};

template<class SpliceGraph, class HyperSet, class Parameters>
class gpu_assembly_queue {
public:
    // sink: the shared result set (the reference's `tmerge`); skip_single_exon: cfg.skip_single_exon_transcripts (assembler.cc:1117);
    // chunk_graphs: graphs a submitting thread collects before it hands them over (capped by batch_graphs)
    gpu_assembly_queue(const Parameters &cfg, ald_tset *sink, bool skip_single_exon = false, int device = 0, int batch_graphs = 65536, int slots = 4, int chunk_graphs = 2048, int pack_threads = 2)
        : gpu_assembly_queue(cfg, sink, skip_single_exon, std::vector<int>(1, device), batch_graphs, slots, chunk_graphs, pack_threads) {}
    // devices: HIP device ordinals (an ordinal may repeat: two GPU threads then share that device); slots: batch objects PER device;
    // pack_threads: threads that copy handed-over chunks into a batch (batches are CUT one at a time, in ticket order, and packed side by side)
    gpu_assembly_queue(const Parameters &cfg, ald_tset *sink, bool skip_single_exon, const std::vector<int> &devices, int batch_graphs = 65536, int slots = 4, int chunk_graphs = 2048, int pack_threads = 2)
        : sink_(sink), skip_(skip_single_exon), batch_graphs_(batch_graphs < 1 ? 1 : batch_graphs)
    {
        if(!sink) throw std::invalid_argument("gpu_assembly_queue: null sink");
        if(devices.empty()) throw std::invalid_argument("gpu_assembly_queue: empty device list");
        if(slots < 1) slots = 1;
        if(pack_threads < 1) pack_threads = 1;
        const int ndev = (int)devices.size();
        chunk_graphs_ = chunk_graphs < 1 ? 1 : chunk_graphs; if(chunk_graphs_ > batch_graphs_) chunk_graphs_ = batch_graphs_;
        ready_cap_ = (long)batch_graphs_ * (slots * ndev + 1);
        ald_params p = stage_params(cfg);
        slots_.resize((size_t)slots * ndev); gpu_q_.resize((size_t)ndev); dev_load_.assign((size_t)ndev, 0);
        for(size_t i = 0; i < slots_.size(); i++) {
            slots_[i].dev = (int)(i / (size_t)slots);
            int rc = ald_batch_create(&p, devices[(size_t)slots_[i].dev], &slots_[i].b);
            if(rc != ALD_OK) { for(auto &q : slots_) if(q.b) ald_batch_destroy(q.b); throw gpu_error(rc, "ald_batch_create"); }
        }
        cv_gpu_.reset(new std::condition_variable[(size_t)ndev]);
        for(int t = 0; t < pack_threads; t++) pack_threads_.emplace_back([this] { pack_loop(); });
        for(int d = 0; d < ndev; d++) gpu_threads_.emplace_back([this, d] { gpu_loop(d); });
        merge_thread_ = std::thread([this] { merge_loop(); });
    }
    ~gpu_assembly_queue()
    {
        try { drain(); } catch(...) {}
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_pack_.notify_all(); for(size_t d = 0; d < gpu_q_.size(); d++) cv_gpu_[d].notify_all(); cv_merge_.notify_all();
        for(auto &t : pack_threads_) t.join();
        for(auto &t : gpu_threads_) t.join();
        merge_thread_.join();
        for(auto &s : slots_) if(s.b) ald_batch_destroy(s.b);
    }
    gpu_assembly_queue(const gpu_assembly_queue &) = delete;
    gpu_assembly_queue &operator=(const gpu_assembly_queue &) = delete;

    // Thread-safe.  The graph's transcripts will carry tid = (ticket << 20) | path index, tickets being handed out chunk by chunk.
    void submit(SpliceGraph &gx, const HyperSet &hx, int sid)
    {
        // the per-graph work stays in the calling thread: the graph is converted straight into the thread's own chunk (its lock is only
        // ever contended by drain())
        lane *L = my_lane();
        std::unique_lock<std::mutex> ll(L->m);
        L->c.append_graph(gx, hx, sid, L->tmp);
        hand_over_if_full(L, ll);
    }
    void submit_staged(const staged_graph &s, int sid)
    {
        lane *L = my_lane();
        std::unique_lock<std::mutex> ll(L->m);
        L->c.append(s, sid);
        hand_over_if_full(L, ll);
    }
private:
    struct lane;
    void hand_over_if_full(lane *L, std::unique_lock<std::mutex> &ll)
    {
        if(L->c.n() < chunk_graphs_) return;
        packed_chunk full; std::swap(full, L->c); L->c.reserve_like(full);
        ll.unlock();
        std::unique_lock<std::mutex> lk(m_);
        while(ready_graphs_ >= ready_cap_ && !err_) cv_space_.wait(lk);            // the pool is ahead of the GPU: hold the submitter back
        if(err_) throw gpu_error(err_, err_msg_.c_str());
        push_chunk(std::move(full));
    }
public:

    // The call shape of assembler::assemble(gx, px, sid) itself (meta/assembler.cc:1075): the graph is queued AS IT IS with its phase set;
    // the pre-steps (extend_strands, boundary grouping, phase projection, hyper_set ctor + filter_nodes) run on the device, in the wave
    // that loads the graph.  A graph on which the reference would have asserted in them is counted in failed_graphs() like any other.
    template<class PhaseSet>
    bool submit_raw(SpliceGraph &gx, const PhaseSet &px, int sid, int max_group_boundary_distance = 10000)
    {
        struct no_hyper_set { std::map<std::vector<int>, int> nodes; } none;
        lane *L = my_lane();
        std::unique_lock<std::mutex> ll(L->m);
        L->c.append_graph(gx, none, sid, L->tmp);
        L->c.last_is_raw(px, max_group_boundary_distance);
        hand_over_if_full(L, ll);
        return true;
    }

    // Hands over every partial chunk and returns when every submitted graph has been merged into the sink.
    void drain()
    {
        std::unique_lock<std::mutex> lk(m_);
        for(auto &kv : lanes_) {
            std::lock_guard<std::mutex> ll(kv.second->m);
            if(kv.second->c.n() > 0) { packed_chunk part; std::swap(part, kv.second->c); push_chunk(std::move(part)); }
        }
        flush_ = true; cv_pack_.notify_all();
        while(!ready_.empty() || gathering_ > 0 || in_flight_ > 0) cv_done_.wait(lk);      // (after an error the stages still run dry: batches are dropped, not merged)
        flush_ = false;
        if(err_) throw gpu_error(err_, err_msg_.c_str());
    }

    long submitted() const { std::lock_guard<std::mutex> lk(m_); return next_; }       // graphs handed over so far (all of them after drain())
    // graphs whose status word was not ALD_ST_OK (the reference would have aborted on an assert, or printed its own skip message)
    long failed_graphs() const { std::lock_guard<std::mutex> lk(m_); return failed_; }
    long batches() const { std::lock_guard<std::mutex> lk(m_); return batches_; }
    // busy seconds of the stages so far: packing chunks into a batch / upload + kernel + download / merge into the sink
    void stage_seconds(double &pack, double &gpu, double &merge) const { std::lock_guard<std::mutex> lk(m_); pack = t_pack_; gpu = t_gpu_; merge = t_merge_; }

private:
    enum { FREE, BUSY };
    struct slot { ald_batch *b = nullptr; std::vector<int32_t> sid; long first = 0; int state = FREE; int dev = 0; bool done = false; };
    struct lane { std::mutex m; packed_chunk c; packed_chunk::scratch tmp; };

    // A thread's lane, looked up once: every later submit() of the thread finds it in thread-local storage (the shared map and its lock
    // were taken once per GRAPH before -- the only lock all submitters shared).  The queue's serial number tells a new queue at the
    // address of a destroyed one from the old one.
    lane *my_lane()
    {
        static thread_local const void *tl_queue = nullptr; static thread_local unsigned long tl_serial = 0; static thread_local lane *tl_lane = nullptr;
        if(tl_queue == (const void*)this && tl_serial == serial_) return tl_lane;
        std::lock_guard<std::mutex> lk(m_);
        std::unique_ptr<lane> &p = lanes_[std::this_thread::get_id()];
        if(!p) p.reset(new lane());
        tl_queue = (const void*)this; tl_serial = serial_; tl_lane = p.get();
        return tl_lane;
    }
    static unsigned long next_serial() { static std::atomic<unsigned long> n(0); return ++n; }
    void push_chunk(packed_chunk &&c)                             // m_ held
    {
        c.first = next_; next_ += c.n(); ready_graphs_ += c.n();
        ready_.push_back(std::move(c));
        if(ready_graphs_ >= batch_graphs_) cv_pack_.notify_one();
    }
    void fail(int rc, const char *what)                          // m_ held; the first error is kept
    {
        if(!err_) { err_ = rc; err_msg_ = std::string(what) + ": " + ald_last_error(); }
        cv_space_.notify_all();
    }
    void pack_loop()                                              // gathers handed-over chunks into a free batch
    {
        for(;;) {
            std::vector<packed_chunk> take; int i = -1;
            {
                std::unique_lock<std::mutex> lk(m_);
                for(;;) {
                    const bool work = ready_graphs_ >= batch_graphs_ || (flush_ && !ready_.empty());
                    if(work) {                                            // a free slot on the device with the fewest batches in flight
                        for(size_t k = 0; k < slots_.size(); k++) if(slots_[k].state == FREE && (i < 0 || dev_load_[(size_t)slots_[k].dev] < dev_load_[(size_t)slots_[(size_t)i].dev])) i = (int)k;
                        if(i >= 0) break;
                    }
                    else if(stop_) return;
                    cv_pack_.wait(lk);
                }
                long got = 0;
                while(!ready_.empty() && got < batch_graphs_) { got += ready_.front().n(); take.push_back(std::move(ready_.front())); ready_.pop_front(); }
                ready_graphs_ -= got; gathering_++; slots_[(size_t)i].state = BUSY; slots_[(size_t)i].done = false; batches_++;
                dev_load_[(size_t)slots_[(size_t)i].dev]++; cut_order_.push_back(i);                      // merged in the order the batches are cut
                cv_space_.notify_all();
            }
            const auto t0 = std::chrono::steady_clock::now();
            slot &S = slots_[(size_t)i];
            S.first = take.front().first; S.sid.clear();
            int rc = ALD_OK;
            for(auto &c : take) { if(rc == ALD_OK) rc = c.add_to(S.b); S.sid.insert(S.sid.end(), c.sid.begin(), c.sid.end()); }
            take.clear();
            const auto t1 = std::chrono::steady_clock::now();
            std::lock_guard<std::mutex> lk(m_);
            if(rc != ALD_OK) fail(rc, "ald_batch_add_packed");
            t_pack_ += std::chrono::duration<double>(t1 - t0).count();
            gathering_--; in_flight_++;
            gpu_q_[(size_t)S.dev].push_back(i); cv_gpu_[(size_t)S.dev].notify_one();
        }
    }
    void gpu_loop(int d)                                          // upload, kernel, download of one batch at a time, on device slot d
    {
        std::deque<int> &mine = gpu_q_[(size_t)d];
        for(;;) {
            int i; bool ok;
            {   // leaves only when the queue is stopping AND no batch cut for this device is still on its way here: the pack thread counts a
                // batch into dev_load_ when it cuts it (and puts it into the merge order) but queues it only after packing
                std::unique_lock<std::mutex> lk(m_);
                while(mine.empty() && !(stop_ && dev_load_[(size_t)d] == 0)) cv_gpu_[(size_t)d].wait(lk);
                if(mine.empty()) return;
                i = mine.front(); mine.pop_front(); ok = !err_;
            }
            slot &S = slots_[(size_t)i];
            const auto t1 = std::chrono::steady_clock::now();
            int rc = ALD_OK; const char *what = "ald_batch_upload";
            if(ok) {
                rc = ald_batch_upload(S.b);
                if(rc == ALD_OK) { what = "ald_batch_run"; rc = ald_batch_run(S.b); }
                if(rc == ALD_OK) { what = "ald_batch_download"; rc = ald_batch_download(S.b); }
            }
            long bad = 0;
            if(ok && rc == ALD_OK) { const int n = (int)S.sid.size(); ald_result_view r; for(int g = 0; g < n; g++) if(ald_batch_get_result(S.b, g, &r) == ALD_OK && r.status != ALD_ST_OK) bad++; }
            const auto t2 = std::chrono::steady_clock::now();
            std::lock_guard<std::mutex> lk(m_);
            if(rc != ALD_OK) fail(rc, what);
            t_gpu_ += std::chrono::duration<double>(t2 - t1).count();
            failed_ += bad;
            S.done = true; dev_load_[(size_t)d]--; cv_merge_.notify_one(); cv_pack_.notify_all();
        }
    }
    void merge_loop()
    {
        for(;;) {
            int i; bool ok;
            {   // the oldest cut batch, once its device is done with it: ticket order, whichever device finishes first
                std::unique_lock<std::mutex> lk(m_);
                while(!(!cut_order_.empty() && slots_[(size_t)cut_order_.front()].done) && !(stop_ && cut_order_.empty())) cv_merge_.wait(lk);
                if(cut_order_.empty()) return;
                i = cut_order_.front(); cut_order_.pop_front(); ok = !err_;
            }
            slot &S = slots_[(size_t)i];
            int rc = ALD_OK;
            const auto t0 = std::chrono::steady_clock::now();
            if(ok) rc = ald_tset_add_batch(sink_, S.b, S.sid.data(), (int64_t)S.first << 20, skip_ ? 1 : 0);      // assembler.cc:1105-1133 for the whole batch
            ald_batch_clear(S.b);
            const auto t1 = std::chrono::steady_clock::now();
            std::lock_guard<std::mutex> lk(m_);
            t_merge_ += std::chrono::duration<double>(t1 - t0).count();
            if(rc != ALD_OK) fail(rc, "ald_tset_add_batch");
            S.state = FREE; S.done = false; in_flight_--;
            cv_pack_.notify_all(); cv_done_.notify_all();
        }
    }

    ald_tset *sink_; bool skip_; int batch_graphs_, chunk_graphs_ = 1; long ready_cap_ = 0; const unsigned long serial_ = next_serial();
    mutable std::mutex m_;
    std::condition_variable cv_pack_, cv_merge_, cv_done_, cv_space_;
    std::unique_ptr<std::condition_variable[]> cv_gpu_;           // one per device slot
    std::vector<slot> slots_; std::vector<std::deque<int>> gpu_q_; std::vector<int> dev_load_; std::deque<int> cut_order_; std::deque<packed_chunk> ready_;
    std::map<std::thread::id, std::unique_ptr<lane>> lanes_;
    long next_ = 0, failed_ = 0, batches_ = 0, ready_graphs_ = 0; int in_flight_ = 0, gathering_ = 0; bool stop_ = false, flush_ = false;
    double t_pack_ = 0, t_gpu_ = 0, t_merge_ = 0;
    int err_ = 0; std::string err_msg_;
    std::thread merge_thread_; std::vector<std::thread> pack_threads_, gpu_threads_;
};

} // namespace aletsch
