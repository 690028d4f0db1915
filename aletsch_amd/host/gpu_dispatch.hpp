// gpu_dispatch.hpp -- queue + flusher in place of the reference's inline per-graph calls.
//
// The reference runs one pool task per cluster (meta/incubator.cc:553-577, 609-637); inside a task every sample graph and the
// merged graph go through assembler::assemble(gx, px, sid) (meta/assembler.cc:296-347, 370, 1075-1136), which builds a scallop,
// decomposes ONE graph and merges its transcripts into the shared transcript_set under `mylock` (assembler.cc:1127-1132).
//
// aletsch::gpu_assembly_queue keeps that call shape for the pool tasks -- submit(gx, hx, sid) where the reference has
// `scallop sx(gx, hx, pa); sx.assemble(); ... tm.add(ts)` -- and batches ACROSS tasks and clusters:
//
//     pool threads --submit()--> filling batch --full--> [GPU thread: upload, kernel, download] --> [merge thread: ald_tset_add_batch]
//
// `slots` batches rotate through the three stages, so staging (in the submitting threads), the kernel and the merge overlap.
// Graphs are merged into the sink in ticket order (the order submit() calls took the queue's lock), batch after batch -- the
// same per-bucket sequence of trans_item::merge calls as a serial run over the tickets, without `mylock` around the decomposition.
// With one submitting thread the result is bit-identical to the serial loop; with several, the ticket order is whatever the pool
// produced, exactly as the reference's merge order is whatever `mylock` produced.
//
// C++11, header only; compiles inside the reference tree with its own types (see gpu_scallop.hpp for the members used).
#pragma once
#include "gpu_scallop.hpp"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>

namespace aletsch {

template<class SpliceGraph, class HyperSet, class Parameters>
class gpu_assembly_queue {
public:
    // sink: the shared result set (the reference's `tmerge`); skip_single_exon: cfg.skip_single_exon_transcripts (assembler.cc:1117)
    gpu_assembly_queue(const Parameters &cfg, ald_tset *sink, bool skip_single_exon = false, int device = 0, int batch_graphs = 65536, int slots = 3)
        : sink_(sink), skip_(skip_single_exon), batch_graphs_(batch_graphs < 1 ? 1 : batch_graphs)
    {
        if(!sink) throw std::invalid_argument("gpu_assembly_queue: null sink");
        if(slots < 1) slots = 1;
        ald_params p = stage_params(cfg);
        slots_.resize((size_t)slots);
        for(auto &s : slots_) {
            int rc = ald_batch_create(&p, device, &s.b);
            if(rc != ALD_OK) { for(auto &q : slots_) if(q.b) ald_batch_destroy(q.b); throw gpu_error(rc, "ald_batch_create"); }
        }
        gpu_thread_ = std::thread([this] { gpu_loop(); });
        merge_thread_ = std::thread([this] { merge_loop(); });
    }
    ~gpu_assembly_queue()
    {
        try { drain(); } catch(...) {}
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_gpu_.notify_all(); cv_merge_.notify_all();
        gpu_thread_.join(); merge_thread_.join();
        for(auto &s : slots_) if(s.b) ald_batch_destroy(s.b);
    }
    gpu_assembly_queue(const gpu_assembly_queue &) = delete;
    gpu_assembly_queue &operator=(const gpu_assembly_queue &) = delete;

    // Thread-safe.  Returns the graph's ticket; its transcripts carry tid = (ticket << 20) | path index.
    long submit(SpliceGraph &gx, const HyperSet &hx, int sid)
    {
        staged_graph s = stage_graph(gx, hx);                     // the per-graph work stays in the calling thread, outside the lock
        ald_graph_view g = s.view();
        std::unique_lock<std::mutex> lk(m_);
        for(;;) {
            if(err_) throw gpu_error(err_, err_msg_.c_str());
            if(fill_ >= 0) break;
            for(size_t i = 0; i < slots_.size(); i++) if(slots_[i].state == FREE) { fill_ = (int)i; slots_[i].state = FILLING; slots_[i].first = next_; break; }
            if(fill_ < 0) cv_free_.wait(lk);                      // every batch is in flight: the pool is ahead of the GPU
        }
        slot &S = slots_[(size_t)fill_];
        int rc = ald_batch_add_graph(S.b, &g);
        if(rc != ALD_OK) throw gpu_error(rc, "ald_batch_add_graph");
        S.sid.push_back((int32_t)sid);
        const long ticket = next_++;
        if((int)S.sid.size() >= batch_graphs_) hand_over();
        return ticket;
    }

    // Flushes the partial batch and returns when every submitted graph has been merged into the sink.
    void drain()
    {
        std::unique_lock<std::mutex> lk(m_);
        if(fill_ >= 0) {
            if(slots_[(size_t)fill_].sid.empty()) { slots_[(size_t)fill_].state = FREE; fill_ = -1; }
            else hand_over();
        }
        while(in_flight_ > 0) cv_done_.wait(lk);
        if(err_) throw gpu_error(err_, err_msg_.c_str());
    }

    long submitted() const { std::lock_guard<std::mutex> lk(m_); return next_; }
    // graphs whose status word was not ALD_ST_OK (the reference would have aborted on an assert, or printed its own skip message)
    long failed_graphs() const { std::lock_guard<std::mutex> lk(m_); return failed_; }
    long batches() const { std::lock_guard<std::mutex> lk(m_); return batches_; }

private:
    enum { FREE, FILLING, QUEUED };
    struct slot { ald_batch *b = nullptr; std::vector<int32_t> sid; long first = 0; int state = FREE; };

    void hand_over()                                              // m_ held
    {
        slots_[(size_t)fill_].state = QUEUED; gpu_q_.push_back(fill_); fill_ = -1; in_flight_++; batches_++;
        cv_gpu_.notify_one();
    }
    void fail(int rc, const char *what)                          // m_ held; the first error is kept
    {
        if(!err_) { err_ = rc; err_msg_ = std::string(what) + ": " + ald_last_error(); }
    }
    void gpu_loop()
    {
        for(;;) {
            int i;
            { std::unique_lock<std::mutex> lk(m_); while(gpu_q_.empty() && !stop_) cv_gpu_.wait(lk); if(gpu_q_.empty()) return; i = gpu_q_.front(); gpu_q_.pop_front(); }
            ald_batch *b = slots_[(size_t)i].b;
            int rc; const char *what = "ald_batch_upload";
            if((rc = ald_batch_upload(b)) == ALD_OK) { what = "ald_batch_run"; rc = ald_batch_run(b); }
            if(rc == ALD_OK) { what = "ald_batch_download"; rc = ald_batch_download(b); }
            long bad = 0;
            if(rc == ALD_OK) { const int n = (int)slots_[(size_t)i].sid.size(); ald_result_view r; for(int g = 0; g < n; g++) if(ald_batch_get_result(b, g, &r) == ALD_OK && r.status != ALD_ST_OK) bad++; }
            std::lock_guard<std::mutex> lk(m_);
            if(rc != ALD_OK) fail(rc, what);
            failed_ += bad;
            merge_q_.push_back(i); cv_merge_.notify_one();
        }
    }
    void merge_loop()
    {
        for(;;) {
            int i; bool ok;
            { std::unique_lock<std::mutex> lk(m_); while(merge_q_.empty() && !stop_) cv_merge_.wait(lk); if(merge_q_.empty()) return; i = merge_q_.front(); merge_q_.pop_front(); ok = !err_; }
            slot &S = slots_[(size_t)i];
            int rc = ALD_OK;
            if(ok) rc = ald_tset_add_batch(sink_, S.b, S.sid.data(), (int64_t)S.first << 20, skip_ ? 1 : 0);      // assembler.cc:1105-1133 for the whole batch
            ald_batch_clear(S.b);
            std::lock_guard<std::mutex> lk(m_);
            if(rc != ALD_OK) fail(rc, "ald_tset_add_batch");
            S.sid.clear(); S.state = FREE; in_flight_--;
            cv_free_.notify_all(); cv_done_.notify_all();
        }
    }

    ald_tset *sink_; bool skip_; int batch_graphs_;
    mutable std::mutex m_;
    std::condition_variable cv_free_, cv_gpu_, cv_merge_, cv_done_;
    std::vector<slot> slots_; std::deque<int> gpu_q_, merge_q_;
    int fill_ = -1; long next_ = 0, failed_ = 0, batches_ = 0; int in_flight_ = 0; bool stop_ = false;
    int err_ = 0; std::string err_msg_;
    std::thread gpu_thread_, merge_thread_;
};

} // namespace aletsch
