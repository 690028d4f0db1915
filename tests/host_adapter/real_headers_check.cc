// real_headers_check.cc -- compile-only: the three host adapters instantiated with the REFERENCE'S OWN types, taken from its headers
// where they lie (-I /root/reference/{rnacore,scallop,util,graph,gtf}); nothing of the reference is copied or built.  Driven by
// tests/test_abi_cpu.py::test_host_adapters_compile_against_the_reference_headers (g++ -std=c++11 -fsyntax-only -Wall; skipped where
// the reference tree is absent).  What it pins: every member the adapters touch -- splice_graph::num_vertices / edges / get_edge_weight
// / get_edge_info / get_vertex_weight / get_vertex_info / strand (rnacore/splice_graph.h:25-143), edge_info's strand / count / abd /
// samples / spAbd (rnacore/edge_info.h:14-35), vertex_info's lpos / rpos / type (rnacore/vertex_info.h:12-43), hyper_set::nodes
// (scallop/hyper_set.h:28-85), phase_set::pmap (rnacore/phase_set.h:21-32), the parameters the path reads (util/parameters.h) and the
// fields of path (rnacore/path.h:14-35) -- exists there under that name with a type the adapter's use of it compiles against.
#include "splice_graph.h"
#include "hyper_set.h"
#include "phase_set.h"
#include "parameters.h"
#include "path.h"
#include "../../aletsch_amd/host/gpu_scallop.hpp"
#include "../../aletsch_amd/host/gpu_dispatch.hpp"

// meta/assembler.cc:1110-1121 -> the two-call shape
template class aletsch::gpu_scallop<splice_graph, hyper_set, parameters, path>;
// the batched form incl. the raw hand-over (row f1: the graph and phase set as assemble(gx, px, sid) receives them)
template class aletsch::gpu_scallop_batch<splice_graph, hyper_set, parameters, path>;
template int aletsch::gpu_scallop_batch<splice_graph, hyper_set, parameters, path>::enqueue_raw<phase_set>(splice_graph &, const phase_set &, int);
// meta/incubator.cc:609-637 / meta/assembler.cc:296-347 -> the dispatch queue, both hand-over forms
template class aletsch::gpu_assembly_queue<splice_graph, hyper_set, parameters>;
template bool aletsch::gpu_assembly_queue<splice_graph, hyper_set, parameters>::submit_raw<phase_set>(splice_graph &, const phase_set &, int, int);

int use_them(splice_graph &gx, hyper_set &hx, phase_set &px, const parameters &cfg, ald_tset *sink)
{
    aletsch::gpu_scallop<splice_graph, hyper_set, parameters, path> sx(gx, hx, cfg, false);
    sx.assemble();
    aletsch::gpu_assembly_queue<splice_graph, hyper_set, parameters> q(cfg, sink, cfg.skip_single_exon_transcripts);
    q.submit(gx, hx, 0); q.submit_raw(gx, px, 1, cfg.max_group_boundary_distance); q.drain();
    return (int)sx.paths.size() + sx.status;
}
