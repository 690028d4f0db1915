import os, sys, numpy as np
os.environ["ALETSCH_DECOMP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aletsch_amd/lib", os.environ.get("PROF_LIB", "libaletsch_decomp_prof.so"))     # PROF_LIB=libaletsch_decomp_<variant>.so: another profiling build (e.g. make VARIANT=profsize VDEFS="-DALD_PROF -DALD_PROF_STAR_BY_SIZE")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import aletsch_amd as A
names = ["load","broken","triv_eval","triv_mut","small_eval","small_mut","unsplit","collect0","g_balance","g_dp","g_splitmerge","g_collect","finish","T_balance","T_pairs","T_setup","M_load","M_add","M_isect","M_mask","M_sums","M_kill","T_hs","T_tail","S5_dup","S5_body","S5_relink","S6_walk","S6_link","S7_tail","SM_kill","SM_reeval"]
import os
CFG = os.environ.get("PROF_CFG", "cfg2")
for n in (20000 if CFG.startswith("cfg2") else 600,):
    pg = A.synth(seed=1002, n_graphs=n, v_min=64, v_max=64, fixed_edges=256, weight_mode=(2 if CFG == "cfg2flow" else 0)) if CFG.startswith("cfg2") else A.synth(seed=1003, n_graphs=n, v_min=400, v_max=512, edges_per_vertex=4)
    with A.DecompBatch(0, trace_events=(600 if CFG.startswith("cfg2") else 6000)) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        tot = np.zeros(len(names)); cnt = 0
        for g in range(0, n, max(1, n // 200)):
            ev = b.trace(g)
            for c, a, bb, v in ev:
                if c >= 100: tot[c - 100] += v
            cnt += 1
        tot /= cnt
        top = tot[:8].sum()                      # load .. collect0 partition a graph's life; g_* are parts of `unsplit`, T_* / M_* parts of `triv_mut`
        print(f"n={n} kernel_ms={b.kernel_ms():.2f}  mean cycles/graph (top-level phases) = {top:.0f}")
        for k, nm in enumerate(names): print(f"   {nm:14s} {tot[k]:12.0f}  {100*tot[k]/top:5.1f}%" + ("" if k < 8 else "   (part of " + ("unsplit: router / extend" if k < 13 else ("triv_mut" if k < 30 else "small_mut")) + ")"))
