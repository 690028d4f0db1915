"""The engine's device code (aletsch_amd/csrc/decomp_device.h, every size class the emulation builds, raw build) compiled for the CPU with
AddressSanitizer + UBSan and run over ordinary, stranded, phased, multi-sample, large, capacity-retry and raw workloads: an index that
leaves a work array, the slab or the record pool, a shift or a conversion that overflows would go unnoticed on the GPU (GPU sanitizers are
not available on the pool) -- here it aborts.  Results are compared with the oracle as everywhere else."""
import os
import subprocess
import sys
import textwrap

import common

SCRIPT = textwrap.dedent('''
    import sys, os, ctypes as C, numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import aletsch_amd as A, common
    from aletsch_amd.packed import PackedGraphs
    common._EMU = C.CDLL(%r); common._EMU.emu_result_free.argtypes = [C.c_void_p]
    for kw in (dict(seed=1002, n_graphs=150, v_min=64, v_max=64, fixed_edges=256),
               dict(seed=3, n_graphs=200, v_min=4, v_max=40, edges_per_vertex=5, n_samples=4, phasing_per_graph=10, strand_mode=1, weight_mode=2),
               dict(seed=4, n_graphs=30, v_min=100, v_max=300, edges_per_vertex=4, n_samples=2, phasing_per_graph=25, layout_mode=1),
               dict(seed=5, n_graphs=3, v_min=400, v_max=520, edges_per_vertex=4),
               dict(seed=6, n_graphs=1, v_min=1100, v_max=1200, edges_per_vertex=3)):
        pg = A.synth(**kw)
        assert not common.compare_results(common.oracle_run(pg, threads=4)[0], common.emu_run(pg)[0], pg.n), kw
    pg = A.synth(seed=77, n_graphs=40, v_min=8, v_max=120, edges_per_vertex=4, phasing_per_graph=5, n_samples=2)      # the largest class (32-bit creation ids)
    assert not common.compare_results(common.oracle_run(pg, threads=4)[0], common.emu_run(pg, force_class=13)[0], pg.n)
    p = A.default_params(); p.max_num_exons = 30
    pg = A.synth(seed=55, n_graphs=60, v_min=10, v_max=60, edges_per_vertex=3, weight_mode=2, phasing_per_graph=2)
    assert not common.compare_results(common.oracle_run(pg, params=p)[0], common.emu_run(pg, params=p)[0], pg.n)
    os.environ["ALD_DEBUG_UNDERCLASS"] = "2"                      # working sets that overflow their class on purpose
    pg = A.synth(seed=78, n_graphs=60, v_min=20, v_max=200, edges_per_vertex=4, phasing_per_graph=3)
    assert not common.compare_results(common.oracle_run(pg, threads=4)[0], common.emu_run(pg)[0], pg.n)
    del os.environ["ALD_DEBUG_UNDERCLASS"]
    rng = np.random.default_rng(5); items = []
    for t in range(120):                                          # raw graphs: the pre-steps in the load phase
        g, phases = common.gene_like_raw(rng, n_runs=int(rng.integers(3, 30)), strand="+-."[t %% 3])
        pgr = PackedGraphs.from_graphs([g]); pgr.edge_rank = np.array(sorted(range(len(g["edges"])), key=lambda q: (g["edges"][q][0], g["edges"][q][1])), np.int32)
        items.append((pgr, phases, int(rng.choice([0, 50, 10000]))))
    r = common.emu_run_raw(items)[0]
    assert (r.status >= 100).sum() < 40
    print("SANITIZED RUN COMPLETE")
''')


def test_engine_emulation_under_sanitizers():
    root = common.ROOT
    out = os.path.join(root, "tests", "_build", "emu_san")
    flags = "-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -DALD_EMU -DALD_RAW_VARIANT -fsanitize=address,undefined -fno-sanitize-recover=undefined -w"
    r = subprocess.run(["make", "-C", os.path.join(root, "tests", "kernel_emu"), "-j8", "OUT=" + out, "CXXFLAGS=" + flags], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    pre = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    env = dict(os.environ, LD_PRELOAD=":".join(pre), ASAN_OPTIONS="detect_leaks=0")
    code = SCRIPT % (root, os.path.join(root, "tests"), os.path.join(out, "libkernel_emu.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "SANITIZED RUN COMPLETE" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
