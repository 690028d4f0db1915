#!/bin/bash
# The committed summaries of one tools/profile_round.sh call (run here, on the files gpurun merged back):
#   bash tools/summarize_round.sh <tag> <profiles/rNN>     -> <dir>/zz_kernel_stats.csv, zz_pmc_traffic.json, zz_pmc_sq.json (each with the `source`
#   block of the run's own bench line: git HEAD, kernel source / library / bench hashes) and profiles/pmc_traffic.json, which bench.py quotes
set -e
tag=$1; out=$2; R=$(cd "$(dirname "$0")/.." && pwd); G=$R/gpurun_out
python3 $R/profiles/summarize.py stats $G/${tag}_stats/s_results.db $out/zz_kernel_stats.csv --skip-first 4 --bench-log $G/${tag}_stats.log > /dev/null
python3 $R/profiles/summarize.py pmc $G/${tag}_pmc1/p_results.db $G/${tag}_pmc2/p_results.db $out/zz_pmc_traffic.json --bench-log $G/${tag}_pmc1.log > /dev/null
python3 $R/profiles/summarize.py sq $G/${tag}_pmc3/p_results.db $G/${tag}_pmc4/p_results.db $out/zz_pmc_sq.json --bench-log $G/${tag}_pmc3.log > /dev/null
python3 - "$out" "$R" <<'PY'
import json, sys
out, R = sys.argv[1], sys.argv[2]
t = json.load(open(out + "/zz_pmc_traffic.json")); q = json.load(open(out + "/zz_pmc_sq.json"))
t.update({"graphs_per_gpu": 100000, "vertices": 64, "edges": 256,
          "derived_from": out.split("/")[-2] + "/" + out.split("/")[-1] + "/zz_pmc_traffic.json + zz_pmc_sq.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_*, separate passes, per launch of ald_decomp_kernel_c1; ONE tools/profile_round.sh call; `source` = the build it was taken on)",
          "instr_per_graph": {k: round(q["per_graph"][k]) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")},
          "lds_bytes_per_workgroup": 8176})
json.dump(t, open(R + "/profiles/pmc_traffic.json", "w"), indent=1)
print(open(out + "/zz_kernel_stats.csv").read().split("\n")[-4:-1])
print("traffic GB %.3f" % (t["traffic_bytes_per_launch"] / 1e9), t["instr_per_graph"], t["source"])
PY
