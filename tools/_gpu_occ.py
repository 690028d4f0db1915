import os, sys, subprocess, json
for k in (4, 8, 10, 12, 14):
    env = dict(os.environ, ALD_WG_PER_CU=str(k))
    r = subprocess.run([sys.executable, "/root/repo/bench.py", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"], capture_output=True, text=True, env=env)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(k, "wg/cu ->", round(d["value"]), "bundles/s  kernel_ms", round(d["roofline"]["kernel_ms"], 1), "grid", d["config"]["grid"], flush=True)
