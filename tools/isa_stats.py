#!/usr/bin/env python3
"""Static instruction mix per device function of one size class (diagnostic):  python tools/_isa_stats.py [class_id] [waves_per_eu]   (ISA_EXTRA="-D..." adds compile flags)"""
import collections, re, subprocess, sys, os
cid = sys.argv[1] if len(sys.argv) > 1 else "1"; waves = sys.argv[2] if len(sys.argv) > 2 else "5"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/isa_c%s.s" % cid
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-fPIC", "-ffp-contract=off", "-DALD_CLASS_ID=" + cid, "-DALD_WAVES_PER_EU=" + waves, *os.environ.get("ISA_EXTRA", "").split(),
                "--cuda-device-only", "-S", "-o", out, os.path.join(root, "aletsch_amd/csrc/decomp_class.hip")], check=True, stderr=subprocess.DEVNULL)
cur = None; cnt = collections.defaultdict(collections.Counter); size = collections.Counter()
for l in open(out):
    m = re.match(r'^(_Z\w+|ald_\w+):', l)
    if m: cur = m.group(1); continue
    if cur is None: continue
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';', '//')): continue
    op = t[0]; size[cur] += 1
    for pre, key in (('scratch_store', 'sst'), ('scratch_load', 'sld'), ('global_store', 'gst'), ('global_load', 'gld'), ('ds_', 'ds'), ('s_swappc', 'call'), ('v_writelane', 'wl'), ('v_', 'v'), ('s_', 's')):
        if op.startswith(pre): cnt[cur][key] += 1; break
for f, n in size.most_common(60):
    d = subprocess.run(['c++filt', f], capture_output=True, text=True).stdout.strip()[:64]
    c = cnt[f]; print(f"{n:6d} v={c['v']:5d} s={c['s']:5d} ds={c['ds']:4d} gld={c['gld']:4d} gst={c['gst']:4d} sld={c['sld']:3d} sst={c['sst']:3d} wl={c['wl']:3d} call={c['call']:3d} {d}")
for l in open(out):
    if re.match(r'^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size)', l): print(l.rstrip())
