#!/usr/bin/env python3
"""Static instructions of a kernel root by the source FUNCTION their line belongs to, and where its SGPR spills (v_writelane / v_readlane
into the spill VGPRs) and scratch accesses sit:  python tools/isa_lines.py <asm from hipcc -gline-tables-only -S> [kernel symbol] [header]"""
import re, collections, bisect, sys, os
asm = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "ald_decomp_kernel_c1"
hdr = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aletsch_amd/csrc/decomp_device.h")
cur = None; fn = None; files = {}
cnt = collections.Counter(); wl = collections.Counter(); rl = collections.Counter(); scr = collections.Counter()
for l in open(asm):
    m = re.match(r'^(_Z\w+|ald_\w+):', l)
    if m: fn = m.group(1); continue
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)); continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';', '//')) or t[0].endswith(':') or fn != kern: continue
    cnt[cur] += 1
    if t[0].startswith('v_writelane'): wl[cur] += 1
    if t[0].startswith('v_readlane') and 'Reload' in l: rl[cur] += 1
    if t[0].startswith('scratch_'): scr[cur] += 1
src = open(hdr).read().split('\n')
marks = [(i + 1, m.group(1)) for i, l in enumerate(src) for m in [re.match(r'^(?:template<[^>]*>\s*)?ALD_(?:INL|FN)\s+[\w:<>]+\s+\**(\w+)\(', l)] if m]
starts = [a for a, _ in marks]
def owner(f, ln):
    name = files.get(f, '?')
    if os.path.basename(hdr) in name:
        i = bisect.bisect_right(starts, ln) - 1
        return marks[i][1] if i >= 0 else '?'
    return os.path.basename(name)
agg = collections.defaultdict(lambda: [0, 0, 0, 0])
for k, n in cnt.items():
    o = owner(*k) if k else '?'; a = agg[o]; a[0] += n; a[1] += wl[k]; a[2] += rl[k]; a[3] += scr[k]
print("%-34s %7s %9s %9s %8s" % ("function", "instrs", "writelane", "reloads", "scratch"))
for o, a in sorted(agg.items(), key=lambda x: -x[1][0])[:45]: print("%-34s %7d %9d %9d %8d" % (o, *a))
print("total", sum(cnt.values()), "writelane", sum(wl.values()), "reloads", sum(rl.values()), "scratch", sum(scr.values()))
print("lines with spill traffic:")
for k in sorted(set(list(wl) + list(rl) + list(scr)), key=lambda k: -(wl[k] + rl[k] + scr[k]))[:40]:
    if k and os.path.basename(hdr) in files.get(k[0], ''): print("  line %5d  wl=%3d rl=%3d scr=%3d  [%s]  %s" % (k[1], wl[k], rl[k], scr[k], owner(*k), src[k[1] - 1].strip()[:110]))
