#!/usr/bin/env python3
"""VGPR spills placed where EXEC is narrowed (diagnostic + test helper, round 4).

A VGPR spill (scratch_store ... Folded Spill) saves only the lanes that are ACTIVE where it stands.  If the register allocator puts it inside a
divergent region -- between an instruction that narrows EXEC (s_and_saveexec_b64, s_mov_b64 exec, s[..], s_and(n2)_b64 exec, ...) and the
s_or_b64 exec, exec, ... that widens it again -- and the value is reloaded after the region, the lanes that were inactive at the spill get
whatever the scratch slot held before.  That is what round 4's faulting test build (-DALD_STARFIX_MAX=1 on the slab twins) did: four spills
behind `s_mov_b64 exec, s[2:3]` of the two-instruction region that clears the sweep's marks, reloaded at the end of the sweep under full EXEC.

  python tools/isa_spill_audit.py file.s [...]        -> one line per offending spill; exit 1 if there is any
A spill offends if (a) it stands inside a region of narrowed EXEC -- regions are matched by the scalar register the old EXEC was saved in --,
(b) the reload that belongs to it (the next access of the same scratch bytes, slots are reused) stands outside that region, and (c) the register
was not written inside the region before the spill (a value defined in the region belongs to the region's lanes).  The walk is over the
assembly text, not a control-flow graph: a heuristic that found exactly the four spills of the faulting build and nothing in 112 other kernels."""
import re, sys

SAVE = re.compile(r'^\s*(?:s_and_saveexec_b64|s_andn2_saveexec_b64|s_or_saveexec_b64)\s+(s\[\d+:\d+\])|^\s*s_mov_b64\s+(s\[\d+:\d+\]),\s*exec\b')
LOOPMASK = re.compile(r'^\s*s_mov_b64\s+(s\[\d+:\d+\]),\s*0\b')
CLOSE = re.compile(r'^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])')
SPILL = re.compile(r'^\s*scratch_store_\w+\s+off,\s*(v\[?[\d:]+\]?),\s*off(?:\s+offset:(\d+))?\s*;.*Folded Spill')
RELOAD = re.compile(r'^\s*scratch_load_\w+\s+(v\[?[\d:]+\]?),\s*off,\s*off(?:\s+offset:(\d+))?\s*;.*Folded Reload')
FUNC = re.compile(r'^(_Z\w+|ald_\w+):')

def vregs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

DEST = re.compile(r'^\s*(v_\w+|global_load_\w+|scratch_load_\w+|ds_read\w*|buffer_load_\w+|flat_load_\w+)\s+(v\[\d+:\d+\]|v\d+\b)')
def dest_regs(line):
    m = DEST.match(line)
    if not m or m.group(1).startswith(('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane')): return set()
    return vregs(m.group(2))

def audit(path, verbose=False):
    """Regions: `s_or_b64 exec, exec, R` closes the region opened by the latest instruction that saved EXEC into R (s_*_saveexec_b64 R / s_mov_b64 R,
    exec) or started a lane-dropping loop with R as its mask (s_mov_b64 R, 0); between the two, EXEC may be narrower than outside."""
    bad = 0
    lines = open(path).read().split('\n')
    fn = None; start = 0
    funcs = []
    for i, l in enumerate(lines):
        m = FUNC.match(l)
        if m:
            if fn is not None: funcs.append((fn, start, i))
            fn = m.group(1); start = i
    if fn is not None: funcs.append((fn, start, len(lines)))
    for fn, a, b in funcs:
        last_save = {}; regions = []; events = []
        label_at = {lines[k].split()[0]: k for k in range(a, b) if lines[k].startswith('.L') and lines[k].split() and lines[k].split()[0].endswith(':')}
        for i in range(a, b):
            l = lines[i]
            m = SAVE.match(l)
            if m:
                # `s_and_saveexec_b64 R, cond` + `s_cbranch_execz LABEL`: the region ends at LABEL (where R is or-ed back), when that lies ahead
                nxt = next((lines[k] for k in range(i + 1, min(b, i + 4)) if lines[k].strip() and not lines[k].lstrip().startswith(';')), '')
                mb = re.match(r'^\s*s_cbranch_execz\s+(\.\w+)', nxt)
                if mb and (mb.group(1) + ':') in label_at and label_at[mb.group(1) + ':'] > i: regions.append((i, label_at[mb.group(1) + ':'])); last_save.pop(m.group(1) or m.group(2), None)
                else: last_save[m.group(1) or m.group(2)] = i
                continue
            mw = re.match(r'^\s*s_\w+\s+(s\[\d+:\d+\])\s*,', l)              # any other scalar write to a saved mask's register: that save is dead
            if mw and mw.group(1) in last_save and not LOOPMASK.match(l) and not re.match(r'^\s*s_or_b64\s+(s\[\d+:\d+\]),\s*s\[\d+:\d+\],\s*\1', l) and not re.match(r'^\s*s_or_b64\s+(s\[\d+:\d+\]),\s*\1', l):
                last_save.pop(mw.group(1), None)
            m = LOOPMASK.match(l)
            if m: last_save.setdefault(m.group(1), i); last_save[m.group(1)] = i; continue
            m = CLOSE.match(l)
            if m:
                if m.group(1) in last_save: regions.append((last_save.pop(m.group(1)), i))
                continue
            m = SPILL.match(l)
            if m: events.append((i, 'S', int(m.group(2) or 0), 4 * len(vregs(m.group(1))))); continue
            m = RELOAD.match(l)
            if m: events.append((i, 'L', int(m.group(2) or 0), 4 * len(vregs(m.group(1)))))
        # scratch slots are reused for different values: the reloads that belong to a spill are the ones that follow it (in text order, wrapping
        # around once for loops) before the next spill to the same bytes
        n = len(events)
        for j, (si, kind, off, size) in enumerate(events):
            if kind != 'S': continue
            inside = [(o, c) for o, c in regions if o < si < c]
            if not inside: continue
            o, c = max(inside, key=lambda r: r[0])          # the innermost region around the spill
            mine = []
            for step in range(1, n):
                li, k2, off2, size2 = events[(j + step) % n]
                if not (off2 < off + size and off < off2 + size2): continue
                if k2 == 'S': break
                mine.append(li)
            outside = [li for li in mine if not (o < li < c)]
            # a value (re)defined inside the region belongs to the region's lanes: spilling it there is what the program means.  Only a value
            # that was live INTO the region -- no write to the register between the region's start and the spill -- loses lanes.
            regs = vregs(SPILL.match(lines[si]).group(1))
            defined_inside = any(regs & dest_regs(lines[k]) for k in range(o + 1, si))
            if outside and not defined_inside:
                bad += 1
                print("  %s: %s: scratch bytes %d..%d are spilled at line %d inside the EXEC region of lines %d..%d (%s) and reloaded outside it, at line(s) %s"
                      % (path, fn, off, off + size - 1, si + 1, o + 1, c + 1, lines[o].strip().split(';')[0].strip(), [x + 1 for x in outside][:4]))
            elif verbose and mine:
                print("  (%s: bytes %d.. spilled at line %d inside region %d..%d, reloaded inside or defined inside)" % (fn, off, si + 1, o + 1, c + 1))
    return bad

if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        total += audit(p)
    print("VGPR spill stores inside a region of narrowed EXEC whose slot is reloaded outside the region:", total)
    sys.exit(1 if total else 0)
