#!/usr/bin/env python
"""Static scan of a gfx950 assembly listing for two wait patterns that serialise memory traffic:

  * a global load whose `s_waitcnt vmcnt(0)` follows within two instructions ("immediately waited"): the load is not
    overlapped with anything -- typically `x = 0; if (lane_ok) x = load` where hipcc folds the bf16 unpacking into the guarded
    block (the remedy is an unconditional load from a clamped address and a select on the unpacked value);
  * a `vmcnt(0)` wait behind a global store with no load in between ("wait after store"): vmcnt counts stores, so the wave
    waits for the store's acknowledgement before it may issue the next one -- typically a coefficient load between stores or a
    wait hipcc re-inserts in every guarded block (the remedy is to issue all loads first and force ONE wait, epi_ready in gemm.hip).

usage: isa_waits.py FILE.s [substring-of-kernel-name ...]     (FILE.s from `hipcc --offload-device-only -S`)
The scan is linear over the listing (it does not follow branches): counts are upper bounds per kernel, good for before / after."""
import re, sys

def scan(path, filters=(), text=None):
    s = open(path).read() if text is None else text
    out = []
    for m in re.finditer(r'^(_Z\S+):\s*;?.*\n', s, re.M):
        name = m.group(1)
        if filters and not any(f in name for f in filters):
            continue
        a = m.end(); b = s.find('.Lfunc_end', a)
        lines = [l.strip() for l in s[a:b].split('\n') if l.strip() and not l.strip().startswith(';')]
        loads = sum(1 for l in lines if l.startswith('global_load') or (l.startswith('buffer_load') and ' lds' not in l))
        stores = sum(1 for l in lines if l.startswith('global_store') or l.startswith('buffer_store'))
        imm = sum(1 for i, l in enumerate(lines) if (l.startswith('global_load') or (l.startswith('buffer_load') and ' lds' not in l))
                  and any('vmcnt(0)' in x for x in lines[i + 1:i + 3]))
        pend = False; after = 0
        for l in lines:
            if l.startswith('global_store') or l.startswith('buffer_store'): pend = True
            elif l.startswith('global_load') or l.startswith('buffer_load'): pend = False
            elif l.startswith('s_waitcnt') and 'vmcnt(0)' in l and pend: after += 1; pend = False
        out.append((name, loads, imm, stores, after))
    return out

if __name__ == '__main__':
    for name, loads, imm, stores, after in scan(sys.argv[1], sys.argv[2:]):
        short = re.sub(r'NS_\d+[A-Za-z]+.*$', '', name)[:70]
        print(f"{short:70s} loads {loads:3d} immediately-waited {imm:3d}   stores {stores:3d} wait-after-store {after:3d}")
