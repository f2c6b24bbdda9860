"""Development aid: VGPR liveness over a kernel's gfx950 assembly (hipcc -S --cuda-device-only): where is the register pressure peak and
what is live there?  A plain backward dataflow over the control-flow graph of labels and s_cbranch / s_branch.
usage: python tools/vgpr_liveness.py file.s <substring of the kernel symbol> [top N]"""
import re, sys

def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[:])v(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out

NO_DEF = ('ds_write', 'ds_store', 'buffer_store', 'global_store', 'flat_store', 'scratch_store', 'v_cmp', 'v_cmpx', 's_', 'v_readfirstlane', 'v_readlane',
          'global_atomic', 'ds_add', 'ds_max', 'ds_min', 'buffer_wbl2', 'buffer_inv', 'ds_gws')

def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    src = open(path).read()
    m = re.search(r'^(\S*%s\S*):' % re.escape(key), src, re.M)
    i = m.end(); j = src.index('.Lfunc_end', i)
    lines = [l.split(';')[0].rstrip() for l in src[i:j].split('\n')]
    ins = []       # (text, defs, uses, label)
    labels = {}
    for l in lines:
        t = l.strip()
        if not t or t.startswith('.') and not t.endswith(':') or t.startswith('//'):
            continue
        if t.endswith(':'):
            labels[t[:-1]] = len(ins)
            continue
        mn = t.split()[0]
        ops = t[len(mn):].split(',')
        d, u = set(), set()
        if mn.startswith(NO_DEF):
            for o in ops: u |= regs(o)
        else:
            d = regs(ops[0]) if ops else set()
            for o in ops[1:]: u |= regs(o)
            if mn.startswith('v_mfma') or 'v_mac' in mn or 'v_fmac' in mn or 'v_dot' in mn and False:
                pass
            if mn.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_movrel', 'v_writelane')): u |= d   # read-modify-write
            if 'dpp' in t or 'row_' in t or 'sdwa' in t.lower(): u |= d       # (bound_ctrl-less DPP keeps old values)
        ins.append((t, d, u))
    n = len(ins)
    succ = [[] for _ in range(n)]
    for k, (t, d, u) in enumerate(ins):
        mn = t.split()[0]
        if mn == 's_branch':
            succ[k] = [labels[t.split()[1]]]
        elif mn.startswith('s_cbranch'):
            succ[k] = [labels[t.split()[1]]] + ([k + 1] if k + 1 < n else [])
        elif mn == 's_endpgm':
            succ[k] = []
        else:
            succ[k] = [k + 1] if k + 1 < n else []
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for k in range(n - 1, -1, -1):
            out = set()
            for s_ in succ[k]: out |= live_in[s_]
            t, d, u = ins[k]
            new = (out - d) | u
            if new != live_in[k]:
                live_in[k] = new; changed = True
    order = sorted(range(n), key=lambda k: -len(live_in[k]))
    print('instructions', n, 'max live', len(live_in[order[0]]))
    seen = []
    for k in order:
        if all(abs(k - s_) > 40 for s_ in seen):
            seen.append(k)
            print('--- at %d: %d live: %s' % (k, len(live_in[k]), ins[k][0]))
            lv = sorted(live_in[k])
            # compress to ranges
            rngs = []; a = b = lv[0]
            for r in lv[1:]:
                if r == b + 1: b = r
                else: rngs.append((a, b)); a = b = r
            rngs.append((a, b))
            print('    ' + ' '.join('v%d' % a if a == b else 'v[%d:%d]' % (a, b) for a, b in rngs))
            if len(seen) >= top: break
    # pressure profile every 100 instructions
    print('profile (max live per 100 instructions):')
    print(' '.join(str(max(len(live_in[k]) for k in range(a, min(n, a + 100)))) for a in range(0, n, 100)))
    return ins, live_in

if __name__ == '__main__':
    main()


def explain(path, key, at=None):
    """for every register live at the peak (or at instruction `at`): the instruction that defined it last (linear order)"""
    sys.argv = [sys.argv[0], path, key, '1']
    ins, live_in = main()
    k = at if at is not None else max(range(len(ins)), key=lambda k: len(live_in[k]))
    by = {}
    for r in sorted(live_in[k]):
        for p in range(k - 1, -1, -1):
            if r in ins[p][1]:
                by.setdefault((p, ins[p][0]), []).append(r); break
        else:
            by.setdefault((-1, 'kernel argument / never defined'), []).append(r)
    for (p, t), rs in sorted(by.items()):
        print('%5d  %-90s %s' % (p, t[:90], rs if len(rs) < 5 else '%d regs from v%d' % (len(rs), rs[0])))
