"""Static instruction mix of k_macro_step by phase: python tools/phase_hist.py <k_macro_step.s>, where the listing is the kernel cut out of
    hipcc <the library's flags> -S --cuda-device-only -DGRIP_MARKS csrc/grip_sim.hip
(the marks build leaves one comment per STAMP site: csrc/grip_physics.h, GRIP_MARKS). Classes: valu, v_mov, v_cndmask, v_cmp, lane (v_readlane / v_writelane), transcendental,
v_permlane32_swap, DPP, scalar ALU, branches, s_waitcnt, s_nop, scalar loads, LDS, vector memory."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
def cls(op):
    if op.startswith('v_mov'): return 'vmov'
    if op.startswith(('v_readlane','v_writelane','v_readfirstlane')): return 'lane'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith(('v_rcp','v_rsq','v_sqrt','v_exp','v_log','v_sin','v_cos')): return 'trans'
    if op.startswith('v_permlane'): return 'swap'
    if op.endswith('_dpp'): return 'dpp'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith(('s_load','s_buffer','s_memtime','s_memrealtime')): return 'smem'
    if op.startswith(('s_cbranch','s_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_','flat_','buffer_','scratch_')): return 'vmem'
    return 'other'
cur = 'pre'; order = [cur]; H = collections.defaultdict(collections.Counter); labels = collections.defaultdict(list)
for l in lines:
    m = re.search(r'==MARK (\d+) line (\d+)', l)
    if m:
        cur = 'after mark %s (line %s)' % (m.group(1), m.group(2)); order.append(cur); continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[cur].append(m.group(1)); continue
    m = re.match(r'^\s+([a-z_0-9]+)', l)
    if m and not l.strip().startswith(';') and not l.strip().startswith('.'):
        H[cur][cls(m.group(1))] += 1
cols = ['valu','vmov','cnd','cmp','lane','trans','swap','dpp','salu','branch','wait','nop','smem','ds','vmem']
print('%-34s' % 'region', ' '.join('%6s' % c for c in cols), '   all')
tot = collections.Counter()
for r in order:
    print('%-34s' % r, ' '.join('%6d' % H[r][c] for c in cols), '%6d' % sum(H[r].values()), ' BBs', len(labels[r]))
    tot.update(H[r])
print('%-34s' % 'TOTAL', ' '.join('%6d' % tot[c] for c in cols), '%6d' % sum(tot.values()))
