"""Distil the aggregated counter passes (tools/pmc_aggregate.py output) into profiles/counters.json, the figures bench.py quotes as
*static*.   usage: python tools/pmc_distill.py <aggregate.json> <name under profiles/ it is committed as> [kernel ...]
Per-state kernels (move generation, step, encode, greedy): n = grid threads / lanes per state, HBM bytes per state = (2 x FETCH_SIZE
+ WRITE_SIZE) KB / n (the guide's doubling rule).  fused_plies_kernel: launches of FOUR plies, every figure divided by four."""
import json, sys

agg, name = sys.argv[1], sys.argv[2]
only = set(sys.argv[3:])
d = json.load(open(agg))
path = 'profiles/counters.json'
c = json.load(open(path))
src = ('profiles/%s (rocprofv3 --pmc, one counter set per pass with --kernel-trace only: tools/pmc_round.sh + tools/pmc_aggregate.py; '
       'FETCH_SIZE doubled per MI355X_MICROARCH.md, confirmed on step_kernel / movegen_kernel / encode_kernel whose read bytes are '
       'known exactly)' % name)
LANES = {'movegen_kernel<false>': 2, 'movegen_kernel<true>': 2, 'movegen_kernel<false, true>': 2, 'step_kernel': 1, 'encode_kernel': 4}     # threads per state
# round 3: the kernel has a second template flag (PACKED); counters.json keeps its keys: <false> = rows, <false, true> = packed, <true> = greedy
RENAME = {'movegen_kernel<false, false>': 'movegen_kernel<false>', 'movegen_kernel<true, false>': 'movegen_kernel<true>',
          'advance_kernel<false>': 'advance_kernel', 'encode_kernel<false>': 'encode_kernel'}      # (round 5: template flags DBG / REQ)
for k, x in list(d.items()):
    k = RENAME.get(k, k)
    if only and k not in only:
        continue
    if k in LANES:
        n = x['grid_threads'] // LANES[k]
        c[k] = dict(n=n, fetch_size_kb=x['FETCH_SIZE'], write_size_kb=x['WRITE_SIZE'],
                    hbm_bytes_per_state=(2 * x['FETCH_SIZE'] + x['WRITE_SIZE']) * 1024.0 / n,
                    insts_valu=x['SQ_INSTS_VALU'], insts_salu=x['SQ_INSTS_SALU'], insts_lds=x['SQ_INSTS_LDS'], waves=x['SQ_WAVES'],
                    wave_cycles_quad=x['SQ_WAVE_CYCLES'], wait_any_quad=x['SQ_WAIT_ANY'], wait_inst_any_quad=x['SQ_WAIT_INST_ANY'],
                    active_inst_any_quad=x['SQ_ACTIVE_INST_ANY'], active_inst_valu_quad=x['SQ_ACTIVE_INST_VALU'],
                    lds_idx_active=x.get('SQ_LDS_IDX_ACTIVE'), lds_bank_conflict=x.get('SQ_LDS_BANK_CONFLICT'),
                    launch_ms=x['launch_us_under_pmc']['SQ_INSTS_VALU'] / 1e3, source=src)
    elif k == 'fused_plies_kernel':
        P = 4.0
        c[k] = dict(plies_per_launch_under_pmc=4, fetch_size_kb=x['FETCH_SIZE'] / P, write_size_kb=x['WRITE_SIZE'] / P,
                    insts_valu=x['SQ_INSTS_VALU'] / P, insts_salu=x['SQ_INSTS_SALU'] / P, insts_lds=x['SQ_INSTS_LDS'] / P,
                    insts_vmem_rd=x['SQ_INSTS_VMEM_RD'] / P, insts_vmem_wr=x['SQ_INSTS_VMEM_WR'] / P,
                    wave_cycles_quad=x['SQ_WAVE_CYCLES'] / P, wait_any_quad=x['SQ_WAIT_ANY'] / P, wait_inst_any_quad=x['SQ_WAIT_INST_ANY'] / P,
                    active_inst_any_quad=x['SQ_ACTIVE_INST_ANY'] / P, active_inst_valu_quad=x['SQ_ACTIVE_INST_VALU'] / P,
                    active_inst_sca_quad=x['SQ_ACTIVE_INST_SCA'] / P, tcc_hit=x['TCC_HIT_sum'] / P, tcc_miss=x['TCC_MISS_sum'] / P,
                    launch_ms=x['launch_us_under_pmc']['SQ_INSTS_VALU'] / 1e3 / P,
                    clock_ghz=x['GRBM_GUI_ACTIVE'] / 8 / (x['launch_us_under_pmc']['GRBM_GUI_ACTIVE'] * 1e3),
                    vgpr=x['vgpr'], lds_bytes=x['lds_bytes'],
                    workload='4096 games x 400 simulations, config 2a; launches of FOUR plies, every figure here divided by four = per ply',
                    source=src)
    elif k.startswith('net_forward_kernel') and 'Cfg<8, 8>' in k:
        # round 5: two forms of the kernel -- <..., true> evaluates the free-running path's REQUEST records (what the headline runs:
        # kept under 'net_forward_kernel'), <..., false> float32 planes (the lock-step path: 'net_forward_kernel_planes')
        c['net_forward_kernel' if ', true>' in k else 'net_forward_kernel_planes'] = dict(n=x['grid_threads'] // 64, fetch_size_kb=x['FETCH_SIZE'], write_size_kb=x['WRITE_SIZE'],
                                       mfma_insts=x['SQ_INSTS_VALU_MFMA_F32'], mfma_busy_cycles=x['SQ_VALU_MFMA_BUSY_CYCLES'],
                                       grbm_gui_active=x['GRBM_GUI_ACTIVE'],
                                       mfma_busy_frac=x['SQ_VALU_MFMA_BUSY_CYCLES'] / (x['GRBM_GUI_ACTIVE'] * 128.0),   # 1024 SIMDs / 8 XCD clocks
                                       insts_valu=x['SQ_INSTS_VALU'], insts_salu=x['SQ_INSTS_SALU'], insts_lds=x['SQ_INSTS_LDS'],
                                       lds_bank_conflict=x['SQ_LDS_BANK_CONFLICT'], lds_idx_active=x['SQ_LDS_IDX_ACTIVE'],
                                       wait_inst_lds_quad=x['SQ_WAIT_INST_LDS'], wave_cycles_quad=x['SQ_WAVE_CYCLES'],
                                       wait_any_quad=x['SQ_WAIT_ANY'], wait_inst_any_quad=x['SQ_WAIT_INST_ANY'],
                                       launch_ms_under_pmc=x['launch_us_under_pmc']['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e3, vgpr=x['vgpr'], source=src)
    elif k in ('advance_kernel', 'boundary_kernel'):
        # the free-running stepped path on plain launches (tools/kernels_once.py `free`): per launch of 2048 slots (one half-batch)
        c[k] = dict(slots=x['grid_threads'] // 64, fetch_size_kb=x['FETCH_SIZE'], write_size_kb=x['WRITE_SIZE'],
                    hbm_bytes_per_launch=(2 * x['FETCH_SIZE'] + x['WRITE_SIZE']) * 1024.0,
                    insts_valu=x['SQ_INSTS_VALU'], insts_salu=x['SQ_INSTS_SALU'], insts_lds=x['SQ_INSTS_LDS'],
                    insts_vmem_rd=x['SQ_INSTS_VMEM_RD'], insts_vmem_wr=x['SQ_INSTS_VMEM_WR'],
                    lanes_per_valu_inst=x['SQ_THREAD_CYCLES_VALU'] / x['SQ_ACTIVE_INST_VALU'],
                    wave_cycles_quad=x['SQ_WAVE_CYCLES'], wait_any_quad=x['SQ_WAIT_ANY'], tcc_hit=x['TCC_HIT_sum'], tcc_miss=x['TCC_MISS_sum'],
                    launch_us_under_pmc=x['launch_us_under_pmc']['SQ_INSTS_VALU'], vgpr=x['vgpr'], lds_bytes=x['lds_bytes'], source=src)
    else:
        continue
    if 'SQ_THREAD_CYCLES_VALU' in x and k in c and 'lanes_per_valu_inst' not in c[k]:
        c[k]['lanes_per_valu_inst'] = x['SQ_THREAD_CYCLES_VALU'] / x['SQ_ACTIVE_INST_VALU']      # active lanes per VALU instruction-cycle, of 64
    print('updated', k)
json.dump(c, open(path, 'w'), indent=1)
