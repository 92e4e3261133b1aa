#!/usr/bin/env python3
"""Summarise rocprofv3 csv output of tools_prof.sh: per-kernel durations and PMC counters."""
import csv, glob, sys, collections, os
root = sys.argv[1]
def short(n):
    n = n.split('(')[0]
    for k in ['k_wave_row', 'k_wave_cp', 'k_wave_col', 'k_solver_row', 'k_cp_row', 'k_col', 'k_rows_r2c', 'k_rows_c2r', 'k_loss_finalize', 'k_iters_done']:
        if k in n:
            tag = k
            if 'PdrP' in n: tag += '<pdr>'
            elif 'PddrP' in n: tag += '<pddr>'
            elif 'AdmmP' in n: tag += '<admm>'
            if k == 'k_wave_col': tag += '<res2>' if ('ILi0E' in n or '<0,' in n) else '<mul>'
            if k in ('k_wave_row', 'k_wave_cp'):
                import re
                m = re.search(r'(Lb[01]ELb[01]E|Li[01]ELb[01]ELb[01]E)', n)
                if m: tag += '<' + m.group(1).replace('Lb','').replace('Li','').replace('E','') + '>'
            if k == 'k_col':
                import re
                m = re.search(r'k_colI[fd]Li(\d)', n) or re.search(r'k_col<\w+, (\d)', n)
                if m: tag += f'<mode{m.group(1)}>'
            return tag
    return n[:60]
for f in glob.glob(os.path.join(root, 'kt', '**', '*kernel_trace.csv'), recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('== kernel trace (us): name count avg min max')
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f'{k:40s} {len(v):5d} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f}')
for p in ['pmc1', 'pmc2', 'pmc3']:
    for f in glob.glob(os.path.join(root, p, '**', '*counter_collection.csv'), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f'== {p}: per-kernel average counter values')
        for k, cs in d.items():
            if 'at::' in k or 'elementwise' in k or 'distribution' in k: continue
            print(k, {c: f'{sum(v)/len(v):.4g}' for c, v in cs.items()}, 'n=', len(next(iter(cs.values()))))
