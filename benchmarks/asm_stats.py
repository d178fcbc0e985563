"""Static instruction mix per kernel of a gfx950 assembly listing (hipcc --offload-device-only -S):

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off --offload-device-only -S nf_spline.hip -o /tmp/nf_spline.s
    python benchmarks/asm_stats.py /tmp/nf_spline.s [name filter]

Counts are over the whole kernel body (loops counted once), so they compare two builds of one kernel, not two kernels."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    kernels, cur = {}, None
    meta = {}
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            cur = None if line.startswith('.Lfunc_end') else cur
        m = re.match(r'^\s*\.(vgpr_count|vgpr_spill_count|sgpr_count|group_segment_fixed_size):\s*(\d+)', line)
        if m and meta.get('_cur'):
            meta[meta['_cur']][m.group(1)] = int(m.group(2))
        m = re.match(r'^\s*\.name:\s*(\S+)', line)
        if m:
            meta['_cur'] = m.group(1)
            meta.setdefault(m.group(1), {})
        if cur is None:
            continue
        m = re.match(r'^\t([a-z_0-9]+)', line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith('v_mfma'): k = 'mfma'
        elif op.startswith('v_exp') or op.startswith('v_log') or op.startswith('v_rcp') or op.startswith('v_sqrt') or op.startswith('v_rsq') or op.startswith('v_sin') or op.startswith('v_cos'): k = 'trans'
        elif op.endswith('_f64') or '_f64_' in op: k = 'f64'
        elif op.startswith('v_'): k = 'valu'
        elif op.startswith('ds_'): k = 'lds'
        elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): k = 'vmem:' + op.split('_')[0]
        elif op.startswith('s_waitcnt'): k = 'waitcnt'
        elif op.startswith('s_barrier'): k = 'barrier'
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): k = 'branch'
        elif op.startswith('s_'): k = 'salu'
        else: continue
        kernels[cur][k] = kernels[cur].get(k, 0) + 1
    names = demangle(list(kernels))
    for n, c in kernels.items():
        d = names[n]
        if filt and filt not in d:
            continue
        if not c:
            continue
        mm = meta.get(n, {})
        print(d[:150])
        print('    vgpr %s spill %s lds %s | ' % (mm.get('vgpr_count'), mm.get('vgpr_spill_count'), mm.get('group_segment_fixed_size')) +
              '  '.join('%s %d' % kv for kv in sorted(c.items())))


if __name__ == '__main__':
    main()
